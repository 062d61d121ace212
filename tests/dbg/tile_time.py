"""Kernel timing of the row path vs the tile path on a configuration-B sized liquid."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hoomd-blue_amd"), os.path.join(ROOT, "tests")]
import torch
from hoomd_amd import kernels as K, _abi as A
from oracle import oracle as orc
n = (128, 128, 64) if len(sys.argv) < 2 else tuple(int(v) for v in sys.argv[1].split("x"))
a = (1 / 0.84) ** (1 / 3)
L = np.array(n) * a
g = [(np.arange(k) + 0.5) * a - l / 2 for k, l in zip(n, L)]
z, y, x = np.meshgrid(g[2], g[1], g[0], indexing="ij")
xyz = np.stack([x.ravel(), y.ravel(), z.ravel()], 1) + np.random.default_rng(0).normal(0, 0.12, (x.size, 3))
xyz = (xyz + L / 2) % L - L / 2
N = len(xyz)
gbox, obox = A.Box.from_L(*L), orc.Box.from_L(*L)
pos_t = torch.from_numpy(orc.pack_pos(xyz, np.zeros(N, int))).cuda()
dim = orc.cell_dimensions(obox, 2.9, 1, (0, 0, 0), 3)
cl = K.cell_list_auto(pos_t, N, 0, gbox, dim, want_idx=True)
print("dim", dim, "cell Nmax", cl["Nmax"])
adj = torch.as_tensor(orc.cell_adj(dim, 1, 3).view(np.int32), device="cuda")
def timeit(f, reps=5):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6
nl = K.nlist_binned(pos_t, N, gbox, [[2.5]], 0.4, cl, adj)
tl = K.nlist_tiles(pos_t, N, gbox, [[2.5]], 0.4, cl, adj)
print("n_neigh equal", bool((nl["n_neigh"] == tl["n_neigh"]).all()), "mean", float(tl["n_neigh"].float().mean()), "Nmax", tl["Nmax"].tolist())
import ctypes as C
_bn = K.nlist_binned
import hoomd_amd._abi as AA
# row build timing: rebuild through the wrapper costs allocations, so time the bare launch
def row_build():
    K.nlist_binned(pos_t, N, gbox, [[2.5]], 0.4, cl, adj)
print("row build (wrapper, ~3 launches + allocs) us", timeit(row_build, 3))
print("tile build us", timeit(lambda: A.check(A.lib.hmd_compute_nlist_tiles(C.byref(tl["args"]), None))))
P = [orc.lj_params(1.0, 1.0)]
for tpp in (4, 8, 16):
    print("tile force tpp", tpp, "us", timeit(lambda: K.pair_forces_tiles("lj", pos_t, N, gbox, tl, P, [[6.25]], compute_virial=False, tpp=tpp)))
import hoomd_amd.kernels as KK
_orig = KK._abi.PairArgs
def staged_only():
    a = tl["_pa"]
    a.block_size = 999
    A.check(fn(C.byref(a), C.byref(tl["geom"]), KK.ptr(tl["tiles"]), KK.ptr(prm_t), None))
    a.block_size = 0
def full_only():
    a = tl["_pa"]
    A.check(fn(C.byref(a), C.byref(tl["geom"]), KK.ptr(tl["tiles"]), KK.ptr(prm_t), None))
fn = A.lib.hmd_compute_pair_forces_lj_tiles
prm_t = torch.tensor([[1.0, 4.0]], dtype=torch.float64, device="cuda")
rc_t = torch.tensor([[6.25]], dtype=torch.float64, device="cuda")
force = torch.zeros((N, 4), dtype=torch.float64, device="cuda")
pa = A.PairArgs()
pa.d_force, pa.N, pa.n_max, pa.d_pos, pa.box = KK.ptr(force), N, N, KK.ptr(pos_t), gbox
pa.d_n_neigh, pa.d_head_list, pa.d_rcutsq, pa.ntypes = KK.ptr(tl["n_neigh"]), KK.ptr(tl["head"]), KK.ptr(rc_t), 1
pa.size_neigh_list = tl["tiles"].numel()
tl["_pa"] = pa
def variant(code):
    def f():
        a = tl["_pa"]; a.block_size = code
        A.check(fn(C.byref(a), C.byref(tl["geom"]), KK.ptr(tl["tiles"]), KK.ptr(prm_t), None))
        a.block_size = 0
    return f
cpos = K.tile_refresh(pos_t, tl)
print("refresh us", timeit(lambda: A.check(A.lib.hmd_tile_refresh(KK.ptr(pos_t), KK.ptr(tl["slot_of"]), KK.ptr(cpos), N, None)), 10))
gfast = A.TileGeom.from_buffer_copy(tl["geom"]); gfast.d_cpos = KK.ptr(cpos); gfast.d_slot_meta = KK.ptr(tl["slot_meta"])
def fast_only():
    A.check(fn(C.byref(tl["_pa"]), C.byref(gfast), KK.ptr(tl["tiles"]), KK.ptr(prm_t), None))
print("tile force FAST staging us", timeit(fast_only, 10))
print("tile force (no wrapper) us", timeit(full_only, 10))
print("row force us", timeit(lambda: K.pair_forces("lj", pos_t, N, gbox, nl["n_neigh"], nl["nlist"], nl["head"], P, [[6.25]], compute_virial=False)))
F1, _ = K.pair_forces_tiles("lj", pos_t, N, gbox, tl, P, [[6.25]], compute_virial=False)
F2, _ = K.pair_forces("lj", pos_t, N, gbox, nl["n_neigh"], nl["nlist"], nl["head"], P, [[6.25]], compute_virial=False)
print("max |dF|", float((F1 - F2).abs().max()), "max |F|", float(F2.abs().max()))
print("tiles_to_rows us", timeit(lambda: K.tiles_to_rows(tl, N, gbox)))
