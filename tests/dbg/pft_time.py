"""Timing of the tile force kernel alone (fast staging, exact capacity) on a jittered lattice."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hoomd-blue_amd"), os.path.join(ROOT, "tests")]
import ctypes as C
import torch
from hoomd_amd import kernels as K, _abi as A
from oracle import oracle as orc
n = (128, 128, 64)
a = (1 / 0.84) ** (1 / 3)
L = np.array(n) * a
g = [(np.arange(k) + 0.5) * a - l / 2 for k, l in zip(n, L)]
z, y, x = np.meshgrid(g[2], g[1], g[0], indexing="ij")
xyz = np.stack([x.ravel(), y.ravel(), z.ravel()], 1) + np.random.default_rng(0).normal(0, 0.12, (x.size, 3))
xyz = (xyz + L / 2) % L - L / 2
# order particles by 4x4x4-site bricks so that neighbors in space are neighbors in memory (as the sorter does)
ijk = np.stack(np.meshgrid(np.arange(n[2]), np.arange(n[1]), np.arange(n[0]), indexing="ij"), -1).reshape(-1, 3)
key = ((ijk[:, 0] // 4) * (n[1] // 4) + ijk[:, 1] // 4) * (n[0] // 4) + ijk[:, 2] // 4
xyz = xyz[np.argsort(key, kind="stable")]
N = len(xyz)
gbox, obox = A.Box.from_L(*L), orc.Box.from_L(*L)
pos_t = torch.from_numpy(orc.pack_pos(xyz, np.zeros(N, int))).cuda()
dim = orc.cell_dimensions(obox, 2.9, 1, (0, 0, 0), 3)
cl = K.cell_list_auto(pos_t, N, 0, gbox, dim, want_idx=True)
adj = torch.as_tensor(orc.cell_adj(dim, 1, 3).view(np.int32), device="cuda")
tl = K.nlist_tiles(pos_t, N, gbox, [[2.5]], 0.4, cl, adj)
cap = int(tl["overflow"][1].item())
print("cell Nmax", cl["Nmax"], "capacity", cap, "mean nn", float(tl["n_neigh"].float().mean()), "Nmax", tl["Nmax"].tolist())
fn = A.lib.hmd_compute_pair_forces_lj_tiles
prm_t = torch.tensor([[4.0, 4.0]], dtype=torch.float64, device="cuda")
rc_t = torch.tensor([[6.25]], dtype=torch.float64, device="cuda")
force = torch.zeros((N, 4), dtype=torch.float64, device="cuda")
pa = A.PairArgs()
pa.d_force, pa.N, pa.n_max, pa.d_pos, pa.box = K.ptr(force), N, N, K.ptr(pos_t), gbox
pa.d_n_neigh, pa.d_head_list, pa.d_rcutsq, pa.ntypes = K.ptr(tl["n_neigh"]), K.ptr(tl["head"]), K.ptr(rc_t), 1
pa.size_neigh_list = tl["tiles"].numel()
cpos = K.tile_refresh(pos_t, tl)
gfast = A.TileGeom.from_buffer_copy(tl["geom"])
gfast.d_cpos, gfast.d_slot_meta, gfast.capacity = K.ptr(cpos), K.ptr(tl["slot_meta"]), cap
for tpp in (0,):
    pa.threads_per_particle = tpp
    for _ in range(3):
        A.check(fn(C.byref(pa), C.byref(gfast), K.ptr(tl["tiles"]), K.ptr(prm_t), None))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        A.check(fn(C.byref(pa), C.byref(gfast), K.ptr(tl["tiles"]), K.ptr(prm_t), None))
    e1.record()
    torch.cuda.synchronize()
    print("tpp", tpp, "tile force us", e0.elapsed_time(e1) / 20 * 1e3)
