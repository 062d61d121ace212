"""Timing of the tile neighbor-list build alone on an SFC-like ordered jittered lattice."""
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
# order particles by 4x4x4-site bricks so that neighbors in space are neighbors in memory
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
print("dim", dim, "cell Nmax", cl["Nmax"], "capacity", int(tl["overflow"][1].item()), "mean nn", float(tl["n_neigh"].float().mean()), "Nmax", tl["Nmax"].tolist())
for _ in range(2):
    A.check(A.lib.hmd_compute_nlist_tiles(C.byref(tl["args"]), None))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    A.check(A.lib.hmd_compute_nlist_tiles(C.byref(tl["args"]), None))
e1.record()
torch.cuda.synchronize()
print("tile build us", e0.elapsed_time(e1) / 10 * 1e3)
# the same with the slot map and the slot-ordered position copy written by the build
slot_of = torch.zeros(N, dtype=torch.int32, device="cuda")
cpos = torch.zeros((tl["slot_meta"].shape[0], 4), dtype=torch.float64, device="cuda")
tl["args"].d_slot_of = slot_of.data_ptr()
tl["args"].geom.d_cpos = cpos.data_ptr()
for _ in range(2):
    A.check(A.lib.hmd_compute_nlist_tiles(C.byref(tl["args"]), None))
torch.cuda.synchronize()
e0.record()
for _ in range(10):
    A.check(A.lib.hmd_compute_nlist_tiles(C.byref(tl["args"]), None))
e1.record()
torch.cuda.synchronize()
print("tile build with slot map + positions us", e0.elapsed_time(e1) / 10 * 1e3)
# launch by launch: on some boxes of the pool the duration alternates between two values
ev = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
ev[0].record()
for i in range(40):
    A.check(A.lib.hmd_compute_nlist_tiles(C.byref(tl["args"]), None))
    ev[i + 1].record()
torch.cuda.synchronize()
print("per launch us", [round(ev[i].elapsed_time(ev[i + 1]) * 1e3) for i in range(40)])
# a long series (BUILD_SERIES launches, back to back): does the duration drift or jump once the device has been busy for seconds?
n_series = int(os.environ.get("BUILD_SERIES", "0"))
if n_series:
    import subprocess
    chunk = 200
    for c0 in range(0, n_series, chunk):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(chunk + 1)]
        ev[0].record()
        for i in range(chunk):
            A.check(A.lib.hmd_compute_nlist_tiles(C.byref(tl["args"]), None))
            ev[i + 1].record()
        torch.cuda.synchronize()
        d = np.array([ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(chunk)])
        print("launches %5d..%5d  median %4.0f  mean %4.0f  max %4.0f  above 470 us: %3d" % (c0, c0 + chunk, np.median(d), d.mean(), d.max(), (d > 470).sum()), flush=True)
    try:
        print(subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp"], capture_output=True, text=True, timeout=20).stdout[-1500:])
    except Exception as e:
        print("rocm-smi:", e)
