"""Timing of the tile neighbor-list build alone on an equilibrated liquid (positions taken from a short run), launch by launch:
does the kernel's duration scatter on liquid data as it does inside the simulation?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hoomd-blue_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "scripts")]
import ctypes as C
import torch
import hoomd_amd as hoomd
from hoomd_amd import kernels as K, _abi as A
from oracle import oracle as orc
import lj_liquid
n = (128, 128, 64)
sim, cell, lj, method = lj_liquid.make_simulation(hoomd, n, thermostat=hoomd.md.methods.thermostats.Berendsen(kT=1.5, tau=0.5))
sim.run(int(os.environ.get("EQUIL", "3000")))
snap = sim.state.get_snapshot()
xyz = np.array(snap.particles.position)
L = np.array(snap.configuration.box[:3])
xyz = (xyz + L / 2) % L - L / 2
del sim, cell, lj, method
# order particles by cell bricks so that neighbors in space are neighbors in memory (the sorter's job in a run)
w = L / np.array([46, 46, 23])
ijk = np.floor((xyz + L / 2) / w).astype(np.int64)
key = ((ijk[:, 2] // 2) * 23 + ijk[:, 1] // 2) * 23 + ijk[:, 0] // 2
xyz = xyz[np.argsort(key, kind="stable")]
N = len(xyz)
gbox, obox = A.Box.from_L(*L), orc.Box.from_L(*L)
pos_t = torch.from_numpy(orc.pack_pos(xyz, np.zeros(N, int))).cuda()
dim = orc.cell_dimensions(obox, 2.9, 1, (0, 0, 0), 3)
cl = K.cell_list_auto(pos_t, N, 0, gbox, dim, want_idx=True)
adj = torch.as_tensor(orc.cell_adj(dim, 1, 3).view(np.int32), device="cuda")
tl = K.nlist_tiles(pos_t, N, gbox, [[2.5]], 0.4, cl, adj)
print("dim", dim, "cell Nmax", cl["Nmax"], "capacity", int(tl["overflow"][1].item()), "mean nn", float(tl["n_neigh"].float().mean()), "Nmax", tl["Nmax"].tolist())
slot_of = torch.zeros(N, dtype=torch.int32, device="cuda")
cpos = torch.zeros((tl["slot_meta"].shape[0], 4), dtype=torch.float64, device="cuda")
tl["args"].d_slot_of = slot_of.data_ptr()
tl["args"].geom.d_cpos = cpos.data_ptr()
for rep in range(3):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(201)]
    ev[0].record()
    for i in range(200):
        A.check(A.lib.hmd_compute_nlist_tiles(C.byref(tl["args"]), None))
        ev[i + 1].record()
    torch.cuda.synchronize()
    d = np.array([ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(200)])
    print("median %4.0f  mean %4.0f  min %4.0f max %4.0f  above 470 us: %3d" % (np.median(d), d.mean(), d.min(), d.max(), (d > 470).sum()), flush=True)
print([round(x) for x in d[:60]])
# the same with the device idle for a while before every launch (the host waits for the previous launch, then spins)
import time
for gap_us in (0, 20, 50, 100, 300, 1000):
    d = []
    for i in range(120):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        while (time.perf_counter() - t0) * 1e6 < gap_us:
            pass
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        A.check(A.lib.hmd_compute_nlist_tiles(C.byref(tl["args"]), None))
        e1.record()
        torch.cuda.synchronize()
        d.append(e0.elapsed_time(e1) * 1e3)
    d = np.array(d)
    print("idle %5d us before each launch: median %4.0f  mean %4.0f  min %4.0f max %4.0f  above 470 us: %3d" % (gap_us, np.median(d), d.mean(), d.min(), d.max(), (d > 470).sum()), flush=True)
