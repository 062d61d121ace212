"""Timing of the tile pair-force kernel alone on the equilibrated liquid of the headline benchmark.
    python tests/dbg/pft_bench.py --make-liquid     # runs the real simulation once, saves /tmp/liquid_pos.npy
    HOOMD_AMD_LIB=.../variants/x/libhoomd_amd.so python tests/dbg/pft_bench.py [--build]   # times one library build
"""
import ctypes as C
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hoomd-blue_amd"), os.path.join(ROOT, "scripts"), os.path.join(ROOT, "tests")]
import torch
n = (128, 128, 64)
SAVE = "/tmp/liquid_pos.npy"
if "--make-liquid" in sys.argv:
    import hoomd_amd as hoomd
    import lj_liquid
    from hoomd_amd.comm import _DevMem
    sim, cell, lj, method = lj_liquid.make_simulation(hoomd, n, thermostat=hoomd.md.methods.thermostats.Berendsen(kT=1.5, tau=0.5))
    sim.run(1300)
    method.thermostat = None
    sim.run(203)  # three steps after a sort: the order in memory is the sorter's
    pd = sim._sysdef.getParticleData()
    pos = torch.as_tensor(_DevMem(pd.pos_ptr, (pd.N, 4), "<f8"), device="cuda").cpu().numpy().copy()
    np.save(SAVE, pos)
    print("saved", pos.shape, "builds", cell.num_builds)
    sys.exit(0)
from hoomd_amd import kernels as K, _abi as A
from oracle import oracle as orc
pos = np.load(SAVE)
N = len(pos)
a = (1 / 0.84) ** (1 / 3)
L = np.array(n) * a
gbox, obox = A.Box.from_L(*L), orc.Box.from_L(*L)
pos_t = torch.from_numpy(pos).cuda()
dim = orc.cell_dimensions(obox, 2.9, 1, (0, 0, 0), 3)
cl = K.cell_list_auto(pos_t, N, 0, gbox, dim, want_idx=True)
adj = torch.as_tensor(orc.cell_adj(dim, 1, 3).view(np.int32), device="cuda")
tl = K.nlist_tiles(pos_t, N, gbox, [[2.5]], 0.4, cl, adj)
cap = int(tl["overflow"][1].item())
nn = tl["n_neigh"].cpu().numpy()
print("lib", os.path.basename(os.path.dirname(A.LIB_PATH)), "cell Nmax", cl["Nmax"], "capacity", cap, "mean nn", nn.mean(), "Nmax", tl["Nmax"].tolist())
fn = A.lib.hmd_compute_pair_forces_lj_tiles
prm_t = torch.tensor([[1.0, 4.0]], dtype=torch.float64, device="cuda")
rc_t = torch.tensor([[6.25]], dtype=torch.float64, device="cuda")
force = torch.zeros((N, 4), dtype=torch.float64, device="cuda")
pa = A.PairArgs()
pa.d_force, pa.N, pa.n_max, pa.d_pos, pa.box = K.ptr(force), N, N, K.ptr(pos_t), gbox
pa.d_n_neigh, pa.d_head_list, pa.d_rcutsq, pa.ntypes = K.ptr(tl["n_neigh"]), K.ptr(tl["head"]), K.ptr(rc_t), 1
cpos = K.tile_refresh(pos_t, tl)
gfast = A.TileGeom.from_buffer_copy(tl["geom"])
gfast.d_cpos, gfast.d_slot_meta, gfast.capacity = K.ptr(cpos), K.ptr(tl["slot_meta"]), cap
def timeit(f, reps=30):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
print("tile force us %.1f" % timeit(lambda: A.check(fn(C.byref(pa), C.byref(gfast), K.ptr(tl["tiles"]), K.ptr(prm_t), None))), "fsum", float(force[:, :3].abs().sum()))
if "--build" in sys.argv:
    def build():
        K.nlist_tiles(pos_t, N, gbox, [[2.5]], 0.4, cl, adj)
    print("tile build (head list + allocations + build) us %.1f" % timeit(build, 5))
