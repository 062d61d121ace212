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
if "--step" in sys.argv:
    # the same launch with the integration pass riding on it (positions and velocities move on; the forces, computed
    # from the slot-ordered copy that is not swapped here, stay the same)
    vel = torch.cat([torch.randn((N, 3), dtype=torch.float64, device="cuda"), torch.ones((N, 1), dtype=torch.float64, device="cuda")], 1).contiguous()
    image = torch.zeros((N, 3), dtype=torch.int32, device="cuda")
    cpos_new = torch.zeros_like(cpos)
    res = torch.zeros(1, dtype=torch.int32, device="cuda")
    rcm = torch.tensor([2.5], dtype=torch.float64, device="cuda")
    pos_w = pos_t.clone()
    st = A.FusedStepArgs()
    st.two.d_vel, st.two.d_net_force, st.two.group_size, st.two.deltaT, st.two.rescale_factor = K.ptr(vel), K.ptr(force), N, 1e-5, 1.0
    st.one.d_pos, st.one.d_vel, st.one.d_image = K.ptr(pos_w), K.ptr(vel), K.ptr(image)
    st.one.group_size, st.one.box, st.one.rescale_factor, st.one.deltaT = N, gbox, 1.0, 1e-5
    c = st.check
    c.d_result, c.d_last_pos, c.d_pos, c.N, c.box = K.ptr(res), K.ptr(tl["last_pos"]), K.ptr(pos_w), N, gbox
    c.d_rcut_max, c.r_buff, c.ntypes, c.lambda_min, c.lam, c.checkn = K.ptr(rcm), 0.4, 1, 1.0, K._s3((1.0, 1.0, 1.0)), 3
    st.d_slot_of, st.d_cpos = K.ptr(tl["slot_of"]), K.ptr(cpos_new)
    fs = A.lib.hmd_compute_pair_forces_lj_tiles_step
    print("tile force + integration pass us %.1f" % timeit(lambda: A.check(fs(C.byref(pa), C.byref(gfast), K.ptr(tl["tiles"]), K.ptr(prm_t), C.byref(st), None))),
          "flag", int(res.item()))
    a2 = A.FusedStepArgs.from_buffer_copy(st)
    a2.d_cpos = K.ptr(cpos_new)
    print("integration pass alone us %.1f" % timeit(lambda: A.check(A.lib.hmd_nve_step_two_one_check(C.byref(a2), None))))
if "--build" in sys.argv:
    def build():
        K.nlist_tiles(pos_t, N, gbox, [[2.5]], 0.4, cl, adj)
    print("tile build (head list + allocations + build) us %.1f" % timeit(build, 5))
