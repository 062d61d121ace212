import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "hoomd-blue_amd"), os.path.join(ROOT, "scripts"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import hoomd_amd as hoomd
from oracle import oracle as orc
import test_gpu_md as T
snap = T.jittered_lattice_snapshot(hoomd, 14, jitter=0.08, seed=9)
rng = np.random.default_rng(0)
if len(sys.argv) > 1 and sys.argv[1] == "perm":
    perm = rng.permutation(snap.particles.N)
    snap.particles.position[:] = snap.particles.position[perm]
    snap.particles.velocity[:] = snap.particles.velocity[perm]
sim, cell, lj, _ = T.make_sim(hoomd, snap)
pos, vel, L = T.oracle_inputs(snap)
ref = orc.Sim(pos, vel, L, r_cut=2.5, r_buff=0.4, dt=0.005)
sim.run(0); ref.run(0)
for step in range(14):
    p_ref, v_ref, img_ref, nf_ref = ref.get()
    s = sim.state.get_snapshot()
    F = lj.forces
    print(step, "dpos", np.abs(s.particles.position - p_ref[:, :3]).max(), "dF", np.abs(F - nf_ref[:, :3]).max(), "Fmax", np.abs(nf_ref[:,:3]).max(),
          "builds", cell.num_builds, ref.stats()["num_builds"], "vmax", np.abs(v_ref[:,:3]).max(), flush=True)
    sim.run(1); ref.run(1)
