"""First time step at which a run with the half steps inside the force kernel differs from one without."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hoomd-blue_amd"), os.path.join(ROOT, "scripts"), os.path.join(ROOT, "tests")]
import hoomd_amd as hoomd
import test_gpu_md as T
snap = T.jittered_lattice_snapshot(hoomd, 18, jitter=0.05, seed=12)
sims = []
for rider in (True, False):
    sim, cell, lj, _ = T.make_sim(hoomd, snap, sorter=False)
    cell.deterministic = True
    sim.run(0)
    sim.operations.integrator._cpp_obj.step_rider = rider
    sims.append((sim, cell, lj))
chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for k in range(20):
    for sim, cell, lj in sims:
        sim.run(chunk)
    a, b = sims[0][0].state.get_snapshot(), sims[1][0].state.get_snapshot()
    dp = np.abs(a.particles.position - b.particles.position).max()
    dv = np.abs(a.particles.velocity - b.particles.velocity).max()
    df = np.abs(sims[0][2].forces - sims[1][2].forces).max()
    print("step", sims[0][0].timestep, "dpos %.3e dvel %.3e dF %.3e" % (dp, dv, df), "builds", sims[0][1].num_builds, sims[1][1].num_builds,
          "rider steps", sims[0][0].operations.integrator._cpp_obj.num_rider_steps)
