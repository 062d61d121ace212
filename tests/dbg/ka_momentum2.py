"""Momentum of the Kob-Andersen mixture step by step at the start of a run (NVE)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hoomd-blue_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "scripts")]
import hoomd_amd as hoomd
import lj_liquid
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
sim, cell, lj, method = lj_liquid.make_ka_simulation(hoomd, (n, n, n), kT=1.0, thermostat=None)
snap = sim.state.get_snapshot()
print("start sum v", np.abs(snap.particles.velocity.sum(0)).max())
prev = None
for step in range(40):
    sim.run(1)
    snap = sim.state.get_snapshot()
    v = snap.particles.velocity
    F = lj.forces
    print(step, "sum v", np.abs(v.sum(0)).max(), "sum F", np.abs(F.sum(0)).max(), "builds", cell.num_builds, flush=True)
