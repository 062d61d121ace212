"""Momentum of the Kob-Andersen mixture over a run: total force and total momentum after every chunk of steps."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hoomd-blue_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "scripts")]
import hoomd_amd as hoomd
import lj_liquid
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
bussi = len(sys.argv) > 2 and sys.argv[2] == "bussi"
sim, cell, lj, method = lj_liquid.make_ka_simulation(hoomd, (n, n, n), kT=1.0, thermostat=hoomd.md.methods.thermostats.Bussi(kT=1.0) if bussi else None)
if bussi:
    thermo = hoomd.md.compute.ThermodynamicProperties(hoomd.filter.All())
    sim.operations.computes.append(thermo)
for chunk in range(12):
    if bussi and chunk >= 6:
        for _ in range(50):
            sim.run(1)
            ke = thermo.kinetic_energy
    else:
        sim.run(50)
    snap = sim.state.get_snapshot()
    v = snap.particles.velocity
    F = lj.forces
    nn = cell._cpp_obj.getNNeighArrayHost() if hasattr(cell._cpp_obj, "getNNeighArrayHost") else None
    print(chunk, "sum v", np.abs(v.sum(0)).max(), "sum F", np.abs(F.sum(0)).max(), "max |F|", np.abs(F).max(), "builds", cell.num_builds, flush=True)
