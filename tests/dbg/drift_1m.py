"""NVE energy drift of the headline liquid at its full size: N = 1,048,576, mode="shift" (the potential is continuous at
the cutoff, so the total energy is what the integrator conserves), 1e5 steps from the equilibrated state, total energy
every 2000 steps.      python tests/dbg/drift_1m.py [steps]"""
import os
import sys
import time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "hoomd-blue_amd"), os.path.join(ROOT, "scripts")]
import hoomd_amd as hoomd
import lj_liquid

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
n = (128, 128, 64)
sim, cell, lj, method = lj_liquid.make_simulation(hoomd, n, thermostat=hoomd.md.methods.thermostats.Berendsen(kT=1.5, tau=0.5), mode="shift")
thermo = hoomd.md.compute.ThermodynamicProperties(hoomd.filter.All())
sim.operations.computes.append(thermo)
sim.run(10000)
method.thermostat = None
sim.run(1000)
N = 128 * 128 * 64
e = [thermo.kinetic_energy + thermo.potential_energy]
b0, t0 = cell.num_builds, time.time()
every = 2000
for k in range(steps // every):
    sim.run(every)
    e.append(thermo.kinetic_energy + thermo.potential_energy)
    if k % 10 == 9:
        print(f"step {every * (k + 1):7d}  E/N {e[-1] / N:+.9f}  drift/N {(e[-1] - e[0]) / N:+.3e}  kT {2 * thermo.kinetic_energy / (3 * N - 3):.5f}", flush=True)
dt = time.time() - t0
e = np.array(e)
print(f"N = {N}, {steps} NVE steps in {dt:.1f} s ({steps / dt:.0f} TPS with the energy read every {every} steps), rebuilds {cell.num_builds - b0}")
print(f"drift per particle: final {abs(e[-1] - e[0]) / N:.3e}, largest excursion {np.abs(e - e[0]).max() / N:.3e}, "
      f"rms fluctuation about the line fit {np.std(e - np.polyval(np.polyfit(np.arange(len(e)), e, 1), np.arange(len(e)))) / N:.3e}")
