import sys, os, time, json
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "hoomd-blue_amd"))
sys.path.insert(0, os.path.dirname(__file__))
import numpy as np
import hoomd_amd as hoomd
from hoomd_amd import _hoomd_amd as cpp
import lj_liquid

n = tuple(int(x) for x in sys.argv[1].split("x")) if len(sys.argv) > 1 else (128, 128, 64)
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 500
period = int(sys.argv[3]) if len(sys.argv) > 3 else 200
tpp = int(sys.argv[4]) if len(sys.argv) > 4 else 0
t0 = time.time()
sim, cell, lj, method = lj_liquid.make_simulation(hoomd, n, thermostat=hoomd.md.methods.thermostats.Berendsen(kT=1.5, tau=0.5), sorter_period=period)
print("setup", time.time() - t0, flush=True)
sim.run(0)
lj._cpp_obj.threads_per_particle = tpp
print("run0", time.time() - t0, "dim", cell.dimensions, "Nmax cell", cell.allocated_particles_per_cell, "nlist Nmax", cell._cpp_obj.getNmaxPerType(), flush=True)
sim.run(steps)
print("melt tps", sim.tps, "builds", cell.num_builds, "shortest", cell.shortest_rebuild, flush=True)
method.thermostat = None
thermo = hoomd.md.compute.ThermodynamicProperties(hoomd.filter.All())
sim.operations.computes.append(thermo)
sim.always_compute_pressure = False
sim.run(0)
cpp.enable_kernel_timing(True)
cpp.reset_kernel_timing()
cell._cpp_obj.resetStats()
sim.run(steps)
tim = cpp.kernel_timing()
cpp.enable_kernel_timing(False)
N = sim.state.N_particles
print("NVE tps (with event timing)", sim.tps, "builds", cell.num_builds, "shortest", cell.shortest_rebuild)
for k, (c, ms) in sorted(tim.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:16s} n={c:6d} total={ms:9.2f} ms avg={1000*ms/c:9.1f} us")
sim.run(steps)
print("NVE tps", sim.tps, "particle-steps/s", sim.tps * N, flush=True)
print("KE/N", thermo.kinetic_energy / N, "PE/N", thermo.potential_energy / N, "T", thermo.kinetic_temperature)
