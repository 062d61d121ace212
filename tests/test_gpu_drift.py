"""NVE energy conservation over 1e5 steps on the GPU against the CPU path over the same 1e5 steps from the same
initial configuration (tests/golden/nve_drift_cpu_1728.npz, written by tests/golden/make_drift_fixture.py with the
oracle).  The two trajectories diverge (chaos, different summation order), so what is compared is the energy
error itself: the final drift and the largest excursion, per particle."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
from test_gpu_md import hoomd, make_sim  # noqa: E402,F401

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_nve_drift_over_1e5_steps_within_the_cpu_paths(hoomd):
    d = np.load(os.path.join(G, "nve_drift_cpu_1728.npz"))
    N = len(d["position"])
    snap = hoomd.Snapshot()
    snap.configuration.box = [d["L"][0], d["L"][1], d["L"][2], 0, 0, 0]
    snap.particles.N = N
    snap.particles.types = ["A"]
    snap.particles.position[:] = d["position"]
    snap.particles.velocity[:] = d["velocity"]
    sim, cell, lj, _ = make_sim(hoomd, snap, r_cut=float(d["r_cut"]), buffer=float(d["r_buff"]), dt=float(d["dt"]), mode="shift", sorter=True)
    thermo = hoomd.md.compute.ThermodynamicProperties(hoomd.filter.All())
    sim.operations.computes.append(thermo)
    sim.run(0)
    e_cpu = d["energies"].sum(axis=1)
    e = [thermo.kinetic_energy + thermo.potential_energy]
    assert np.isclose(e[0], e_cpu[0], rtol=1e-10)
    every = int(d["every"])
    for _ in range(len(e_cpu) - 1):
        sim.run(every)
        e.append(thermo.kinetic_energy + thermo.potential_energy)
    assert sim.timestep == 100000
    e = np.array(e)
    drift_gpu, drift_cpu = abs(e[-1] - e[0]) / N, abs(e_cpu[-1] - e_cpu[0]) / N
    worst_gpu, worst_cpu = np.abs(e - e[0]).max() / N, np.abs(e_cpu - e_cpu[0]).max() / N
    print(f"drift per particle over 1e5 steps: GPU {drift_gpu:.3e}, CPU {drift_cpu:.3e}; largest excursion GPU {worst_gpu:.3e}, CPU {worst_cpu:.3e}; "
          f"rebuilds GPU {cell.num_builds}, CPU {int(d['num_builds'])}")
    # the bound: twice the CPU path's figure plus a margin of 2e-4 kT per particle
    assert drift_gpu < 2.0 * drift_cpu + 2e-4
    assert worst_gpu < 2.0 * worst_cpu + 2e-4
    # the rebuild cadence is a property of the dynamics, not of the trajectory: within a few percent
    assert abs(cell.num_builds - int(d["num_builds"])) < 0.05 * int(d["num_builds"])
    assert abs(np.array(sim.operations.integrator.linear_momentum)).max() < 1e-7
