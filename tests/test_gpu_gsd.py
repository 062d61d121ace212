"""hoomd.write.GSD / Simulation.create_state_from_gsd mirrors on a running simulation
(hoomd/write/gsd.py, hoomd/simulation.py:180-220)."""
import os
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "scripts"))


@pytest.fixture(scope="module")
def hoomd():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a MI355X")
    import hoomd_amd
    return hoomd_amd


def test_trajectory_writer_and_restart(hoomd, tmp_path):
    import lj_liquid
    from hoomd_amd.gsd_file import GSDFile
    sim, cell, wca, fene, _ = lj_liquid.make_kg_simulation(hoomd, (16, 8, 8), chain_length=8, sorter_period=20)
    fn = str(tmp_path / "traj.gsd")
    w = hoomd.write.GSD(trigger=hoomd.trigger.Periodic(10), filename=fn, mode="wb", dynamic=["property", "momentum"])
    sim.operations.writers.append(w)
    sim.run(35, write_at_start=True)
    assert sim.timestep == 35
    with GSDFile(fn, "r") as f:
        assert f.nframes == 4
        assert [int(f.read_chunk(k, "configuration/step")[0]) for k in range(4)] == [0, 10, 20, 30]
        assert "bonds/group" in f.chunk_names(0) and "bonds/group" not in f.chunk_names(1)
        assert "particles/velocity" in f.chunk_names(3)
    # a one-frame file of the present state, then a new simulation from it
    one = str(tmp_path / "restart.gsd")
    hoomd.write.GSD.write(sim.state, one)
    snap = sim.state.get_snapshot()
    sim2 = hoomd.Simulation(device=hoomd.device.GPU(), seed=3)
    sim2.create_state_from_gsd(one)
    assert sim2.timestep == 35
    s2 = sim2.state.get_snapshot()
    assert s2.particles.N == snap.particles.N and s2.bonds.N == snap.bonds.N
    assert np.array_equal(s2.bonds.group, snap.bonds.group) and s2.bonds.types == ["polymer"]
    assert np.abs(s2.particles.position - snap.particles.position).max() < 1e-5      # 32-bit on disk
    assert np.abs(s2.particles.velocity - snap.particles.velocity).max() < 1e-5
    assert np.array_equal(s2.particles.image, snap.particles.image)
    # the restarted system runs with the same forces: bonds and exclusions came through the file
    cell2 = hoomd.md.nlist.Cell(buffer=0.4)
    wca2 = hoomd.md.pair.LJ(nlist=cell2, default_r_cut=2.0 ** (1.0 / 6.0), mode="shift")
    wca2.params[("A", "A")] = dict(epsilon=1.0, sigma=1.0)
    fene2 = hoomd.md.bond.FENEWCA()
    fene2.params["polymer"] = dict(k=30.0, r0=1.5, epsilon=1.0, sigma=1.0, delta=0.0)
    sim2.operations.integrator = hoomd.md.Integrator(dt=0.005, methods=[hoomd.md.methods.ConstantVolume(filter=hoomd.filter.All())],
                                                     forces=[wca2, fene2])
    sim2.run(0)
    assert np.abs(fene2.forces - fene.forces).max() < 2e-2 * np.abs(fene.forces).max()
    assert cell2._cpp_obj.getNumExclusions() == snap.bonds.N
    sim2.run(10)
    assert sim2.timestep == 45
    # frames can be picked by number; the last frame of the trajectory holds step 30
    sim3 = hoomd.Simulation(device=hoomd.device.GPU(), seed=3)
    sim3.create_state_from_gsd(fn, frame=-1)
    assert sim3.timestep == 30 and sim3.state.get_snapshot().bonds.N == snap.bonds.N
