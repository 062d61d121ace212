"""End-to-end GPU tests through the Python API (the mirror of hoomd.md.nlist.Cell / pair.LJ /
Integrator / methods.ConstantVolume), checked against the CPU oracle and the reference's own
test recipes.  Tolerances: forces/energies 1e-6 relative (as the project states), neighbor
pair sets exact."""
import json
import os
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "scripts"))
G = os.path.join(HERE, "golden")

from oracle import oracle as orc  # noqa: E402


@pytest.fixture(scope="module")
def hoomd():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a MI355X")
    import hoomd_amd
    return hoomd_amd


def two_particle_snapshot(hoomd, d=1.0, L=20.0, types=("A",)):
    """hoomd/conftest.py:166-200."""
    s = hoomd.Snapshot()
    s.configuration.box = [L, L, L, 0, 0, 0]
    s.particles.N = 2
    s.particles.types = list(types)
    s.particles.position[:] = [[-d / 2, 0, 0.1], [d / 2, 0, 0.1]]
    return s


def jittered_lattice_snapshot(hoomd, n, jitter=0.1, seed=3, kT=1.5, ntypes=1):
    import lj_liquid
    snap = lj_liquid.make_snapshot(hoomd, n, kT=kT, seed=seed)
    rng = np.random.default_rng(seed)
    snap.particles.position[:] += rng.normal(0, jitter, snap.particles.position.shape)
    if ntypes > 1:
        snap.particles.types = ["A", "B"][:ntypes]
        snap.particles.typeid[:] = rng.integers(0, ntypes, snap.particles.N)
    snap.wrap()
    return snap


def make_sim(hoomd, snap, r_cut=2.5, buffer=0.4, dt=0.005, mode="none", sorter=False, thermostat=None, params=None,
             tail_correction=False, seed=7):
    sim = hoomd.Simulation(device=hoomd.device.GPU(), seed=seed)
    sim.create_state_from_snapshot(snap)
    cell = hoomd.md.nlist.Cell(buffer=buffer)
    lj = hoomd.md.pair.LJ(nlist=cell, default_r_cut=r_cut, mode=mode, tail_correction=tail_correction)
    types = snap.particles.types
    for i, a in enumerate(types):
        for b in types[i:]:
            lj.params[(a, b)] = (params or {}).get((a, b), dict(epsilon=1.0, sigma=1.0))
    method = hoomd.md.methods.ConstantVolume(filter=hoomd.filter.All(), thermostat=thermostat)
    sim.operations.integrator = hoomd.md.Integrator(dt=dt, methods=[method], forces=[lj])
    if not sorter:
        sim.operations.tuners.clear()
    return sim, cell, lj, method


def oracle_inputs(snap):
    pos = orc.pack_pos(snap.particles.position, snap.particles.typeid)
    vel = np.concatenate([snap.particles.velocity, snap.particles.mass[:, None]], axis=1)
    L = snap.configuration.box[:3]
    return pos, vel, L


def test_force_energy_accuracy_reference_values(hoomd):
    """hoomd/md/pytest/test_potential.py:1008-1044 with the LJ rows of forces_and_energies.json."""
    with open(os.path.join(G, "lj_forces_and_energies.json")) as f:
        fe = json.load(f)
    for params, forces, energies in zip(fe["LJ"]["params"], fe["LJ"]["forces"], fe["LJ"]["energies"]):
        snap = two_particle_snapshot(hoomd, d=0.75)
        sim, cell, lj, _ = make_sim(hoomd, snap, params={("A", "A"): params})
        sim.run(0)
        for d, f_ref, e_ref in zip(fe["distances"], forces, energies):
            s = sim.state.get_snapshot()
            s.particles.position[0] = [0, 0, 0.1]
            s.particles.position[1] = [0, 0, d + 0.1]
            sim.state.set_snapshot(s)
            assert np.isclose(lj.energies.sum(), e_ref, rtol=5e-6, atol=1e-6 if e_ref == 0 else 0)
            assert np.allclose(lj.forces[0], np.array([0, 0, f_ref]), rtol=1e-5, atol=1e-8)
            assert np.isclose(lj.energy, e_ref, rtol=5e-6, atol=1e-6 if e_ref == 0 else 0)


def test_pair_list_of_the_7x7x7_lattice(hoomd):
    """hoomd/md/pytest/test_nlist.py:221-372 (true_pair_list.json)."""
    with open(os.path.join(G, "true_pair_list.json")) as f:
        truth = {tuple(sorted(p)) for p in json.load(f)}
    g = np.arange(-3.5, 3.5) + 0.5
    x, y, z = np.meshgrid(g, g, g)
    snap = hoomd.Snapshot()
    snap.configuration.box = [7, 7, 7, 0, 0, 0]
    snap.particles.N = 343
    snap.particles.types = ["A"]
    snap.particles.position[:] = np.vstack((x.flatten(), y.flatten(), z.flatten())).T
    sim, cell, lj, _ = make_sim(hoomd, snap, r_cut=1.1, buffer=0.0)
    sim.run(0)
    assert {tuple(p) for p in cell.pair_list.tolist()} == truth
    local = cell.local_pair_list
    assert len(local) == 1029
    arrays = cell.cpu_local_nlist_arrays
    assert np.all(arrays["n_neigh"] == 6) and not arrays["half_nlist"]
    assert cell.dimensions == (6, 6, 6)
    # the same buffers on the device, zero-copy (hoomd/md/pytest/test_nlist.py:430-573: gpu_local_nlist_arrays)
    import torch
    with cell.gpu_local_nlist_arrays as data:
        assert not data.half_nlist
        # (unsigned 64- and 32-bit buffers, seen through signed views of the same width for the index arithmetic)
        head = torch.as_tensor(data.head_list, device="cuda").view(torch.int64)
        nn = torch.as_tensor(data.n_neigh, device="cuda").view(torch.int32)
        nl = torch.as_tensor(data.nlist, device="cuda").view(torch.int32)
        assert head.data_ptr() == cell._cpp_obj.head_list_ptr and nl.data_ptr() == cell._cpp_obj.nlist_ptr
        assert head.shape == (343,) and nn.shape == (343,) and nl.numel() >= int((head[-1] + nn[-1].to(head.dtype)).item())
        assert np.array_equal(head.cpu().numpy().view(np.uint64), arrays["head_list"])
        assert np.array_equal(nn.cpu().numpy().view(np.uint32), arrays["n_neigh"])
        # pairs (i < j) built on the device from the three buffers are the reference's list
        i = torch.repeat_interleave(torch.arange(343, device="cuda"), nn.to(torch.int64))
        offs = torch.arange(i.numel(), device="cuda") - torch.repeat_interleave(torch.cumsum(nn.to(torch.int64), 0) - nn.to(torch.int64), nn.to(torch.int64))
        j = nl.to(torch.int64)[head.to(torch.int64)[i] + offs]
        pairs = {(int(a), int(b)) for a, b in zip(i.tolist(), j.tolist()) if a < b}
        assert pairs == {tuple(sorted(p)) for p in local.tolist()}
        with pytest.raises(RuntimeError):
            cell.gpu_local_nlist_arrays
    with pytest.raises(RuntimeError):
        data.head_list  # the views die with the block


@pytest.mark.parametrize("mode", ["none", "shift", "xplor"])
def test_config_A_forces_and_pairs_match_oracle(hoomd, mode):
    """Configuration A: 40^3 lattice (N = 64,000), rho 0.84, r_cut 2.5, buffer 0.4."""
    snap = jittered_lattice_snapshot(hoomd, 40 if mode == "none" else 16)
    N = snap.particles.N
    sim, cell, lj, _ = make_sim(hoomd, snap, mode=mode)
    if mode == "xplor":
        lj.r_on[("A", "A")] = 2.0
    sim.run(0)
    pos, vel, L = oracle_inputs(snap)
    obox = orc.Box.from_L(*L)
    half = orc.nlist_binned(pos, N, obox, [[2.5]], 0.4, storage_full=False)
    shift = {"none": orc.SHIFT_NONE, "shift": orc.SHIFT_SHIFT, "xplor": orc.SHIFT_XPLOR}[mode]
    F_ref, V_ref = orc.pair_forces(orc.EV_LJ, pos, N, obox, half["n_neigh"], half["nlist"], half["head"],
                                   [orc.lj_params(1.0, 1.0)], [[6.25]], [[4.0]], shift_mode=shift, third_law=True)
    F = lj.forces
    scale = np.abs(F_ref[:, :3]).max()
    assert np.abs(F - F_ref[:, :3]).max() / scale < 1e-6 * 1e-4
    assert np.allclose(lj.energies, F_ref[:, 3], rtol=1e-9, atol=1e-9)
    assert np.isclose(lj.energy, F_ref[:, 3].sum(), rtol=1e-10)
    V = lj.virials
    assert np.abs(V - V_ref.T).max() / np.abs(V_ref).max() < 1e-10
    if mode == "none":
        want = set()
        for i, s in enumerate(orc.neighbor_sets(half["n_neigh"], half["nlist"], half["head"])):
            want.update((i, int(j)) for j in s)
        assert {tuple(p) for p in cell.pair_list.tolist()} == want
        assert cell.dimensions == tuple(orc.cell_dimensions(obox, 2.9))


def test_short_trajectory_and_rebuild_cadence_match_oracle(hoomd):
    snap = jittered_lattice_snapshot(hoomd, 20, jitter=0.05)
    sim, cell, lj, _ = make_sim(hoomd, snap)
    pos, vel, L = oracle_inputs(snap)
    ref = orc.Sim(pos, vel, L, r_cut=2.5, r_buff=0.4, dt=0.005)
    sim.run(60)
    ref.run(60)
    p_ref, v_ref, img_ref, nf_ref = ref.get()
    s = sim.state.get_snapshot()
    assert np.abs(s.particles.position - p_ref[:, :3]).max() < 1e-9
    assert np.abs(s.particles.velocity - v_ref[:, :3]).max() < 1e-8
    assert np.array_equal(s.particles.image, img_ref)
    # the acceleration array is current at the end of a run although fused passes do not store it (unit mass)
    assert np.abs(s.particles.acceleration - nf_ref[:, :3]).max() < 1e-7 * max(1.0, np.abs(nf_ref[:, :3]).max())
    st = ref.stats()
    assert cell.num_builds == st["num_builds"] and cell.shortest_rebuild == st["shortest_rebuild"]
    assert sim.timestep == 60


@pytest.mark.parametrize("sorter", [False, True])
def test_half_steps_inside_the_force_kernel_change_no_bit(hoomd, sorter):
    """An NVE run with one pair force takes its half steps inside the force kernel (one kernel per time step); with the
    switch off the integration pass runs as a kernel of its own.  Same forces, same arithmetic: the trajectories agree
    bit for bit, through rebuilds, sorts and runs cut into pieces, and so do the rebuild statistics."""
    snap = jittered_lattice_snapshot(hoomd, 18, jitter=0.05, seed=12)
    out = []
    for rider in (True, False):
        sim, cell, lj, _ = make_sim(hoomd, snap, sorter=sorter)
        cell.deterministic = True  # (the order of a cell's members, and with it the order of the sums, is the same in both runs)
        if sorter:
            sim.operations.tuners[0].trigger = 7
        sim.run(0)
        sim.operations.integrator._cpp_obj.step_rider = rider
        sim.run(33)
        sim.run(1)
        sim.run(26)
        s = sim.state.get_snapshot()
        n_ride = sim.operations.integrator._cpp_obj.num_rider_steps
        assert (n_ride > 30) if rider else (n_ride == 0)
        out.append((s.particles.position.copy(), s.particles.velocity.copy(), s.particles.image.copy(), lj.forces.copy(), lj.energies.copy(),
                    cell.num_builds, cell.shortest_rebuild, s.particles.acceleration.copy()))
    for x, y in zip(out[0], out[1]):
        assert np.array_equal(x, y)
    assert out[0][5] >= 6  # rebuilds happened


def test_nve_energy_drift_not_worse_than_cpu_path(hoomd):
    snap = jittered_lattice_snapshot(hoomd, 16, jitter=0.05, kT=1.0)
    N = snap.particles.N
    sim, cell, lj, _ = make_sim(hoomd, snap, mode="shift", sorter=True)
    sim.operations.tuners[0].trigger = 100
    thermo = hoomd.md.compute.ThermodynamicProperties(hoomd.filter.All())
    sim.operations.computes.append(thermo)
    pos, vel, L = oracle_inputs(snap)
    ref = orc.Sim(pos, vel, L, r_cut=2.5, r_buff=0.4, dt=0.005, shift_mode=orc.SHIFT_SHIFT)
    sim.run(0)
    ref.run(0)
    e0 = thermo.kinetic_energy + thermo.potential_energy
    ke, pe = ref.energies()
    assert np.isclose(e0, ke + pe, rtol=1e-10)
    sim.run(1500)
    ref.run(1500)
    e1 = thermo.kinetic_energy + thermo.potential_energy
    ke1, pe1 = ref.energies()
    drift_gpu = abs(e1 - e0) / N
    drift_cpu = abs((ke1 + pe1) - (ke + pe)) / N
    assert drift_gpu < 2e-3
    assert drift_gpu < 2.0 * drift_cpu + 2e-4
    assert abs(np.array(sim.operations.integrator.linear_momentum)).max() < 1e-8
    assert np.isclose(thermo.degrees_of_freedom, 3 * N - 3)


def test_particle_sorter_keeps_the_physics(hoomd):
    snap = jittered_lattice_snapshot(hoomd, 14, jitter=0.05, seed=9)
    rng = np.random.default_rng(0)
    perm = rng.permutation(snap.particles.N)
    snap.particles.position[:] = snap.particles.position[perm]
    snap.particles.velocity[:] = snap.particles.velocity[perm]
    a, cell_a, lj_a, _ = make_sim(hoomd, snap, sorter=False)
    b, cell_b, lj_b, _ = make_sim(hoomd, snap, sorter=True)
    b.operations.tuners[0].trigger = 5
    a.run(0)
    b.run(0)
    assert np.allclose(lj_a.forces, lj_b.forces, rtol=1e-11, atol=1e-10)
    a.run(12)
    b.run(12)
    sa, sb = a.state.get_snapshot(), b.state.get_snapshot()
    assert np.abs(sa.particles.position - sb.particles.position).max() < 1e-10
    assert np.allclose(lj_a.forces, lj_b.forces, rtol=1e-8, atol=1e-8)
    # the lists were built at different steps, so they differ in the buffer shell only: both hold every pair inside r_cut
    x = sa.particles.position
    d = x[:, None, :] - x[None, :, :]
    d -= np.round(d / sa.configuration.box[0]) * sa.configuration.box[0]
    i, j = np.nonzero(np.triu((d * d).sum(-1) <= 2.5**2, 1))
    inside = set(zip(i.tolist(), j.tolist()))
    assert inside <= {tuple(p) for p in cell_a.pair_list.tolist()}
    assert inside <= {tuple(p) for p in cell_b.pair_list.tolist()}
    assert cell_b.num_builds > cell_a.num_builds  # every sort forces a rebuild


def test_two_types_per_pair_tables(hoomd):
    snap = jittered_lattice_snapshot(hoomd, 12, ntypes=2, seed=21)
    N = snap.particles.N
    params = {("A", "A"): dict(epsilon=1.0, sigma=1.0), ("A", "B"): dict(epsilon=1.5, sigma=0.8),
              ("B", "B"): dict(epsilon=0.5, sigma=0.88)}
    sim, cell, lj, _ = make_sim(hoomd, snap, params=params, sorter=True)
    lj.r_cut[("A", "B")] = 2.0
    lj.r_cut[("B", "B")] = 2.2
    sim.run(0)
    assert lj.params[("B", "A")]["epsilon"] == pytest.approx(1.5) and lj.r_cut[("A", "B")] == 2.0
    pos, vel, L = oracle_inputs(snap)
    obox = orc.Box.from_L(*L)
    r_cut = [[2.5, 2.0], [2.0, 2.2]]
    P = [orc.lj_params(1.0, 1.0), orc.lj_params(0.8, 1.5), orc.lj_params(0.8, 1.5), orc.lj_params(0.88, 0.5)]
    half = orc.nlist_binned(pos, N, obox, r_cut, 0.4, storage_full=False)
    F_ref, _ = orc.pair_forces(orc.EV_LJ, pos, N, obox, half["n_neigh"], half["nlist"], half["head"], P, np.square(r_cut),
                               third_law=True, compute_virial=False)
    assert np.abs(lj.forces - F_ref[:, :3]).max() / np.abs(F_ref[:, :3]).max() < 1e-10
    sim.run(20)
    assert np.isfinite(lj.forces).all()


def test_tail_correction_values(hoomd):
    """hoomd/md/pytest/test_potential.py:1202-1286."""
    sigma, epsilon, d_pair, r_cut = 1.0, 0.5, 1.5, 2.0
    v1 = np.array([0.46168675, -0.21020661, 0.21240303])
    res = {}
    for tc in (True, False):
        snap = two_particle_snapshot(hoomd, d=d_pair)
        snap.particles.velocity[0] = v1
        snap.particles.velocity[1] = -v1
        sim, cell, lj, _ = make_sim(hoomd, snap, r_cut=r_cut, tail_correction=tc, params={("A", "A"): dict(sigma=sigma, epsilon=epsilon)})
        sim.always_compute_pressure = True
        th = hoomd.md.compute.ThermodynamicQuantities(filter=hoomd.filter.All())
        sim.operations.computes.append(th)
        sim.run(0)
        res[tc] = (th.potential_energy, th.pressure, sim)
    N, volume = 2, 20.0**3
    rho = N / volume
    lj1, lj2 = 4 * epsilon * sigma**12, 4 * epsilon * sigma**6
    dE = 2 * N * np.pi * rho * (lj1 / 9 / r_cut**9 - lj2 / 3 / r_cut**3)
    dP = 4 / 6 * rho**2 * np.pi * (lj1 * 4 / 3 / r_cut**9 - lj2 * 2 / r_cut**3)
    e_lj = 4 * epsilon * ((sigma / d_pair)**12 - (sigma / d_pair)**6)
    ke = 0.5 * (v1 @ v1) * 2
    f = 24 * epsilon / d_pair * (2 * (sigma / d_pair)**12 - (sigma / d_pair)**6)
    p = (2 * ke / 3 + f * d_pair / 3) / volume
    np.testing.assert_allclose(res[False][0], e_lj)
    np.testing.assert_allclose(res[True][0], e_lj + dE)
    np.testing.assert_allclose(res[False][1], p)
    np.testing.assert_allclose(res[True][1], p + dP)
    # a shifted potential cannot carry a tail correction (test_potential.py:1165-1187)
    snap = two_particle_snapshot(hoomd, d=d_pair)
    sim, cell, lj, _ = make_sim(hoomd, snap, mode="shift", tail_correction=True)
    with pytest.raises(RuntimeError):
        sim.run(0)


def test_thermostats(hoomd):
    snap = jittered_lattice_snapshot(hoomd, 12, jitter=0.05, kT=0.8)
    for make in (lambda: hoomd.md.methods.thermostats.Berendsen(kT=1.6, tau=0.1), lambda: hoomd.md.methods.thermostats.Bussi(kT=1.6)):
        sim, cell, lj, method = make_sim(hoomd, snap, thermostat=make())
        th = hoomd.md.compute.ThermodynamicProperties(hoomd.filter.All())
        sim.operations.computes.append(th)
        sim.run(600)
        temps = []
        for _ in range(10):
            sim.run(20)
            temps.append(th.kinetic_temperature)
        assert abs(np.mean(temps) - 1.6) < 0.08
        method.thermostat = None
        sim.run(10)


def test_bussi_factor_worked_out_on_the_device_changes_no_bit(hoomd, monkeypatch):
    """With the Bussi thermostat the kinetic energy stays on the device: the second half step takes the sums along, the
    first half step of the next time step divides the random target energy by them, and the host never waits.  The same
    arithmetic as the path through the host (HOOMD_AMD_NO_DEFERRED_THERMOSTAT): trajectories and logged energies agree
    bit for bit, through rebuilds and runs cut into pieces."""
    snap = jittered_lattice_snapshot(hoomd, 14, jitter=0.05, kT=0.8, seed=3)
    out = []
    for host_path in (False, True):
        if host_path:
            monkeypatch.setenv("HOOMD_AMD_NO_DEFERRED_THERMOSTAT", "1")
        else:
            monkeypatch.delenv("HOOMD_AMD_NO_DEFERRED_THERMOSTAT", raising=False)
        sim, cell, lj, method = make_sim(hoomd, snap, thermostat=hoomd.md.methods.thermostats.Bussi(kT=1.6))
        cell.deterministic = True  # (the order of a cell's members, and with it the order of the force sums, is the same in both runs)
        th = hoomd.md.compute.ThermodynamicProperties(hoomd.filter.All())
        sim.operations.computes.append(th)
        sim.run(41)
        ke_mid = th.kinetic_energy
        sim.run(1)
        sim.run(30)
        s = sim.state.get_snapshot()
        m = method._cpp_obj
        assert (m.num_deferred_factors == 0 and m.num_steps_with_sums == 0) if host_path else (m.num_deferred_factors == 72 and m.num_steps_with_sums >= 65)
        out.append((s.particles.position.copy(), s.particles.velocity.copy(), s.particles.image.copy(), lj.forces.copy(), np.array(ke_mid),
                    np.array(th.kinetic_energy), np.array(th.potential_energy), cell.num_builds))
    for x, y in zip(out[0], out[1]):
        assert np.array_equal(x, y)
    assert out[0][7] >= 4 and abs(2 * out[0][5] / (3 * snap.particles.N - 3) - 1.6) < 0.15


def test_api_errors_and_state(hoomd):
    snap = two_particle_snapshot(hoomd, d=1.2, types=("A", "B"))
    sim = hoomd.Simulation(device=hoomd.device.GPU(), seed=1)
    cell = hoomd.md.nlist.Cell(buffer=0.4)
    lj = hoomd.md.pair.LJ(nlist=cell, default_r_cut=2.5)
    with pytest.raises(hoomd.error.DataAccessError):
        lj.forces
    with pytest.raises(hoomd.error.DataAccessError):
        cell.num_builds
    with pytest.raises(RuntimeError):
        sim.run(1)  # no state yet
    sim.create_state_from_snapshot(snap)
    with pytest.raises(RuntimeError):
        sim.create_state_from_snapshot(snap)
    lj.params[("A", "A")] = dict(epsilon=1, sigma=1)
    sim.operations.integrator = hoomd.md.Integrator(dt=0.005, methods=[hoomd.md.methods.ConstantVolume(hoomd.filter.All())], forces=[lj])
    with pytest.raises(hoomd.error.IncompleteSpecificationError):
        sim.run(0)  # (A,B) and (B,B) are missing
    lj.params[(["A", "B"], "B")] = dict(epsilon=2, sigma=1.1)
    sim.run(0)
    assert lj.params[("A", "B")]["sigma"] == pytest.approx(1.1)
    with pytest.raises(KeyError):
        lj.params[("A", "C")] = dict(epsilon=1, sigma=1)
    with pytest.raises(ValueError):
        lj.mode = "bogus"
    with pytest.raises(NotImplementedError):
        hoomd.device.CPU()
    assert sim.state.N_particles == 2 and sim.state.particle_types == ["A", "B"]
    assert sim.state.box == hoomd.Box.cube(20.0)
    cell.buffer = 0.5
    assert cell.buffer == 0.5
    sim.run(3)
    assert sim.timestep == 3 and sim.tps > 0
    sim.state.thermalize_particle_momenta(hoomd.filter.All(), kT=1.5)
    s = sim.state.get_snapshot()
    assert np.abs(s.particles.velocity.sum(0)).max() < 1e-12


def test_nan_and_out_of_box_are_reported(hoomd):
    snap = two_particle_snapshot(hoomd, d=1.0)
    sim, cell, lj, _ = make_sim(hoomd, snap)
    sim.run(0)
    s = sim.state.get_snapshot()
    s.particles.position[1] = [30.0, 0, 0]
    sim.state.set_snapshot(s)
    with pytest.raises(RuntimeError, match="no longer in the simulation box"):
        sim.run(1)
    s.particles.position[1] = [np.nan, 0, 0]
    sim.state.set_snapshot(s)
    with pytest.raises(RuntimeError, match="NaN"):
        sim.run(1)
    # box too small for the cutoff (hoomd/md/NeighborList.cc:535-561)
    small = two_particle_snapshot(hoomd, d=1.0, L=5.0)
    sim2, _, _, _ = make_sim(hoomd, small)
    with pytest.raises(RuntimeError, match="too small"):
        sim2.run(0)


@pytest.mark.parametrize("distance", np.linspace(0.1, 2.0, 3))
@pytest.mark.parametrize("k", [0.5, 2.0, 5.0])
@pytest.mark.parametrize("mode", ["none", "shift"])
def test_example_plugin_force_and_energy(hoomd, distance, k, mode):
    """example_plugins/pair_plugin/pytest/test_example_pair.py:15-81 (all three sigmas per case)."""
    from hoomd_amd import pair_plugin
    for sigma in (0.5, 1.0, 1.5):
        sim = hoomd.Simulation(device=hoomd.device.GPU(), seed=1)
        sim.create_state_from_snapshot(two_particle_snapshot(hoomd, d=distance))
        cell = hoomd.md.nlist.Cell(buffer=0.4)
        ex = pair_plugin.pair.ExamplePair(cell, default_r_cut=sigma, mode=mode)
        ex.params[("A", "A")] = dict(k=k, sigma=sigma)
        sim.operations.integrator = hoomd.md.Integrator(dt=0.001, methods=[hoomd.md.methods.ConstantVolume(hoomd.filter.All())], forces=[ex])
        sim.run(0)
        snap = sim.state.get_snapshot()
        dx = snap.particles.position[1] - snap.particles.position[0]
        dr = np.linalg.norm(dx)
        if dr >= sigma:
            f, e = np.zeros(3), 0.0
        else:
            f = k * (sigma - dr) * dx / dr
            e = 0.5 * k * (sigma - dr)**2  # the shift 0.5 k (r_cut - sigma)^2 vanishes at r_cut = sigma
        np.testing.assert_array_almost_equal(ex.forces, [-f, f], decimal=6)
        np.testing.assert_array_almost_equal(ex.energies, [e / 2, e / 2], decimal=6)
        assert ex.params[("A", "A")] == dict(k=k, sigma=sigma)


def test_autotuner_interface_answers(hoomd):
    """hoomd/operation.py:386-476: kernel_parameters / is_tuning_complete / tune_kernel_parameters exist on every
    operation; before attaching they raise as the reference does."""
    snap = jittered_lattice_snapshot(hoomd, 10, jitter=0.05)
    sim, cell, lj, method = make_sim(hoomd, snap)
    with pytest.raises(hoomd.error.DataAccessError):
        lj.kernel_parameters
    with pytest.raises(RuntimeError):
        cell.tune_kernel_parameters()
    sim.run(0)
    F0 = lj.forces
    for op in (cell, lj, method, sim.operations.integrator):
        assert op.is_tuning_complete is True
        assert isinstance(op.kernel_parameters, dict)
        op.tune_kernel_parameters()
        op.kernel_parameters = op.kernel_parameters   # saving and restoring the tuned values round-trips
    assert lj.kernel_parameters["pair_force"] == (256, 0)
    lj.kernel_parameters = {"pair_force": (256, 16)}
    assert lj.kernel_parameters["pair_force"] == (256, 16) and lj._cpp_obj.threads_per_particle == 16
    sim.run(1)
    sim2, cell2, lj2, _ = make_sim(hoomd, snap)
    sim2.run(1)
    assert np.abs(lj.forces - lj2.forces).max() < 1e-9 * max(1.0, np.abs(F0).max())
    with pytest.raises(ValueError):
        lj.kernel_parameters = {"pair_force": (128, 8)}
    with pytest.raises(ValueError):
        cell.kernel_parameters = {"no_such_kernel": (1,)}


@pytest.mark.parametrize("mode", ["none", "shift", "xplor"])
def test_compute_energy_between_sets(hoomd, mode):
    """hoomd/md/pair/pair.py:90-124: even against odd tags; the two halves of the total energy by set."""
    snap = jittered_lattice_snapshot(hoomd, 8, jitter=0.08, seed=4)
    N = snap.particles.N
    sim, cell, lj, _ = make_sim(hoomd, snap, mode=mode)
    if mode == "xplor":
        lj.r_on[("A", "A")] = 2.0
    with pytest.raises(hoomd.error.DataAccessError):
        lj.compute_energy([0], [1])
    sim.run(0)
    tags = np.arange(N, dtype=np.int32)
    even, odd = tags[0::2], tags[1::2]
    u_eo = lj.compute_energy(even, odd)
    # brute force over all even-odd pairs: minimum image, LJ with the mode's energy modification
    # (hoomd/md/PotentialPair.h:912-976, hoomd/md/EvaluatorPairLJ.h:165-207)
    pos, _, L = oracle_inputs(snap)

    def pair_energy(ti, tj):
        d = pos[ti, None, :3] - pos[None, tj, :3]
        d -= np.rint(d / L) * L
        rsq = (d * d).sum(-1)
        r6 = 1.0 / rsq**3
        e = 4.0 * (r6 * r6 - r6)
        rc2, ron2 = 6.25, 4.0
        if mode == "shift":
            e -= 4.0 * (1.0 / rc2**6 - 1.0 / rc2**3)
        if mode == "xplor":
            sw = (rsq - rc2) ** 2 * (rc2 + 2 * rsq - 3 * ron2) / (rc2 - ron2) ** 3
            e = np.where(rsq >= ron2, e * sw, e)
        return np.where(rsq < rc2, e, 0.0)

    want = pair_energy(even, odd).sum()
    assert np.isclose(u_eo, want, rtol=1e-11)
    assert np.isclose(lj.compute_energy(odd, even), u_eo, rtol=1e-12)
    assert lj.compute_energy([], odd) == 0.0
    # one particle against everybody else = twice its per-particle energy
    i = 17
    others = np.delete(tags, i)
    assert np.isclose(lj.compute_energy([i], others), 2 * lj.energies[i], rtol=1e-10)
    assert np.isclose(lj.compute_energy([i], others), pair_energy(np.array([i]), others).sum(), rtol=1e-11)
