"""Bonded path through the Python API (mirror of hoomd.md.bond.FENEWCA + nlist exclusions): the
reference's own bond test recipe (hoomd/md/pytest/test_bond.py) and a small Kremer-Grest melt checked
against the CPU oracle."""
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


def bonded_pair_snapshot(hoomd, d=0.969, L=5.0):
    """hoomd/md/pytest/test_bond.py:100-112."""
    s = hoomd.Snapshot()
    s.configuration.box = [L, L, L, 0, 0, 0]
    s.particles.N = 2
    s.particles.types = ["A"]
    s.particles.position[:] = [[-d / 2, 0, 0.1], [d / 2, 0, 0.1]]
    s.bonds.N = 1
    s.bonds.types = ["A-A"]
    s.bonds.typeid[0] = 0
    s.bonds.group[0] = (0, 1)
    return s


def test_reference_bond_recipe(hoomd):
    with open(os.path.join(G, "bond_fenewca_cases.json")) as f:
        cases = json.load(f)
    for c in cases:
        sim = hoomd.Simulation(device=hoomd.device.GPU(), seed=1)
        sim.create_state_from_snapshot(bonded_pair_snapshot(hoomd, c["distance"]))
        potential = hoomd.md.bond.FENEWCA()
        potential.params["A-A"] = c["params"]
        sim.operations.integrator = hoomd.md.Integrator(dt=0.005, forces=[potential])
        sim.run(0)
        for key, v in c["params"].items():
            assert potential.params["A-A"][key] == pytest.approx(v)
        assert sum(potential.energies) == pytest.approx(c["energy"], rel=1e-2)
        np.testing.assert_allclose(potential.forces[0], [c["force"], 0.0, 0.0], rtol=1e-2, atol=1e-5)
        np.testing.assert_allclose(potential.forces[1], [-c["force"], 0.0, 0.0], rtol=1e-2, atol=1e-5)
        snap = sim.state.get_snapshot()
        assert snap.bonds.N == 1 and snap.bonds.types == ["A-A"] and tuple(snap.bonds.group[0]) == (0, 1)


def test_api_errors(hoomd):
    sim = hoomd.Simulation(device=hoomd.device.GPU(), seed=1)
    sim.create_state_from_snapshot(bonded_pair_snapshot(hoomd))
    potential = hoomd.md.bond.FENEWCA()
    with pytest.raises(ValueError):
        potential.params["A-A"] = dict(k=1.0)
    sim.operations.integrator = hoomd.md.Integrator(dt=0.005, forces=[potential])
    with pytest.raises(hoomd.error.IncompleteSpecificationError):
        sim.run(0)
    # a bond stretched past r0: the reference raises "bond out of bounds"
    sim2 = hoomd.Simulation(device=hoomd.device.GPU(), seed=1)
    sim2.create_state_from_snapshot(bonded_pair_snapshot(hoomd, d=1.7))
    p2 = hoomd.md.bond.FENEWCA()
    p2.params["A-A"] = dict(k=30.0, r0=1.5, epsilon=1.0, sigma=1.0, delta=0.0)
    sim2.operations.integrator = hoomd.md.Integrator(dt=0.005, forces=[p2])
    with pytest.raises(RuntimeError, match="out of bounds"):
        sim2.run(1)


def wca_brute_force(x, L, bonded, r_cut):
    """Shifted LJ over all pairs that are not bonded (minimum image, orthorhombic)."""
    N = len(x)
    F = np.zeros((N, 3))
    E = 0.0
    ecut = 4.0 * (r_cut ** -12 - r_cut ** -6)
    for i in range(N):
        d = x[i] - x
        d -= np.round(d / L) * L
        r2 = (d * d).sum(1)
        m = (r2 < r_cut * r_cut) & (np.arange(N) != i)
        m[list(bonded[i])] = False
        r2i = 1.0 / r2[m]
        r6i = r2i ** 3
        F[i] = (d[m] * (r2i * r6i * (48.0 * r6i - 24.0))[:, None]).sum(0)
        E += 0.5 * (4.0 * r6i * (r6i - 1.0) - ecut).sum()
    return F, E


@pytest.mark.parametrize("tiles", [False, True])
def test_kremer_grest_melt_forces_exclusions_and_energy_conservation(hoomd, tiles):
    import lj_liquid
    n, chain = (16, 10, 10), 8
    sim, cell, wca, fene, _ = lj_liquid.make_kg_simulation(hoomd, n, chain_length=chain, sorter_period=20)
    thermo = hoomd.md.compute.ThermodynamicProperties(hoomd.filter.All())
    sim.operations.computes.append(thermo)
    if tiles:
        # about three particles per cell: index rows by default; the tile path must give the same (both builds leave
        # the excluded partners out themselves)
        cell._attach(sim)
        cell._cpp_obj.tile_min_occupancy = 0.0
    sim.run(0)
    snap = sim.state.get_snapshot()
    N = snap.particles.N
    assert snap.bonds.N == N // chain * (chain - 1)
    assert cell._cpp_obj.getNumExclusions() == snap.bonds.N
    assert cell._cpp_obj.tiles_active == tiles
    L = np.array(snap.configuration.box[:3])
    bonded = [set() for _ in range(N)]
    for a, b in snap.bonds.group:
        bonded[a].add(int(b))
        bonded[b].add(int(a))
    # no bonded pair is in the neighbor list; every other pair inside the cutoff is
    pairs = {tuple(p) for p in cell.pair_list.tolist()}
    assert not any((min(a, b), max(a, b)) in pairs for a, b in snap.bonds.group.tolist())
    x = snap.particles.position
    r_cut = 2.0 ** (1.0 / 6.0)
    F_pair, E_pair = wca_brute_force(x, L, bonded, r_cut)
    assert np.abs(wca.forces - F_pair).max() <= 1e-6 * np.abs(F_pair).max()
    assert wca.energy == pytest.approx(E_pair, rel=1e-6)
    pos = orc.pack_pos(x, snap.particles.typeid)
    F_bond, _ = orc.bond_forces_fene(pos, np.arange(N), N, orc.Box.from_L(*L), snap.bonds.group, snap.bonds.typeid,
                                     [orc.fene_params(30.0, 1.5, 1.0, 1.0, 0.0)])
    assert np.abs(fene.forces - F_bond[:, :3]).max() <= 1e-6 * np.abs(F_bond[:, :3]).max()
    assert fene.energy == pytest.approx(F_bond[:, 3].sum(), rel=1e-6)
    # NVE with the sorter reordering particles every 20 steps: the bond table follows the order
    e0 = thermo.kinetic_energy + thermo.potential_energy
    assert thermo.potential_energy == pytest.approx(E_pair + F_bond[:, 3].sum(), rel=1e-6)
    sim.run(400)
    e1 = thermo.kinetic_energy + thermo.potential_energy
    assert abs(e1 - e0) / N < 2e-3
    snap1 = sim.state.get_snapshot()
    d = snap1.particles.position[snap1.bonds.group[:, 0]] - snap1.particles.position[snap1.bonds.group[:, 1]]
    d -= np.round(d / L) * L
    assert np.sqrt((d * d).sum(1)).max() < 1.5
    F_bond1, _ = orc.bond_forces_fene(orc.pack_pos(snap1.particles.position, snap1.particles.typeid), np.arange(N), N,
                                      orc.Box.from_L(*L), snap1.bonds.group, snap1.bonds.typeid, [orc.fene_params(30.0, 1.5, 1.0, 1.0, 0.0)])
    assert np.abs(fene.forces - F_bond1[:, :3]).max() <= 1e-6 * np.abs(F_bond1[:, :3]).max()
