"""CPU oracle vs the reference's own arithmetic and fixtures (no GPU needed).

* ref_header_vectors.json was printed by a program that compiles the reference's headers where
  they lie (oracle/ref_harness.cc) -> bit-exact comparison.
* lj_forces_and_energies.json / true_pair_list.json are the reference's own data files.
* reference_cases.json holds the hand-built cases of the reference's C++ unit tests.
"""
import json
import os

import numpy as np
import pytest

from oracle import oracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))
G = os.path.join(HERE, "golden")


def fh(x):
    return float.fromhex(x)


def v3(x):
    return np.array([fh(a) for a in x])


@pytest.fixture(scope="module")
def vec():
    with open(os.path.join(G, "ref_header_vectors.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def cases():
    with open(os.path.join(G, "reference_cases.json")) as f:
        return json.load(f)


def test_lj_evaluator_bit_exact(vec):
    for c in vec["lj"]:
        p = orc.lj_params(fh(c["sigma"]), fh(c["epsilon"]))
        assert p == (fh(c["sigma_6"]), fh(c["epsilon_x_4"]))
        ok, f, e = orc.eval_pair(orc.EV_LJ, fh(c["rsq"]), fh(c["rcutsq"]), p, c["shift"])
        assert ok == bool(c["evaluated"])
        assert f == fh(c["force_divr"]) and e == fh(c["pair_eng"])
        le, lp = orc.eval_lrc(orc.EV_LJ, fh(c["rcutsq"]), p)
        assert le == fh(c["lrc_energy"]) and lp == fh(c["lrc_pressure"])


def test_example_evaluator_bit_exact(vec):
    for c in vec["example"]:
        ok, f, e = orc.eval_pair(orc.EV_EXAMPLE, fh(c["rsq"]), fh(c["rcutsq"]), (fh(c["k"]), fh(c["sigma"])),
                                 c["shift"])
        assert ok == bool(c["evaluated"])
        assert f == fh(c["force_divr"]) and e == fh(c["pair_eng"])


def test_fene_evaluator_bit_exact(vec):
    # hoomd/md/EvaluatorBondFENE.h compiled from the reference's header
    assert len(vec["fene"]) == 300
    for c in vec["fene"]:
        p = [float.fromhex(c[k]) for k in ("k", "r_0", "sigma_6", "epsilon_x_4", "delta")]
        ok, f, e = orc.eval_bond_fene(float.fromhex(c["rsq"]), p)
        assert ok == bool(c["evaluated"])
        if ok:
            assert f == float.fromhex(c["force_divr"]) and e == float.fromhex(c["bond_eng"])


def test_fenewca_against_reference_test_values():
    # the FENEWCA rows of hoomd/md/pytest/test_bond.py: two particles 0.969 apart along x in a box of 5
    with open(os.path.join(G, "bond_fenewca_cases.json")) as f:
        cases = json.load(f)
    assert len(cases) == 3
    for c in cases:
        d = c["distance"]
        pos = np.array([[-d / 2, 0, 0.1, 0], [d / 2, 0, 0.1, 0]], dtype=np.float64)
        box = orc.Box.from_L(5, 5, 5)
        force, virial = orc.bond_forces_fene(pos, [0, 1], 2, box, [[0, 1]], [0], [orc.fene_params(**c["params"])])
        assert force[:, 3].sum() == pytest.approx(c["energy"], rel=1e-2)
        np.testing.assert_allclose(force[0, :3], [c["force"], 0, 0], rtol=1e-2, atol=1e-5)
        np.testing.assert_allclose(force[1, :3], [-c["force"], 0, 0], rtol=1e-2, atol=1e-5)
        # virial: sum over both ends of 1/2 dx dx F/r = dx^2 F/r with dx = x1 - x0
        assert virial[0].sum() == pytest.approx(-c["force"] * d, rel=1e-2)


def test_bond_out_of_bounds_is_an_error():
    pos = np.array([[-1.0, 0, 0, 0], [1.0, 0, 0, 0]])
    with pytest.raises(RuntimeError):
        orc.bond_forces_fene(pos, [0, 1], 2, orc.Box.from_L(10, 10, 10), [[0, 1]], [0], [orc.fene_params(30.0, 1.5, 1.0, 1.0, 0.0)])


def test_box_arithmetic_bit_exact(vec):
    for b in vec["boxes"]:
        L = v3(b["L"])
        t = v3(b["tilt"])
        box = orc.Box.from_L(L[0], L[1], L[2], t[0], t[1], t[2], b["periodic"])
        assert np.array_equal(orc.nearest_plane_distance(box), v3(b["nearest_plane_distance"]))
        for c in b["cases"]:
            assert np.array_equal(orc.make_fraction(box, v3(c["frac_in"]), v3(c["ghost_width"])), v3(c["frac"]))
            assert np.array_equal(orc.min_image(box, v3(c["v"])), v3(c["min_image"]))
            w, img = orc.wrap(box, v3(c["wrap_in"]), c["wrap_img_in"])
            assert np.array_equal(w, v3(c["wrap_out"]))
            assert list(img) == c["wrap_img_out"]


def test_min_image_rint_equals_branch_form_within_one_image(vec):
    # the device form (round-to-nearest image) and the host form agree bit for bit for |v| < L
    # in orthorhombic boxes except exactly on the +L/2 face, where the tie rounds to even.
    for b in vec["boxes"]:
        L = v3(b["L"])
        t = v3(b["tilt"])
        if np.any(t != 0) or L[2] == 0:
            continue
        box = orc.Box.from_L(L[0], L[1], L[2], 0, 0, 0, b["periodic"])
        for c in b["cases"]:
            v = v3(c["v"])
            if np.any(np.abs(np.abs(v) - L / 2) < 1e-12):
                continue
            assert np.array_equal(orc.min_image(box, v, rint=True), orc.min_image(box, v))


def test_indexers(vec):
    for w, h, i, j, idx, n in vec["index2d"]:
        assert orc.index2d(w, i, j) == idx and w * h == n
    for w, h, d, i, j, k, idx, n in vec["index3d"]:
        assert orc.index3d(w, h, i, j, k) == idx and w * h * d == n


def test_lj_against_reference_mathematica_values():
    """hoomd/md/pytest/test_potential.py:1008-1044 semantics: two particles on the z axis,
    F on particle 0 = forces[i] * z_hat, sum of energies = energies[i], rtol 5e-6."""
    with open(os.path.join(G, "lj_forces_and_energies.json")) as f:
        fe = json.load(f)
    box = orc.Box.from_L(20, 20, 20)
    for params, forces, energies in zip(fe["LJ"]["params"], fe["LJ"]["forces"], fe["LJ"]["energies"]):
        for d, f_ref, e_ref in zip(fe["distances"], forces, energies):
            pos = orc.pack_pos([[0, 0, 0.1], [0, 0, d + 0.1]], [0, 0])
            nl = orc.nlist_binned(pos, 2, box, [[2.5]], 0.4, storage_full=True)
            for third_law, full in ((False, True), (True, False)):
                nl = orc.nlist_binned(pos, 2, box, [[2.5]], 0.4, storage_full=full)
                F, _ = orc.pair_forces(orc.EV_LJ, pos, 2, box, nl["n_neigh"], nl["nlist"], nl["head"],
                                       [orc.lj_params(params["sigma"], params["epsilon"])], [[2.5**2]],
                                       third_law=third_law)
                assert np.isclose(F[:, 3].sum(), e_ref, rtol=5e-6, atol=1e-6 if e_ref == 0 else 0)
                assert np.allclose(F[0, :3], [0, 0, f_ref], rtol=1e-5, atol=1e-8)
                assert np.allclose(F[1, :3], [0, 0, -f_ref], rtol=1e-5, atol=1e-8)


def _lattice(n, a):
    g = (np.arange(-n / 2, n / 2) + 0.5) * a
    # hoomd/conftest.py:203-263 lattice_snapshot_factory ordering (meshgrid 'ij' then reshape)
    x, y, z = np.meshgrid(g, g, g)
    return np.stack([x.ravel(), y.ravel(), z.ravel()], axis=1)


def test_true_pair_list():
    """hoomd/md/pytest/test_nlist.py:221-372: 7^3 simple cubic lattice, a = 1, r_cut = 1.1,
    buffer = 0: the 1029 nearest-neighbour pairs."""
    with open(os.path.join(G, "true_pair_list.json")) as f:
        truth = {tuple(sorted(p)) for p in json.load(f)}
    n, a = 7, 1.0
    xyz = _lattice(n, a)
    box = orc.Box.from_L(n * a, n * a, n * a)
    pos = orc.pack_pos(xyz, np.zeros(len(xyz), dtype=int))
    nl = orc.nlist_binned(pos, len(xyz), box, [[1.1]], 0.0, storage_full=False)
    pairs = set()
    for i, s in enumerate(orc.neighbor_sets(nl["n_neigh"], nl["nlist"], nl["head"])):
        for j in s:
            pairs.add(tuple(sorted((i, int(j)))))
    assert len(pairs) == 1029
    assert pairs == truth
    nl_full = orc.nlist_binned(pos, len(xyz), box, [[1.1]], 0.0, storage_full=True)
    assert np.all(nl_full["n_neigh"] == 6)


def test_cell_dimensions_and_adjacency(cases):
    for c in cases["cell_dimensions"]:
        box = orc.Box.from_L(*c["L"])
        assert list(orc.cell_dimensions(box, c["width"])) == c["dim"]
    for c in cases["cell_adjacency"]:
        box = orc.Box.from_L(*c["L"])
        dim = orc.cell_dimensions(box, c["width"])
        adj = orc.cell_adj(dim, c["radius"])
        assert adj.shape[1] == c["n_adj"]
        assert np.all(adj == np.arange(c["n_adj"], dtype=np.uint32)[None, :])


def test_cell_small(cases):
    c = cases["cell_small"]
    box = orc.Box.from_L(*c["L"])
    dim = orc.cell_dimensions(box, c["width"])
    assert list(dim) == c["dim"]
    pos = orc.pack_pos(c["pos"], c["type"])
    r = orc.cell_list_auto(pos, 8, 0, box, dim, flag_mode=0)
    assert r["conditions"][1:] == (0, 0)
    total = 0
    for cell in c["cells"]:
        ci = orc.index3d(dim[0], dim[1], *cell["cell"])
        sz = r["cell_size"][ci]
        assert sz == len(cell["members"])
        got = sorted(orc.unpack_int_w(r["xyzf"][ci, :sz, 3]).tolist())
        assert got == cell["members"]
        for s in range(sz):
            m = int(orc.unpack_int_w(r["xyzf"][ci, s:s + 1, 3])[0])
            assert np.array_equal(r["xyzf"][ci, s, :3], np.array(c["pos"][m], dtype=float))
        total += sz
    assert total == 8 and r["cell_size"].sum() == 8
    # charge flag + type/body
    r = orc.cell_list_auto(pos, 8, 0, box, dim, flag_mode=1, charge=c["charge"], body=c["body"],
                           want_type_body=True)
    for cell in c["cells"]:
        ci = orc.index3d(dim[0], dim[1], *cell["cell"])
        for s in range(r["cell_size"][ci]):
            q = r["xyzf"][ci, s, 3]
            m = c["charge"].index(q)
            assert m in cell["members"]
            assert list(r["type_body"][ci, s]) == [c["type"][m], c["body"][m]]


def test_cell_list_random_10000():
    """hoomd/test/test_cell_list.cc:521-569: every particle appears exactly once."""
    rng = np.random.default_rng(7)
    L = np.array([20.0, 31.0, 17.5])
    xyz = (rng.random((10000, 3)) - 0.5) * L
    box = orc.Box.from_L(*L)
    dim = orc.cell_dimensions(box, 3.0)
    r = orc.cell_list_auto(orc.pack_pos(xyz, np.zeros(10000, int)), 10000, 0, box, dim, want_idx=True)
    assert r["cell_size"].sum() == 10000
    seen = np.concatenate([r["idx"][c, :r["cell_size"][c]] for c in range(len(r["cell_size"]))])
    assert np.array_equal(np.sort(seen), np.arange(10000))


def test_cell_list_flags_nan_and_out_of_box():
    box = orc.Box.from_L(10, 10, 10)
    dim = orc.cell_dimensions(box, 2.0)
    pos = orc.pack_pos([[0, 0, 0], [np.nan, 0, 0], [0, 6.0, 0]], [0, 0, 0])
    r = orc.compute_cell_list(pos, 3, 0, box, dim, 4)
    assert r["conditions"] == (0, 2, 3)
    # an out-of-box ghost is ignored silently (hoomd/CellList.cc:464-467)
    r = orc.compute_cell_list(pos[[0, 2]], 1, 1, box, dim, 4)
    assert r["conditions"] == (0, 0, 0) and r["cell_size"].sum() == 1


def test_nlist_reference_cases(cases):
    for c in cases["nlist"]:
        box = orc.Box.from_L(*c["L"])
        pos = orc.pack_pos(c["pos"], c["type"])
        n = len(c["pos"])
        nl = orc.nlist_binned(pos, n, box, c["r_cut"], c["r_buff"], storage_full=(c["mode"] == "full"))
        sets = orc.neighbor_sets(nl["n_neigh"], nl["nlist"], nl["head"])
        if "neighbors" in c:
            for i, want in enumerate(c["neighbors"]):
                if want is not None:
                    assert sets[i].tolist() == want, (c["name"], i)
        if "n_neigh" in c:
            assert nl["n_neigh"].tolist() == c["n_neigh"], c["name"]


def test_nlist_half_is_upper_triangle_of_full():
    rng = np.random.default_rng(3)
    L = 12.0
    xyz = (rng.random((600, 3)) - 0.5) * L
    typ = rng.integers(0, 2, 600)
    box = orc.Box.from_L(L, L, L)
    pos = orc.pack_pos(xyz, typ)
    rc = [[2.0, 1.5], [1.5, -1.0]]
    full = orc.nlist_binned(pos, 600, box, rc, 0.3, storage_full=True)
    half = orc.nlist_binned(pos, 600, box, rc, 0.3, storage_full=False)
    fs = orc.neighbor_sets(full["n_neigh"], full["nlist"], full["head"])
    hs = orc.neighbor_sets(half["n_neigh"], half["nlist"], half["head"])
    for i in range(600):
        assert np.array_equal(hs[i], fs[i][fs[i] > i])
        for j in fs[i]:
            assert i in fs[int(j)]
            assert not (typ[i] == 1 and typ[j] == 1)   # r_cut <= 0 disables the pair
    # brute-force check of the full list with the same arithmetic
    for i in range(0, 600, 37):
        want = []
        for j in range(600):
            if j == i:
                continue
            r = rc[typ[i]][typ[j]]
            if r <= 0:
                continue
            d = orc.min_image(box, xyz[i] - xyz[j])
            if d[0] * d[0] + d[1] * d[1] + d[2] * d[2] <= (r + 0.3) * (r + 0.3):
                want.append(j)
        assert fs[i].tolist() == want


def test_third_law_matches_full_list_forces():
    rng = np.random.default_rng(11)
    L = 9.0
    n = 5
    g = (np.arange(n) + 0.5) * (L / n) - L / 2
    xyz = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3) + rng.normal(0, 0.08, (n**3, 3))
    N = len(xyz)
    box = orc.Box.from_L(L, L, L)
    pos = orc.pack_pos(xyz, np.zeros(N, int))
    P = [orc.lj_params(1.0, 1.0)]
    for mode in (orc.SHIFT_NONE, orc.SHIFT_SHIFT, orc.SHIFT_XPLOR):
        full = orc.nlist_binned(pos, N, box, [[2.5]], 0.4, storage_full=True)
        half = orc.nlist_binned(pos, N, box, [[2.5]], 0.4, storage_full=False)
        Ff, Vf = orc.pair_forces(orc.EV_LJ, pos, N, box, full["n_neigh"], full["nlist"], full["head"], P,
                                 [[6.25]], [[4.0]], shift_mode=mode, third_law=False)
        Fh, Vh = orc.pair_forces(orc.EV_LJ, pos, N, box, half["n_neigh"], half["nlist"], half["head"], P,
                                 [[6.25]], [[4.0]], shift_mode=mode, third_law=True)
        assert np.allclose(Ff, Fh, rtol=1e-10, atol=1e-10)
        assert np.allclose(Vf, Vh, rtol=1e-10, atol=1e-10)
        assert abs(Ff[:, :3].sum(0)).max() < 1e-9


def test_needs_updating_state_machine():
    """hoomd/md/NeighborList.cc:1325-1418."""
    s = orc.NlistState.new(0.4)
    calls = []

    def chk(result):
        def f():
            calls.append(1)
            return result
        return f

    # first call at t=0: last_checked == 0 and a forced update is pending -> true, not counted
    assert s.needs_updating(0, chk(False)) is True
    assert s.num_builds == 0 and not calls
    # again within the same step: cached result (false: nothing was recorded)
    assert s.needs_updating(0, chk(True)) is False
    # t=1: delay satisfied (1 >= 0 + 1) -> distance check runs
    assert s.needs_updating(1, chk(False)) is False and len(calls) == 1
    assert s.needs_updating(1, chk(True)) is False and len(calls) == 1    # cached within the step
    assert s.needs_updating(5, chk(True)) is True
    assert s.num_builds == 1 and s.shortest_rebuild == 5
    assert s.needs_updating(6, chk(False)) is False
    s.force_update = 1
    assert s.needs_updating(7, chk(False)) is True
    assert s.forced_updates == 1 and s.last_updated_tstep == 7
    # rebuild_check_delay: no check until last_updated + delay
    s.rebuild_check_delay = 3
    n0 = len(calls)
    assert s.needs_updating(8, chk(True)) is False and s.needs_updating(9, chk(True)) is False
    assert len(calls) == n0
    assert s.needs_updating(10, chk(True)) is True and s.dangerous_updates == 1
    # tiny buffer: always rebuild without a check
    s2 = orc.NlistState.new(0.0)
    s2.needs_updating(0, chk(False))
    n0 = len(calls)
    assert s2.needs_updating(1, chk(False)) is True and len(calls) == n0


def test_integrator_steps_and_energy_conservation():
    n = 6
    a = (1 / 0.84) ** (1 / 3)
    L = n * a
    g = (np.arange(n) + 0.5) * a - L / 2
    xyz = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    rng = np.random.default_rng(1234)
    v = rng.normal(0, np.sqrt(1.5), xyz.shape)
    v -= v.mean(0)
    N = len(xyz)
    pos = orc.pack_pos(xyz, np.zeros(N, int))
    vel = np.concatenate([v, np.ones((N, 1))], axis=1)
    sim = orc.Sim(pos, vel, (L, L, L), r_cut=2.5, r_buff=0.4, dt=0.002, shift_mode=orc.SHIFT_SHIFT)
    sim.run(0)
    ke0, pe0 = sim.energies()
    sim.run(400)
    ke1, pe1 = sim.energies()
    assert abs((ke1 + pe1) - (ke0 + pe0)) / N < 2e-3
    st = sim.stats()
    assert st["timestep"] == 400 and st["num_builds"] >= 2
    p, vv, img, nf = sim.get()
    assert np.all(np.abs(p[:, :3]) <= L / 2 + 1e-12)
    assert abs((vv[:, :3] * vv[:, 3:4]).sum(0)).max() < 1e-9     # momentum conserved
