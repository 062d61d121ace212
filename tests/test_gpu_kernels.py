"""GPU parity tests: every HIP driver, called through the C ABI, against the CPU oracle on the
same seeded inputs.  Bit-exact for cell membership, neighbor sets, head lists, flags and the
integrator half steps; forces/virials within 1e-6 relative (asserted much tighter)."""
import json
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import oracle as orc  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
G = os.path.join(HERE, "golden")
F_RTOL = 1e-6  # tolerance the project states for forces and virials


@pytest.fixture(scope="module")
def K():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a MI355X: torch.cuda.is_available() is False")
    from hoomd_amd import kernels
    return kernels


@pytest.fixture(scope="module")
def A():
    from hoomd_amd import _abi
    return _abi


def liquid(n, rho=0.84, jitter=0.12, seed=5, ntypes=1):
    a = (1.0 / rho) ** (1.0 / 3.0)
    L = n * a
    g = (np.arange(n) + 0.5) * a - L / 2
    xyz = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    rng = np.random.default_rng(seed)
    xyz = xyz + rng.normal(0, jitter, xyz.shape)
    xyz = (xyz + L / 2) % L - L / 2
    typ = rng.integers(0, ntypes, len(xyz)) if ntypes > 1 else np.zeros(len(xyz), dtype=int)
    return xyz, typ, L


def boxes(A, L, **kw):
    if np.isscalar(L):
        L = (L, L, L)
    return A.Box.from_L(*L, **kw), orc.Box.from_L(*L, **kw)


def cell_sets(cell_size, xyzf):
    """per cell: sorted tuple of (index, x, y, z) -- order independent and bit exact"""
    out = []
    for c in range(len(cell_size)):
        sz = int(cell_size[c])
        idx = orc.unpack_int_w(xyzf[c, :sz, 3])
        rows = sorted((int(i), *map(float, xyzf[c, s, :3])) for s, i in enumerate(idx))
        out.append(tuple(rows))
    return out


def gpu_nlist(K, A, pos_t, N, gbox, obox, r_cut, r_buff, n_ghost=0, ghost_width=(0, 0, 0), body=None, filter_body=False,
              ndim=3):
    _, _, rmax, _ = orc.update_rlist(np.asarray(r_cut, dtype=float), r_buff)
    dim = orc.cell_dimensions(obox, rmax + r_buff, 1, ghost_width, ndim)
    body_t = torch.as_tensor(np.asarray(body, dtype=np.int64).astype(np.uint32).view(np.int32), device="cuda") if body is not None else None
    cl = K.cell_list_auto(pos_t, N, n_ghost, gbox, dim, ghost_width, body=body_t)
    adj = torch.as_tensor(orc.cell_adj(dim, 1, ndim).view(np.int32), device="cuda")
    nl = K.nlist_binned(pos_t, N, gbox, r_cut, r_buff, cl, adj, n_ghost, ghost_width, body=body_t, filter_body=filter_body)
    return nl, cl, dim


def sets_of(nl, N):
    return orc.neighbor_sets(nl["n_neigh"].cpu().numpy().view(np.uint32)[:N], nl["nlist"].cpu().numpy().view(np.uint32),
                             nl["head"].cpu().numpy().view(np.uint64)[:N])


# ------------------------------------------------------------------ cell list
@pytest.mark.parametrize("L,width,gw", [((20.0, 31.0, 17.5), 3.0, (0, 0, 0)), ((12.0, 12.0, 12.0), 2.9, (0, 0, 0)),
                                        ((16.0, 9.0, 11.0), 2.0, (1.5, 0, 1.5))])
def test_cell_list_matches_oracle_bit_exact(K, A, L, width, gw):
    rng = np.random.default_rng(17)
    N, n_ghost = 9000, (700 if any(gw) else 0)
    xyz = (rng.random((N, 3)) - 0.5) * np.array(L)
    periodic = tuple(0 if g else 1 for g in gw)
    if n_ghost:
        gx = (rng.random((n_ghost, 3)) - 0.5) * (np.array(L) + 2 * np.array(gw) * 1.05)  # some fall outside
        xyz = np.concatenate([xyz, gx])
    typ = rng.integers(0, 3, len(xyz))
    gbox, obox = boxes(A, L, periodic=periodic)
    dim = orc.cell_dimensions(obox, width, 1, gw)
    pos = orc.pack_pos(xyz, typ)
    ref = orc.cell_list_auto(pos, N, n_ghost, obox, dim, gw, want_type_body=True)
    got = K.cell_list_auto(torch.from_numpy(pos).cuda(), N, n_ghost, gbox, dim, gw, want_idx=True)
    assert got["Nmax"] == ref["Nmax"]
    cs = got["cell_size"].cpu().numpy().view(np.uint32)
    assert np.array_equal(cs, ref["cell_size"])
    assert cell_sets(cs, got["xyzf"].cpu().numpy()) == cell_sets(ref["cell_size"], ref["xyzf"])
    assert got["conditions"].cpu().numpy().tolist() == [0, 0, 0]
    # type_body and idx travel with the same slot
    xyzf = got["xyzf"].cpu().numpy()
    tb = got["type_body"].cpu().numpy().view(np.uint32)
    idx = got["idx"].cpu().numpy().view(np.uint32)
    for c in range(0, len(cs), 7):
        for s in range(cs[c]):
            i = int(orc.unpack_int_w(xyzf[c, s:s + 1, 3])[0])
            assert idx[c, s] == i and tb[c, s, 0] == typ[i] and tb[c, s, 1] == orc.NO_BODY


def test_cell_list_overflow_nan_and_out_of_box_flags(K, A):
    gbox, obox = boxes(A, 10.0)
    dim = orc.cell_dimensions(obox, 2.0)
    xyz = np.zeros((40, 3))
    xyz[:, 0] = np.linspace(-0.4, 0.4, 40)  # all in one cell
    pos = torch.from_numpy(orc.pack_pos(xyz, np.zeros(40, int))).cuda()
    r = K.compute_cell_list(pos, 40, 0, gbox, dim, 8)
    assert r["conditions"].cpu().numpy().tolist() == [40, 0, 0]
    assert int(r["cell_size"].sum()) == 40
    bad = orc.pack_pos([[0, 0, 0], [np.nan, 0, 0], [0, 6.0, 0]], [0, 0, 0])
    r = K.compute_cell_list(torch.from_numpy(bad).cuda(), 3, 0, gbox, dim, 4)
    assert r["conditions"].cpu().numpy().tolist() == [0, 2, 3]
    # an out-of-box ghost is ignored silently
    r = K.compute_cell_list(torch.from_numpy(bad[[0, 2]]).cuda(), 1, 1, gbox, dim, 4)
    assert r["conditions"].cpu().numpy().tolist() == [0, 0, 0] and int(r["cell_size"].sum()) == 1


def test_cell_list_flag_modes_and_reference_small_case(K, A):
    with open(os.path.join(G, "reference_cases.json")) as f:
        c = json.load(f)["cell_small"]
    gbox, obox = boxes(A, c["L"])
    dim = orc.cell_dimensions(obox, c["width"])
    assert list(dim) == c["dim"]
    pos = torch.from_numpy(orc.pack_pos(c["pos"], c["type"])).cuda()
    charge = torch.tensor(c["charge"], dtype=torch.float64).cuda()
    body = torch.tensor(c["body"], dtype=torch.int32).cuda()
    r = K.cell_list_auto(pos, 8, 0, gbox, dim, flag_charge=True, charge=charge, body=body)
    cs = r["cell_size"].cpu().numpy()
    xyzf = r["xyzf"].cpu().numpy()
    tb = r["type_body"].cpu().numpy()
    for cell in c["cells"]:
        ci = orc.index3d(dim[0], dim[1], *cell["cell"])
        assert cs[ci] == len(cell["members"])
        for s in range(cs[ci]):
            m = c["charge"].index(xyzf[ci, s, 3])
            assert m in cell["members"] and tb[ci, s].tolist() == [c["type"][m], c["body"][m]]
    r = K.cell_list_auto(pos, 8, 0, gbox, dim, flag_type=True)
    xyzf = r["xyzf"].cpu().numpy()
    ci = orc.index3d(dim[0], dim[1], 2, 2, 3)
    assert orc.unpack_int_w(xyzf[ci, :1, 3])[0] == 2  # type of particle 1


def test_sort_cell_list_orders_members_by_index(K, A):
    xyz, typ, L = liquid(14, seed=2)
    gbox, obox = boxes(A, L)
    dim = orc.cell_dimensions(obox, 2.9)
    N = len(xyz)
    perm = np.random.default_rng(0).permutation(N)
    pos = torch.from_numpy(orc.pack_pos(xyz[perm], typ)).cuda()
    cl = K.cell_list_auto(pos, N, 0, gbox, dim, want_idx=True)
    before = cell_sets(cl["cell_size"].cpu().numpy().view(np.uint32), cl["xyzf"].cpu().numpy())
    K.sort_cell_list(cl)
    cs = cl["cell_size"].cpu().numpy().view(np.uint32)
    xyzf = cl["xyzf"].cpu().numpy()
    assert cell_sets(cs, xyzf) == before
    idx = cl["idx"].cpu().numpy().view(np.uint32)
    for c in range(len(cs)):
        members = orc.unpack_int_w(xyzf[c, :cs[c], 3])
        assert np.all(np.diff(members.astype(np.int64)) > 0)
        assert np.array_equal(idx[c, :cs[c]], members)


# ------------------------------------------------------------------ neighbor list
def test_head_list_matches_oracle(K):
    rng = np.random.default_rng(1)
    for N, nt in [(1, 1), (777, 1), (5000, 3), (70000, 4)]:
        typ = rng.integers(0, nt, N)
        pos = orc.pack_pos(rng.random((N, 3)), typ)
        Nmax = rng.integers(1, 9, nt).astype(np.uint32) * 4
        head_ref, total_ref = orc.build_head_list(pos, N, Nmax)
        head, total = K.build_head_list(torch.from_numpy(pos).cuda(), N, torch.as_tensor(Nmax.view(np.int32)).cuda())
        assert total == total_ref
        assert np.array_equal(head.cpu().numpy().view(np.uint64)[:N], head_ref)


def test_nlist_reference_hand_built_cases(K, A):
    with open(os.path.join(G, "reference_cases.json")) as f:
        cases = json.load(f)["nlist"]
    for c in cases:
        if c["mode"] != "full":
            continue  # the GPU class builds full lists only (hoomd/md/NeighborListGPUBinned.cc:51-54)
        gbox, obox = boxes(A, c["L"])
        n = len(c["pos"])
        pos = torch.from_numpy(orc.pack_pos(c["pos"], c["type"])).cuda()
        nl, _, _ = gpu_nlist(K, A, pos, n, gbox, obox, c["r_cut"], c["r_buff"])
        sets = sets_of(nl, n)
        if "neighbors" in c:
            for i, want in enumerate(c["neighbors"]):
                if want is not None:
                    assert sets[i].tolist() == want, (c["name"], i)
        if "n_neigh" in c:
            assert nl["n_neigh"].cpu().numpy()[:n].tolist() == c["n_neigh"], c["name"]


def test_nlist_true_pair_list(K, A):
    with open(os.path.join(G, "true_pair_list.json")) as f:
        truth = {tuple(sorted(p)) for p in json.load(f)}
    g = (np.arange(-3.5, 3.5) + 0.5)
    xyz = np.stack([m.ravel() for m in np.meshgrid(g, g, g)], 1)
    gbox, obox = boxes(A, 7.0)
    pos = torch.from_numpy(orc.pack_pos(xyz, np.zeros(343, int))).cuda()
    nl, _, _ = gpu_nlist(K, A, pos, 343, gbox, obox, [[1.1]], 0.0)
    pairs = set()
    for i, s in enumerate(sets_of(nl, 343)):
        assert len(s) == 6
        for j in s:
            pairs.add(tuple(sorted((i, int(j)))))
    assert pairs == truth


@pytest.mark.parametrize("n,ntypes", [(12, 1), (16, 2), (40, 1)])
def test_nlist_sets_match_oracle_bit_exact(K, A, n, ntypes):
    xyz, typ, L = liquid(n, ntypes=ntypes)
    N = len(xyz)
    gbox, obox = boxes(A, L)
    r_cut = [[2.5]] if ntypes == 1 else [[2.5, 2.0], [2.0, -1.0]]
    pos = orc.pack_pos(xyz, typ)
    ref = orc.nlist_binned(pos, N, obox, r_cut, 0.4, storage_full=True)
    nl, _, _ = gpu_nlist(K, A, torch.from_numpy(pos).cuda(), N, gbox, obox, r_cut, 0.4)
    assert np.array_equal(nl["n_neigh"].cpu().numpy().view(np.uint32)[:N], ref["n_neigh"])
    assert np.array_equal(nl["Nmax"].cpu().numpy().view(np.uint32), ref["Nmax"])
    got = sets_of(nl, N)
    want = orc.neighbor_sets(ref["n_neigh"], ref["nlist"], ref["head"])
    for i in range(N):
        assert np.array_equal(got[i], want[i]), i
    lp = nl["last_pos"].cpu().numpy()
    assert np.array_equal(lp[:N, :3], pos[:, :3])


def test_nlist_triclinic_and_2d_and_body_filter(K, A):
    rng = np.random.default_rng(9)
    # triclinic
    L, tilt = (14.0, 12.0, 13.0), dict(xy=0.2, xz=-0.1, yz=0.15)
    gbox, obox = A.Box.from_L(*L, **tilt), orc.Box.from_L(*L, **tilt)
    N = 1500
    f = rng.random((N, 3)) - 0.5
    xyz = np.stack([f[:, 0] * L[0] + tilt["xy"] * f[:, 1] * L[1] + tilt["xz"] * f[:, 2] * L[2],
                    f[:, 1] * L[1] + tilt["yz"] * f[:, 2] * L[2], f[:, 2] * L[2]], 1)
    pos = orc.pack_pos(xyz, np.zeros(N, int))
    ref = orc.nlist_binned(pos, N, obox, [[2.0]], 0.3, storage_full=True)
    nl, _, _ = gpu_nlist(K, A, torch.from_numpy(pos).cuda(), N, gbox, obox, [[2.0]], 0.3)
    got, want = sets_of(nl, N), orc.neighbor_sets(ref["n_neigh"], ref["nlist"], ref["head"])
    assert all(np.array_equal(a, b) for a, b in zip(got, want))
    # 2D box (Lz = 0): hoomd/md/test/test_neighborlist.cc:1347-1518
    gbox, obox = A.Box.from_L(20.0, 20.0, 0.0), orc.Box.from_L(20.0, 20.0, 0.0)
    xyz = np.concatenate([(rng.random((800, 2)) - 0.5) * 20.0, np.zeros((800, 1))], 1)
    pos = orc.pack_pos(xyz, np.zeros(800, int))
    ref = orc.nlist_binned(pos, 800, obox, [[1.5]], 0.2, storage_full=True, ndim=2)
    nl, _, dim = gpu_nlist(K, A, torch.from_numpy(pos).cuda(), 800, gbox, obox, [[1.5]], 0.2, ndim=2)
    assert dim[2] == 1
    got, want = sets_of(nl, 800), orc.neighbor_sets(ref["n_neigh"], ref["nlist"], ref["head"])
    assert all(np.array_equal(a, b) for a, b in zip(got, want))
    # body filter: hoomd/md/test/test_neighborlist.cc:942-1057
    xyz, typ, L = liquid(8, seed=4)
    N = len(xyz)
    body = np.where(rng.random(N) < 0.5, rng.integers(0, 20, N), orc.NO_BODY).astype(np.int64)
    gbox, obox = boxes(A, L)
    pos = orc.pack_pos(xyz, typ)
    ref = orc.nlist_binned(pos, N, obox, [[2.0]], 0.3, storage_full=True, body=body, filter_body=True)
    nl, _, _ = gpu_nlist(K, A, torch.from_numpy(pos).cuda(), N, gbox, obox, [[2.0]], 0.3, body=body, filter_body=True)
    got, want = sets_of(nl, N), orc.neighbor_sets(ref["n_neigh"], ref["nlist"], ref["head"])
    assert all(np.array_equal(a, b) for a, b in zip(got, want))
    assert sum(len(s) for s in got) < sum(len(s) for s in sets_of(gpu_nlist(K, A, torch.from_numpy(pos).cuda(), N, gbox, obox, [[2.0]], 0.3)[0], N))


def test_nlist_with_ghost_layer_non_periodic_box(K, A):
    """Local box of a decomposed run: non-periodic in x, ghosts (index >= N) in the padded cells."""
    rng = np.random.default_rng(21)
    L = (12.0, 13.0, 14.0)
    gw = (2.4, 0.0, 0.0)
    N, ng = 1800, 700
    own = (rng.random((N, 3)) - 0.5) * np.array(L)
    gh = (rng.random((ng, 3)) - 0.5) * np.array(L)
    gh[:, 0] = np.where(rng.random(ng) < 0.5, L[0] / 2 + rng.random(ng) * gw[0], -L[0] / 2 - rng.random(ng) * gw[0])
    xyz = np.concatenate([own, gh])
    gbox, obox = boxes(A, L, periodic=(0, 1, 1))
    pos = orc.pack_pos(xyz, np.zeros(len(xyz), int))
    ref = orc.nlist_binned(pos, N, obox, [[2.0]], 0.4, storage_full=True, n_ghost=ng, ghost_width=gw)
    nl, _, _ = gpu_nlist(K, A, torch.from_numpy(pos).cuda(), N, gbox, obox, [[2.0]], 0.4, n_ghost=ng, ghost_width=gw)
    got, want = sets_of(nl, N), orc.neighbor_sets(ref["n_neigh"], ref["nlist"], ref["head"])
    assert all(np.array_equal(a, b) for a, b in zip(got, want))
    assert max(int(s.max()) for s in got if len(s)) >= N  # ghosts do appear as neighbors


def test_nlist_dense_cell_multiple_candidate_passes(K, A):
    """More than 768 candidates around one cell and more than 64 particles in it."""
    rng = np.random.default_rng(2)
    N = 1400
    xyz = (rng.random((N, 3)) - 0.5) * 1.8  # one clump
    gbox, obox = boxes(A, 12.0)
    pos = orc.pack_pos(xyz, np.zeros(N, int))
    ref = orc.nlist_binned(pos, N, obox, [[1.0]], 0.2, storage_full=True)
    nl, _, _ = gpu_nlist(K, A, torch.from_numpy(pos).cuda(), N, gbox, obox, [[1.0]], 0.2)
    got, want = sets_of(nl, N), orc.neighbor_sets(ref["n_neigh"], ref["nlist"], ref["head"])
    assert all(np.array_equal(a, b) for a, b in zip(got, want))


def test_distance_check_matches_oracle(K, A):
    xyz, typ, L = liquid(12, ntypes=2, seed=8)
    N = len(xyz)
    gbox, obox = boxes(A, L)
    pos = orc.pack_pos(xyz, typ)
    rcm = [2.5, 2.0]
    rng = np.random.default_rng(0)
    for scale, lam in [(0.05, (1, 1, 1)), (0.11, (1, 1, 1)), (0.124, (1, 1, 1)), (0.05, (0.99, 1.0, 1.01))]:
        d = rng.normal(0, 1, (N, 3))
        d *= scale / np.linalg.norm(d, axis=1)[:, None] * rng.random((N, 1))
        cur = pos.copy()
        cur[:, :3] += d
        want = orc.distance_check(cur, pos, N, obox, rcm, 0.25, lam)
        got = K.nlist_check(torch.from_numpy(cur).cuda(), torch.from_numpy(pos).cuda(), N, gbox, rcm, 0.25, lam, checkn=7)
        assert got == (7 if want else 0)
    # a particle that crossed the periodic boundary did not "move" a box length
    cur = pos.copy()
    cur[0, 0] = L / 2 - 1e-3
    last = pos.copy()
    last[0, 0] = -L / 2 + 1e-3
    assert K.nlist_check(torch.from_numpy(cur).cuda(), torch.from_numpy(last).cuda(), N, gbox, rcm, 0.25) == 0


def test_exclusion_filter_matches_oracle(K, A):
    xyz, typ, L = liquid(9, seed=6)
    N = len(xyz)
    gbox, obox = boxes(A, L)
    pos = orc.pack_pos(xyz, typ)
    nl, _, _ = gpu_nlist(K, A, torch.from_numpy(pos).cuda(), N, gbox, obox, [[2.5]], 0.4)
    n_neigh = nl["n_neigh"].cpu().numpy().view(np.uint32)[:N].copy()
    nlist = nl["nlist"].cpu().numpy().view(np.uint32).copy()
    head = nl["head"].cpu().numpy().view(np.uint64)[:N].copy()
    # exclusions (i,i+1),(i,i+2) as hoomd/md/test/test_neighborlist.cc:1070-1090
    n_ex = np.zeros(N, dtype=np.uint32)
    ex = np.zeros((4, N), dtype=np.uint32)
    for i in range(N):
        for j in (i + 1, i + 2):
            if j < N:
                for a_, b_ in ((i, j), (j, i)):
                    ex[n_ex[a_], a_] = b_
                    n_ex[a_] += 1
    K.nlist_filter(N, nl["head"], torch.from_numpy(n_ex.view(np.int32)).cuda(), torch.from_numpy(ex.view(np.int32)).cuda(),
                   nl["n_neigh"], nl["nlist"])
    orc.nlist_filter(N, head, n_ex, ex, n_neigh, nlist)
    assert np.array_equal(nl["n_neigh"].cpu().numpy().view(np.uint32)[:N], n_neigh)
    got = sets_of(nl, N)
    want = orc.neighbor_sets(n_neigh, nlist, head)
    assert all(np.array_equal(a, b) for a, b in zip(got, want))
    assert all((i + 1) not in s and (i + 2) not in s for i, s in enumerate(got))


# ------------------------------------------------------------------ pair forces
def _rel_err(a, b):
    scale = np.maximum(np.abs(b), 1e-300)
    return np.max(np.abs(a - b) / np.maximum(scale, np.abs(b).max() * 1e-9))


@pytest.mark.parametrize("shift_mode", [0, 1, 2])
@pytest.mark.parametrize("ntypes", [1, 2])
def test_lj_forces_match_oracle(K, A, shift_mode, ntypes):
    xyz, typ, L = liquid(14, ntypes=ntypes, seed=3 + ntypes)
    N = len(xyz)
    gbox, obox = boxes(A, L)
    pos = orc.pack_pos(xyz, typ)
    if ntypes == 1:
        r_cut, params = [[2.5]], [orc.lj_params(1.0, 1.0)]
        ron = [[2.0]]
    else:
        r_cut = [[2.5, 2.0], [2.0, 2.2]]
        params = [orc.lj_params(1.0, 1.0), orc.lj_params(0.8, 1.5), orc.lj_params(0.8, 1.5), orc.lj_params(0.88, 0.5)]
        ron = [[2.0, 1.5], [1.5, 2.4]]  # last: r_on > r_cut -> plain shift (hoomd/md/PotentialPair.h:693-697)
    rcutsq = np.square(r_cut)
    ronsq = np.square(ron)
    pos_t = torch.from_numpy(pos).cuda()
    nl, _, _ = gpu_nlist(K, A, pos_t, N, gbox, obox, r_cut, 0.4)
    half = orc.nlist_binned(pos, N, obox, r_cut, 0.4, storage_full=False)
    F_ref, V_ref = orc.pair_forces(orc.EV_LJ, pos, N, obox, half["n_neigh"], half["nlist"], half["head"], params, rcutsq,
                                   ronsq, shift_mode=shift_mode, third_law=True, compute_virial=True)
    for tpp in (1, 8, 64):
        F, V = K.pair_forces("lj", pos_t, N, gbox, nl["n_neigh"], nl["nlist"], nl["head"], params, rcutsq, ronsq,
                             shift_mode=shift_mode, compute_virial=True, tpp=tpp)
        F, V = F.cpu().numpy(), V.cpu().numpy()
        assert _rel_err(F[:, :3], F_ref[:, :3]) < F_RTOL * 1e-4
        assert _rel_err(F[:, 3], F_ref[:, 3]) < F_RTOL * 1e-4
        assert _rel_err(V, V_ref) < F_RTOL * 1e-4
    # without the virial flag the force output is unchanged
    F2, _ = K.pair_forces("lj", pos_t, N, gbox, nl["n_neigh"], nl["nlist"], nl["head"], params, rcutsq, ronsq,
                          shift_mode=shift_mode, compute_virial=False)
    assert _rel_err(F2.cpu().numpy(), F_ref) < F_RTOL * 1e-4


def test_lj_reference_mathematica_values(K, A):
    with open(os.path.join(G, "lj_forces_and_energies.json")) as f:
        fe = json.load(f)
    gbox, obox = boxes(A, 20.0)
    for params, forces, energies in zip(fe["LJ"]["params"], fe["LJ"]["forces"], fe["LJ"]["energies"]):
        for d, f_ref, e_ref in zip(fe["distances"], forces, energies):
            pos = torch.from_numpy(orc.pack_pos([[0, 0, 0.1], [0, 0, d + 0.1]], [0, 0])).cuda()
            nl, _, _ = gpu_nlist(K, A, pos, 2, gbox, obox, [[2.5]], 0.4)
            F, _ = K.pair_forces("lj", pos, 2, gbox, nl["n_neigh"], nl["nlist"], nl["head"],
                                 [orc.lj_params(params["sigma"], params["epsilon"])], [[6.25]])
            F = F.cpu().numpy()
            assert np.isclose(F[:, 3].sum(), e_ref, rtol=5e-6, atol=1e-6 if e_ref == 0 else 0)
            assert np.allclose(F[0, :3], [0, 0, f_ref], rtol=1e-5, atol=1e-8)


def test_example_plugin_evaluator(K, A):
    """example_plugins/pair_plugin/pytest/test_example_pair.py:15-81: harmonic pair, 3 distances x
    3 k x 3 sigma x {none, shift}, checked against the closed form to 6 decimals -- plus the oracle."""
    gbox, obox = boxes(A, 20.0)
    for d in (0.5, 0.75, 1.0):
        for k in (1.0, 20.0, 100.0):
            for sigma in (0.5, 1.0, 1.5):
                for mode in (0, 1):
                    r_cut = 2.0
                    pos = torch.from_numpy(orc.pack_pos([[0, 0, 0.1], [0, 0, d + 0.1]], [0, 0])).cuda()
                    nl, _, _ = gpu_nlist(K, A, pos, 2, gbox, obox, [[r_cut]], 0.4)
                    F, _ = K.pair_forces("example", pos, 2, gbox, nl["n_neigh"], nl["nlist"], nl["head"], [(k, sigma)],
                                         [[r_cut**2]], shift_mode=mode)
                    F = F.cpu().numpy()
                    e = 0.5 * k * (sigma - d) ** 2 - (0.5 * k * (sigma - r_cut) ** 2 if mode else 0.0)
                    np.testing.assert_array_almost_equal(F[0, :3], [0, 0, -k * (sigma - d)], decimal=6)
                    np.testing.assert_almost_equal(F[:, 3].sum(), e, decimal=6)
    xyz, typ, L = liquid(8, seed=12)
    N = len(xyz)
    gbox, obox = boxes(A, L)
    pos = orc.pack_pos(xyz, typ)
    pos_t = torch.from_numpy(pos).cuda()
    nl, _, _ = gpu_nlist(K, A, pos_t, N, gbox, obox, [[1.6]], 0.3)
    ref = orc.nlist_binned(pos, N, obox, [[1.6]], 0.3, storage_full=True)
    F_ref, V_ref = orc.pair_forces(orc.EV_EXAMPLE, pos, N, obox, ref["n_neigh"], ref["nlist"], ref["head"], [(30.0, 1.2)], [[2.56]])
    F, V = K.pair_forces("example", pos_t, N, gbox, nl["n_neigh"], nl["nlist"], nl["head"], [(30.0, 1.2)], [[2.56]])
    assert _rel_err(F.cpu().numpy(), F_ref) < F_RTOL * 1e-4 and _rel_err(V.cpu().numpy(), V_ref) < F_RTOL * 1e-4


def test_pair_force_rejects_bad_arguments(K, A):
    import ctypes as C
    a = A.PairArgs()
    a.threads_per_particle = 3
    assert A.lib.hmd_compute_pair_forces_lj(C.byref(a), None, None) < 0
    with pytest.raises(A.KernelError):
        A.check(A.lib.hmd_compute_pair_forces_lj(C.byref(a), None, None), "x")


# ------------------------------------------------------------------ integrator
def test_half_steps_bit_exact_vs_oracle(K, A):
    xyz, typ, L = liquid(10, seed=14)
    N = len(xyz)
    rng = np.random.default_rng(3)
    gbox, obox = boxes(A, L)
    pos = orc.pack_pos(xyz, typ)
    pos[:8, 0] = L / 2 - 1e-4  # these will cross the +x face
    pos[8:16, 2] = -L / 2 + 1e-4
    vel = np.concatenate([rng.normal(0, 1.3, (N, 3)), rng.uniform(0.5, 2.0, (N, 1))], 1)
    vel[:8, 0] = 1.0
    vel[8:16, 2] = -1.0
    accel = rng.normal(0, 5.0, (N, 3))
    image = rng.integers(-2, 3, (N, 3)).astype(np.int32)
    nf = np.concatenate([rng.normal(0, 7.0, (N, 3)), rng.normal(-3, 1, (N, 1))], 1)
    group = rng.permutation(N)[: N // 2].astype(np.uint32)
    for grp, rescale in ((None, 1.0), (group, 0.97)):
        p, v, a_, im = pos.copy(), vel.copy(), accel.copy(), image.copy()
        orc.step_one(p, v, a_, im, obox, 0.005, rescale, grp)
        tp, tv, ta, ti = (torch.from_numpy(x.copy()).cuda() for x in (pos, vel, accel, image))
        tg = torch.from_numpy(grp.view(np.int32)).cuda() if grp is not None else None
        K.step_one(tp, tv, ta, ti, gbox, 0.005, rescale, tg)
        assert np.array_equal(tp.cpu().numpy(), p) and np.array_equal(tv.cpu().numpy(), v)
        assert np.array_equal(ti.cpu().numpy(), im)
        assert not np.array_equal(im, image)  # some images changed
        v2, a2 = v.copy(), a_.copy()
        orc.step_two(v2, a2, nf, 0.005, rescale, grp)
        tnf = torch.from_numpy(nf).cuda()
        K.step_two(tv, ta, tnf, 0.005, rescale, tg)
        assert np.array_equal(tv.cpu().numpy(), v2) and np.array_equal(ta.cpu().numpy(), a2)
    acc = K.compute_accelerations(torch.from_numpy(vel).cuda(), torch.from_numpy(nf).cuda())
    assert np.array_equal(acc.cpu().numpy(), orc.compute_accel(vel, nf))


def test_fused_half_steps_and_check_match_the_separate_drivers(K, A):
    """step two(t) + step one(t+1) + distance check in one kernel: bit-identical to the three drivers."""
    xyz, typ, L = liquid(10, seed=15, ntypes=2)
    N = len(xyz)
    rng = np.random.default_rng(4)
    gbox, obox = boxes(A, L)
    pos = orc.pack_pos(xyz, typ)
    pos[:8, 0] = L / 2 - 1e-4
    vel = np.concatenate([rng.normal(0, 1.3, (N, 3)), rng.uniform(0.5, 2.0, (N, 1))], 1)
    vel[:8, 0] = 1.0
    accel = rng.normal(0, 5.0, (N, 3))
    image = rng.integers(-2, 3, (N, 3)).astype(np.int32)
    nf = np.concatenate([rng.normal(0, 7.0, (N, 3)), rng.normal(-3, 1, (N, 1))], 1)
    last = pos.copy()
    rcut_max = [2.5, 2.0]
    for dt, expect_move in ((0.005, False), (0.08, True)):
        p, v, a_, im = pos.copy(), vel.copy(), accel.copy(), image.copy()
        orc.step_two(v, a_, nf, dt, 1.0, None)
        orc.step_one(p, v, a_, im, obox, dt, 1.0, None)
        tp, tv, ta, ti, tnf, tl = (torch.from_numpy(x.copy()).cuda() for x in (pos, vel, accel, image, nf, last))
        flag, pub = K.fused_step(tp, tv, ta, ti, tnf, gbox, dt, last_pos=tl, rcut_max=rcut_max, r_buff=0.4, checkn=7, publish=True)
        assert np.array_equal(tp.cpu().numpy(), p) and np.array_equal(tv.cpu().numpy(), v)
        assert np.array_equal(ta.cpu().numpy(), a_) and np.array_equal(ti.cpu().numpy(), im)
        want = K.nlist_check(torch.from_numpy(p).cuda(), tl, N, gbox, rcut_max, 0.4, checkn=7)
        assert flag == want and (flag == 7) == expect_move
        assert pub == (7 << 32) | flag
    # a group, no check
    group = rng.permutation(N)[: N // 2].astype(np.uint32)
    p, v, a_, im = pos.copy(), vel.copy(), accel.copy(), image.copy()
    orc.step_two(v, a_, nf, 0.005, 1.0, group)
    orc.step_one(p, v, a_, im, obox, 0.005, 1.0, group)
    tp, tv, ta, ti, tnf = (torch.from_numpy(x.copy()).cuda() for x in (pos, vel, accel, image, nf))
    assert K.fused_step(tp, tv, ta, ti, tnf, gbox, 0.005, group=torch.from_numpy(group.view(np.int32)).cuda()) is None
    assert np.array_equal(tp.cpu().numpy(), p) and np.array_equal(tv.cpu().numpy(), v) and np.array_equal(ta.cpu().numpy(), a_)


def test_half_steps_add_two_force_arrays_like_the_net_force_pass(K, A):
    """d_net_force2: the second half step, alone and joined with the next first half step, adds two force arrays where it
    reads them; bit-identical to hmd_integrator_sum_net_force followed by the same pass on the sum."""
    xyz, typ, L = liquid(10, seed=31)
    N = len(xyz)
    rng = np.random.default_rng(8)
    gbox, obox = boxes(A, L)
    pos = orc.pack_pos(xyz, typ)
    vel = np.concatenate([rng.normal(0, 1.3, (N, 3)), rng.uniform(0.5, 2.0, (N, 1))], 1)
    image = np.zeros((N, 3), dtype=np.int32)
    f1 = np.concatenate([rng.normal(0, 7.0, (N, 3)), rng.normal(-3, 1, (N, 1))], 1)
    f2 = np.concatenate([rng.normal(0, 30.0, (N, 3)), rng.normal(5, 1, (N, 1))], 1)
    f2[:5, :3] = -f1[:5, :3]  # sums that cancel
    tf1, tf2 = torch.from_numpy(f1).cuda(), torch.from_numpy(f2).cuda()
    nf, _ = K.sum_net_force([tf1, tf2])
    # the second half step alone
    va, vb = torch.from_numpy(vel.copy()).cuda(), torch.from_numpy(vel.copy()).cuda()
    aa, ab = torch.zeros((N, 3), dtype=torch.float64, device="cuda"), torch.zeros((N, 3), dtype=torch.float64, device="cuda")
    K.step_two(va, aa, nf, 0.005)
    K.step_two(vb, ab, tf1, 0.005, net_force2=tf2)
    assert np.array_equal(va.cpu().numpy(), vb.cpu().numpy()) and np.array_equal(aa.cpu().numpy(), ab.cpu().numpy())
    # joined with the next first half step
    out = []
    for kw in (dict(net_force=nf), dict(net_force=tf1, net_force2=tf2)):
        tp, tv, ta, ti = (torch.from_numpy(x.copy()).cuda() for x in (pos, vel, np.zeros((N, 3)), image))
        K.fused_step(tp, tv, ta, ti, kw["net_force"], gbox, 0.005, net_force2=kw.get("net_force2"))
        out.append((tp.cpu().numpy(), tv.cpu().numpy(), ta.cpu().numpy(), ti.cpu().numpy()))
    for x, y in zip(*out):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("grouped", [False, True])
def test_second_half_step_takes_the_thermo_sums_along(K, A, grouped):
    """hmd_nvt_rescale_step_two_thermo: the second half step and the sums over the velocities it leaves in one pass,
    bit-identical to hmd_nvt_rescale_step_two followed by hmd_compute_thermo (same tree of partial sums); what the
    pass cannot take along is declined."""
    import ctypes as C
    rng = np.random.default_rng(77)
    N = 300_001  # more than one trip of the grid-stride loop, a ragged last one
    vel = np.concatenate([rng.normal(0, 1.3, (N, 3)), rng.uniform(0.5, 2.0, (N, 1))], 1)
    nf = np.concatenate([rng.normal(0, 7.0, (N, 3)), rng.normal(-3, 1, (N, 1))], 1)
    group = torch.from_numpy(np.sort(rng.choice(N, N // 3, replace=False)).astype(np.int32)).cuda() if grouped else None
    tf = torch.from_numpy(nf).cuda()
    va, vb = torch.from_numpy(vel.copy()).cuda(), torch.from_numpy(vel.copy()).cuda()
    aa, ab = torch.zeros((N, 3), dtype=torch.float64, device="cuda"), torch.zeros((N, 3), dtype=torch.float64, device="cuda")
    K.step_two(va, aa, tf, 0.005, rescale=0.97, group=group)
    sums_a = K.thermo(va, tf, group=group)
    sums_b = K.step_two_thermo(vb, ab, tf, 0.005, rescale=0.97, group=group).cpu().numpy()
    assert np.array_equal(va.cpu().numpy(), vb.cpu().numpy()) and np.array_equal(aa.cpu().numpy(), ab.cpu().numpy())
    assert np.array_equal(sums_a, sums_b) and sums_b[0] > 0
    # sums over other arrays than the half step's, or with virials: HMD_ERR_UNSUPPORTED
    s, t = A.StepTwoArgs(), A.ThermoArgs()
    s.d_vel, s.d_accel, s.d_net_force, s.group_size, s.deltaT, s.rescale_factor = K.ptr(vb), K.ptr(ab), K.ptr(tf), N, 0.005, 1.0
    out, scratch = torch.zeros(8, dtype=torch.float64, device="cuda"), torch.zeros(A.lib.hmd_thermo_scratch_bytes(), dtype=torch.uint8, device="cuda")
    t.d_out, t.d_scratch, t.d_vel, t.d_net_force, t.group_size = K.ptr(out), K.ptr(scratch), K.ptr(va), K.ptr(tf), N
    assert A.lib.hmd_nvt_rescale_step_two_thermo(C.byref(s), C.byref(t), None) == A.ERR_UNSUPPORTED
    t.d_vel, t.d_net_virial = K.ptr(vb), K.ptr(scratch)
    assert A.lib.hmd_nvt_rescale_step_two_thermo(C.byref(s), C.byref(t), None) == A.ERR_UNSUPPORTED
    torch.cuda.synchronize()
    assert np.array_equal(vb.cpu().numpy(), va.cpu().numpy())  # nothing ran


def test_first_half_step_reads_its_factor_from_the_device(K, A):
    """d_rescale_ke: the factor sqrt(rescale_gamma / kinetic energy) is worked out by the kernel from the sum the thermo
    pass left on the device; the same bits as with the factor the host computes from a copy of that sum, in the first
    half step alone and joined with the distance check."""
    xyz, typ, L = liquid(10, seed=16)
    N = len(xyz)
    rng = np.random.default_rng(5)
    gbox, obox = boxes(A, L)
    pos = orc.pack_pos(xyz, typ)
    vel = np.concatenate([rng.normal(0, 1.3, (N, 3)), rng.uniform(0.5, 2.0, (N, 1))], 1)
    accel = rng.normal(0, 5.0, (N, 3))
    image = np.zeros((N, 3), dtype=np.int32)
    nf = torch.zeros((N, 4), dtype=torch.float64, device="cuda")
    sums = torch.from_numpy(K.thermo(torch.from_numpy(vel).cuda(), nf)).cuda()
    gamma = 1.2345678901234567 * 1.5 * N
    factor = float(np.sqrt(np.float64(gamma) / np.float64(sums[0].item())))
    assert 0.5 < factor < 2.0
    out = []
    for kw in (dict(rescale=factor), dict(rescale=123.0, rescale_ke=sums, rescale_gamma=gamma)):
        tp, tv, ta, ti = (torch.from_numpy(x.copy()).cuda() for x in (pos, vel, accel, image))
        K.step_one(tp, tv, ta, ti, gbox, 0.005, **kw)
        out.append((tp.cpu().numpy(), tv.cpu().numpy(), ti.cpu().numpy()))
    for x, y in zip(*out):
        assert np.array_equal(x, y)
    out2 = []
    for kw in (dict(rescale_one=factor), dict(rescale_one=123.0, rescale_ke=sums, rescale_gamma=gamma)):
        tp, tv, ta, ti = (torch.from_numpy(x.copy()).cuda() for x in (pos, vel, accel, image))
        last = torch.from_numpy(pos.copy()).cuda()
        flag = K.fused_step(tp, tv, ta, ti, nf, gbox, 0.005, last_pos=last, rcut_max=[2.5], r_buff=0.4, step_two=False, **kw)
        out2.append((tp.cpu().numpy(), tv.cpu().numpy(), ti.cpu().numpy(), np.array(flag[0])))
    for x, y in zip(*out2):
        assert np.array_equal(x, y)
    assert np.array_equal(out[0][0], out2[0][0])


def test_first_half_step_with_check_and_slot_copy_matches_the_separate_drivers(K, A):
    """hmd_nvt_rescale_step_one_check: step one with a thermostat's rescaling factor + distance check + slot-ordered
    position copy in one kernel, bit-identical to the drivers run one after the other."""
    xyz, typ, L = liquid(10, seed=16, ntypes=2)
    N = len(xyz)
    rng = np.random.default_rng(5)
    gbox, obox = boxes(A, L)
    pos = orc.pack_pos(xyz, typ)
    pos[:8, 0] = L / 2 - 1e-4
    vel = np.concatenate([rng.normal(0, 1.3, (N, 3)), rng.uniform(0.5, 2.0, (N, 1))], 1)
    vel[:8, 0] = 1.0
    accel = rng.normal(0, 5.0, (N, 3))
    image = rng.integers(-2, 3, (N, 3)).astype(np.int32)
    nf = np.zeros((N, 4))
    last = pos.copy()
    rcut_max = [2.5, 2.0]
    slot_of = rng.permutation(2 * N)[:N].astype(np.int32)
    for dt, scale, expect_move in ((0.005, 0.97, False), (0.08, 1.04, True)):
        p, v, a_, im = pos.copy(), vel.copy(), accel.copy(), image.copy()
        orc.step_one(p, v, a_, im, obox, dt, scale, None)
        tp, tv, ta, ti, tnf, tl = (torch.from_numpy(x.copy()).cuda() for x in (pos, vel, accel, image, nf, last))
        cpos = torch.zeros((2 * N, 4), dtype=torch.float64, device="cuda")
        flag, pub = K.fused_step(tp, tv, ta, ti, tnf, gbox, dt, rescale_one=scale, last_pos=tl, rcut_max=rcut_max, r_buff=0.4, checkn=9, publish=True,
                                 step_two=False, slot_of=torch.from_numpy(slot_of).cuda(), cpos=cpos)
        assert np.array_equal(tp.cpu().numpy(), p) and np.array_equal(tv.cpu().numpy(), v)
        assert np.array_equal(ta.cpu().numpy(), accel) and np.array_equal(ti.cpu().numpy(), im)
        want = K.nlist_check(torch.from_numpy(p).cuda(), tl, N, gbox, rcut_max, 0.4, checkn=9)
        assert flag == want and (flag == 9) == expect_move and pub == (9 << 32) | flag
        assert np.array_equal(cpos.cpu().numpy()[slot_of], p)


def test_sum_net_force_and_thermo(K, A):
    rng = np.random.default_rng(4)
    N = 5000
    fs = [rng.normal(0, 1, (N, 4)) for _ in range(3)]
    vs = [rng.normal(0, 1, (6, N)) for _ in range(3)]
    nf_ref, nv_ref = orc.sum_net_force(fs, vs)
    nf, nv = K.sum_net_force([torch.from_numpy(f).cuda() for f in fs], [torch.from_numpy(v).cuda() for v in vs])
    assert np.array_equal(nf.cpu().numpy(), nf_ref) and np.array_equal(nv.cpu().numpy(), nv_ref)
    # accumulate on top (clear = False)
    nf2, _ = K.sum_net_force([torch.from_numpy(fs[0]).cuda()], None, net_force=nf.clone(), clear=False)
    assert np.array_equal(nf2.cpu().numpy(), nf_ref + fs[0])
    vel = np.concatenate([rng.normal(0, 1.2, (N, 3)), rng.uniform(0.5, 2, (N, 1))], 1)
    ke, pe = orc.thermo(vel, nf_ref)
    out = K.thermo(torch.from_numpy(vel).cuda(), nf, nv)
    assert np.isclose(out[0], ke, rtol=1e-12) and np.isclose(out[1], pe, rtol=1e-11, atol=1e-9)
    assert np.isclose(out[2], (nv_ref[0] + nv_ref[3] + nv_ref[5]).sum(), rtol=1e-10, atol=1e-9)
    assert np.allclose(out[3:6], (vel[:, :3] * vel[:, 3:4]).sum(0), rtol=1e-10, atol=1e-9)
    out2 = K.thermo(torch.from_numpy(vel).cuda(), nf, nv)
    assert np.array_equal(out, out2)  # fixed reduction tree: reproducible


# ------------------------------------------------------------------ reorder + halo helpers
def test_cell_order_permutation_and_gather(K, A):
    xyz, typ, L = liquid(12, seed=30)
    N = len(xyz)
    rng = np.random.default_rng(1)
    perm = rng.permutation(N)
    xyz = xyz[perm]
    gbox, obox = boxes(A, L)
    dim = orc.cell_dimensions(obox, 2.9)
    pos = torch.from_numpy(orc.pack_pos(xyz, typ)).cuda()
    cl = K.cell_list_auto(pos, N, 0, gbox, dim)
    order, offs = K.cell_order(cl, N)
    o = order.cpu().numpy()
    assert np.array_equal(np.sort(o), np.arange(N))
    cs = cl["cell_size"].cpu().numpy()
    of = offs.cpu().numpy()
    assert np.array_equal(of, np.concatenate([[0], np.cumsum(cs)]))
    xyzf = cl["xyzf"].cpu().numpy()
    for c in range(0, len(cs), 5):
        assert np.array_equal(o[of[c]:of[c + 1]], orc.unpack_int_w(xyzf[c, :cs[c], 3]))
    newpos = K.gather_rows(pos, order)
    assert np.array_equal(newpos.cpu().numpy(), pos.cpu().numpy()[o])
    img = torch.from_numpy(rng.integers(-3, 3, (N, 3)).astype(np.int32)).cuda()
    assert np.array_equal(K.gather_rows(img, order).cpu().numpy(), img.cpu().numpy()[o])
    acc = torch.from_numpy(rng.normal(0, 1, (N, 3))).cuda()
    assert np.array_equal(K.gather_rows(acc, order).cpu().numpy(), acc.cpu().numpy()[o])
    tag = K.gather_rows(torch.arange(N, dtype=torch.int32).cuda(), order)
    rtag = K.build_rtag(tag)
    assert np.array_equal(tag.cpu().numpy()[rtag.cpu().numpy()], np.arange(N))


def test_ghost_plan_and_pack(K, A):
    rng = np.random.default_rng(12)
    L = (10.0, 12.0, 14.0)
    gbox, obox = boxes(A, L, periodic=(0, 0, 1))
    N = 4000
    xyz = (rng.random((N, 3)) - 0.5) * np.array(L)
    typ = rng.integers(0, 2, N)
    pos = orc.pack_pos(xyz, typ)
    rg = np.array([2.9, 2.4])
    plan = K.ghost_plan(torch.from_numpy(pos).cuda(), N, gbox, rg, mask=0b001111).cpu().numpy()
    want = np.zeros(N, dtype=np.int32)
    for i in range(N):
        f = orc.make_fraction(obox, xyz[i])
        g = rg[typ[i]] / np.array(L)
        for d in range(3):
            if f[d] >= 1.0 - g[d]:
                want[i] |= 1 << (2 * d)
            if f[d] < g[d]:
                want[i] |= 1 << (2 * d + 1)
    assert np.array_equal(plan, want & 0b001111)
    send = torch.from_numpy(rng.integers(0, N, 900).astype(np.int32)).cuda()
    sid = rng.integers(0, 27, 900).astype(np.uint8)
    table = np.array([[(i % 3 - 1) * L[0], ((i // 3) % 3 - 1) * L[1], (i // 9 - 1) * L[2]] for i in range(27)])
    out = K.pack_ghost_positions(torch.from_numpy(pos).cuda(), send, torch.from_numpy(sid).cuda(), torch.from_numpy(table).cuda())
    s = send.cpu().numpy()
    exp = pos[s].copy()
    exp[:, :3] += table[sid]
    assert np.array_equal(out.cpu().numpy(), exp)


def test_migrate_dest_and_vote_rows(K, A):
    import ctypes as C
    rng = np.random.default_rng(21)
    glo, gL, grid, dec, rank = np.array([-6.0, -7.0, -8.0]), np.array([12.0, 14.0, 16.0]), (2, 1, 3), (1, 0, 1), 4
    N = 5000
    # some particles outside the global box along the decomposed directions (they wrap), none along y
    xyz = glo + rng.random((N, 3)) * gL
    xyz[:200, 0] += gL[0] * rng.choice([-1, 1], 200) * 0.3
    xyz[200:400, 2] -= gL[2] * 0.2
    pos = torch.from_numpy(orc.pack_pos(xyz, np.zeros(N, dtype=int))).cuda()
    dest = torch.full((N,), -1, dtype=torch.int32, device="cuda")
    n_leave = torch.zeros(1, dtype=torch.int32, device="cuda")
    a = A.MigrateDestArgs()
    a.d_pos, a.N, a.rank = pos.data_ptr(), N, rank
    a.global_lo[:], a.global_L[:], a.grid[:], a.decomposed[:] = list(glo), list(gL), list(grid), list(dec)
    a.d_dest, a.d_n_leave = dest.data_ptr(), n_leave.data_ptr()
    A.check(A.lib.hmd_migrate_dest(C.byref(a), None), "hmd_migrate_dest")
    f = (xyz - glo) / gL
    w = np.where(np.array(dec, dtype=bool), np.floor(f), 0.0)
    c = np.clip(np.floor((f - w) * np.array(grid)).astype(np.int64), 0, np.array(grid) - 1)
    want = (c[:, 2] * grid[1] + c[:, 1]) * grid[0] + c[:, 0]
    assert np.array_equal(dest.cpu().numpy(), want)
    assert int(n_leave.item()) == int((want != rank).sum())
    assert A.lib.hmd_migrate_dest(None, None) == A.ERR_INVALID_ARGUMENT

    # pack with vote rows, then unpack: positions land where the map says, the flag becomes the maximum
    n_send, n_flag = 700, 3
    send = torch.from_numpy(rng.integers(0, N, n_send).astype(np.int32)).cuda()
    sid = torch.from_numpy(rng.integers(0, 27, n_send).astype(np.uint8)).cuda()
    table = torch.from_numpy(rng.normal(size=(27, 3))).cuda()
    rows = rng.permutation(n_send + n_flag)
    out_row = torch.from_numpy(rows[:n_send].astype(np.int32)).cuda()
    flag_rows = torch.from_numpy(rows[n_send:].astype(np.int32)).cuda()
    flag = torch.tensor([41], dtype=torch.int32, device="cuda")
    buf = torch.full((n_send + n_flag, 4), float("nan"), dtype=torch.float64, device="cuda")
    p = A.PackVoteArgs()
    p.pack.d_out, p.pack.d_pos, p.pack.d_send_idx, p.pack.d_send_shift_id = buf.data_ptr(), pos.data_ptr(), send.data_ptr(), sid.data_ptr()
    p.pack.d_shift_table, p.pack.n_send = table.data_ptr(), n_send
    p.d_out_row, p.d_flag_word, p.d_flag_rows, p.n_flag_rows = out_row.data_ptr(), flag.data_ptr(), flag_rows.data_ptr(), n_flag
    A.check(A.lib.hmd_pack_ghost_positions_vote(C.byref(p), None), "hmd_pack_ghost_positions_vote")
    exp = pos[send.long()].clone()
    exp[:, :3] += table[sid.long()]
    assert torch.equal(buf[out_row.long()], exp)
    assert torch.all(buf[flag_rows.long(), 0].view(torch.int64) & 0xFFFFFFFF == 41)
    # a "received" buffer: the same rows, two of the flag rows carrying other ranks' words (one larger)
    buf[flag_rows.long()[0], 0] = torch.tensor([57], dtype=torch.int64, device="cuda").view(torch.float64)[0]
    buf[flag_rows.long()[1], 0] = torch.tensor([12], dtype=torch.int64, device="cuda").view(torch.float64)[0]
    in_map = torch.full((n_send + n_flag,), -1, dtype=torch.int32, device="cuda")
    in_map[out_row.long()] = torch.arange(n_send, dtype=torch.int32, device="cuda")
    ghost = torch.zeros((n_send, 4), dtype=torch.float64, device="cuda")
    u = A.UnpackVoteArgs()
    u.d_in, u.d_map, u.n_rows, u.d_ghost_pos, u.d_flag_word = buf.data_ptr(), in_map.data_ptr(), n_send + n_flag, ghost.data_ptr(), flag.data_ptr()
    A.check(A.lib.hmd_unpack_ghost_positions(C.byref(u), None), "hmd_unpack_ghost_positions")
    assert torch.equal(ghost, exp) and int(flag.item()) == 57
    # publish: stamp and word arrive together in pinned memory
    pinned = torch.zeros(1, dtype=torch.int64).pin_memory()
    A.check(A.lib.hmd_publish_word(flag.data_ptr(), pinned.data_ptr(), 9, None), "hmd_publish_word")
    torch.cuda.synchronize()
    assert int(pinned.item()) == (9 << 32) | 57


def test_ghost_send_list_keys(K, A):
    import ctypes as C
    rng = np.random.default_rng(33)
    L = (10.0, 12.0, 14.0)
    gbox, obox = boxes(A, L, periodic=(0, 0, 1))
    N = 6000
    xyz = (rng.random((N, 3)) - 0.5) * np.array(L)
    typ = rng.integers(0, 2, N)
    pos = torch.from_numpy(orc.pack_pos(xyz, typ)).cuda()
    rg = torch.tensor([2.5, 2.0], dtype=torch.float64, device="cuda")
    # directions of a domain decomposed along x and y: 8 combinations of the x and y plan bits
    masks = [1, 2, 4, 8, 1 | 4, 1 | 8, 2 | 4, 2 | 8]
    dir_mask = torch.tensor(masks, dtype=torch.int32, device="cuda")
    plan = torch.zeros(N, dtype=torch.int32, device="cuda")
    counters = torch.zeros(1 + len(masks), dtype=torch.int32, device="cuda")
    a = A.GhostSendListArgs()
    a.ghost_plan.d_plan, a.ghost_plan.d_pos, a.ghost_plan.N, a.ghost_plan.box = plan.data_ptr(), pos.data_ptr(), N, gbox
    a.ghost_plan.d_r_ghost, a.ghost_plan.ntypes, a.ghost_plan.mask = rg.data_ptr(), 2, 0b001111
    a.d_dir_mask, a.n_dir = dir_mask.data_ptr(), len(masks)
    a.d_total, a.d_dir_count = counters.data_ptr(), counters.data_ptr() + 4
    want_plan = K.ghost_plan(pos, N, gbox, rg.cpu().numpy(), mask=0b001111).cpu().numpy()
    want = sorted((k << 32) | i for k, m in enumerate(masks) for i in np.nonzero((want_plan & m) == m)[0].tolist())
    # a buffer that is too small reports the need; the full one holds exactly the expected keys
    for cap in (10, len(want) + 5):
        keys = torch.full((cap,), -1, dtype=torch.int64, device="cuda")
        counters.zero_()
        a.d_keys, a.capacity = keys.data_ptr(), cap
        A.check(A.lib.hmd_ghost_send_list(C.byref(a), None), "hmd_ghost_send_list")
        c = counters.tolist()
        assert c[0] == len(want)
        assert c[1:] == [int(((want_plan & m) == m).sum()) for m in masks]
    assert np.array_equal(plan.cpu().numpy(), want_plan)
    assert sorted(keys[:len(want)].tolist()) == want and int(keys[len(want)]) == -1


@pytest.mark.parametrize("N,frac_out", [(5000, 0.03), (3000, 0.6), (700, 0.0)])
def test_migration_kernels(K, A, N, frac_out):
    """Owner keys + counts, pack, gap filling and unpack against the same steps done with array operations
    (the expressions of the CPU-tensor path of ParticleExchange.migrate)."""
    import ctypes as C
    rng = np.random.default_rng(5 + N)
    glo, gL, grid, dec, rank = np.array([-6.0, -7.0, -8.0]), np.array([12.0, 14.0, 16.0]), (2, 1, 3), (1, 0, 1), 3
    world = 6
    lo = glo + np.array([1, 0, 1]) * gL / np.array(grid)     # rank 3 = cell (1, 0, 1)
    xyz = lo + rng.random((N, 3)) * gL / np.array(grid)
    out = rng.random(N) < frac_out
    xyz[out] += rng.normal(0, 4.0, (int(out.sum()), 3)) * np.array(dec)   # some leave, also across the periodic boundary
    pos = orc.pack_pos(xyz, rng.integers(0, 3, N))
    vel, accel = rng.normal(size=(N, 4)), rng.normal(size=(N, 3))
    image = rng.integers(-3, 4, (N, 3)).astype(np.int32)
    tag = rng.permutation(N).astype(np.int32)
    body = np.where(rng.random(N) < 0.5, -1, tag).astype(np.int32)
    charge, diameter = rng.normal(size=N), rng.random(N) + 0.5
    host = dict(pos=pos, vel=vel, accel=accel, image=image, tag=tag, body=body, charge=charge, diameter=diameter)
    extra = 400  # room for incoming particles
    dev = {}
    for k, v in host.items():
        t = torch.zeros((N + extra,) + v.shape[1:], dtype=torch.from_numpy(v).dtype, device="cuda")
        t[:N] = torch.from_numpy(v).cuda()
        dev[k] = t

    def arrays():
        p = A.PDataArraysStruct()
        p.d_pos, p.d_vel, p.d_accel, p.d_image = dev["pos"].data_ptr(), dev["vel"].data_ptr(), dev["accel"].data_ptr(), dev["image"].data_ptr()
        p.d_tag, p.d_body, p.d_charge, p.d_diameter = dev["tag"].data_ptr(), dev["body"].data_ptr(), dev["charge"].data_ptr(), dev["diameter"].data_ptr()
        return p

    # ---- owners, keys and counts per owner ----
    dest = torch.full((N,), -1, dtype=torch.int32, device="cuda")
    counters = torch.zeros(1 + world, dtype=torch.int32, device="cuda")
    keys = torch.zeros(N, dtype=torch.int64, device="cuda")
    a = A.MigrateDestArgs()
    a.d_pos, a.N, a.rank = dev["pos"].data_ptr(), N, rank
    a.global_lo[:], a.global_L[:], a.grid[:], a.decomposed[:] = list(glo), list(gL), list(grid), list(dec)
    a.d_dest, a.d_n_leave, a.d_rank_count, a.n_ranks = dest.data_ptr(), counters.data_ptr(), counters.data_ptr() + 4, world
    a.d_keys, a.capacity = keys.data_ptr(), N
    A.check(A.lib.hmd_migrate_dest(C.byref(a), None), "hmd_migrate_dest")
    f = (xyz - glo) / gL
    w = np.where(np.array(dec, dtype=bool), np.floor(f), 0.0)
    c = np.clip(np.floor((f - w) * np.array(grid)).astype(np.int64), 0, np.array(grid) - 1)
    want_dest = (c[:, 2] * grid[1] + c[:, 1]) * grid[0] + c[:, 0]
    leave = want_dest != rank
    n_leave = int(leave.sum())
    cnt = counters.cpu().numpy()
    assert np.array_equal(dest.cpu().numpy(), want_dest)
    assert cnt[0] == n_leave and np.array_equal(cnt[1:], np.bincount(want_dest[leave], minlength=world))
    skeys = torch.sort(keys[:n_leave])[0]
    li = np.nonzero(leave)[0]
    li = li[np.argsort(want_dest[li], kind="stable")]
    assert np.array_equal(skeys.cpu().numpy(), (want_dest[li].astype(np.int64) << 32) | li)
    # a key buffer that is too small: the count still says how many there are
    a.capacity = max(n_leave // 2, 1)
    counters.zero_()
    A.check(A.lib.hmd_migrate_dest(C.byref(a), None), "hmd_migrate_dest")
    assert int(counters[0].item()) == n_leave

    # ---- pack ----
    payload = torch.full((max(n_leave, 1), A.MIGRATE_ROW_WORDS), -7, dtype=torch.int64, device="cuda")
    pk = A.MigratePackArgs()
    pk.arrays, pk.d_keys, pk.n, pk.d_out = arrays(), skeys.data_ptr(), n_leave, payload.data_ptr()
    pk.global_lo[:], pk.global_L[:], pk.decomposed[:] = list(glo), list(gL), list(dec)
    A.check(A.lib.hmd_migrate_pack(C.byref(pk), None), "hmd_migrate_pack")
    p = pos[li].copy()
    p[:, :3] -= w[li] * gL
    exp = np.concatenate([p.view(np.int64), vel[li].view(np.int64), accel[li].view(np.int64), image[li].astype(np.int64) + w[li].astype(np.int64),
                          tag[li].astype(np.int64)[:, None], body[li].astype(np.int64)[:, None], charge[li].view(np.int64)[:, None],
                          diameter[li].view(np.int64)[:, None]], axis=1)
    assert np.array_equal(payload.cpu().numpy()[:n_leave], exp)

    # ---- the same rows as fixed-size messages with headers; a slot that is too small drops the excess ----
    counts = np.bincount(want_dest[leave], minlength=world)
    for K in (int(counts.max()) + 3, max(int(counts.max()) // 2, 1)):
        fixed = torch.full((world * (K + 1), A.MIGRATE_ROW_WORDS), -7, dtype=torch.int64, device="cuda")
        pk.slot_rows, pk.n_ranks, pk.d_out = K, world, fixed.data_ptr()
        for r in range(world + 1):
            pk.send_off[r] = int(counts[:r].sum())
        A.check(A.lib.hmd_migrate_pack(C.byref(pk), None), "hmd_migrate_pack")
        fx = fixed.cpu().numpy().reshape(world, K + 1, -1)
        for r in range(world):
            assert fx[r, 0, 0] == counts[r] and fx[r, 0, 1] == counts.max() and fx[r, 0, 2] == K
            take = min(int(counts[r]), K)
            first = int(counts[:r].sum())
            assert np.array_equal(fx[r, 1:1 + take], exp[first:first + take])
            assert (fx[r, 1 + take:] == -7).all()
    pk.slot_rows, pk.d_out = 0, payload.data_ptr()
    if n_leave and n_leave <= extra:
        # unpack from the roomy fixed-size layout = unpack from the back-to-back rows
        K = int(counts.max()) + 3
        fixed = torch.zeros((world * (K + 1), A.MIGRATE_ROW_WORDS), dtype=torch.int64, device="cuda")
        pk.slot_rows, pk.d_out = K, fixed.data_ptr()
        A.check(A.lib.hmd_migrate_pack(C.byref(pk), None), "hmd_migrate_pack")
        pk.slot_rows, pk.d_out = 0, payload.data_ptr()
        probe = {k: torch.zeros_like(v) for k, v in dev.items()}
        uf = A.MigrateUnpackArgs()
        uf.arrays.d_pos, uf.arrays.d_vel, uf.arrays.d_accel, uf.arrays.d_image = probe["pos"].data_ptr(), probe["vel"].data_ptr(), probe["accel"].data_ptr(), probe["image"].data_ptr()
        uf.arrays.d_tag, uf.arrays.d_body, uf.arrays.d_charge, uf.arrays.d_diameter = probe["tag"].data_ptr(), probe["body"].data_ptr(), probe["charge"].data_ptr(), probe["diameter"].data_ptr()
        uf.d_in, uf.first, uf.n, uf.slot_rows, uf.n_ranks = fixed.data_ptr(), 5, n_leave, K, world
        for r in range(world + 1):
            uf.recv_off[r] = int(counts[:r].sum())
        A.check(A.lib.hmd_migrate_unpack(C.byref(uf), None), "hmd_migrate_unpack")
        assert np.array_equal(probe["tag"].cpu().numpy()[5:5 + n_leave], tag[li])
        assert np.array_equal(probe["pos"].cpu().numpy()[5:5 + n_leave].view(np.int64), exp[:, 0:4])
        assert np.array_equal(probe["diameter"].cpu().numpy()[5:5 + n_leave], diameter[li])

    # ---- fill the gaps: the stayers, each with all of its fields, end up in [0, n_keep) ----
    n_keep = N - n_leave
    scratch = torch.zeros(max(2 * n_leave, 1), dtype=torch.int32, device="cuda")
    fl = A.MigrateFillArgs()
    fl.arrays, fl.d_dest, fl.rank, fl.N, fl.d_keys, fl.n_leave, fl.d_scratch = arrays(), dest.data_ptr(), rank, N, skeys.data_ptr(), n_leave, scratch.data_ptr()
    A.check(A.lib.hmd_migrate_fill(C.byref(fl), None), "hmd_migrate_fill")
    got = {k: v.cpu().numpy() for k, v in dev.items()}
    stay = np.nonzero(~leave)[0]
    assert sorted(got["tag"][:n_keep]) == sorted(tag[stay])
    by_tag = np.empty(N, dtype=np.int64)
    by_tag[tag] = np.arange(N)
    src = by_tag[got["tag"][:n_keep]]
    for k, v in host.items():
        assert np.array_equal(got[k][:n_keep], v[src]), k
    # stayers in front of the boundary did not move
    front = stay[stay < n_keep]
    assert np.array_equal(got["tag"][front], tag[front])

    # ---- unpack: the leavers' rows come back as new particles behind the stayers ----
    n_in = min(n_leave, extra)
    u = A.MigrateUnpackArgs()
    u.arrays, u.d_in, u.first, u.n = arrays(), payload.data_ptr(), n_keep, n_in
    A.check(A.lib.hmd_migrate_unpack(C.byref(u), None), "hmd_migrate_unpack")
    got = {k: v.cpu().numpy() for k, v in dev.items()}
    s = slice(n_keep, n_keep + n_in)
    assert np.array_equal(got["pos"][s].view(np.int64), exp[:n_in, 0:4])
    assert np.array_equal(got["vel"][s], vel[li[:n_in]]) and np.array_equal(got["accel"][s], accel[li[:n_in]])
    assert np.array_equal(got["image"][s].astype(np.int64), exp[:n_in, 11:14])
    assert np.array_equal(got["tag"][s], tag[li[:n_in]]) and np.array_equal(got["body"][s], body[li[:n_in]])
    assert np.array_equal(got["charge"][s], charge[li[:n_in]]) and np.array_equal(got["diameter"][s], diameter[li[:n_in]])
    assert A.lib.hmd_migrate_pack(None, None) == A.ERR_INVALID_ARGUMENT


def test_ghost_exchange_helper_kernels(K, A):
    """Key splitting, the first exchange's pack / unpack (tag in the upper half of w) and the row maps of the
    exchange that carries the vote, each against the same thing written with array operations."""
    import ctypes as C
    rng = np.random.default_rng(77)
    # ---- keys -> index and direction arrays ----
    n = 3000
    idx = rng.integers(0, 2**31 - 1, n).astype(np.int64)
    d = rng.integers(0, 26, n).astype(np.int64)
    keys = torch.from_numpy(np.sort((d << 32) | idx)).cuda()
    si = torch.zeros(n, dtype=torch.int32, device="cuda")
    sd = torch.zeros(n, dtype=torch.uint8, device="cuda")
    A.check(A.lib.hmd_ghost_split_keys(keys.data_ptr(), n, si.data_ptr(), sd.data_ptr(), None), "hmd_ghost_split_keys")
    kk = keys.cpu().numpy()
    assert np.array_equal(si.cpu().numpy(), (kk & 0xFFFFFFFF).astype(np.int32)) and np.array_equal(sd.cpu().numpy(), (kk >> 32).astype(np.uint8))

    # ---- first exchange: pack with tag, unpack with defaults ----
    N, n_send = 2000, 900
    pos = orc.pack_pos(rng.normal(size=(N, 3)) * 5, rng.integers(0, 4, N))
    tag = rng.permutation(N).astype(np.int32)
    send = rng.integers(0, N, n_send).astype(np.int32)
    sid = rng.integers(0, 27, n_send).astype(np.uint8)
    table = rng.normal(size=(27, 3))
    pos_t, tag_t, send_t, sid_t, table_t = (torch.from_numpy(v).cuda() for v in (pos, tag, send, sid, table))
    out = torch.zeros((n_send, 4), dtype=torch.float64, device="cuda")
    pa = A.PackPosArgs()
    pa.d_out, pa.d_pos, pa.d_send_idx, pa.d_send_shift_id, pa.d_shift_table, pa.n_send = out.data_ptr(), pos_t.data_ptr(), send_t.data_ptr(), sid_t.data_ptr(), table_t.data_ptr(), n_send
    A.check(A.lib.hmd_pack_ghost_init(C.byref(pa), tag_t.data_ptr(), None), "hmd_pack_ghost_init")
    first, total = 50, 50 + n_send
    g = dict(pos=torch.full((total, 4), 9.0, dtype=torch.float64, device="cuda"), image=torch.full((total, 3), 9, dtype=torch.int32, device="cuda"),
             tag=torch.full((total,), 9, dtype=torch.int32, device="cuda"), body=torch.full((total,), 9, dtype=torch.int32, device="cuda"),
             charge=torch.full((total,), 9.0, dtype=torch.float64, device="cuda"), diameter=torch.full((total,), 9.0, dtype=torch.float64, device="cuda"))
    u = A.UnpackGhostInitArgs()
    u.d_in, u.n, u.first = out.data_ptr(), n_send, first
    u.arrays.d_pos, u.arrays.d_image, u.arrays.d_tag, u.arrays.d_body = g["pos"].data_ptr(), g["image"].data_ptr(), g["tag"].data_ptr(), g["body"].data_ptr()
    u.arrays.d_charge, u.arrays.d_diameter = g["charge"].data_ptr(), g["diameter"].data_ptr()
    A.check(A.lib.hmd_unpack_ghost_init(C.byref(u), None), "hmd_unpack_ghost_init")
    exp = pos[send].copy()
    exp[:, :3] += table[sid]
    assert np.array_equal(g["pos"].cpu().numpy()[first:].view(np.int64), exp.view(np.int64))   # w: the type id, nothing else
    assert np.array_equal(g["tag"].cpu().numpy()[first:], tag[send])
    assert (g["image"][first:] == 0).all() and (g["body"][first:] == -1).all() and (g["charge"][first:] == 0).all() and (g["diameter"][first:] == 1).all()
    assert (g["pos"][:first] == 9).all() and (g["tag"][:first] == 9).all()

    # ---- row maps of the exchange with vote rows ----
    for world, rank in ((2, 0), (2, 1), (4, 2), (8, 7), (8, 0)):
        send_c = [int(v) for v in rng.integers(0, 40, world)]
        recv_c = [int(v) for v in rng.integers(0, 40, world)]
        send_c[rank] = recv_c[rank] = 0
        if world == 4:
            send_c[1] = recv_c[3] = 0   # an empty message still gets its vote row
        out_row, flag_rows, in_map, off, ghost = [], [], [], 0, 0
        for r in range(world):
            out_row += list(range(off, off + send_c[r]))
            if r != rank:
                flag_rows.append(off + send_c[r])
            off += send_c[r] + (r != rank)
        for r in range(world):
            in_map += list(range(ghost, ghost + recv_c[r]))
            if r != rank:
                in_map.append(-1)
            ghost += recv_c[r]
        a = A.VotePlanArgs()
        a.n_ranks, a.rank = world, rank
        for r in range(world + 1):
            a.send_off[r], a.recv_off[r] = sum(send_c[:r]), sum(recv_c[:r])
        o = torch.full((max(sum(send_c), 1),), -5, dtype=torch.int32, device="cuda")
        f = torch.full((world - 1,), -5, dtype=torch.int32, device="cuda")
        im = torch.full((len(in_map),), -5, dtype=torch.int32, device="cuda")
        a.d_out_row, a.d_flag_rows, a.d_in_map = o.data_ptr(), f.data_ptr(), im.data_ptr()
        A.check(A.lib.hmd_vote_plan(C.byref(a), None), "hmd_vote_plan")
        assert o.cpu().tolist()[:sum(send_c)] == out_row and f.cpu().tolist() == flag_rows and im.cpu().tolist() == in_map
    a.n_ranks = 65
    assert A.lib.hmd_vote_plan(C.byref(a), None) == A.ERR_INVALID_ARGUMENT


@pytest.mark.parametrize("n,bits", [(0, 37), (1, 37), (777, 37), (200000, 37), (5000, 64), (4096, 33)])
def test_sort_keys(K, A, n, bits):
    """hmd_sort_keys: the send-list keys (destination << 32 | particle index) in ascending order."""
    rng = np.random.default_rng(n + bits)
    if bits == 64:
        keys = rng.integers(0, 2**62, n, dtype=np.int64)
    else:
        keys = (rng.integers(0, 2**(bits - 32), n, dtype=np.int64) << 32) | rng.integers(0, 2**21, n, dtype=np.int64)
    out = K.sort_keys(torch.from_numpy(keys).cuda(), bits).cpu().numpy()
    assert np.array_equal(out, np.sort(keys))


def test_probe_tells_streams_that_run_beside_the_default_stream_from_those_behind_it():
    """hmd_stream_runs_beside_default: of a handful of created streams at least one has a hardware queue of its own
    (its one-wave kernel finishes while the default stream's long kernel runs); the answer for one stream is stable."""
    import ctypes as C
    from hoomd_amd import _abi
    words = torch.zeros(2, dtype=torch.int64, device="cuda")
    streams = [torch.cuda.Stream() for _ in range(5)]
    answers = []
    for s in streams:
        got = []
        for _ in range(2):
            o = C.c_int(-1)
            _abi.check(_abi.lib.hmd_stream_runs_beside_default(C.c_void_p(s.cuda_stream), _abi.ptr(words), C.byref(o)), "probe")
            got.append(o.value)
        assert got[0] == got[1] and got[0] in (0, 1)
        answers.append(got[0])
    assert sum(answers) >= 1
    o = C.c_int(-1)
    assert _abi.lib.hmd_stream_runs_beside_default(None, _abi.ptr(words), C.byref(o)) == _abi.ERR_INVALID_ARGUMENT


def test_packs_that_read_their_sizes_on_the_device(K, A):
    """hmd_migrate_pack_slots_dev / hmd_pack_ghost_init_slots_dev write the very buffers hmd_migrate_pack /
    hmd_pack_ghost_init_slots write when the host passes the counts in; a list that outgrew its key array is announced
    in every header; the *_and header kernels hand the host headers and counts in one piece."""
    import ctypes as C
    rng = np.random.default_rng(31)
    # ---- migration ----
    N, world, rank = 5000, 6, 3
    glo, gL, grid, dec = np.array([-6.0, -7.0, -8.0]), np.array([12.0, 14.0, 16.0]), (2, 1, 3), (1, 0, 1)
    lo = glo + np.array([1, 0, 1]) * gL / np.array(grid)
    xyz = lo + rng.random((N, 3)) * gL / np.array(grid)
    out = rng.random(N) < 0.1
    xyz[out] += rng.normal(0, 4.0, (int(out.sum()), 3)) * np.array(dec)
    dev = dict(pos=torch.from_numpy(orc.pack_pos(xyz, rng.integers(0, 3, N))).cuda(), vel=torch.from_numpy(rng.normal(size=(N, 4))).cuda(),
               accel=torch.from_numpy(rng.normal(size=(N, 3))).cuda(), image=torch.from_numpy(rng.integers(-3, 4, (N, 3)).astype(np.int32)).cuda(),
               tag=torch.from_numpy(rng.permutation(N).astype(np.int32)).cuda(), body=torch.full((N,), -1, dtype=torch.int32, device="cuda"),
               charge=torch.from_numpy(rng.normal(size=N)).cuda(), diameter=torch.from_numpy(rng.random(N) + 0.5).cuda())
    p = A.PDataArraysStruct()
    p.d_pos, p.d_vel, p.d_accel, p.d_image = dev["pos"].data_ptr(), dev["vel"].data_ptr(), dev["accel"].data_ptr(), dev["image"].data_ptr()
    p.d_tag, p.d_body, p.d_charge, p.d_diameter = dev["tag"].data_ptr(), dev["body"].data_ptr(), dev["charge"].data_ptr(), dev["diameter"].data_ptr()
    dest = torch.zeros(N, dtype=torch.int32, device="cuda")
    for cap in (2048, 64):   # roomy, and too small for the leavers
        counters = torch.zeros(1 + world, dtype=torch.int32, device="cuda")
        keys = torch.full((cap,), -1, dtype=torch.int64, device="cuda")   # all ones: a destination no rank has
        a = A.MigrateDestArgs()
        a.d_pos, a.N, a.rank = dev["pos"].data_ptr(), N, rank
        a.global_lo[:], a.global_L[:], a.grid[:], a.decomposed[:] = list(glo), list(gL), list(grid), list(dec)
        a.d_dest, a.d_n_leave, a.d_rank_count, a.n_ranks = dest.data_ptr(), counters.data_ptr(), counters.data_ptr() + 4, world
        a.d_keys, a.capacity = keys.data_ptr(), cap
        A.check(A.lib.hmd_migrate_dest(C.byref(a), None), "hmd_migrate_dest")
        cnt = counters.cpu().numpy()
        n_leave = int(cnt[0])
        assert (n_leave <= cap) == (cap == 2048) and n_leave > 64
        skeys = K.sort_keys(keys, 32 + 3)   # 3 bits hold ranks 0..5 and the padding's 7
        assert (skeys.cpu().numpy()[min(n_leave, cap):] >> 32 & 7 == 7).all()
        slot = int(cnt[1:].max()) + 2
        pk = A.MigratePackArgs()
        pk.arrays, pk.d_keys = p, skeys.data_ptr()
        pk.global_lo[:], pk.global_L[:], pk.decomposed[:] = list(glo), list(gL), list(dec)
        pk.slot_rows, pk.n_ranks = slot, world
        got = torch.full((world * (slot + 1), A.MIGRATE_ROW_WORDS), -7, dtype=torch.int64, device="cuda")
        pk.d_out = got.data_ptr()
        A.check(A.lib.hmd_migrate_pack_slots_dev(C.byref(pk), counters.data_ptr(), counters.data_ptr() + 4, cap, None), "hmd_migrate_pack_slots_dev")
        g = got.cpu().numpy().reshape(world, slot + 1, -1)
        if n_leave <= cap:
            want = torch.full_like(got, -7)
            pk.d_out, pk.n = want.data_ptr(), n_leave
            for r in range(world + 1):
                pk.send_off[r] = int(cnt[1:1 + r].sum())
            A.check(A.lib.hmd_migrate_pack(C.byref(pk), None), "hmd_migrate_pack")
            assert torch.equal(got, want)
        else:
            assert (g[:, 0, 1] == 0xFFFFFFFF).all() and np.array_equal(g[:, 0, 0], cnt[1:])
        words = torch.zeros(2 * world + 1 + world, dtype=torch.int32, device="cuda")
        pub = torch.zeros(64, dtype=torch.int64, device="cuda")
        A.check(A.lib.hmd_migrate_slot_headers_and(got.data_ptr(), slot, world, counters.data_ptr(), 1 + world, words.data_ptr(), pub.data_ptr(), 77, None), "headers")
        assert np.array_equal(pub.cpu().numpy()[:len(words)], (77 << 32) | words.cpu().numpy().view(np.uint32).astype(np.int64))
        w = words.cpu().numpy().view(np.uint32)
        assert np.array_equal(w[0:2 * world:2], cnt[1:]) and np.array_equal(w[2 * world:], cnt.view(np.uint32))
        assert (w[1:2 * world:2] == (0xFFFFFFFF if n_leave > cap else cnt[1:].max())).all()

    # ---- ghost layer: rows per direction -> rows per destination rank, messages with room from last time ----
    n_dir, world = 7, 4
    dir_dest = np.array([0, 0, 1, 1, 1, 3, 3], dtype=np.int32)      # directions sorted by destination; nothing for rank 2
    dir_count = rng.integers(0, 300, n_dir).astype(np.int32)
    total = int(dir_count.sum())
    Np = 4000
    pos = orc.pack_pos(rng.normal(size=(Np, 3)) * 5, rng.integers(0, 4, Np))
    tag = rng.permutation(Np).astype(np.int32)
    table = rng.normal(size=(n_dir, 3))
    pos_t, tag_t, table_t = torch.from_numpy(pos).cuda(), torch.from_numpy(tag).cuda(), torch.from_numpy(table).cuda()
    rows = np.bincount(dir_dest, weights=dir_count, minlength=world).astype(np.int64)
    for cap_keys, tight in ((total + 500, False), (total + 500, True), (total // 2, False)):
        n_have = min(total, cap_keys)
        send = torch.from_numpy(np.concatenate([rng.integers(0, Np, n_have), np.full(cap_keys - n_have, -1)]).astype(np.int32)).cuda()
        sid = torch.from_numpy(np.concatenate([np.repeat(np.arange(n_dir), dir_count)[:n_have], np.full(cap_keys - n_have, 255)]).astype(np.uint8)).cuda()
        s = A.GhostSlots()
        s.n_ranks = world
        row0 = 0
        for r in range(world):
            s.cap[r] = int(rows[r]) // 2 if (tight and r == 1) else int(rows[r]) + 10
            s.row0[r] = row0
            row0 += s.cap[r] + 1
        counters = torch.from_numpy(np.concatenate([[total], dir_count]).astype(np.int32)).cuda()
        dd = torch.from_numpy(dir_dest).cuda()
        pa = A.PackPosArgs()
        got = torch.full((row0, 4), -3.0, dtype=torch.float64, device="cuda")
        pa.d_out, pa.d_pos, pa.d_send_idx, pa.d_send_shift_id, pa.d_shift_table = got.data_ptr(), pos_t.data_ptr(), send.data_ptr(), sid.data_ptr(), table_t.data_ptr()
        A.check(A.lib.hmd_pack_ghost_init_slots_dev(C.byref(pa), tag_t.data_ptr(), C.byref(s), counters.data_ptr(), counters.data_ptr() + 4, dd.data_ptr(), n_dir,
                                                    cap_keys, None, None), "hmd_pack_ghost_init_slots_dev")
        # the same with the send list taken from sorted keys (and stored on the way)
        skeys = ((sid.to(torch.int64) << 32) | (send.to(torch.int64) & 0xFFFFFFFF)).contiguous()
        send2, sid2, got2 = torch.zeros_like(send), torch.zeros_like(sid), torch.full_like(got, -3.0)
        pa.d_out, pa.d_send_idx, pa.d_send_shift_id = got2.data_ptr(), send2.data_ptr(), sid2.data_ptr()
        A.check(A.lib.hmd_pack_ghost_init_slots_dev(C.byref(pa), tag_t.data_ptr(), C.byref(s), counters.data_ptr(), counters.data_ptr() + 4, dd.data_ptr(), n_dir,
                                                    cap_keys, skeys.data_ptr(), None), "hmd_pack_ghost_init_slots_dev")
        assert torch.equal(got2.view(torch.int64), got.view(torch.int64))
        assert torch.equal(send2[:n_have], send[:n_have]) and torch.equal(sid2[:n_have], sid[:n_have])
        pa.d_out, pa.d_send_idx, pa.d_send_shift_id = got.data_ptr(), send.data_ptr(), sid.data_ptr()
        flag = (1 if tight else 0) | (2 if total > cap_keys else 0)
        hdr = got.cpu().numpy().view(np.uint64)
        for r in range(world):
            assert int(hdr[s.row0[r], 0]) == rows[r] and int(hdr[s.row0[r], 1]) == flag
        if total <= cap_keys:
            want = torch.full_like(got, -3.0)
            pa.d_out, pa.n_send = want.data_ptr(), total
            s.flag = flag
            for r in range(world + 1):
                s.off[r] = int(rows[:r].sum())
            A.check(A.lib.hmd_pack_ghost_init_slots(C.byref(pa), tag_t.data_ptr(), C.byref(s), None), "hmd_pack_ghost_init_slots")
            assert torch.equal(got.view(torch.int64), want.view(torch.int64))
        words = torch.zeros(2 * world + 1 + n_dir, dtype=torch.int32, device="cuda")
        A.check(A.lib.hmd_ghost_slot_headers_and(got.data_ptr(), C.byref(s), counters.data_ptr(), 1 + n_dir, words.data_ptr(), None, 0, None), "headers")
        w = words.cpu().numpy()
        assert np.array_equal(w[0:2 * world:2], rows) and (w[1:2 * world:2] == flag).all() and np.array_equal(w[2 * world:], counters.cpu().numpy())
    assert A.lib.hmd_migrate_pack_slots_dev(None, None, None, 0, None) == A.ERR_INVALID_ARGUMENT
    # counters to zero and keys to the padding in one launch
    c = torch.full((9,), 5, dtype=torch.int32, device="cuda")
    k = torch.zeros(1000, dtype=torch.int64, device="cuda")
    A.check(A.lib.hmd_comm_prep(c.data_ptr(), 7, k.data_ptr(), 999, None), "hmd_comm_prep")
    assert c.cpu().tolist() == [0] * 7 + [5, 5] and (k[:999] == -1).all() and int(k[999]) == 0
