"""The neighbor-list build with the fp32 pre-test must give, bit for bit, the sets of the exact fp64 test
(hoomd/md/NeighborListBinned.cc:150-157): pairs engineered to sit at r_list, one ulp to either side of it and
all through the band the pre-test cannot decide are compared with the CPU oracle and with the same kernel family
run without the pre-test."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import oracle as orc  # noqa: E402


def same_sets(got, want):
    return len(got) == len(want) and all(np.array_equal(g, w) for g, w in zip(got, want))


@pytest.fixture(scope="module")
def K():
    from hoomd_amd import kernels
    return kernels


@pytest.fixture(scope="module")
def A():
    from hoomd_amd import _abi
    return _abi


def _system(seed, L, n_free, n_pairs, r_list, typ=None):
    """Random particles plus partners placed at r_list * (1 + eps) from some of them, eps from 0 and single ulps up
    to 1e-5, in random directions, wrapped into the box (so some pairs straddle the periodic boundary).  With types
    (typ: type ids of the free particles followed by the partners'), r_list is the matrix and every engineered pair
    sits at the r_list of its own type pair."""
    rng = np.random.default_rng(seed)
    Lb = np.asarray(L, dtype=np.float64)
    free = (rng.random((n_free, 3)) - 0.5) * Lb
    eps = np.concatenate([[0.0], np.array([1, -1, 2, -2, 4, -4]) * 2.220446049250313e-16,
                          [s * 10.0 ** e for e in range(-14, -4) for s in (1.0, -1.0)]])
    src = rng.integers(0, n_free, n_pairs)
    d = rng.normal(size=(n_pairs, 3))
    d /= np.linalg.norm(d, axis=1)[:, None]
    # a third of the directions along an axis: the squared distance is then a single rounded product
    ax = rng.integers(0, 3, n_pairs)
    axis_dir = np.zeros((n_pairs, 3))
    axis_dir[np.arange(n_pairs), ax] = rng.choice([-1.0, 1.0], n_pairs)
    d = np.where((np.arange(n_pairs) % 3 == 0)[:, None], axis_dir, d)
    if typ is not None:
        r_list = np.asarray(r_list)[typ[src], typ[n_free:]]
    dist = r_list * (1.0 + eps[rng.integers(0, len(eps), n_pairs)])
    partners = free[src] + d * dist[:, None]
    xyz = np.concatenate([free, partners])
    xyz = (xyz + Lb / 2) % Lb - Lb / 2
    return xyz


@pytest.mark.parametrize("L,seed", [((21.0, 21.0, 21.0), 1), ((30.0, 12.5, 17.0), 2)])
def test_prefiltered_build_equals_exact_test(K, A, L, seed):
    r_cut, r_buff = 2.5, 0.4
    r_list = r_cut + r_buff
    n_free = int(0.6 * L[0] * L[1] * L[2])
    xyz = _system(seed, L, n_free, 4000, r_list)
    N = len(xyz)
    pos = orc.pack_pos(xyz, np.zeros(N, dtype=int))
    obox, gbox = orc.Box.from_L(*L), A.Box.from_L(*L)
    dim = orc.cell_dimensions(obox, r_list)
    assert min(dim) >= 3
    ref = orc.nlist_binned(pos, N, obox, [[r_cut]], r_buff, storage_full=True)
    ref_sets = orc.neighbor_sets(ref["n_neigh"], ref["nlist"], ref["head"])
    # the engineered pairs do sit where the pre-test cannot decide: some are neighbors, some are not
    d2 = []
    for i in range(n_free, N, 7):
        dx = xyz[:n_free] - xyz[i]
        dx -= np.asarray(L) * np.rint(dx / np.asarray(L))
        r2 = np.einsum("ij,ij->i", dx, dx)
        d2.append(r2[np.argmin(np.abs(r2 - r_list ** 2))])
    d2 = np.array(d2)
    assert (np.abs(d2 - r_list ** 2) < 1e-9).sum() > 50 and (d2 <= r_list ** 2).any() and (d2 > r_list ** 2).any()

    pos_t = torch.from_numpy(pos).cuda()
    cl = K.cell_list_auto(pos_t, N, 0, gbox, dim, want_idx=True)
    adj = torch.as_tensor(orc.cell_adj(dim).view(np.int32), device="cuda")

    def sets_of(nl):
        return orc.neighbor_sets(nl["n_neigh"].cpu().numpy().view(np.uint32)[:N], nl["nlist"].cpu().numpy().view(np.uint32),
                                 nl["head"].cpu().numpy().view(np.uint64)[:N])

    # "exact": the general kernel (several particle types, body filter, triclinic boxes), chosen here by the
    # environment switch.  It shifts the candidates of a periodic image once when it loads them and decides pairs
    # whose squared distance comes within rounding of r_list^2 again with the CPU path's (p_i - p_j) - L.
    results = {}
    for mode in ("pretest", "exact"):
        if mode == "exact":
            os.environ["HOOMD_AMD_NLIST_EXACT_ONLY"] = "1"
        try:
            rows = K.nlist_binned(pos_t, N, gbox, [[r_cut]], r_buff, cl, adj)
            tl = K.nlist_tiles(pos_t, N, gbox, [[r_cut]], r_buff, cl, adj)
            tl["nlist"] = K.tiles_to_rows(tl, N, gbox)
        finally:
            os.environ.pop("HOOMD_AMD_NLIST_EXACT_ONLY", None)
        results[mode] = (sets_of(rows), sets_of(tl))
        assert np.array_equal(rows["n_neigh"].cpu().numpy().view(np.uint32)[:N], ref["n_neigh"]), mode
        assert np.array_equal(tl["n_neigh"].cpu().numpy().view(np.uint32)[:N], ref["n_neigh"]), mode
        assert np.array_equal(rows["last_pos"].cpu().numpy()[:N, :3], pos[:, :3]) and np.array_equal(tl["last_pos"].cpu().numpy()[:N, :3], pos[:, :3])
    for mode, (s_rows, s_tiles) in results.items():
        assert same_sets(s_rows, ref_sets), mode
        assert same_sets(s_tiles, ref_sets), mode


def test_prefilter_handles_coincident_particles_and_disabled_pairs(K, A):
    """Two particles at the same position are each other's neighbors (the particle itself is removed by its place in
    the candidate list, not by a zero distance); r_cut <= 0 switches the pair off."""
    L = (12.0, 12.0, 12.0)
    rng = np.random.default_rng(5)
    xyz = (rng.random((1200, 3)) - 0.5) * np.asarray(L)
    xyz[1] = xyz[0]
    xyz[700] = xyz[0]
    N = len(xyz)
    pos = orc.pack_pos(xyz, np.zeros(N, dtype=int))
    obox, gbox = orc.Box.from_L(*L), A.Box.from_L(*L)
    dim = orc.cell_dimensions(obox, 2.9)
    pos_t = torch.from_numpy(pos).cuda()
    cl = K.cell_list_auto(pos_t, N, 0, gbox, dim, want_idx=True)
    adj = torch.as_tensor(orc.cell_adj(dim).view(np.int32), device="cuda")
    for r_cut in (2.5, -1.0):
        ref = orc.nlist_binned(pos, N, obox, [[r_cut]], 0.4, storage_full=True)
        # the cell width is fixed by the caller's cell list, as in the reference when one consumer turns its pair off
        nl = K.nlist_binned(pos_t, N, gbox, [[r_cut]], 0.4, cl, adj)
        got = orc.neighbor_sets(nl["n_neigh"].cpu().numpy().view(np.uint32)[:N], nl["nlist"].cpu().numpy().view(np.uint32),
                                nl["head"].cpu().numpy().view(np.uint64)[:N])
        assert same_sets(got, orc.neighbor_sets(ref["n_neigh"], ref["nlist"], ref["head"]))
        if r_cut > 0:
            assert 1 in got[0] and 700 in got[0] and 0 in got[1] and 0 not in got[0]
        else:
            assert all(len(s) == 0 for s in got)


@pytest.mark.parametrize("seed", [3, 4])
def test_prefiltered_build_with_two_types(K, A, seed):
    """Two particle types: every type pair has its own r_list (and its own undecided band); engineered pairs sit at
    the r_list of their own pair, one pair of types is switched off in the second case."""
    L = (20.0, 17.0, 23.0)
    r_buff = 0.4
    r_cut = np.array([[2.5, 2.0], [2.0, 2.2]]) if seed == 3 else np.array([[2.5, 1.9], [1.9, -1.0]])
    r_list = np.where(r_cut > 0, r_cut + r_buff, 1.7)  # (pairs of the switched-off kind are placed somewhere inside the cell width)
    n_free, n_pairs = int(0.6 * L[0] * L[1] * L[2]), 4000
    rng = np.random.default_rng(100 + seed)
    typ = rng.integers(0, 2, n_free + n_pairs)
    xyz = _system(seed, L, n_free, n_pairs, r_list, typ)
    N = len(xyz)
    pos = orc.pack_pos(xyz, typ)
    obox, gbox = orc.Box.from_L(*L), A.Box.from_L(*L)
    dim = orc.cell_dimensions(obox, 2.9)
    ref = orc.nlist_binned(pos, N, obox, r_cut, r_buff, storage_full=True)
    ref_sets = orc.neighbor_sets(ref["n_neigh"], ref["nlist"], ref["head"])
    pos_t = torch.from_numpy(pos).cuda()
    cl = K.cell_list_auto(pos_t, N, 0, gbox, dim, want_idx=True)
    adj = torch.as_tensor(orc.cell_adj(dim).view(np.int32), device="cuda")
    rows = K.nlist_binned(pos_t, N, gbox, r_cut, r_buff, cl, adj)
    tl = K.nlist_tiles(pos_t, N, gbox, r_cut, r_buff, cl, adj)
    tl["nlist"] = K.tiles_to_rows(tl, N, gbox)
    for nl in (rows, tl):
        assert np.array_equal(nl["n_neigh"].cpu().numpy().view(np.uint32)[:N], ref["n_neigh"])
        got = orc.neighbor_sets(nl["n_neigh"].cpu().numpy().view(np.uint32)[:N], nl["nlist"].cpu().numpy().view(np.uint32),
                                nl["head"].cpu().numpy().view(np.uint64)[:N])
        assert same_sets(got, ref_sets)
    assert np.array_equal(np.sort(tl["Nmax"].cpu().numpy().view(np.uint32)), np.sort(ref["Nmax"]))


@pytest.mark.parametrize("L,seed,excl", [((20.0, 17.0, 23.0), 7, False), ((9.5, 14.0, 6.2), 8, True), ((31.0, 5.0, 12.0), 9, True)])
def test_block_build_for_sparse_cells(K, A, L, seed, excl):
    """Short cutoff, about three particles per cell: the build that gives a wave a block of 2x2x2 cells and the 4x4x4
    cells around it as candidates.  Same sets as the CPU path, pairs engineered onto r_list included, odd grids and a
    grid of three cells along one axis (the region then holds two images of the same cell), with and without
    exclusions, as index rows and as tile rows."""
    r_cut, r_buff = 2.0 ** (1.0 / 6.0), 0.4
    r_list = r_cut + r_buff
    n_free = int(0.6 * L[0] * L[1] * L[2])
    xyz = _system(seed, L, n_free, n_free // 4, r_list)
    N = len(xyz)
    pos = orc.pack_pos(xyz, np.zeros(N, dtype=int))
    obox, gbox = orc.Box.from_L(*L), A.Box.from_L(*L)
    dim = orc.cell_dimensions(obox, r_list)
    assert min(dim) >= 3
    ref = orc.nlist_binned(pos, N, obox, [[r_cut]], r_buff, storage_full=True)
    n_neigh, nlist = ref["n_neigh"].copy(), ref["nlist"].copy()
    ex_t = None
    if excl:
        n_ex = np.zeros(N, dtype=np.uint32)
        ex = np.zeros((2, N), dtype=np.uint32)
        for i in range(0, N - 1, 3):  # "bonds" between consecutive particles and between engineered partners
            j = i + 1
            ex[n_ex[i], i], ex[n_ex[j], j] = j, i
            n_ex[i] += 1
            n_ex[j] += 1
        orc.nlist_filter(N, ref["head"], n_ex, ex, n_neigh, nlist)
        ex_t = (torch.from_numpy(n_ex.view(np.int32)).cuda(), torch.from_numpy(ex.view(np.int32)).cuda())
    want = orc.neighbor_sets(n_neigh, nlist, ref["head"])
    pos_t = torch.from_numpy(pos).cuda()
    cl = K.cell_list_auto(pos_t, N, 0, gbox, dim, want_idx=True)
    assert cl["Nmax"] <= 12  # the build under test is the one chosen
    adj = torch.as_tensor(orc.cell_adj(dim).view(np.int32), device="cuda")
    rows = K.nlist_binned(pos_t, N, gbox, [[r_cut]], r_buff, cl, adj, exclusions=ex_t)
    tl = K.nlist_tiles(pos_t, N, gbox, [[r_cut]], r_buff, cl, adj, exclusions=ex_t)
    tl["nlist"] = K.tiles_to_rows(tl, N, gbox)
    for nl in (rows, tl):
        assert np.array_equal(nl["n_neigh"].cpu().numpy().view(np.uint32)[:N], n_neigh)
        got = orc.neighbor_sets(nl["n_neigh"].cpu().numpy().view(np.uint32)[:N], nl["nlist"].cpu().numpy().view(np.uint32),
                                nl["head"].cpu().numpy().view(np.uint64)[:N])
        assert same_sets(got, want)
        assert np.array_equal(nl["last_pos"].cpu().numpy()[:N, :3], pos[:, :3])
    # the same through the kernel that walks cell by cell
    os.environ["HOOMD_AMD_NLIST_NO_BLOCK_BUILD"] = "1"
    try:
        rows2 = K.nlist_binned(pos_t, N, gbox, [[r_cut]], r_buff, cl, adj, exclusions=ex_t)
    finally:
        os.environ.pop("HOOMD_AMD_NLIST_NO_BLOCK_BUILD", None)
    assert np.array_equal(rows2["n_neigh"].cpu().numpy(), rows["n_neigh"].cpu().numpy())


def _wrap_triclinic(xyz, L, xy, xz, yz):
    """Into the box spanned by a1 = (Lx, 0, 0), a2 = (xy Ly, Ly, 0), a3 = (xz Lz, yz Lz, Lz), centred on the origin."""
    h = np.array([[L[0], xy * L[1], xz * L[2]], [0.0, L[1], yz * L[2]], [0.0, 0.0, L[2]]])
    f = np.linalg.solve(h, xyz.T).T + 0.5
    f -= np.floor(f)
    f = np.clip(f, 1e-9, 1.0 - 1e-9)
    return (h @ (f - 0.5).T).T


@pytest.mark.parametrize("case", ["three_types", "body_filter", "triclinic", "two_cells_wide", "triclinic_two_types"])
def test_general_build_decides_pairs_at_r_list_like_the_cpu_path(K, A, case):
    """The build kernel that serves everything the pre-test kernel does not -- three or more particle types, the body
    filter, triclinic boxes (images of adjacent cells fixed up front) and grids with fewer than three cells along a
    periodic direction (minimum image per pair) -- on pairs engineered onto r_list, one ulp to either side of it and
    through 1e-14 ... 1e-5 around it, many of them across periodic faces: the sets are the CPU oracle's
    (hoomd/md/NeighborListBinned.cc:150-157 with hoomd/BoxDim.h:306-372), bit for bit."""
    r_buff = 0.4
    tilt = (0.0, 0.0, 0.0)
    L = (21.0, 18.0, 23.0)
    body = None
    nt = 1
    r_cut = np.array([[2.5]])
    if case == "three_types":
        nt = 3
        r_cut = np.array([[2.5, 2.0, 2.2], [2.0, 1.8, -1.0], [2.2, -1.0, 2.4]])
    elif case == "triclinic":
        tilt = (0.3, -0.2, 0.15)
    elif case == "triclinic_two_types":
        nt, tilt = 2, (-0.25, 0.1, 0.35)
        r_cut = np.array([[2.5, 2.1], [2.1, 1.9]])
    elif case == "two_cells_wide":
        L = (21.0, 18.0, 6.5)   # two cells along z: every cell sees both images of its z neighbor
    r_list = np.where(r_cut > 0, r_cut + r_buff, 1.7)
    n_free, n_pairs = int(0.6 * L[0] * L[1] * L[2]), 4000
    rng = np.random.default_rng(40 + ["three_types", "body_filter", "triclinic", "two_cells_wide", "triclinic_two_types"].index(case))
    typ = rng.integers(0, nt, n_free + n_pairs)
    xyz = _system(11 + len(case), L, n_free, n_pairs, r_list if nt > 1 else float(r_list[0, 0]), typ if nt > 1 else None)
    if any(tilt):
        xyz = _wrap_triclinic(xyz, L, *tilt)
    N = len(xyz)
    if case == "body_filter":
        # rigid-body ids: free particles in bodies of three, every fourth engineered partner in its source's body
        body = np.full(N, 0xFFFFFFFF, dtype=np.uint32)
        body[: n_free - n_free % 3] = np.repeat(np.arange(n_free // 3, dtype=np.uint32), 3)
    pos = orc.pack_pos(xyz, typ if nt > 1 else np.zeros(N, dtype=int))
    obox, gbox = orc.Box.from_L(*L, *tilt), A.Box.from_L(*L, *tilt)
    dim = orc.cell_dimensions(obox, float(r_list.max()) if nt > 1 else 2.9)
    assert (min(dim) >= 3) == (case != "two_cells_wide")
    ref = orc.nlist_binned(pos, N, obox, r_cut, r_buff, storage_full=True, body=body, filter_body=body is not None)
    ref_sets = orc.neighbor_sets(ref["n_neigh"], ref["nlist"], ref["head"])
    # engineered pairs land on both sides of r_list^2 within rounding, also across the periodic faces
    h = np.array([[L[0], tilt[0] * L[1], tilt[1] * L[2]], [0.0, L[1], tilt[2] * L[2]], [0.0, 0.0, L[2]]])
    close = across = 0
    for i in range(n_free, N, 5):
        d = xyz[:n_free] - xyz[i]
        f = np.linalg.solve(h, d.T).T
        wrapped = np.rint(f)
        d = (h @ (f - wrapped).T).T
        r2 = np.einsum("ij,ij->i", d, d)
        rl = r_list[typ[i], typ[:n_free]] if nt > 1 else r_list[0, 0]
        k = np.argmin(np.abs(r2 - rl ** 2))
        if abs(r2[k] - (rl[k] if nt > 1 else rl) ** 2) < 1e-9:
            close += 1
            across += bool(wrapped[k].any())
    assert close > 50 and across > 5

    pos_t = torch.from_numpy(pos).cuda()
    body_t = torch.from_numpy(body.view(np.int32)).cuda() if body is not None else None
    cl = K.cell_list_auto(pos_t, N, 0, gbox, dim, want_idx=True, **({"body": body_t} if body is not None else {}))
    adj = torch.as_tensor(orc.cell_adj(dim).view(np.int32), device="cuda")
    os.environ["HOOMD_AMD_NLIST_EXACT_ONLY"] = "1"  # (one and two types would take the pre-test kernel in orthorhombic boxes)
    try:
        lists = [K.nlist_binned(pos_t, N, gbox, r_cut, r_buff, cl, adj, body=body_t, filter_body=body is not None)]
        if case == "three_types":  # (tile rows exist for orthorhombic boxes)
            tl = K.nlist_tiles(pos_t, N, gbox, r_cut, r_buff, cl, adj)
            tl["nlist"] = K.tiles_to_rows(tl, N, gbox)
            lists.append(tl)
    finally:
        os.environ.pop("HOOMD_AMD_NLIST_EXACT_ONLY", None)
    for nl in lists:
        assert np.array_equal(nl["n_neigh"].cpu().numpy().view(np.uint32)[:N], ref["n_neigh"])
        got = orc.neighbor_sets(nl["n_neigh"].cpu().numpy().view(np.uint32)[:N], nl["nlist"].cpu().numpy().view(np.uint32),
                                nl["head"].cpu().numpy().view(np.uint64)[:N])
        assert same_sets(got, ref_sets)
