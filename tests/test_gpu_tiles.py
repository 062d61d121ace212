"""GPU parity tests of the tile path (LDS-staged neighbor list and pair forces): the expanded tile
rows must be the oracle's neighbor sets exactly, forces/energies/virials within 1e-6 relative (observed
1e-10), and every output of the row-based build (n_neigh, Nmax growth, last_pos) must be identical."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import oracle as orc  # noqa: E402
from test_gpu_kernels import K, A, liquid, boxes, _rel_err, F_RTOL  # noqa: E402,F401


def tile_nlist(K, A, pos_t, N, gbox, obox, r_cut, r_buff, n_ghost=0, ghost_width=(0, 0, 0), capacity=None):
    _, _, rmax, _ = orc.update_rlist(np.asarray(r_cut, dtype=float), r_buff)
    dim = orc.cell_dimensions(obox, rmax + r_buff, 1, ghost_width, 3)
    cl = K.cell_list_auto(pos_t, N, n_ghost, gbox, dim, ghost_width, want_idx=True)
    adj = torch.as_tensor(orc.cell_adj(dim, 1, 3).view(np.int32), device="cuda")
    tl = K.nlist_tiles(pos_t, N, gbox, r_cut, r_buff, cl, adj, capacity=capacity)
    return tl, cl, dim


def expanded_sets(K, tl, N, gbox):
    nlist = K.tiles_to_rows(tl, N, gbox)
    return orc.neighbor_sets(tl["n_neigh"].cpu().numpy().view(np.uint32)[:N], nlist.cpu().numpy().view(np.uint32),
                             tl["head"].cpu().numpy().view(np.uint64)[:N])


@pytest.mark.parametrize("n,ntypes", [(9, 1), (12, 1), (13, 2), (16, 2), (40, 1), (12, 3)])
def test_tile_neighbor_sets_match_oracle(K, A, n, ntypes):
    # 9 -> 3 cells per side (every region wraps onto the whole box), 13 -> odd grid (4 cells, last block half empty)
    xyz, typ, L = liquid(n, ntypes=ntypes)
    N = len(xyz)
    gbox, obox = boxes(A, L)
    r_cut = [[2.5]] if ntypes == 1 else [[2.5, 2.0], [2.0, -1.0]] if ntypes == 2 else [[2.5, 2.0, 2.3], [2.0, -1.0, 1.8], [2.3, 1.8, 2.4]]
    pos = orc.pack_pos(xyz, typ)
    ref = orc.nlist_binned(pos, N, obox, r_cut, 0.4, storage_full=True)
    tl, _, dim = tile_nlist(K, A, torch.from_numpy(pos).cuda(), N, gbox, obox, r_cut, 0.4)
    assert np.array_equal(tl["n_neigh"].cpu().numpy().view(np.uint32)[:N], ref["n_neigh"])
    assert np.array_equal(tl["Nmax"].cpu().numpy().view(np.uint32), ref["Nmax"])
    got = expanded_sets(K, tl, N, gbox)
    want = orc.neighbor_sets(ref["n_neigh"], ref["nlist"], ref["head"])
    for i in range(N):
        assert np.array_equal(got[i], want[i]), i
    assert np.array_equal(tl["last_pos"].cpu().numpy()[:N], pos)


@pytest.mark.parametrize("n,ntypes,height", [(12, 1, 4), (10, 2, 7), (24, 1, 2)])
def test_builds_leave_out_exclusions(K, A, n, ntypes, height):
    """Rows built with exclusions equal the CPU path's build followed by its filter pass
    (hoomd/md/NeighborListGPU.cu:158-223): chain-like exclusions (i, i+1), (i, i+2) as
    hoomd/md/test/test_neighborlist.cc:1070-1090, more partners than the kernels keep in registers in one case, and
    both build kernels (one type: with the fp32 pre-test; two types: the general one)."""
    xyz, typ, L = liquid(n, ntypes=ntypes, seed=17)
    N = len(xyz)
    gbox, obox = boxes(A, L)
    r_cut = [[2.5]] if ntypes == 1 else [[2.5, 2.0], [2.0, 2.2]]
    pos = orc.pack_pos(xyz, typ)
    n_ex = np.zeros(N, dtype=np.uint32)
    ex = np.zeros((height, N), dtype=np.uint32)
    reach = height // 2
    for i in range(N):
        if i % 5 == 4:
            continue  # some particles without exclusions
        for j in range(i + 1, i + 1 + reach):
            if j < N and j % 5 != 4:
                for a_, b_ in ((i, j), (j, i)):
                    if n_ex[a_] < height:
                        ex[n_ex[a_], a_] = b_
                        n_ex[a_] += 1
    ref = orc.nlist_binned(pos, N, obox, r_cut, 0.4, storage_full=True)
    n_neigh, nlist = ref["n_neigh"].copy(), ref["nlist"].copy()
    orc.nlist_filter(N, ref["head"], n_ex, ex, n_neigh, nlist)
    assert (n_neigh < ref["n_neigh"]).sum() > N // 2
    want = orc.neighbor_sets(n_neigh, nlist, ref["head"])
    pos_t = torch.from_numpy(pos).cuda()
    dim = orc.cell_dimensions(obox, 2.9)
    cl = K.cell_list_auto(pos_t, N, 0, gbox, dim, want_idx=True)
    adj = torch.as_tensor(orc.cell_adj(dim).view(np.int32), device="cuda")
    ex_t = (torch.from_numpy(n_ex.view(np.int32)).cuda(), torch.from_numpy(ex.view(np.int32)).cuda())
    tl = K.nlist_tiles(pos_t, N, gbox, r_cut, 0.4, cl, adj, exclusions=ex_t)
    # the row build with the same exclusions
    rows = K.nlist_binned(pos_t, N, gbox, r_cut, 0.4, cl, adj, exclusions=ex_t)
    assert np.array_equal(rows["n_neigh"].cpu().numpy().view(np.uint32)[:N], n_neigh)
    got_rows = orc.neighbor_sets(rows["n_neigh"].cpu().numpy().view(np.uint32)[:N], rows["nlist"].cpu().numpy().view(np.uint32),
                                 rows["head"].cpu().numpy().view(np.uint64)[:N])
    assert all(np.array_equal(g, w) for g, w in zip(got_rows, want))
    assert np.array_equal(tl["n_neigh"].cpu().numpy().view(np.uint32)[:N], n_neigh)
    got = expanded_sets(K, tl, N, gbox)
    for i in range(N):
        assert np.array_equal(got[i], want[i]), i
    # forces from the filtered rows (the pairs left out contribute nothing)
    P = [orc.lj_params(1.0, 1.0)] * (ntypes * ntypes)
    F_ref, _ = orc.pair_forces(orc.EV_LJ, pos, N, obox, n_neigh, nlist, ref["head"], P, np.square(r_cut), compute_virial=False)
    F, _ = K.pair_forces_tiles("lj", pos_t, N, gbox, tl, P, np.square(r_cut), compute_virial=False)
    assert np.abs(F.cpu().numpy()[:N] - F_ref).max() / np.abs(F_ref).max() < 1e-9


def test_tile_triclinic_box(K, A):
    rng = np.random.default_rng(9)
    L, tilt = (14.0, 12.0, 13.0), dict(xy=0.2, xz=-0.1, yz=0.15)
    gbox, obox = A.Box.from_L(*L, **tilt), orc.Box.from_L(*L, **tilt)
    N = 1500
    f = rng.random((N, 3)) - 0.5
    xyz = np.stack([f[:, 0] * L[0] + tilt["xy"] * f[:, 1] * L[1] + tilt["xz"] * f[:, 2] * L[2],
                    f[:, 1] * L[1] + tilt["yz"] * f[:, 2] * L[2], f[:, 2] * L[2]], 1)
    pos = orc.pack_pos(xyz, np.zeros(N, int))
    ref = orc.nlist_binned(pos, N, obox, [[2.0]], 0.3, storage_full=True)
    pos_t = torch.from_numpy(pos).cuda()
    tl, _, _ = tile_nlist(K, A, pos_t, N, gbox, obox, [[2.0]], 0.3)
    got, want = expanded_sets(K, tl, N, gbox), orc.neighbor_sets(ref["n_neigh"], ref["nlist"], ref["head"])
    assert all(np.array_equal(a, b) for a, b in zip(got, want))
    half = orc.nlist_binned(pos, N, obox, [[2.0]], 0.3, storage_full=False)
    # random gas: clip the force of near-overlapping pairs out of the comparison by using a soft potential
    F_ref, V_ref = orc.pair_forces(orc.EV_EXAMPLE, pos, N, obox, half["n_neigh"], half["nlist"], half["head"], [(2.0, 2.0)], [[4.0]],
                                   third_law=True, compute_virial=True)
    F, V = K.pair_forces_tiles("example", pos_t, N, gbox, tl, [(2.0, 2.0)], [[4.0]], compute_virial=True)
    assert _rel_err(F.cpu().numpy(), F_ref) < F_RTOL * 1e-4
    assert _rel_err(V.cpu().numpy(), V_ref) < F_RTOL * 1e-4


def test_tile_ghost_layer_non_periodic_box(K, A):
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
    pos_t = torch.from_numpy(pos).cuda()
    tl, _, _ = tile_nlist(K, A, pos_t, N, gbox, obox, [[2.0]], 0.4, n_ghost=ng, ghost_width=gw)
    got, want = expanded_sets(K, tl, N, gbox), orc.neighbor_sets(ref["n_neigh"], ref["nlist"], ref["head"])
    assert all(np.array_equal(a, b) for a, b in zip(got, want))
    F_ref, _ = orc.pair_forces(orc.EV_EXAMPLE, pos, N, obox, ref["n_neigh"], ref["nlist"], ref["head"], [(1.5, 2.0)], [[4.0]],
                               third_law=False, compute_virial=False)
    for fast in (False, True):
        F, _ = K.pair_forces_tiles("example", pos_t, N, gbox, tl, [(1.5, 2.0)], [[4.0]], compute_virial=False, fast=fast)
        assert _rel_err(F.cpu().numpy()[:N], F_ref[:N]) < F_RTOL * 1e-4


@pytest.mark.parametrize("shift_mode", [0, 1, 2])
@pytest.mark.parametrize("ntypes", [1, 2, 3])
def test_tile_lj_forces_match_oracle(K, A, shift_mode, ntypes):
    # (rows for two and three types carry the neighbors' types in their entries; two types take the kernels that keep both
    # parameter sets in registers, three the ones with tables in LDS)
    xyz, typ, L = liquid(14, ntypes=ntypes, seed=3 + ntypes)
    N = len(xyz)
    gbox, obox = boxes(A, L)
    pos = orc.pack_pos(xyz, typ)
    if ntypes == 1:
        r_cut, params, ron = [[2.5]], [orc.lj_params(1.0, 1.0)], [[2.0]]
    elif ntypes == 3:
        r_cut = [[2.5, 2.0, 2.3], [2.0, 2.2, 1.8], [2.3, 1.8, 2.4]]
        eps = [[1.0, 0.8, 1.2], [0.8, 0.88, 0.6], [1.2, 0.6, 0.9]]
        sig = [[1.0, 1.5, 0.9], [1.5, 0.5, 1.1], [0.9, 1.1, 0.95]]
        params = [orc.lj_params(eps[i][j], sig[i][j]) for i in range(3) for j in range(3)]
        ron = [[2.0, 1.5, 2.0], [1.5, 2.4, 1.2], [2.0, 1.2, 2.0]]
    else:
        r_cut = [[2.5, 2.0], [2.0, 2.2]]
        params = [orc.lj_params(1.0, 1.0), orc.lj_params(0.8, 1.5), orc.lj_params(0.8, 1.5), orc.lj_params(0.88, 0.5)]
        ron = [[2.0, 1.5], [1.5, 2.4]]
    rcutsq, ronsq = np.square(r_cut), np.square(ron)
    pos_t = torch.from_numpy(pos).cuda()
    tl, _, _ = tile_nlist(K, A, pos_t, N, gbox, obox, r_cut, 0.4)
    half = orc.nlist_binned(pos, N, obox, r_cut, 0.4, storage_full=False)
    F_ref, V_ref = orc.pair_forces(orc.EV_LJ, pos, N, obox, half["n_neigh"], half["nlist"], half["head"], params, rcutsq,
                                   ronsq, shift_mode=shift_mode, third_law=True, compute_virial=True)
    for tpp, fast in ((4, False), (0, False), (0, True), (4, True)):
        F, V = K.pair_forces_tiles("lj", pos_t, N, gbox, tl, params, rcutsq, ronsq, shift_mode=shift_mode, compute_virial=True, tpp=tpp, fast=fast)
        assert _rel_err(F.cpu().numpy()[:, :3], F_ref[:, :3]) < F_RTOL * 1e-4
        assert _rel_err(F.cpu().numpy()[:, 3], F_ref[:, 3]) < F_RTOL * 1e-4
        assert _rel_err(V.cpu().numpy(), V_ref) < F_RTOL * 1e-4
    F2, _ = K.pair_forces_tiles("lj", pos_t, N, gbox, tl, params, rcutsq, ronsq, shift_mode=shift_mode, compute_virial=False)
    assert _rel_err(F2.cpu().numpy(), F_ref) < F_RTOL * 1e-4


def test_group_numbering_of_grids_too_large_for_the_lds(K, A, monkeypatch):
    """Grids with more blocks than the numbering workgroup keeps in LDS get their group counts from a launch of their
    own (tile_block_counts_kernel); forced here on a small grid: same numbering, same rows."""
    xyz, typ, L = liquid(14, seed=9)
    N = len(xyz)
    gbox, obox = boxes(A, L)
    pos_t = torch.from_numpy(orc.pack_pos(xyz, typ)).cuda()
    dim = orc.cell_dimensions(obox, 2.9, 1, (0, 0, 0), 3)
    adj = torch.as_tensor(orc.cell_adj(dim, 1, 3).view(np.int32), device="cuda")
    out = []
    for limit in (None, "0"):
        if limit is not None:
            monkeypatch.setenv("HOOMD_AMD_BG_LDS_LIMIT", limit)
        cl = K.cell_list_auto(pos_t, N, 0, gbox, dim, want_idx=True)
        tl = K.nlist_tiles(pos_t, N, gbox, [[2.5]], 0.4, cl, adj)
        out.append((tl["block_group"].cpu().numpy().copy(), tl["n_neigh"].cpu().numpy().copy(), expanded_sets(K, tl, N, gbox)))
    assert out[0][0][-1] > 0 and np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    assert all(np.array_equal(x, y) for x, y in zip(out[0][2], out[1][2])) and len(out[0][2]) == N


@pytest.mark.parametrize("ntypes", [1, 2, 3])
def test_tile_forces_with_densely_filled_cells(K, A, ntypes):
    """More than 32 members in most cells (density 1.3: the staging takes the members past the 32nd of a cell in extra
    trips) and rows longer than the pieces fetched a group ahead: forces, energies and virials against the oracle."""
    xyz, typ, L = liquid(16, rho=1.3, jitter=0.08, ntypes=ntypes, seed=40 + ntypes)
    N = len(xyz)
    gbox, obox = boxes(A, L)
    pos = orc.pack_pos(xyz, typ)
    sig = [[1.0, 0.8, 0.9], [0.8, 0.88, 0.85], [0.9, 0.85, 0.95]]
    eps = [[1.0, 1.5, 1.2], [1.5, 0.5, 0.7], [1.2, 0.7, 0.9]]
    r_cut = [[2.5 * sig[i][j] for j in range(ntypes)] for i in range(ntypes)]
    params = [orc.lj_params(eps[i][j], sig[i][j]) for i in range(ntypes) for j in range(ntypes)]
    rcutsq = np.square(r_cut)
    pos_t = torch.from_numpy(pos).cuda()
    tl, cl, _ = tile_nlist(K, A, pos_t, N, gbox, obox, r_cut, 0.4)
    assert (cl["cell_size"].cpu().numpy() > 32).mean() > 0.3
    assert tl["n_neigh"].cpu().numpy().view(np.uint32)[:N].max() > (128 if ntypes == 1 else 96)  # (longer than the pieces fetched a group ahead)
    half = orc.nlist_binned(pos, N, obox, r_cut, 0.4, storage_full=False)
    F_ref, V_ref = orc.pair_forces(orc.EV_LJ, pos, N, obox, half["n_neigh"], half["nlist"], half["head"], params, rcutsq,
                                   np.zeros_like(rcutsq), shift_mode=0, third_law=True, compute_virial=True)
    F, V = K.pair_forces_tiles("lj", pos_t, N, gbox, tl, params, rcutsq, np.zeros_like(rcutsq), shift_mode=0, compute_virial=True)
    F2, _ = K.pair_forces_tiles("lj", pos_t, N, gbox, tl, params, rcutsq, np.zeros_like(rcutsq), shift_mode=0, compute_virial=False)
    # At this density single pair terms are 1e4 to 1e5 times a net force component, and the order of the sum follows the
    # order of a cell's members, which the cell list's slot counters decide anew in every run: rounding alone moves a
    # component by 1e-11 to 1e-10 of its value, so the bound here is 1e-9 (a missed or doubled pair shows at 1e-5).
    tol = F_RTOL * 1e-3
    assert _rel_err(F.cpu().numpy(), F_ref) < tol and _rel_err(V.cpu().numpy(), V_ref) < tol
    assert _rel_err(F2.cpu().numpy(), F_ref) < tol


@pytest.mark.parametrize("jitter", [0.0, 0.05])
def test_tile_rows_laid_out_by_type(K, A, jitter, monkeypatch):
    """Two types: the build lays a row out as [type 0 ..., dummy up to a multiple of four, type 1 ...] and the force kernel
    chooses its parameters per chunk of four entries.  An 80:20 mixture at density 1.2 with three cutoffs, on the lattice
    (every row of a type as long as the longest: the rows end exactly at their capacity; 37 particles per cell: more
    candidates than one pass of the build holds, so it sweeps once per type) and off it: neighbor sets and counts are the
    oracle's, the capacities too (the dummy entries do not count), forces to 1e-10 of the largest with either kernel."""
    n, rho = 20, 1.2
    a = (1.0 / rho) ** (1.0 / 3.0)
    L = n * a
    g = (np.arange(n) + 0.5) * a - L / 2
    xyz = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    rng = np.random.default_rng(3)
    if jitter:
        xyz = xyz + rng.normal(0, jitter, xyz.shape)
    xyz = (xyz + L / 2) % L - L / 2
    typ = (rng.random(len(xyz)) < 0.2).astype(int)
    N = len(xyz)
    gbox, obox = boxes(A, L)
    pos = orc.pack_pos(xyz, typ)
    r_cut = [[2.5, 2.0], [2.0, 2.2]]
    P = [orc.lj_params(1.0, 1.0), orc.lj_params(0.8, 1.5), orc.lj_params(0.8, 1.5), orc.lj_params(0.88, 0.5)]
    pos_t = torch.from_numpy(pos).cuda()
    tl, cl, _ = tile_nlist(K, A, pos_t, N, gbox, obox, r_cut, 0.4)
    assert tl["geom"].rows_by_type == 1
    assert 27 * cl["cell_size"].cpu().numpy().max() > 14 * 64  # (more than one pass)
    ref = orc.nlist_binned(pos, N, obox, r_cut, 0.4, storage_full=True)
    assert np.array_equal(tl["n_neigh"].cpu().numpy().view(np.uint32)[:N], ref["n_neigh"])
    assert np.array_equal(tl["Nmax"].cpu().numpy().view(np.uint32), ref["Nmax"])
    sets = expanded_sets(K, tl, N, gbox)
    rsets = orc.neighbor_sets(ref["n_neigh"], ref["nlist"], ref["head"])
    assert all(np.array_equal(s, r) for s, r in zip(sets, rsets))
    # what the rows hold: per home particle the entries counted in the slot table = listed neighbors + the dummy entries
    lay = tl["slot_meta"].cpu().numpy().view(np.uint32)[tl["slot_of"].cpu().numpy()[:N], 1]
    extra = lay.astype(np.int64) - ref["n_neigh"]
    assert extra.min() >= 0 and extra.max() <= 3 and extra.max() > 0
    half = orc.nlist_binned(pos, N, obox, r_cut, 0.4, storage_full=False)
    rcutsq = np.square(r_cut)
    F_ref, _ = orc.pair_forces(orc.EV_LJ, pos, N, obox, half["n_neigh"], half["nlist"], half["head"], P, rcutsq, np.zeros_like(rcutsq),
                               shift_mode=0, third_law=True, compute_virial=False)
    scale = np.abs(F_ref).max()
    for per_pair in (False, True):
        if per_pair:
            monkeypatch.setenv("HOOMD_AMD_PFT_NO_BY_TYPE", "1")  # the kernel that reads the type of every entry
        F, _ = K.pair_forces_tiles("lj", pos_t, N, gbox, tl, P, rcutsq, np.zeros_like(rcutsq), shift_mode=0, compute_virial=False)
        assert np.abs(F.cpu().numpy() - F_ref).max() < 1e-10 * scale


def test_tile_path_limits_are_reported(K, A):
    gbox, _ = boxes(A, 5.0)
    xyz = (np.random.default_rng(1).random((50, 3)) - 0.5) * 5.0
    pos_t = torch.from_numpy(orc.pack_pos(xyz, np.zeros(50, int))).cuda()
    cl = K.cell_list_auto(pos_t, 50, 0, gbox, (2, 2, 2), want_idx=True)
    with pytest.raises(A.KernelError, match="at least 3 cells"):
        K.tile_geom(cl, gbox)
    # a tile smaller than the densest region: the build reports it instead of writing out of bounds
    xyz, typ, L = liquid(12)
    gbox, obox = boxes(A, L)
    pos_t = torch.from_numpy(orc.pack_pos(xyz, typ)).cuda()
    with pytest.raises(A.KernelError, match="capacity"):
        tile_nlist(K, A, pos_t, len(xyz), gbox, obox, [[2.5]], 0.4, capacity=600)


def test_tile_forces_after_particles_wrap_between_builds(K, A):
    """Between two builds the integrator wraps particles that leave the box: they jump by a box vector
    while keeping their cell-list slot.  The tile kernel must see them in the right image."""
    xyz, typ, L = liquid(12, seed=8)
    N = len(xyz)
    gbox, obox = boxes(A, L)
    pos = orc.pack_pos(xyz, typ)
    pos_t = torch.from_numpy(pos).cuda()
    tl, cl, dim = tile_nlist(K, A, pos_t, N, gbox, obox, [[2.5]], 0.4)
    nlist = K.tiles_to_rows(tl, N, gbox)
    rng = np.random.default_rng(3)
    moved = xyz + rng.normal(0, 0.05, xyz.shape)
    near = np.abs(xyz) > L / 2 - 0.45         # push the ones near a face across it
    moved = np.where(near, moved + np.sign(xyz) * 0.5, moved)
    wrapped = (moved + L / 2) % L - L / 2
    assert (np.abs(wrapped - moved) > 1).any()
    pos2_t = torch.from_numpy(orc.pack_pos(wrapped, typ)).cuda()
    P = [orc.lj_params(1.0, 1.0)]
    F_rows, V_rows = K.pair_forces("lj", pos2_t, N, gbox, tl["n_neigh"], nlist, tl["head"], P, [[6.25]], compute_virial=True)
    for fast in (False, True):
        F_tile, V_tile = K.pair_forces_tiles("lj", pos2_t, N, gbox, tl, P, [[6.25]], compute_virial=True, fast=fast)
        assert _rel_err(F_tile.cpu().numpy(), F_rows.cpu().numpy()) < F_RTOL * 1e-4
        assert _rel_err(V_tile.cpu().numpy(), V_rows.cpu().numpy()) < F_RTOL * 1e-4


@pytest.mark.parametrize("fast,wlo,whi,rho", [(False, (1, 0, 1), (-2, -1, -1), 0.84), (True, (1, 0, 1), (-2, -1, -1), 0.84), (True, (0, 1, 0), (-1, -2, -2), 0.84),
                                              (True, (1, 1, 1), (1, 1, 1), 0.84), (True, (1, 0, 1), (-2, -1, -1), 1.3)])
def test_tile_forces_by_block_window(K, A, fast, wlo, whi, rho):
    """The blocks inside a window plus the blocks outside it give, bit for bit, what one launch over all
    blocks gives; each launch writes only the particles of its own blocks.  (Density 1.3: the kernel version with eight
    waves per workgroup.)"""
    xyz, typ, L = liquid(20 if rho < 1 else 24, rho=rho, jitter=0.12 if rho < 1 else 0.08, seed=5)
    N = len(xyz)
    gbox, obox = boxes(A, L)
    pos_t = torch.from_numpy(orc.pack_pos(xyz, typ)).cuda()
    tl, _, dim = tile_nlist(K, A, pos_t, N, gbox, obox, [[2.5]], 0.4)
    nb = [(d + 1) // 2 for d in dim]
    assert min(nb) >= 3
    P = [orc.lj_params(1.0, 1.0)]
    F_all, V_all = K.pair_forces_tiles("lj", pos_t, N, gbox, tl, P, [[6.25]], compute_virial=True, fast=fast)
    lo, hi = wlo, tuple(h if h > 0 else nb[d] + h for d, h in enumerate(whi))
    F_in, V_in = K.pair_forces_tiles("lj", pos_t, N, gbox, tl, P, [[6.25]], compute_virial=True, fast=fast, window=(1, lo, hi))
    written = ~torch.isnan(F_in[:, 0])
    assert 0 < int(written.sum()) < N
    assert torch.equal(F_in[written], F_all[written])
    F, V = K.pair_forces_tiles("lj", pos_t, N, gbox, tl, P, [[6.25]], compute_virial=True, fast=fast, window=(2, lo, hi), into=(F_in.clone(), V_in.clone()))
    assert torch.equal(F, F_all) and torch.equal(V, V_all)
    # the outside launch alone leaves the inside particles untouched
    F_out, _ = K.pair_forces_tiles("lj", pos_t, N, gbox, tl, P, [[6.25]], compute_virial=False, fast=fast, window=(2, lo, hi))
    assert torch.equal(torch.isnan(F_out[:, 0]), written)
    with pytest.raises(A.KernelError, match="window"):
        K.pair_forces_tiles("lj", pos_t, N, gbox, tl, P, [[6.25]], window=(1, (0, 0, 0), (nb[0], 0, 0)))
    with pytest.raises(A.KernelError, match="block_window_mode"):
        K.pair_forces_tiles("lj", pos_t, N, gbox, tl, P, [[6.25]], window=(3, (0, 0, 0), (0, 0, 0)))


@pytest.mark.parametrize("n,ntypes,dt,edges", [(12, 1, 0.005, False), (12, 1, 0.005, True), (16, 1, 0.04, True), (13, 2, 0.005, True)])
def test_tile_force_kernel_carries_the_integration_pass(K, A, n, ntypes, dt, edges):
    """hmd_compute_pair_forces_lj_tiles_step against the two entry points it joins: forces, positions (with particles
    wrapped through the box faces), velocities, images, the slot-ordered copy and the distance check, bit for bit; and a
    launch that its predicate switches off leaves everything alone."""
    xyz, typ, L = liquid(n, ntypes=ntypes, seed=21)
    N = len(xyz)
    gbox, obox = boxes(A, L)
    rng = np.random.default_rng(6)
    pos = orc.pack_pos(xyz, typ)
    # some particles right at the faces, moving outwards (they overlap others: large forces, the list goes stale)
    edge = rng.permutation(N)[: 40 if edges else 0]
    axis = rng.integers(0, 3, len(edge))
    side = rng.choice([-1.0, 1.0], len(edge))
    pos[edge, axis] = side * (L / 2 - 1e-5)
    pos_t = torch.from_numpy(pos).cuda()
    r_cut = [[2.5]] if ntypes == 1 else [[2.5, 2.0], [2.0, 2.2]]
    tl, cl, dim = tile_nlist(K, A, pos_t, N, gbox, obox, r_cut, 0.4)
    vel = np.concatenate([rng.normal(0, 1.3, (N, 3)), rng.uniform(0.5, 2.0, (N, 1)) if ntypes == 2 else np.ones((N, 1))], 1)
    vel[edge, axis] = side * 1.5
    image = rng.integers(-2, 3, (N, 3)).astype(np.int32)
    P = [orc.lj_params(1.0, 1.0)] if ntypes == 1 else [orc.lj_params(1.0, 1.0), orc.lj_params(1.5, 0.8), orc.lj_params(1.5, 0.8), orc.lj_params(0.5, 0.88)]
    rcsq = (np.asarray(r_cut) ** 2).tolist()
    rcut_max = [max(r) for r in r_cut]
    # the two entry points one after the other
    p1, v1, i1 = pos_t.clone(), torch.from_numpy(vel).cuda(), torch.from_numpy(image).cuda()
    F1, _ = K.pair_forces_tiles("lj", p1, N, gbox, tl, P, rcsq, compute_virial=False, fast=True)
    cpos1 = K.tile_refresh(p1, tl)
    flag1, _ = K.fused_step(p1, v1, None, i1, F1, gbox, dt, last_pos=tl["last_pos"], rcut_max=rcut_max, r_buff=0.4, checkn=7, slot_of=tl["slot_of"], cpos=cpos1)
    assert not edges or (i1.cpu().numpy() != image).any()  # particles did cross the faces
    # the joined kernel
    p2, v2, i2 = pos_t.clone(), torch.from_numpy(vel).cuda(), torch.from_numpy(image).cuda()
    F2, cpos2, flag2 = K.pair_forces_tiles_step(p2, v2, i2, N, gbox, tl, P, rcsq, dt, last_pos=tl["last_pos"], rcut_max=rcut_max, r_buff=0.4, checkn=7)
    assert np.array_equal(F2.cpu().numpy()[:N], F1.cpu().numpy()[:N])
    assert np.array_equal(p2.cpu().numpy(), p1.cpu().numpy()) and np.array_equal(v2.cpu().numpy(), v1.cpu().numpy())
    assert np.array_equal(i2.cpu().numpy(), i1.cpu().numpy())
    slots = tl["slot_of"].cpu().numpy().view(np.uint32)[:N]
    assert np.array_equal(cpos2.cpu().numpy()[slots], cpos1.cpu().numpy()[slots])
    assert flag2 == flag1
    # switched off: nothing moves
    p3, v3, i3 = pos_t.clone(), torch.from_numpy(vel).cuda(), torch.from_numpy(image).cuda()
    word = torch.full((1,), 5, dtype=torch.int32, device="cuda")
    F3, cpos3, flag3 = K.pair_forces_tiles_step(p3, v3, i3, N, gbox, tl, P, rcsq, dt, last_pos=tl["last_pos"], rcut_max=rcut_max, r_buff=0.4, checkn=7,
                                                skip=(word, 5))
    assert np.array_equal(p3.cpu().numpy(), pos) and np.array_equal(v3.cpu().numpy(), vel) and flag3 == 0
    assert np.isnan(F3.cpu().numpy()).all() and np.isnan(cpos3.cpu().numpy()).all()


def test_step_pass_is_declined_outside_its_case(K, A):
    """A triclinic box, a rescaling factor or a displacement limit: hmd_pair_forces_tiles_step_supported says no (and names
    the reason) instead of running something else than the two entry points would."""
    import ctypes as C
    a, s = A.PairArgs(), A.FusedStepArgs()
    gbox = A.Box.from_L(10.0, 10.0, 10.0)
    a.N = 100
    dummy = torch.zeros(8, dtype=torch.float64, device="cuda")
    s.one.d_pos = s.one.d_vel = s.two.d_vel = s.one.d_image = K.ptr(dummy)
    s.one.group_size = s.two.group_size = 100
    s.one.rescale_factor = s.two.rescale_factor = 1.0
    s.one.deltaT = s.two.deltaT = 0.005
    s.one.box = gbox
    assert A.lib.hmd_pair_forces_tiles_step_supported(C.byref(a), C.byref(s)) == 0
    s.one.rescale_factor = 0.99
    assert A.lib.hmd_pair_forces_tiles_step_supported(C.byref(a), C.byref(s)) != 0 and b"rescal" in A.lib.hmd_last_error()
    s.one.rescale_factor = 1.0
    s.one.box = A.Box.from_L(10.0, 10.0, 10.0, 0.1, 0.0, 0.0)
    assert A.lib.hmd_pair_forces_tiles_step_supported(C.byref(a), C.byref(s)) != 0 and b"orthorhombic" in A.lib.hmd_last_error()
    s.one.box = gbox
    a.compute_virial = 1
    assert A.lib.hmd_pair_forces_tiles_step_supported(C.byref(a), C.byref(s)) != 0
