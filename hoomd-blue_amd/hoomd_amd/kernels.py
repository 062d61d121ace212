"""Thin torch-tensor front end of the C ABI (include/hoomd_amd.h).

Every function here builds the argument block of one driver and launches it on the current HIP
stream of the tensors' device.  Device memory is plain ``torch`` tensors; nothing in this module
computes anything itself.  Used by the Python-level halo exchange and by the GPU parity tests.
"""
import ctypes as C

import numpy as np
import torch

from . import _abi
from ._abi import Box, Index2D, Index3D, Scalar3, check, ptr


def _stream(t):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _s3(v):
    return Scalar3(float(v[0]), float(v[1]), float(v[2]))


def pack_pos(xyz, typeid=None, device="cuda"):
    """(N,3) float64 + type ids -> (N,4) tensor with the id bit-stuffed into w."""
    xyz = np.asarray(xyz, dtype=np.float64).reshape(-1, 3)
    out = np.zeros((xyz.shape[0], 4), dtype=np.float64)
    out[:, :3] = xyz
    if typeid is not None:
        w = (np.asarray(typeid, dtype=np.int64) & 0xFFFFFFFF)
        out[:, 3] = w.view(np.float64) if w.ndim else w
    return torch.from_numpy(out).to(device)


def int_w(col):
    """Low 32 bits of each float64 of a (host) array."""
    a = np.ascontiguousarray(col, dtype=np.float64)
    return (a.view(np.int64) & 0xFFFFFFFF).astype(np.uint32)


def compute_cell_list(pos, N, n_ghost, box, dim, Nmax, ghost_width=(0, 0, 0), flag_charge=False, flag_type=False,
                      charge=None, body=None, want_type_body=True, want_idx=False):
    dev = pos.device
    ncells = dim[0] * dim[1] * dim[2]
    cell_size = torch.zeros(ncells, dtype=torch.int32, device=dev)
    xyzf = torch.zeros((ncells, Nmax, 4), dtype=torch.float64, device=dev)
    tb = torch.zeros((ncells, Nmax, 2), dtype=torch.int32, device=dev) if want_type_body else None
    idx = torch.zeros((ncells, Nmax), dtype=torch.int32, device=dev) if want_idx else None
    cond = torch.zeros(3, dtype=torch.int32, device=dev)
    a = _abi.CellListArgs()
    a.d_cell_size, a.d_xyzf, a.d_type_body, a.d_cell_idx = ptr(cell_size), ptr(xyzf), ptr(tb), ptr(idx)
    a.d_conditions, a.d_pos, a.d_charge, a.d_body = ptr(cond), ptr(pos), ptr(charge), ptr(body)
    a.N, a.n_ghost, a.Nmax = N, n_ghost, Nmax
    a.flag_charge, a.flag_type = int(flag_charge), int(flag_type)
    a.box = box
    a.ci = Index3D(*dim)
    a.cli = Index2D(Nmax, ncells)
    a.ghost_width = _s3(ghost_width)
    a.clear_cell_size = 1
    check(_abi.lib.hmd_compute_cell_list(C.byref(a), _stream(pos)), "hmd_compute_cell_list")
    return dict(cell_size=cell_size, xyzf=xyzf, type_body=tb, idx=idx, conditions=cond, Nmax=Nmax, dim=tuple(dim))


def cell_list_auto(pos, N, n_ghost, box, dim, ghost_width=(0, 0, 0), **kw):
    """Grow Nmax until nothing overflows (hoomd/CellList.cc:184-199)."""
    ncells = dim[0] * dim[1] * dim[2]
    Nmax = max(1, int(np.ceil((N + n_ghost) / ncells)))
    while True:
        r = compute_cell_list(pos, N, n_ghost, box, dim, Nmax, ghost_width, **kw)
        c = r["conditions"].cpu().numpy().astype(np.uint32)
        if c[0] > Nmax:
            Nmax = int(c[0])
            continue
        return r


def sort_cell_list(cl, flag_is_index=True):
    a = _abi.SortCellListArgs()
    a.d_cell_size, a.d_xyzf, a.d_type_body, a.d_cell_idx = ptr(cl["cell_size"]), ptr(cl["xyzf"]), ptr(cl["type_body"]), ptr(cl["idx"])
    a.ci = Index3D(*cl["dim"])
    a.cli = Index2D(cl["Nmax"], cl["cell_size"].numel())
    a.flag_is_index = int(flag_is_index)
    check(_abi.lib.hmd_sort_cell_list(C.byref(a), _stream(cl["xyzf"])), "hmd_sort_cell_list")


def build_head_list(pos, N, Nmax_t):
    dev = pos.device
    head = torch.zeros(max(N, 1), dtype=torch.int64, device=dev)
    req = torch.zeros(1, dtype=torch.int64, device=dev)
    ws_bytes = _abi.lib.hmd_nlist_head_list_workspace_bytes(N)
    ws = torch.zeros(ws_bytes, dtype=torch.uint8, device=dev)
    a = _abi.HeadListArgs()
    a.d_head_list, a.d_req_size_nlist, a.d_Nmax, a.d_pos = ptr(head), ptr(req), ptr(Nmax_t), ptr(pos)
    a.N, a.ntypes = N, Nmax_t.numel()
    a.d_workspace, a.workspace_bytes = ptr(ws), ws_bytes
    check(_abi.lib.hmd_nlist_build_head_list(C.byref(a), _stream(pos)), "hmd_nlist_build_head_list")
    return head, int(req.item())


def nlist_binned(pos, N, box, r_cut, r_buff, cl, adj, n_ghost=0, ghost_width=(0, 0, 0), body=None, filter_body=False, exclusions=None):
    """Full protocol: head list -> build -> grow Nmax on overflow -> rebuild
    (hoomd/md/NeighborList.cc:406-434).  ``cl`` from compute_cell_list, ``adj`` (ncells, n_adj) int32."""
    dev = pos.device
    r_cut_t = torch.as_tensor(np.asarray(r_cut, dtype=np.float64), device=dev).contiguous()
    nt = r_cut_t.shape[0]
    Nmax_t = torch.full((nt,), 4, dtype=torch.int32, device=dev)
    ncells = cl["cell_size"].numel()
    last_pos = torch.zeros((max(N, 1), 4), dtype=torch.float64, device=dev)
    n_builds = 0
    while True:
        head, total = build_head_list(pos, N, Nmax_t)
        nlist = torch.zeros(max(total, 1), dtype=torch.int32, device=dev)
        n_neigh = torch.zeros(max(N, 1), dtype=torch.int32, device=dev)
        cond = torch.zeros(nt, dtype=torch.int32, device=dev)
        a = _abi.NlistBinnedArgs()
        a.d_nlist, a.d_n_neigh, a.d_last_updated_pos, a.d_conditions = ptr(nlist), ptr(n_neigh), ptr(last_pos), ptr(cond)
        a.d_Nmax, a.d_head_list, a.d_pos, a.d_body, a.N = ptr(Nmax_t), ptr(head), ptr(pos), ptr(body), N
        a.d_cell_size, a.d_cell_xyzf, a.d_cell_type_body, a.d_cell_adj = ptr(cl["cell_size"]), ptr(cl["xyzf"]), ptr(cl["type_body"]), ptr(adj)
        a.ci = Index3D(*cl["dim"])
        a.cli = Index2D(cl["Nmax"], ncells)
        a.cadji = Index2D(adj.shape[1], ncells)
        a.box = box
        a.d_r_cut, a.r_buff, a.ntypes = ptr(r_cut_t), float(r_buff), nt
        a.filter_body = int(filter_body)
        a.ghost_width = _s3(ghost_width)
        if exclusions is not None:  # (n_ex, ex_list) by particle index, ex_list of shape (height, pitch)
            a.d_n_ex_idx, a.d_ex_list_idx = ptr(exclusions[0]), ptr(exclusions[1])
            a.exli = Index2D(exclusions[1].shape[1], exclusions[1].shape[0])
        check(_abi.lib.hmd_compute_nlist_binned(C.byref(a), _stream(pos)), "hmd_compute_nlist_binned")
        n_builds += 1
        c = cond.cpu().numpy().astype(np.int64)
        nm = Nmax_t.cpu().numpy().astype(np.int64)
        over = c > nm
        if over.any():
            # hoomd/md/NeighborList.cc:1601-1617
            nm[over] = np.where(c[over] > 4, (c[over] + 3) & ~3, 4)
            Nmax_t = torch.as_tensor(nm.astype(np.int32), device=dev)
            continue
        break
    return dict(n_neigh=n_neigh, nlist=nlist, head=head, Nmax=Nmax_t, last_pos=last_pos, n_builds=n_builds)


def nlist_check(pos, last_pos, N, box, rcut_max, r_buff, lam=(1.0, 1.0, 1.0), checkn=1):
    dev = pos.device
    res = torch.zeros(1, dtype=torch.int32, device=dev)
    rcm = torch.as_tensor(np.asarray(rcut_max, dtype=np.float64), device=dev)
    a = _abi.NlistCheckArgs()
    a.d_result, a.d_last_pos, a.d_pos, a.N, a.box = ptr(res), ptr(last_pos), ptr(pos), N, box
    a.d_rcut_max, a.r_buff, a.ntypes = ptr(rcm), float(r_buff), rcm.numel()
    a.lambda_min = float(min(lam))
    a.lam = _s3(lam)
    a.checkn = checkn
    check(_abi.lib.hmd_nlist_needs_update_check(C.byref(a), _stream(pos)), "hmd_nlist_needs_update_check")
    return int(res.item())


def nlist_filter(N, head, n_ex, ex_list, n_neigh, nlist):
    """ex_list: (n_ex_max, pitch) int32 with pitch >= N (Index2D(pitch, height))."""
    a = _abi.NlistFilterArgs()
    a.d_n_neigh, a.d_nlist, a.d_head_list, a.d_n_ex, a.d_ex_list = ptr(n_neigh), ptr(nlist), ptr(head), ptr(n_ex), ptr(ex_list)
    a.exli = Index2D(ex_list.shape[1], ex_list.shape[0])
    a.N = N
    check(_abi.lib.hmd_nlist_filter(C.byref(a), _stream(nlist)), "hmd_nlist_filter")


def pair_forces(evaluator, pos, N, box, n_neigh, nlist, head, params, rcutsq, ronsq=None, shift_mode=0,
                compute_virial=True, tpp=0, n_total=None):
    dev = pos.device
    n_total = pos.shape[0] if n_total is None else n_total
    prm = torch.as_tensor(np.asarray(params, dtype=np.float64).reshape(-1, 2), device=dev).contiguous()
    rc = torch.as_tensor(np.asarray(rcutsq, dtype=np.float64), device=dev).contiguous()
    nt = rc.shape[0]
    ro = torch.as_tensor(np.asarray(ronsq if ronsq is not None else np.zeros((nt, nt)), dtype=np.float64), device=dev).contiguous()
    force = torch.full((n_total, 4), float("nan"), dtype=torch.float64, device=dev)
    virial = torch.full((6, n_total), float("nan"), dtype=torch.float64, device=dev) if compute_virial else None
    a = _abi.PairArgs()
    a.d_force, a.d_virial, a.virial_pitch = ptr(force), ptr(virial), n_total
    a.N, a.n_max, a.d_pos, a.box = N, n_total, ptr(pos), box
    a.d_n_neigh, a.d_nlist, a.d_head_list = ptr(n_neigh), ptr(nlist), ptr(head)
    a.d_rcutsq, a.d_ronsq, a.size_neigh_list, a.ntypes = ptr(rc), ptr(ro), nlist.numel(), nt
    a.shift_mode, a.compute_virial, a.threads_per_particle = shift_mode, int(compute_virial), tpp
    fn = {"lj": _abi.lib.hmd_compute_pair_forces_lj, "example": _abi.lib.hmd_compute_pair_forces_example}[evaluator]
    check(fn(C.byref(a), ptr(prm), _stream(pos)), "hmd_compute_pair_forces_" + evaluator)
    return force, virial


def tile_geom(cl, box, ntypes=1, capacity=None):
    """Tile geometry of a cell list built with want_idx=True; raises KernelError when unsupported."""
    g = _abi.TileGeom()
    g.d_cell_size, g.d_cell_idx = ptr(cl["cell_size"]), ptr(cl["idx"])
    g.ci = Index3D(*cl["dim"])
    g.cli = Index2D(cl["Nmax"], cl["cell_size"].numel())
    cap = C.c_uint(0)
    check(_abi.lib.hmd_nlist_tiles_supported(C.byref(g.ci), C.byref(box), cl["Nmax"], ntypes, C.byref(cap)), "hmd_nlist_tiles_supported")
    g.capacity = cap.value if capacity is None else capacity
    return g


def nlist_tiles(pos, N, box, r_cut, r_buff, cl, adj, capacity=None, exclusions=None):
    """Tile build with the head-list / overflow protocol of nlist_binned.  exclusions: (n_ex, ex_list) by particle
    index, ex_list of shape (height, pitch) with pitch >= N: left out of the rows as they are built."""
    dev = pos.device
    r_cut_t = torch.as_tensor(np.asarray(r_cut, dtype=np.float64), device=dev).contiguous()
    nt = r_cut_t.shape[0]
    geom = tile_geom(cl, box, nt, capacity)
    Nmax_t = torch.full((nt,), 4, dtype=torch.int32, device=dev)
    last_pos = torch.zeros((max(N, 1), 4), dtype=torch.float64, device=dev)
    overflow = torch.zeros(2, dtype=torch.int32, device=dev)
    n_builds = 0
    n_members = int(pos.shape[0])  # owned + ghost particles in the cell list
    nblocks = ((cl["dim"][0] + 1) // 2) * ((cl["dim"][1] + 1) // 2) * ((cl["dim"][2] + 1) // 2)
    block_group = torch.zeros(nblocks + 1, dtype=torch.int32, device=dev)
    geom.d_block_group = ptr(block_group)
    while True:
        head, total = build_head_list(pos, N, Nmax_t)
        geom.row_pieces = (int(Nmax_t.max().item()) + (3 if nt == 2 else 0) + 15) // 16  # (rows by type: up to three dummy entries)
        entries = _abi.lib.hmd_nlist_tiles_storage_entries(C.byref(geom.ci), n_members, geom.row_pieces)
        # (not zeros: an entry the build should have written and did not must show -- 5 is the slot of a real candidate)
        tiles = torch.full((entries,), 5, dtype=torch.int16, device=dev)
        overflow.zero_()
        n_neigh = torch.zeros(max(N, 1), dtype=torch.int32, device=dev)
        cond = torch.zeros(nt, dtype=torch.int32, device=dev)
        a = _abi.NlistTilesArgs()
        a.geom = geom
        a.d_cell_xyzf, a.d_cell_type_body, a.d_cell_adj = ptr(cl["xyzf"]), ptr(cl["type_body"]), ptr(adj)
        a.cadji = Index2D(adj.shape[1], cl["cell_size"].numel())
        slot_meta = torch.zeros((cl["cell_size"].numel() * cl["Nmax"], 4), dtype=torch.int32, device=dev)
        a.d_slot_meta = ptr(slot_meta)
        a.d_tile_nlist, a.d_n_neigh, a.d_last_updated_pos, a.d_conditions = ptr(tiles), ptr(n_neigh), ptr(last_pos), ptr(cond)
        a.d_Nmax, a.d_head_list, a.d_pos, a.N, a.box = ptr(Nmax_t), ptr(head), ptr(pos), N, box
        a.d_r_cut, a.r_buff, a.ntypes, a.d_tile_overflow = ptr(r_cut_t), float(r_buff), nt, ptr(overflow)
        a.tile_nlist_entries = entries
        if exclusions is not None:
            a.d_n_ex_idx, a.d_ex_list_idx = ptr(exclusions[0]), ptr(exclusions[1])
            a.exli = Index2D(exclusions[1].shape[1], exclusions[1].shape[0])
        check(_abi.lib.hmd_compute_nlist_tiles(C.byref(a), _stream(pos)), "hmd_compute_nlist_tiles")
        # (two types: rows laid out by type; the force entry points read this from the geometry)
        geom.rows_by_type = a.geom.rows_by_type = int(_abi.lib.hmd_nlist_tiles_rows_by_type(C.byref(a)))
        n_builds += 1
        if int(overflow[0].item()):
            raise _abi.KernelError("tile capacity exceeded")
        c = cond.cpu().numpy().astype(np.int64)
        nm = Nmax_t.cpu().numpy().astype(np.int64)
        over = c > nm
        if over.any():
            nm[over] = np.where(c[over] > 4, (c[over] + 3) & ~3, 4)
            Nmax_t = torch.as_tensor(nm.astype(np.int32), device=dev)
            continue
        # slot of every cell-list member, for the slot-ordered position copy of the fast staging path
        slot_of = torch.zeros(pos.shape[0], dtype=torch.int32, device=dev)
        check(_abi.lib.hmd_tile_slots(C.byref(geom), ptr(slot_of), _stream(pos)), "hmd_tile_slots")
        # geom points into the cell list's tensors: keep them alive with the result
        return dict(slot_meta=slot_meta, slot_of=slot_of, tiles=tiles, n_neigh=n_neigh, head=head, last_pos=last_pos, Nmax=Nmax_t, n_builds=n_builds, geom=geom, args=a,
                    cell_list=cl, adj=adj, r_cut=r_cut_t, overflow=overflow, block_group=block_group, nlist_size=total, ntypes=nt)


def tiles_to_rows(tl, N, box):
    dev = tl["tiles"].device
    nlist = torch.full((max(tl["nlist_size"], 4),), -1, dtype=torch.int32, device=dev)
    a = _abi.TilesToRowsArgs()
    a.geom = tl["geom"]
    a.d_tile_nlist, a.d_n_neigh, a.d_head_list, a.d_nlist, a.N, a.box = ptr(tl["tiles"]), ptr(tl["n_neigh"]), ptr(tl["head"]), ptr(nlist), N, box
    a.ntypes = tl["ntypes"]
    check(_abi.lib.hmd_nlist_tiles_to_rows(C.byref(a), _stream(nlist)), "hmd_nlist_tiles_to_rows")
    return nlist


def tile_refresh(pos, tl):
    """Slot-ordered copy of the current positions (hmd_tile_refresh)."""
    cpos = torch.full((tl["slot_meta"].shape[0], 4), float("nan"), dtype=torch.float64, device=pos.device)
    check(_abi.lib.hmd_tile_refresh(ptr(pos), ptr(tl["slot_of"]), ptr(cpos), pos.shape[0], _stream(pos)), "hmd_tile_refresh")
    return cpos


def pair_forces_tiles(evaluator, pos, N, box, tl, params, rcutsq, ronsq=None, shift_mode=0, compute_virial=True, tpp=0, n_total=None, fast=False,
                      window=None, into=None):
    """window: (mode, lo[3], hi[3]) restricts the launch to the blocks inside (1) or outside (2) a window of the
    grid of 2x2x2-cell blocks; into: (force, virial) arrays of an earlier call to write into."""
    dev = pos.device
    n_total = pos.shape[0] if n_total is None else n_total
    prm = torch.as_tensor(np.asarray(params, dtype=np.float64).reshape(-1, 2), device=dev).contiguous()
    rc = torch.as_tensor(np.asarray(rcutsq, dtype=np.float64), device=dev).contiguous()
    nt = rc.shape[0]
    ro = torch.as_tensor(np.asarray(ronsq if ronsq is not None else np.zeros((nt, nt)), dtype=np.float64), device=dev).contiguous()
    if into is not None:
        force, virial = into
    else:
        force = torch.full((n_total, 4), float("nan"), dtype=torch.float64, device=dev)
        virial = torch.full((6, n_total), float("nan"), dtype=torch.float64, device=dev) if compute_virial else None
    a = _abi.PairArgs()
    a.d_force, a.d_virial, a.virial_pitch = ptr(force), ptr(virial), n_total
    a.N, a.n_max, a.d_pos, a.box = N, n_total, ptr(pos), box
    if window is not None:
        a.block_window_mode = window[0]
        for d in range(3):
            a.block_lo[d], a.block_hi[d] = window[1][d], window[2][d]
    a.d_n_neigh, a.d_nlist, a.d_head_list = ptr(tl["n_neigh"]), None, ptr(tl["head"])
    a.d_rcutsq, a.d_ronsq, a.size_neigh_list, a.ntypes = ptr(rc), ptr(ro), tl["tiles"].numel(), nt
    a.shift_mode, a.compute_virial, a.threads_per_particle = shift_mode, int(compute_virial), tpp
    fn = {"lj": _abi.lib.hmd_compute_pair_forces_lj_tiles, "example": _abi.lib.hmd_compute_pair_forces_example_tiles}[evaluator]
    geom = _abi.TileGeom.from_buffer_copy(tl["geom"])
    # the kernel stages the region from the slot-ordered copy of the positions and the per-slot row info
    cpos = tile_refresh(pos, tl)
    geom.d_cpos, geom.d_slot_meta = ptr(cpos), ptr(tl["slot_meta"])
    if fast:
        geom.capacity = max(int(tl["overflow"][1].item()), 1)  # the largest region the build saw: the exact LDS need
    check(fn(C.byref(a), C.byref(geom), ptr(tl["tiles"]), ptr(prm), _stream(pos)), "hmd_compute_pair_forces_%s_tiles" % evaluator)
    return force, virial


def pair_forces_tiles_step(pos, vel, image, N, box, tl, params, rcutsq, dt, last_pos=None, rcut_max=None, r_buff=0.0, checkn=1, shift_mode=0,
                           skip=None):
    """hmd_compute_pair_forces_lj_tiles_step: the tile pair force with the second half step, the next first half step, the
    distance check and the slot-ordered copy riding on it.  pos, vel and image are updated in place; returns (force, the
    new slot-ordered positions, flag word).  skip: (flag tensor, value) switches the launch off when flag == value."""
    dev = pos.device
    prm = torch.as_tensor(np.asarray(params, dtype=np.float64).reshape(-1, 2), device=dev).contiguous()
    rc = torch.as_tensor(np.asarray(rcutsq, dtype=np.float64), device=dev).contiguous()
    nt = rc.shape[0]
    ro = torch.zeros((nt, nt), dtype=torch.float64, device=dev)
    force = torch.full((pos.shape[0], 4), float("nan"), dtype=torch.float64, device=dev)
    a = _abi.PairArgs()
    a.d_force, a.N, a.n_max, a.d_pos, a.box = ptr(force), N, pos.shape[0], ptr(pos), box
    a.d_n_neigh, a.d_nlist, a.d_head_list = ptr(tl["n_neigh"]), None, ptr(tl["head"])
    a.d_rcutsq, a.d_ronsq, a.size_neigh_list, a.ntypes = ptr(rc), ptr(ro), tl["tiles"].numel(), nt
    a.shift_mode, a.compute_virial = shift_mode, 0
    if skip is not None:
        a.d_skip_flag, a.skip_value = ptr(skip[0]), skip[1]
    geom = _abi.TileGeom.from_buffer_copy(tl["geom"])
    cpos = tile_refresh(pos, tl)
    cpos_new = torch.full_like(cpos, float("nan"))
    geom.d_cpos, geom.d_slot_meta = ptr(cpos), ptr(tl["slot_meta"])
    geom.capacity = max(int(tl["overflow"][1].item()), 1)
    s = _abi.FusedStepArgs()
    s.two.d_vel, s.two.d_net_force, s.two.group_size, s.two.deltaT, s.two.rescale_factor = ptr(vel), ptr(force), N, float(dt), 1.0
    s.one.d_pos, s.one.d_vel, s.one.d_image = ptr(pos), ptr(vel), ptr(image)
    s.one.group_size, s.one.box, s.one.rescale_factor, s.one.deltaT = N, box, 1.0, float(dt)
    res = None
    if last_pos is not None:
        res = torch.zeros(1, dtype=torch.int32, device=dev)
        rcm = torch.as_tensor(np.asarray(rcut_max, dtype=np.float64), device=dev)
        c = s.check
        c.d_result, c.d_last_pos, c.d_pos, c.N, c.box = ptr(res), ptr(last_pos), ptr(pos), N, box
        c.d_rcut_max, c.r_buff, c.ntypes, c.lambda_min, c.lam, c.checkn = ptr(rcm), float(r_buff), rcm.numel(), 1.0, _s3((1.0, 1.0, 1.0)), checkn
    s.d_slot_of, s.d_cpos = ptr(tl["slot_of"]), ptr(cpos_new)
    check(_abi.lib.hmd_pair_forces_tiles_step_supported(C.byref(a), C.byref(s)), "hmd_pair_forces_tiles_step_supported")
    check(_abi.lib.hmd_compute_pair_forces_lj_tiles_step(C.byref(a), C.byref(geom), ptr(tl["tiles"]), ptr(prm), C.byref(s), _stream(pos)),
          "hmd_compute_pair_forces_lj_tiles_step")
    return force, cpos_new, (int(res.item()) if res is not None else None)


def bond_table(tag, rtag, N, tag_n_bonds, tag_blist):
    """Per-particle bond table in index order from the one keyed by tag (hmd_update_bond_table).
    tag_blist: (max_bonds, n_tags, 2) int32 of (partner tag, bond type)."""
    dev = tag.device
    max_b, n_tags = tag_blist.shape[0], tag_blist.shape[1]
    n_bonds = torch.zeros(max(N, 1), dtype=torch.int32, device=dev)
    blist = torch.zeros((max(max_b, 1), max(N, 1), 2), dtype=torch.int32, device=dev)
    flags = torch.zeros(1, dtype=torch.int32, device=dev)
    a = _abi.BondTableArgs()
    a.N, a.n_total, a.d_tag, a.d_rtag = N, tag.shape[0], ptr(tag), ptr(rtag)
    a.d_tag_n_bonds, a.d_tag_blist, a.tag_blist_idx = ptr(tag_n_bonds), ptr(tag_blist), Index2D(n_tags, max_b)
    a.d_n_bonds, a.d_blist, a.blist_idx, a.d_flags = ptr(n_bonds), ptr(blist), Index2D(max(N, 1), max_b), ptr(flags)
    check(_abi.lib.hmd_update_bond_table(C.byref(a), _stream(tag)), "hmd_update_bond_table")
    return n_bonds, blist, int(flags.item())


def bond_forces_fene(pos, N, box, n_bonds, blist, params, compute_virial=True):
    """params: rows of (k, r_0, epsilon_x_4, sigma_6, delta) per bond type."""
    dev = pos.device
    prm = torch.as_tensor(np.asarray(params, dtype=np.float64).reshape(-1, 5), device=dev).contiguous()
    force = torch.full((max(N, 1), 4), float("nan"), dtype=torch.float64, device=dev)
    virial = torch.full((6, max(N, 1)), float("nan"), dtype=torch.float64, device=dev) if compute_virial else None
    flags = torch.zeros(1, dtype=torch.int32, device=dev)
    a = _abi.BondArgs()
    a.d_force, a.d_virial, a.virial_pitch, a.N, a.d_pos, a.box = ptr(force), ptr(virial), max(N, 1), N, ptr(pos), box
    a.d_blist, a.blist_idx, a.d_n_bonds = ptr(blist), Index2D(blist.shape[1], blist.shape[0]), ptr(n_bonds)
    a.n_bond_types, a.d_flags, a.compute_virial = prm.shape[0], ptr(flags), int(compute_virial)
    check(_abi.lib.hmd_compute_bond_forces_fene(C.byref(a), ptr(prm), _stream(pos)), "hmd_compute_bond_forces_fene")
    return force, virial, int(flags.item())


def sum_net_force(forces, virials=None, net_force=None, net_virial=None, clear=True):
    dev = forces[0].device
    n = forces[0].shape[0]
    nf = net_force if net_force is not None else torch.zeros((n, 4), dtype=torch.float64, device=dev)
    nv = net_virial if net_virial is not None else (torch.zeros((6, n), dtype=torch.float64, device=dev) if virials else None)
    a = _abi.SumNetForceArgs()
    a.d_net_force, a.d_net_virial, a.net_virial_pitch = ptr(nf), ptr(nv), n
    for k, f in enumerate(forces):
        a.force_list.f[k] = f.data_ptr()
        if virials:
            a.force_list.v[k] = virials[k].data_ptr()
            a.force_list.vpitch[k] = n
    a.nparticles, a.clear, a.compute_virial = n, int(clear), int(bool(virials))
    check(_abi.lib.hmd_integrator_sum_net_force(C.byref(a), _stream(nf)), "hmd_integrator_sum_net_force")
    return nf, nv


def step_one(pos, vel, accel, image, box, dt, rescale=1.0, group=None, limit=None, rescale_ke=None, rescale_gamma=0.0):
    """rescale_ke: a device tensor whose first double is the kinetic energy; the factor is then sqrt(rescale_gamma / ke)"""
    a = _abi.StepOneArgs()
    a.d_rescale_ke, a.rescale_gamma = ptr(rescale_ke), float(rescale_gamma)
    a.d_pos, a.d_vel, a.d_accel, a.d_image, a.d_group_members = ptr(pos), ptr(vel), ptr(accel), ptr(image), ptr(group)
    a.group_size = pos.shape[0] if group is None else group.numel()
    a.box, a.rescale_factor, a.deltaT = box, float(rescale), float(dt)
    a.use_limit = int(limit is not None)
    a.maximum_displacement = float(limit or 0.0)
    check(_abi.lib.hmd_nvt_rescale_step_one(C.byref(a), _stream(pos)), "hmd_nvt_rescale_step_one")


def step_two(vel, accel, net_force, dt, rescale=1.0, group=None, net_force2=None):
    a = _abi.StepTwoArgs()
    a.d_vel, a.d_accel, a.d_group_members, a.d_net_force = ptr(vel), ptr(accel), ptr(group), ptr(net_force)
    a.d_net_force2 = ptr(net_force2)
    a.group_size = vel.shape[0] if group is None else group.numel()
    a.deltaT, a.rescale_factor = float(dt), float(rescale)
    check(_abi.lib.hmd_nvt_rescale_step_two(C.byref(a), _stream(vel)), "hmd_nvt_rescale_step_two")


def fused_step(pos, vel, accel, image, net_force, box, dt, rescale_two=1.0, rescale_one=1.0, group=None, last_pos=None, rcut_max=None,
               r_buff=0.0, checkn=1, publish=False, step_two=True, slot_of=None, cpos=None, net_force2=None, rescale_ke=None, rescale_gamma=0.0):
    """hmd_nve_step_two_one_check (step_two=False: hmd_nvt_rescale_step_one_check, the first half step alone); returns
    (flag word, published 64-bit word) when the check is requested."""
    dev = pos.device
    a = _abi.FusedStepArgs()
    n = pos.shape[0] if group is None else group.numel()
    a.two.d_vel, a.two.d_accel, a.two.d_group_members, a.two.d_net_force = ptr(vel), ptr(accel), ptr(group), ptr(net_force)
    a.two.group_size, a.two.deltaT, a.two.rescale_factor = n, float(dt), float(rescale_two)
    a.two.d_net_force2 = ptr(net_force2)
    a.one.d_pos, a.one.d_vel, a.one.d_accel, a.one.d_image, a.one.d_group_members = ptr(pos), ptr(vel), ptr(accel), ptr(image), ptr(group)
    a.one.group_size, a.one.box, a.one.rescale_factor, a.one.deltaT = n, box, float(rescale_one), float(dt)
    a.one.d_rescale_ke, a.one.rescale_gamma = ptr(rescale_ke), float(rescale_gamma)
    res = pub = None
    if last_pos is not None:
        res = torch.zeros(1, dtype=torch.int32, device=dev)
        rcm = torch.as_tensor(np.asarray(rcut_max, dtype=np.float64), device=dev)
        c = a.check
        c.d_result, c.d_last_pos, c.d_pos, c.N, c.box = ptr(res), ptr(last_pos), ptr(pos), pos.shape[0], box
        c.d_rcut_max, c.r_buff, c.ntypes, c.lambda_min, c.lam, c.checkn = ptr(rcm), float(r_buff), rcm.numel(), 1.0, _s3((1.0, 1.0, 1.0)), checkn
        if publish:
            pub = torch.zeros(1, dtype=torch.int64, device=dev)
            c.d_published, c.stamp = ptr(pub), checkn
    if slot_of is not None:
        a.d_slot_of, a.d_cpos = ptr(slot_of), ptr(cpos)
    if step_two:
        check(_abi.lib.hmd_nve_step_two_one_check(C.byref(a), _stream(pos)), "hmd_nve_step_two_one_check")
    else:
        check(_abi.lib.hmd_nvt_rescale_step_one_check(C.byref(a), _stream(pos)), "hmd_nvt_rescale_step_one_check")
    if res is None:
        return None
    return int(res.item()), (int(pub.item()) if pub is not None else None)


def compute_accelerations(vel, net_force):
    accel = torch.zeros((vel.shape[0], 3), dtype=torch.float64, device=vel.device)
    check(_abi.lib.hmd_compute_accelerations(ptr(accel), ptr(vel), ptr(net_force), vel.shape[0], _stream(vel)),
          "hmd_compute_accelerations")
    return accel


def thermo(vel, net_force, net_virial=None, group=None):
    dev = vel.device
    out = torch.zeros(8, dtype=torch.float64, device=dev)
    scratch = torch.zeros(_abi.lib.hmd_thermo_scratch_bytes(), dtype=torch.uint8, device=dev)
    a = _abi.ThermoArgs()
    a.d_out, a.d_scratch, a.d_vel, a.d_net_force, a.d_net_virial = ptr(out), ptr(scratch), ptr(vel), ptr(net_force), ptr(net_virial)
    a.net_virial_pitch = vel.shape[0]
    a.d_group_members = ptr(group)
    a.group_size = vel.shape[0] if group is None else group.numel()
    check(_abi.lib.hmd_compute_thermo(C.byref(a), _stream(vel)), "hmd_compute_thermo")
    return out.cpu().numpy()


def step_two_thermo(vel, accel, net_force, dt, rescale=1.0, group=None):
    """hmd_nvt_rescale_step_two_thermo: the second half step and the thermo sums of the new velocities in one pass;
    returns the eight sums as a device tensor."""
    dev = vel.device
    out = torch.zeros(8, dtype=torch.float64, device=dev)
    scratch = torch.zeros(_abi.lib.hmd_thermo_scratch_bytes(), dtype=torch.uint8, device=dev)
    s = _abi.StepTwoArgs()
    s.d_vel, s.d_accel, s.d_group_members, s.d_net_force = ptr(vel), ptr(accel), ptr(group), ptr(net_force)
    s.group_size = vel.shape[0] if group is None else group.numel()
    s.deltaT, s.rescale_factor = float(dt), float(rescale)
    a = _abi.ThermoArgs()
    a.d_out, a.d_scratch, a.d_vel, a.d_net_force = ptr(out), ptr(scratch), ptr(vel), ptr(net_force)
    a.net_virial_pitch = vel.shape[0]
    a.d_group_members, a.group_size = ptr(group), s.group_size
    check(_abi.lib.hmd_nvt_rescale_step_two_thermo(C.byref(s), C.byref(a), _stream(vel)), "hmd_nvt_rescale_step_two_thermo")
    return out


def cell_order(cl, N):
    dev = cl["xyzf"].device
    ncells = cl["cell_size"].numel()
    order = torch.zeros(max(N, 1), dtype=torch.int32, device=dev)
    offs = torch.zeros(ncells + 1, dtype=torch.int32, device=dev)
    wsb = _abi.lib.hmd_cell_order_workspace_bytes(ncells)
    ws = torch.zeros(wsb, dtype=torch.uint8, device=dev)
    a = _abi.CellOrderArgs()
    a.d_order, a.d_cell_offset, a.d_cell_size, a.d_cell_xyzf = ptr(order), ptr(offs), ptr(cl["cell_size"]), ptr(cl["xyzf"])
    a.ci = Index3D(*cl["dim"])
    a.cli = Index2D(cl["Nmax"], ncells)
    a.N = N
    a.d_workspace, a.workspace_bytes = ptr(ws), wsb
    check(_abi.lib.hmd_cell_order(C.byref(a), _stream(order)), "hmd_cell_order")
    return order, offs


def gather_rows(src, order):
    out = torch.empty_like(src)
    eb = src.element_size() * (src.shape[1] if src.dim() == 2 else 1)
    check(_abi.lib.hmd_gather_rows(ptr(out), ptr(src), ptr(order), order.numel(), eb, _stream(src)), "hmd_gather_rows")
    return out


def sort_keys(keys, key_bits=64):
    """Ascending sort of int64 keys whose low key_bits bits are significant (hmd_sort_keys)."""
    out = torch.empty_like(keys)
    n = keys.numel()
    ws = torch.empty(max(int(_abi.lib.hmd_sort_keys_workspace_bytes(n)), 1), dtype=torch.uint8, device=keys.device)
    check(_abi.lib.hmd_sort_keys(ptr(keys), ptr(out), n, key_bits, ptr(ws), ws.numel(), _stream(keys)), "hmd_sort_keys")
    return out


def build_rtag(tag):
    rtag = torch.zeros_like(tag)
    check(_abi.lib.hmd_build_rtag(ptr(rtag), ptr(tag), tag.numel(), _stream(tag)), "hmd_build_rtag")
    return rtag


def ghost_plan(pos, N, box, r_ghost, mask):
    dev = pos.device
    plan = torch.zeros(max(N, 1), dtype=torch.int32, device=dev)
    rg = torch.as_tensor(np.asarray(r_ghost, dtype=np.float64), device=dev)
    a = _abi.GhostPlanArgs()
    a.d_plan, a.d_pos, a.N, a.box, a.d_r_ghost, a.ntypes, a.mask = ptr(plan), ptr(pos), N, box, ptr(rg), rg.numel(), mask
    check(_abi.lib.hmd_make_ghost_exchange_plan(C.byref(a), _stream(pos)), "hmd_make_ghost_exchange_plan")
    return plan


def pack_ghost_positions(pos, send_idx, shift_id, shift_table, out=None):
    n = send_idx.numel()
    if out is None:
        out = torch.empty((n, 4), dtype=torch.float64, device=pos.device)
    a = _abi.PackPosArgs()
    a.d_out, a.d_pos, a.d_send_idx, a.d_send_shift_id, a.d_shift_table, a.n_send = ptr(out), ptr(pos), ptr(send_idx), ptr(shift_id), ptr(shift_table), n
    check(_abi.lib.hmd_pack_ghost_positions(C.byref(a), _stream(pos)), "hmd_pack_ghost_positions")
    return out
