"""ctypes mirror of include/hoomd_amd.h.

Loads ``libhoomd_amd.so`` (the gfx950 kernel-driver library) from the package directory and
declares every struct and entry point of the C ABI.  There is no fallback: if the library is
missing the import fails, and nothing here computes on the CPU.
"""
import ctypes as C
import os

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# HOOMD_AMD_LIB: an experimental build of the library (make -C csrc variant ...), for kernel timing scripts only
LIB_PATH = os.environ.get("HOOMD_AMD_LIB") or os.path.join(_PKG_ROOT, "libhoomd_amd.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build the HIP kernel library first "
        "(python -c 'import __graft_entry__ as g; g.build()' or make -C hoomd-blue_amd/csrc)")



def _share_the_hip_runtime_with_torch():
    """PyTorch's ROCm wheels carry their own copy of the HIP runtime (torch/lib/libamdhip64.so, the same SONAME as the
    system's) and ask for it by a name the dynamic loader does not match against an already loaded system copy: with
    this library loaded first and torch second, the process would hold two HIP runtimes, and device pointers, streams
    and RCCL communicators of one mean nothing to the other.  Loading torch's copy first makes everything that follows
    -- this library, the host module, torch itself -- resolve to that one runtime.  Without an installed torch the
    system's runtime is used."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(path):
        C.CDLL(path, mode=C.RTLD_GLOBAL)


_share_the_hip_runtime_with_torch()
lib = C.CDLL(LIB_PATH)

NO_BODY = 0xFFFFFFFF
OK = 0
ERR_INVALID_ARGUMENT, ERR_UNSUPPORTED, ERR_TOO_LARGE = -1, -2, -3

vp = C.c_void_p
u32 = C.c_uint
f64 = C.c_double


class Scalar3(C.Structure):
    _fields_ = [("x", f64), ("y", f64), ("z", f64)]


class Box(C.Structure):
    _fields_ = [("lo", f64 * 3), ("hi", f64 * 3), ("L", f64 * 3), ("Linv", f64 * 3), ("xy", f64), ("xz", f64),
                ("yz", f64), ("nearest_plane", f64 * 3), ("periodic", C.c_ubyte * 3), ("pad_", C.c_ubyte * 5)]

    @classmethod
    def from_L(cls, Lx, Ly, Lz, xy=0.0, xz=0.0, yz=0.0, periodic=(1, 1, 1)):
        b = cls()
        lib.hmd_box_init(C.byref(b), f64(Lx), f64(Ly), f64(Lz), f64(xy), f64(xz), f64(yz))
        for d in range(3):
            b.periodic[d] = int(periodic[d])
        return b

    @classmethod
    def from_lo_hi(cls, lo, hi, periodic=(1, 1, 1)):
        b = cls()
        lib.hmd_box_init(C.byref(b), f64(1), f64(1), f64(1), f64(0), f64(0), f64(0))
        lib.hmd_box_set_lo_hi(C.byref(b), (f64 * 3)(*lo), (f64 * 3)(*hi))
        for d in range(3):
            b.periodic[d] = int(periodic[d])
        return b


class Index2D(C.Structure):
    _fields_ = [("w", u32), ("h", u32)]


class Index3D(C.Structure):
    _fields_ = [("w", u32), ("h", u32), ("d", u32)]


class CellListArgs(C.Structure):
    _fields_ = [("d_cell_size", vp), ("d_xyzf", vp), ("d_type_body", vp), ("d_cell_orientation", vp),
                ("d_cell_idx", vp), ("d_conditions", vp), ("d_pos", vp), ("d_orientation", vp), ("d_charge", vp),
                ("d_diameter", vp), ("d_body", vp), ("N", u32), ("n_ghost", u32), ("Nmax", u32),
                ("flag_charge", C.c_int), ("flag_type", C.c_int), ("box", Box), ("ci", Index3D), ("cli", Index2D),
                ("ghost_width", Scalar3), ("block_size", u32), ("clear_cell_size", C.c_int)]


class SortCellListArgs(C.Structure):
    _fields_ = [("d_cell_size", vp), ("d_xyzf", vp), ("d_type_body", vp), ("d_cell_idx", vp), ("ci", Index3D),
                ("cli", Index2D), ("flag_is_index", C.c_int)]


class NlistCheckArgs(C.Structure):
    _fields_ = [("d_result", vp), ("d_last_pos", vp), ("d_pos", vp), ("N", u32), ("box", Box), ("d_rcut_max", vp),
                ("r_buff", f64), ("ntypes", u32), ("lambda_min", f64), ("lam", Scalar3), ("checkn", u32),
                ("d_published", vp), ("stamp", u32)]


class HeadListArgs(C.Structure):
    _fields_ = [("d_head_list", vp), ("d_req_size_nlist", vp), ("d_Nmax", vp), ("d_pos", vp), ("N", u32),
                ("ntypes", u32), ("d_workspace", vp), ("workspace_bytes", C.c_size_t)]


class NlistBinnedArgs(C.Structure):
    _fields_ = [("d_nlist", vp), ("d_n_neigh", vp), ("d_last_updated_pos", vp), ("d_conditions", vp), ("d_Nmax", vp),
                ("d_head_list", vp), ("d_pos", vp), ("d_body", vp), ("N", u32), ("d_cell_size", vp),
                ("d_cell_xyzf", vp), ("d_cell_idx", vp), ("d_cell_type_body", vp), ("d_cell_adj", vp),
                ("ci", Index3D), ("cli", Index2D), ("cadji", Index2D), ("box", Box), ("d_r_cut", vp), ("r_buff", f64),
                ("ntypes", u32), ("threads_per_particle", u32), ("block_size", u32), ("filter_body", C.c_int),
                ("ghost_width", Scalar3), ("d_n_ex_idx", vp), ("d_ex_list_idx", vp), ("exli", Index2D)]


class NlistFilterArgs(C.Structure):
    _fields_ = [("d_n_neigh", vp), ("d_nlist", vp), ("d_head_list", vp), ("d_n_ex", vp), ("d_ex_list", vp),
                ("exli", Index2D), ("N", u32), ("block_size", u32)]


class UpdateExclusionArgs(C.Structure):
    _fields_ = [("d_tag", vp), ("d_rtag", vp), ("d_n_ex_tag", vp), ("d_ex_list_tag", vp),
                ("ex_list_tag_indexer", Index2D), ("d_n_ex_idx", vp), ("d_ex_list_idx", vp),
                ("ex_list_indexer", Index2D), ("N", u32)]


class BondTableArgs(C.Structure):
    _fields_ = [("N", u32), ("n_total", u32), ("d_tag", vp), ("d_rtag", vp), ("d_tag_n_bonds", vp), ("d_tag_blist", vp),
                ("tag_blist_idx", Index2D), ("d_n_bonds", vp), ("d_blist", vp), ("blist_idx", Index2D), ("d_flags", vp)]


class BondArgs(C.Structure):
    _fields_ = [("d_force", vp), ("d_virial", vp), ("virial_pitch", C.c_size_t), ("N", u32), ("d_pos", vp), ("box", Box),
                ("d_blist", vp), ("blist_idx", Index2D), ("d_n_bonds", vp), ("n_bond_types", u32), ("d_flags", vp),
                ("compute_virial", C.c_int)]


class PairArgs(C.Structure):
    _fields_ = [("d_force", vp), ("d_virial", vp), ("virial_pitch", C.c_size_t), ("N", u32), ("n_max", u32),
                ("d_pos", vp), ("d_charge", vp), ("box", Box), ("d_n_neigh", vp), ("d_nlist", vp), ("d_head_list", vp),
                ("d_rcutsq", vp), ("d_ronsq", vp), ("size_neigh_list", C.c_size_t), ("ntypes", u32),
                ("block_size", u32), ("shift_mode", u32), ("compute_virial", u32), ("threads_per_particle", u32),
                ("d_skip_flag", vp), ("skip_value", u32), ("publish_stamp", u32), ("d_published", vp),
                ("block_window_mode", u32), ("block_lo", u32 * 3), ("block_hi", u32 * 3)]


class ForceList(C.Structure):
    _fields_ = [("f", vp * 6), ("t", vp * 6), ("v", vp * 6), ("vpitch", C.c_size_t * 6)]


class SumNetForceArgs(C.Structure):
    _fields_ = [("d_net_force", vp), ("d_net_virial", vp), ("net_virial_pitch", C.c_size_t), ("d_net_torque", vp),
                ("force_list", ForceList), ("nparticles", u32), ("clear", C.c_int), ("compute_virial", C.c_int)]


class StepOneArgs(C.Structure):
    _fields_ = [("d_pos", vp), ("d_vel", vp), ("d_accel", vp), ("d_image", vp), ("d_group_members", vp),
                ("group_size", u32), ("box", Box), ("block_size", u32), ("rescale_factor", f64), ("deltaT", f64),
                ("use_limit", C.c_int), ("maximum_displacement", f64), ("d_rescale_ke", vp), ("rescale_gamma", f64)]


class StepTwoArgs(C.Structure):
    _fields_ = [("d_vel", vp), ("d_accel", vp), ("d_group_members", vp), ("group_size", u32), ("d_net_force", vp),
                ("block_size", u32), ("deltaT", f64), ("rescale_factor", f64), ("d_net_force2", vp)]


class ThermoArgs(C.Structure):
    _fields_ = [("d_out", vp), ("d_scratch", vp), ("d_vel", vp), ("d_net_force", vp), ("d_net_virial", vp),
                ("net_virial_pitch", C.c_size_t), ("d_group_members", vp), ("group_size", u32), ("d_out_host", vp)]


class CellOrderArgs(C.Structure):
    _fields_ = [("d_order", vp), ("d_cell_offset", vp), ("d_cell_size", vp), ("d_cell_xyzf", vp), ("ci", Index3D),
                ("cli", Index2D), ("N", u32), ("d_workspace", vp), ("workspace_bytes", C.c_size_t)]


class GhostPlanArgs(C.Structure):
    _fields_ = [("d_plan", vp), ("d_pos", vp), ("N", u32), ("box", Box), ("d_r_ghost", vp), ("ntypes", u32),
                ("mask", u32)]


class MigrateDestArgs(C.Structure):
    _fields_ = [("d_pos", vp), ("N", u32), ("global_lo", f64 * 3), ("global_L", f64 * 3), ("grid", u32 * 3),
                ("decomposed", C.c_ubyte * 3), ("pad_", C.c_ubyte), ("rank", u32), ("d_dest", vp), ("d_n_leave", vp),
                ("d_keys", vp), ("capacity", u32), ("d_rank_count", vp), ("n_ranks", u32)]


class PDataArraysStruct(C.Structure):
    _fields_ = [("d_pos", vp), ("d_vel", vp), ("d_accel", vp), ("d_image", vp), ("d_tag", vp), ("d_body", vp),
                ("d_charge", vp), ("d_diameter", vp)]


MIGRATE_ROW_WORDS = 18


class MigratePackArgs(C.Structure):
    _fields_ = [("arrays", PDataArraysStruct), ("d_keys", vp), ("n", u32), ("global_lo", f64 * 3), ("global_L", f64 * 3),
                ("decomposed", C.c_ubyte * 3), ("pad_", C.c_ubyte), ("d_out", vp), ("slot_rows", u32), ("n_ranks", u32),
                ("send_off", u32 * 65)]


class MigrateFillArgs(C.Structure):
    _fields_ = [("arrays", PDataArraysStruct), ("d_dest", vp), ("rank", u32), ("N", u32), ("d_keys", vp), ("n_leave", u32),
                ("d_scratch", vp)]


class UnpackGhostInitArgs(C.Structure):
    _fields_ = [("d_in", vp), ("n", u32), ("first", u32), ("arrays", PDataArraysStruct)]


class GhostSlots(C.Structure):
    _fields_ = [("n_ranks", u32), ("flag", u32), ("off", u32 * 65), ("row0", u32 * 64), ("cap", u32 * 64)]


VOTE_PLAN_MAX_RANKS = 64


class VotePlanArgs(C.Structure):
    _fields_ = [("n_ranks", u32), ("rank", u32), ("send_off", u32 * (VOTE_PLAN_MAX_RANKS + 1)), ("recv_off", u32 * (VOTE_PLAN_MAX_RANKS + 1)),
                ("d_out_row", vp), ("d_flag_rows", vp), ("d_in_map", vp)]


class MigrateUnpackArgs(C.Structure):
    _fields_ = [("arrays", PDataArraysStruct), ("d_in", vp), ("first", u32), ("n", u32), ("slot_rows", u32), ("n_ranks", u32),
                ("recv_off", u32 * 65)]


class GhostSendListArgs(C.Structure):
    _fields_ = [("ghost_plan", GhostPlanArgs), ("d_dir_mask", vp), ("n_dir", u32), ("d_keys", vp), ("capacity", u32),
                ("d_total", vp), ("d_dir_count", vp)]


class PackPosArgs(C.Structure):
    _fields_ = [("d_out", vp), ("d_pos", vp), ("d_send_idx", vp), ("d_send_shift_id", vp), ("d_shift_table", vp),
                ("n_send", u32)]


class PackVoteArgs(C.Structure):
    _fields_ = [("pack", PackPosArgs), ("d_out_row", vp), ("d_flag_word", vp), ("d_flag_rows", vp), ("n_flag_rows", u32)]


class UnpackVoteArgs(C.Structure):
    _fields_ = [("d_in", vp), ("d_map", vp), ("n_rows", u32), ("d_ghost_pos", vp), ("d_flag_word", vp)]


class FusedStepArgs(C.Structure):
    _fields_ = [("two", StepTwoArgs), ("one", StepOneArgs), ("check", NlistCheckArgs), ("d_slot_of", vp), ("d_cpos", vp)]


class TileGeom(C.Structure):
    _fields_ = [("d_cell_size", vp), ("d_cell_idx", vp), ("ci", Index3D), ("cli", Index2D), ("capacity", u32), ("d_cpos", vp),
                ("d_slot_meta", vp), ("d_block_group", vp), ("row_pieces", u32), ("rows_by_type", u32)]


class NlistTilesArgs(C.Structure):
    _fields_ = [("geom", TileGeom), ("d_cell_xyzf", vp), ("d_cell_type_body", vp), ("d_cell_adj", vp), ("cadji", Index2D),
                ("d_tile_nlist", vp), ("d_slot_meta", vp), ("d_slot_of", vp), ("d_n_neigh", vp), ("d_last_updated_pos", vp), ("d_conditions", vp),
                ("d_Nmax", vp), ("d_head_list", vp), ("d_pos", vp), ("N", u32), ("box", Box), ("d_r_cut", vp), ("r_buff", f64),
                ("ntypes", u32), ("d_tile_overflow", vp), ("tile_nlist_entries", C.c_size_t), ("d_n_ex_idx", vp), ("d_ex_list_idx", vp),
                ("exli", Index2D)]


class TilesToRowsArgs(C.Structure):
    _fields_ = [("geom", TileGeom), ("d_tile_nlist", vp), ("d_n_neigh", vp), ("d_head_list", vp), ("d_nlist", vp), ("N", u32),
                ("box", Box), ("ntypes", u32)]


# every symbol the header declares, with its signature
_SIGNATURES = {
    "hmd_box_init": (None, [C.POINTER(Box), f64, f64, f64, f64, f64, f64]),
    "hmd_box_set_lo_hi": (None, [C.POINTER(Box), C.POINTER(f64), C.POINTER(f64)]),
    "hmd_abi_version": (C.c_int, []),
    "hmd_last_error": (C.c_char_p, []),
    "hmd_compute_cell_list": (C.c_int, [C.POINTER(CellListArgs), vp]),
    "hmd_sort_cell_list": (C.c_int, [C.POINTER(SortCellListArgs), vp]),
    "hmd_nlist_needs_update_check": (C.c_int, [C.POINTER(NlistCheckArgs), vp]),
    "hmd_nlist_head_list_workspace_bytes": (C.c_size_t, [u32]),
    "hmd_nlist_build_head_list": (C.c_int, [C.POINTER(HeadListArgs), vp]),
    "hmd_nlist_build_head_list_noalloc": (C.c_int, [C.POINTER(HeadListArgs), vp]),
    "hmd_compute_nlist_binned": (C.c_int, [C.POINTER(NlistBinnedArgs), vp]),
    "hmd_nlist_filter": (C.c_int, [C.POINTER(NlistFilterArgs), vp]),
    "hmd_update_exclusion_list": (C.c_int, [C.POINTER(UpdateExclusionArgs), vp]),
    "hmd_pack_ghost_positions_vote": (C.c_int, [C.POINTER(PackVoteArgs), vp]),
    "hmd_unpack_ghost_positions": (C.c_int, [C.POINTER(UnpackVoteArgs), vp]),
    "hmd_ghost_send_list": (C.c_int, [C.POINTER(GhostSendListArgs), vp]),
    "hmd_migrate_dest": (C.c_int, [C.POINTER(MigrateDestArgs), vp]),
    "hmd_ghost_split_keys": (C.c_int, [vp, u32, vp, vp, vp]),
    "hmd_pack_ghost_init": (C.c_int, [C.POINTER(PackPosArgs), vp, vp]),
    "hmd_unpack_ghost_init": (C.c_int, [C.POINTER(UnpackGhostInitArgs), vp]),
    "hmd_pack_ghost_init_slots": (C.c_int, [C.POINTER(PackPosArgs), vp, C.POINTER(GhostSlots), vp]),
    "hmd_ghost_slot_headers": (C.c_int, [vp, C.POINTER(GhostSlots), vp, vp]),
    "hmd_unpack_ghost_init_slots": (C.c_int, [C.POINTER(UnpackGhostInitArgs), C.POINTER(GhostSlots), vp]),
    "hmd_migrate_slot_headers": (C.c_int, [vp, u32, u32, vp, vp]),
    "hmd_migrate_pack_slots_dev": (C.c_int, [C.POINTER(MigratePackArgs), vp, vp, u32, vp]),
    "hmd_migrate_slot_headers_and": (C.c_int, [vp, u32, u32, vp, u32, vp, vp, u32, vp]),
    "hmd_pack_ghost_init_slots_dev": (C.c_int, [C.POINTER(PackPosArgs), vp, C.POINTER(GhostSlots), vp, vp, vp, u32, u32, vp, vp]),
    "hmd_ghost_slot_headers_and": (C.c_int, [vp, C.POINTER(GhostSlots), vp, u32, vp, vp, u32, vp]),
    "hmd_comm_prep": (C.c_int, [vp, u32, vp, u32, vp]),
    "hmd_vote_plan": (C.c_int, [C.POINTER(VotePlanArgs), vp]),
    "hmd_migrate_pack": (C.c_int, [C.POINTER(MigratePackArgs), vp]),
    "hmd_migrate_fill": (C.c_int, [C.POINTER(MigrateFillArgs), vp]),
    "hmd_migrate_unpack": (C.c_int, [C.POINTER(MigrateUnpackArgs), vp]),
    "hmd_publish_word": (C.c_int, [vp, vp, u32, vp]),
    "hmd_publish_words": (C.c_int, [vp, u32, vp, u32, vp]),
    "hmd_stream_runs_beside_default": (C.c_int, [vp, vp, C.POINTER(C.c_int)]),
    "hmd_update_bond_table": (C.c_int, [C.POINTER(BondTableArgs), vp]),
    "hmd_compute_bond_forces_fene": (C.c_int, [C.POINTER(BondArgs), vp, vp]),
    "hmd_compute_pair_forces_lj": (C.c_int, [C.POINTER(PairArgs), vp, vp]),
    "hmd_compute_pair_forces_example": (C.c_int, [C.POINTER(PairArgs), vp, vp]),
    "hmd_integrator_sum_net_force": (C.c_int, [C.POINTER(SumNetForceArgs), vp]),
    "hmd_nvt_rescale_step_one": (C.c_int, [C.POINTER(StepOneArgs), vp]),
    "hmd_nvt_rescale_step_two": (C.c_int, [C.POINTER(StepTwoArgs), vp]),
    "hmd_compute_accelerations": (C.c_int, [vp, vp, vp, u32, vp]),
    "hmd_thermo_scratch_bytes": (C.c_size_t, []),
    "hmd_compute_thermo": (C.c_int, [C.POINTER(ThermoArgs), vp]),
    "hmd_nvt_rescale_step_two_thermo": (C.c_int, [C.POINTER(StepTwoArgs), C.POINTER(ThermoArgs), vp]),
    "hmd_cell_order_workspace_bytes": (C.c_size_t, [u32]),
    "hmd_cell_order": (C.c_int, [C.POINTER(CellOrderArgs), vp]),
    "hmd_gather_rows": (C.c_int, [vp, vp, vp, u32, u32, vp]),
    "hmd_build_rtag": (C.c_int, [vp, vp, u32, vp]),
    "hmd_sort_keys_workspace_bytes": (C.c_size_t, [u32]),
    "hmd_sort_keys": (C.c_int, [vp, vp, u32, u32, vp, C.c_size_t, vp]),
    "hmd_make_ghost_exchange_plan": (C.c_int, [C.POINTER(GhostPlanArgs), vp]),
    "hmd_pack_ghost_positions": (C.c_int, [C.POINTER(PackPosArgs), vp]),
    "hmd_nve_step_two_one_check": (C.c_int, [C.POINTER(FusedStepArgs), vp]),
    "hmd_nvt_rescale_step_one_check": (C.c_int, [C.POINTER(FusedStepArgs), vp]),
    "hmd_nlist_tiles_supported": (C.c_int, [C.POINTER(Index3D), C.POINTER(Box), u32, u32, C.POINTER(u32)]),
    "hmd_nlist_tiles_storage_entries": (C.c_size_t, [C.POINTER(Index3D), u32, u32]),
    "hmd_tile_slots": (C.c_int, [C.POINTER(TileGeom), vp, vp]),
    "hmd_tile_refresh": (C.c_int, [vp, vp, vp, u32, vp]),
    "hmd_compute_nlist_tiles": (C.c_int, [C.POINTER(NlistTilesArgs), vp]),
    "hmd_nlist_tiles_to_rows": (C.c_int, [C.POINTER(TilesToRowsArgs), vp]),
    "hmd_nlist_tiles_rows_by_type": (C.c_int, [C.POINTER(NlistTilesArgs)]),
    "hmd_compute_pair_forces_lj_tiles": (C.c_int, [C.POINTER(PairArgs), C.POINTER(TileGeom), vp, vp, vp]),
    "hmd_compute_pair_forces_example_tiles": (C.c_int, [C.POINTER(PairArgs), C.POINTER(TileGeom), vp, vp, vp]),
    "hmd_compute_pair_forces_lj_tiles_step": (C.c_int, [C.POINTER(PairArgs), C.POINTER(TileGeom), vp, vp, C.POINTER(FusedStepArgs), vp]),
    "hmd_pair_forces_tiles_step_supported": (C.c_int, [C.POINTER(PairArgs), C.POINTER(FusedStepArgs)]),
}

for _name, (_res, _args) in _SIGNATURES.items():
    _f = getattr(lib, _name)
    _f.restype = _res
    _f.argtypes = _args


class KernelError(RuntimeError):
    pass


def check(code, what=""):
    """Raise on a non-zero return of a driver call."""
    if code == 0:
        return
    if code < 0:
        raise KernelError(f"{what}: {lib.hmd_last_error().decode()} (code {code})")
    raise KernelError(f"{what}: HIP error {code}")


def ptr(t):
    """Device (or host) address of a torch tensor / numpy array as c_void_p; None -> NULL."""
    if t is None:
        return None
    if hasattr(t, "data_ptr"):
        return C.c_void_p(t.data_ptr())
    return C.c_void_p(t.ctypes.data)
