"""The C-ABI library loads on a CPU-only box, exports every symbol the header declares, and the
ctypes mirrors agree with the C struct layouts.  No kernel is launched here."""
import ctypes as C
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "hoomd_amd.h")


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(hmd_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from hoomd_amd import _abi
    names = _declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(_abi.lib, n), f"{n} declared in include/hoomd_amd.h but not exported"
        assert n in _abi._SIGNATURES, f"{n} has no ctypes signature"
    assert _abi.lib.hmd_abi_version() == 1


def test_ctypes_struct_layouts_match_the_header():
    from hoomd_amd import _abi
    pairs = {
        "hmd_box_t": _abi.Box, "hmd_index2d_t": _abi.Index2D, "hmd_index3d_t": _abi.Index3D,
        "hmd_cell_list_args_t": _abi.CellListArgs, "hmd_sort_cell_list_args_t": _abi.SortCellListArgs,
        "hmd_nlist_check_args_t": _abi.NlistCheckArgs, "hmd_head_list_args_t": _abi.HeadListArgs,
        "hmd_nlist_binned_args_t": _abi.NlistBinnedArgs, "hmd_nlist_filter_args_t": _abi.NlistFilterArgs,
        "hmd_update_exclusion_args_t": _abi.UpdateExclusionArgs, "hmd_pair_args_t": _abi.PairArgs,
        "hmd_force_list_t": _abi.ForceList, "hmd_sum_net_force_args_t": _abi.SumNetForceArgs,
        "hmd_step_one_args_t": _abi.StepOneArgs, "hmd_step_two_args_t": _abi.StepTwoArgs,
        "hmd_thermo_args_t": _abi.ThermoArgs, "hmd_cell_order_args_t": _abi.CellOrderArgs,
        "hmd_ghost_plan_args_t": _abi.GhostPlanArgs, "hmd_pack_pos_args_t": _abi.PackPosArgs,
        "hmd_migrate_dest_args_t": _abi.MigrateDestArgs, "hmd_migrate_pack_args_t": _abi.MigratePackArgs,
        "hmd_migrate_fill_args_t": _abi.MigrateFillArgs, "hmd_unpack_ghost_init_args_t": _abi.UnpackGhostInitArgs,
        "hmd_vote_plan_args_t": _abi.VotePlanArgs, "hmd_migrate_unpack_args_t": _abi.MigrateUnpackArgs, "hmd_pack_vote_args_t": _abi.PackVoteArgs,
        "hmd_unpack_vote_args_t": _abi.UnpackVoteArgs, "hmd_ghost_send_list_args_t": _abi.GhostSendListArgs, "hmd_bond_table_args_t": _abi.BondTableArgs, "hmd_bond_args_t": _abi.BondArgs,
        "hmd_tile_geom_t": _abi.TileGeom, "hmd_nlist_tiles_args_t": _abi.NlistTilesArgs, "hmd_tiles_to_rows_args_t": _abi.TilesToRowsArgs,
    }
    lines = ['#include "hoomd_amd.h"', "#include <stdio.h>", "#include <stddef.h>", "int main(void){"]
    for cname, ct in pairs.items():
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        last = ct._fields_[-1][0]
        cfield = {"lam": "lambda"}.get(last, last)
        lines.append(f'printf("{cname}.{last} %zu\\n", offsetof({cname}, {cfield}));')
    lines.append("return 0;}")
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.c")
        open(src, "w").write("\n".join(lines))
        exe = os.path.join(d, "t")
        subprocess.check_call(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
        out = subprocess.check_output([exe]).decode().split("\n")
    got = dict(l.split() for l in out if l.strip())
    for cname, ct in pairs.items():
        assert int(got[cname]) == C.sizeof(ct), cname
        last = ct._fields_[-1][0]
        assert int(got[f"{cname}.{last}"]) == getattr(ct, last).offset, f"{cname}.{last}"


def test_box_init_matches_oracle_box():
    from hoomd_amd import _abi
    from oracle import oracle as orc
    for args in [(42.394, 42.394, 42.394, 0, 0, 0), (12.0, 9.0, 11.0, 0.3, -0.2, 0.15), (8.0, 8.0, 0.0, 0.1, 0, 0)]:
        b = _abi.Box.from_L(*args)
        o = orc.Box.from_L(*args)
        for f in ("lo", "hi", "L", "Linv"):
            assert list(getattr(b, f)) == list(getattr(o, f))
        assert (b.xy, b.xz, b.yz) == (o.xy, o.xz, o.yz)
        assert list(b.nearest_plane) == list(orc.nearest_plane_distance(o))


def test_argument_validation_without_a_gpu():
    """Rejected argument blocks return a negative code before anything is launched."""
    from hoomd_amd import _abi
    a = _abi.PairArgs()
    assert _abi.lib.hmd_compute_pair_forces_lj(C.byref(a), None, None) == _abi.ERR_INVALID_ARGUMENT
    assert b"required" in _abi.lib.hmd_last_error()
    c = _abi.CellListArgs()
    assert _abi.lib.hmd_compute_cell_list(C.byref(c), None) == _abi.ERR_INVALID_ARGUMENT
    assert _abi.lib.hmd_gather_rows(C.c_void_p(8), C.c_void_p(16), C.c_void_p(8), 4, 7, None) == _abi.ERR_INVALID_ARGUMENT


def test_one_hip_runtime_whatever_the_import_order():
    """PyTorch's wheel carries its own copy of the HIP runtime; imported after this package it must not become a
    second runtime in the process (device pointers, streams and RCCL communicators are shared between the two)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); %s\n"
            "libs = sorted({l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l}); print(len(libs), libs)")
    for order in ("import hoomd_amd, torch", "import torch, hoomd_amd"):
        r = subprocess.run([sys.executable, "-c", code % (os.path.join(root, "hoomd-blue_amd"), order)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        assert r.stdout.split()[0] == "1", (order, r.stdout)
