"""Regenerates the committed golden fixtures.  Needs /root/reference (build container only).

* ref_header_vectors.json  -- printed by oracle/_ref/ref_harness, a program that compiles the
  reference's own header arithmetic (BoxDim.h, Index1D.h, EvaluatorPairLJ.h,
  EvaluatorPairExample.h, EvaluatorBondFENE.h) where it lies under /root/reference.  Doubles are
  C99 hex floats.
* ref_written.gsd -- a two-frame "hoomd"-schema file written through the reference's own file library
  (hoomd/extern/gsd.c, compiled into oracle/_ref/libgsd_ref.so), read back by tests/test_gsd.py.
* bond_fenewca_cases.json -- the FENEWCA rows of the parameter table in the reference's
  hoomd/md/pytest/test_bond.py (two particles 0.969 apart, expected force and energy to 1e-2).
* lj_forces_and_energies.json -- the "LJ" rows of the reference's data file
  hoomd/md/pytest/forces_and_energies.json (Mathematica values at r = 0.75 and 1.5).
* true_pair_list.json -- the reference's data file hoomd/md/pytest/true_pair_list.json
  (1029 tag pairs of the 7x7x7 simple-cubic lattice at r_cut = 1.1).

Run:  python tests/golden/make_golden.py
"""
import json
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


def main():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
    out = subprocess.check_output([os.path.join(ROOT, "oracle", "_ref", "ref_harness")])
    data = json.loads(out)  # validates
    with open(os.path.join(HERE, "ref_header_vectors.json"), "w") as f:
        json.dump(data, f, separators=(",", ":"))

    with open(os.path.join(REF, "hoomd/md/pytest/forces_and_energies.json")) as f:
        fe = json.load(f)
    with open(os.path.join(HERE, "lj_forces_and_energies.json"), "w") as f:
        json.dump({"LJ": fe["LJ"], "distances": [0.75, 1.5], "r_cut": 2.5}, f, indent=1)

    # the FENEWCA rows of bond_test_parameters (hoomd/md/pytest/test_bond.py:38-58), read as text
    import re
    text = open(os.path.join(REF, "hoomd/md/pytest/test_bond.py")).read()
    rows = re.findall(r"hoomd\.md\.bond\.FENEWCA,\s*dict\(\),\s*dict\(([^)]*)\),\s*([-0-9.e]+),\s*([-0-9.e]+),", text)
    cases = []
    for params, force, energy in rows:
        d = {k.strip(): float(v) for k, v in (kv.split("=") for kv in params.split(","))}
        cases.append({"params": d, "distance": 0.969, "force": float(force), "energy": float(energy)})
    assert len(cases) == 3
    with open(os.path.join(HERE, "bond_fenewca_cases.json"), "w") as f:
        json.dump(cases, f, indent=1)

    # a small GSD file written by the reference's library
    import ctypes as C
    import numpy as np
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libgsd_ref.so"))
    h = C.create_string_buffer(4096)
    out_gsd = os.path.join(HERE, "ref_written.gsd")
    assert lib.gsd_create_and_open(h, out_gsd.encode(), b"golden", b"hoomd", C.c_uint32((1 << 16) | 4), 3, 0) == 0

    def put(name, a, t):
        a = np.ascontiguousarray(a)
        n, m = (a.shape[0], a.shape[1]) if a.ndim == 2 else (a.shape[0], 1)
        assert lib.gsd_write_chunk(h, name.encode(), t, C.c_uint64(n), C.c_uint32(m), 0, a.ctypes.data_as(C.c_void_p)) == 0

    types = np.zeros((2, 4), dtype=np.uint8)
    types[0, :1] = list(b"A")
    types[1, :3] = list(b"Bee")
    btypes = np.frombuffer(b"polymer\0", dtype=np.uint8).reshape(1, 8)
    for frame, step in enumerate((100, 200)):
        put("configuration/step", np.array([step], dtype=np.uint64), 4)
        put("configuration/box", np.array([10, 12, 14, 0, 0, 0], dtype=np.float32), 9)
        put("particles/N", np.array([5], dtype=np.uint32), 3)
        put("particles/position", np.arange(15, dtype=np.float32).reshape(5, 3) * 0.25 + frame, 9)
        if frame == 0:
            put("configuration/dimensions", np.array([3], dtype=np.uint8), 1)
            put("particles/types", types, 1)
            put("particles/typeid", np.array([0, 1, 0, 1, 1], dtype=np.uint32), 3)
            put("bonds/N", np.array([2], dtype=np.uint32), 3)
            put("bonds/types", btypes, 1)
            put("bonds/typeid", np.array([0, 0], dtype=np.uint32), 3)
            put("bonds/group", np.array([[0, 1], [1, 2]], dtype=np.uint32), 3)
        assert lib.gsd_end_frame(h) == 0
    assert lib.gsd_close(h) == 0

    shutil.copyfile(os.path.join(REF, "hoomd/md/pytest/true_pair_list.json"),
                    os.path.join(HERE, "true_pair_list.json"))
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
