#!/bin/bash
# A/B timing of library variants on one box: tests/dbg/ab.sh <variant> [<variant> ...]  ("base" = the regular build)
V=$GRAFT_REPO_ROOT/hoomd-blue_amd/variants
timeout -k 10 200 python tests/dbg/pft_bench.py --make-liquid 2>&1 | tail -1 || exit 1
for rep in 1 2; do
  for v in "$@"; do
    if [ "$v" = base ]; then unset HOOMD_AMD_LIB; else export HOOMD_AMD_LIB=$V/$v/libhoomd_amd.so; fi
    echo -n "$v: "; timeout -k 10 100 python tests/dbg/pft_bench.py $AB_ARGS 2>&1 | tail -${AB_LINES:-1} || exit 1
  done
done
