#!/bin/bash
# A/B timing of the neighbor build alone (build_time.py) for library variants: tests/dbg/ab_build.sh base <variant> ...
V=$GRAFT_REPO_ROOT/hoomd-blue_amd/variants
for rep in 1 2; do
  for v in "$@"; do
    if [ "$v" = base ]; then unset HOOMD_AMD_LIB; else export HOOMD_AMD_LIB=$V/$v/libhoomd_amd.so; fi
    echo "$v: "; timeout -k 10 100 python tests/dbg/build_time.py 2>&1 | grep "tile build with\|per launch" | cut -c1-260 || exit 1
  done
done
