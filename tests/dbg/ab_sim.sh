#!/bin/bash
# A/B timing of library variants on the whole time step: tests/dbg/ab_sim.sh <variant> [...]  ("base" = the regular build)
# (the host module finds libhoomd_amd.so through its run path, which LD_LIBRARY_PATH precedes; HOOMD_AMD_LIB serves ctypes)
V=$GRAFT_REPO_ROOT/hoomd-blue_amd/variants
for rep in ${AB_REPS:-1 2}; do
  for v in "$@"; do
    if [ "$v" = base ]; then unset HOOMD_AMD_LIB LD_LIBRARY_PATH; else export HOOMD_AMD_LIB=$V/$v/libhoomd_amd.so LD_LIBRARY_PATH=$V/$v; fi
    echo -n "$v: "; timeout -k 10 200 python bench.py --gpus 1 --steps 2000 --warmup 100 --no-cpu-baseline $AB_ARGS 2>gpurun_out/ab_sim_$v.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms/step %.4f' % d['ms_per_step'], 'kernel us %.1f' % d['roofline']['avg_launch_us'], d['roofline']['launches'])" || exit 1
  done
done
