#!/bin/bash
# A/B timing of several environment settings on the whole time step: tests/dbg/ab_envs.sh "A=1" "B=1 C=2" ...   ("base" = none;
# bench arguments in AB_ARGS, repetitions in AB_REPS)
for rep in ${AB_REPS:-1 2}; do
  for v in base "$@"; do
    echo -n "$v: "
    if [ "$v" = base ]; then timeout -k 10 200 python bench.py --gpus 1 --steps 2000 --warmup 100 --no-cpu-baseline $AB_ARGS 2>gpurun_out/ab_env.err; else env $v timeout -k 10 200 python bench.py --gpus 1 --steps 2000 --warmup 100 --no-cpu-baseline $AB_ARGS 2>gpurun_out/ab_env.err; fi | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ms/step %.4f' % d['ms_per_step'], 'kernel us %.1f' % d['roofline']['avg_launch_us'], d['roofline']['launches'], 'rebuilds', d['config']['rebuilds_in_timed_region'])" || exit 1
  done
done
