#!/bin/bash
# The bench lines profiles/ keeps for a tree: tests/dbg/bench_lines.sh <tag>  ->  gpurun_out/<tag>_lines/*.json
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-lines}_lines; mkdir -p $O
run() { n=$1; shift; timeout -k 10 400 python bench.py "$@" > $O/$n.json 2> $O/$n.err; python -c "
import json,sys; d=json.load(open('$O/$n.json')); print('$n', round(d['ms_per_step'],4), '%.3e'%d['value'], round(d['roofline']['avg_launch_us'],1), round(d['roofline']['frac'],3))"; }
run bench_default
run bench_driver_cmd_steps20_warmup5 --gpus 1 --steps 20 --warmup 5
run bench_driver_cmd_steps20_warmup5_v2 --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline
run bench_dd_one_rank_rccl_cpp --force-dd --no-cpu-baseline
run bench_ka4m --workload ka --lattice 128x128x256 --no-cpu-baseline
run bench_kg2m_chains64_rho085 --workload kg --lattice 128x128x128 --no-cpu-baseline
run bench_lj8m_one_gpu --lattice 256x256x128 --steps 1000 --warmup 200 --equil 3000 --no-cpu-baseline
