#!/bin/bash
# Everything the measurement section of the design notes quotes, from one GPU box: the kernel-trace summary of the
# driver's bench command, the issue-side counters of the hot kernels, and their HBM traffic.
#     scripts/profile_round.sh <tag>      (from the repository root; results under gpurun_out/<tag>_*)
set -e
tag=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${tag}_trace" -o t -- python3 "$ROOT/bench.py" --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline \
    > "$OUT/${tag}_trace.json" 2> "$OUT/${tag}_trace.err"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${tag}_trace2k" -o t -- python3 "$ROOT/bench.py" --gpus 1 --steps 2000 --warmup 100 --no-cpu-baseline \
    > "$OUT/${tag}_trace2k.json" 2> "$OUT/${tag}_trace2k.err"
cd "$ROOT"
scripts/pmc_kernels.sh "$tag" > "$OUT/${tag}_pmc.txt" 2>&1
scripts/measure_traffic.sh > "$OUT/${tag}_traffic.txt" 2>&1
