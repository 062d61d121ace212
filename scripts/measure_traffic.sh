#!/bin/bash
# HBM traffic of the hot kernels from the L2 memory-side counters: one rocprofv3 --pmc pass per counter
# (FETCH_SIZE and WRITE_SIZE do not fit one pass), each over a short bench run.  Run on the GPU box from
# the repository root; writes gpurun_out/traffic/{fetch,write}/ and gpurun_out/traffic/summary.json.
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/traffic
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o t -- python3 "$ROOT/bench.py" --steps 150 --warmup 50 --equil 800 --no-cpu-baseline > "$OUT/fetch.log" 2>&1
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o t -- python3 "$ROOT/bench.py" --steps 150 --warmup 50 --equil 800 --no-cpu-baseline > "$OUT/write.log" 2>&1
cd "$ROOT"
python3 scripts/summarize_traffic.py "$OUT"
