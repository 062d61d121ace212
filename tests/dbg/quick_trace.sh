#!/bin/bash
# Kernel-trace summary of a 600-step bench run: tests/dbg/quick_trace.sh <tag> [bench arguments]
tag=$1; shift
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 280 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/${tag}_qt" -o t -- python3 "$ROOT/bench.py" --gpus 1 --steps 600 --warmup 50 --no-cpu-baseline "$@" \
    > "$ROOT/gpurun_out/${tag}_qt.json" 2> "$ROOT/gpurun_out/${tag}_qt.err" || exit 1
cd "$ROOT"
python3 scripts/trace_summary.py gpurun_out/${tag}_qt 600 gpurun_out/${tag}_qt_summary.json > gpurun_out/${tag}_qt_summary.txt 2>&1; python3 -c "
import json,sys
d=json.load(open('gpurun_out/${tag}_qt_summary.json'))
print(open('gpurun_out/${tag}_qt.json').read()[:200])
for k,v in list(d['kernels'].items())[:12]: print('%-60s n=%5d avg=%8.1f sd=%6.1f per_step=%7.2f'%(k[:60],v['launches'],v['avg_us'],v['stddev_us'],v['total_us_per_step']))
"
