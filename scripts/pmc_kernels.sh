#!/bin/bash
# Issue-side counters of the hot kernels over a short bench run: one rocprofv3 --pmc pass per counter set
# (a pass holds at most 8 SQ counters).  Run on the GPU box from the repository root:
#     scripts/pmc_kernels.sh <tag> [bench.py arguments]
# writes gpurun_out/pmc_<tag>/{p1,p2}/ and gpurun_out/pmc_<tag>/summary.json (per kernel: counter -> mean per launch).
set -e
tag=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$tag
mkdir -p "$OUT"
ARGS="--steps 60 --warmup 20 --equil 1000 --no-cpu-baseline $*"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT \
    --output-format csv -d "$OUT/p1" -o p -- python3 "$ROOT/bench.py" $ARGS > "$OUT/p1.log" 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR \
    --output-format csv -d "$OUT/p2" -o p -- python3 "$ROOT/bench.py" $ARGS > "$OUT/p2.log" 2>&1
cd "$ROOT"
python3 - "$OUT" <<'PY'
import collections, csv, glob, json, sys
out = sys.argv[1]
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for p in ("p1", "p2"):
    for f in glob.glob(f"{out}/{p}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:100]
            vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, v in vals.items():
    if not any(s in k for s in ("pair_force", "nlist_", "fused_step", "cell_list", "tile_")):
        continue
    res[k] = {}
    for c, xs in v.items():
        # launches predicated off by a stale neighbor list do next to nothing: not part of the mean
        big = [x for x in xs if x >= 0.2 * max(xs)] if max(xs) > 0 else xs
        res[k][c] = sum(big) / len(big)
        res[k]["launches_counted"] = len(big)
json.dump(res, open(f"{out}/summary.json", "w"), indent=1, sort_keys=True)
for k, v in res.items():
    print(k)
    for c, x in sorted(v.items()):
        print(f"    {c:28s} {x:14.4g}")
PY
