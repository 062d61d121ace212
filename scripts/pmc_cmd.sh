#!/bin/bash
# Issue-side counters of every kernel of an arbitrary python script (two rocprofv3 --pmc passes):
#     scripts/pmc_cmd.sh <tag> <script.py> [args]      (environment variables of the caller are inherited)
# -> gpurun_out/pmc_<tag>/summary.json and a table on stdout (mean per launch, launches under 20 % of the largest dropped)
set -e
tag=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$tag
mkdir -p "$OUT"
SCRIPT=$(realpath "$1"); shift
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT \
    --output-format csv -d "$OUT/p1" -o p -- python3 "$SCRIPT" "$@" > "$OUT/p1.log" 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR \
    --output-format csv -d "$OUT/p2" -o p -- python3 "$SCRIPT" "$@" > "$OUT/p2.log" 2>&1
timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_WAVES_EQ_64 SQ_INSTS_SMEM \
    --output-format csv -d "$OUT/p3" -o p -- python3 "$SCRIPT" "$@" > "$OUT/p3.log" 2>&1 || true
cd "$ROOT"
python3 - "$OUT" <<'PY'
import collections, csv, glob, json, sys
out = sys.argv[1]
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for p in ("p1", "p2", "p3"):
    for f in glob.glob(f"{out}/{p}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:90]
            vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, v in vals.items():
    if not any(s in k for s in ("pair_force", "nlist_", "fused_step", "cell_list", "tile_")):
        continue
    res[k] = {}
    for c, xs in v.items():
        big = [x for x in xs if x >= 0.2 * max(xs)] if max(xs) > 0 else xs
        res[k][c] = sum(big) / len(big)
        res[k]["launches_counted"] = len(big)
json.dump(res, open(f"{out}/summary.json", "w"), indent=1, sort_keys=True)
for k, v in res.items():
    print(k)
    for c, x in sorted(v.items()):
        print(f"    {c:28s} {x:14.4g}")
PY
