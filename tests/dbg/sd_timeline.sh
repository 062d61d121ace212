#!/bin/bash
# kernel timeline of a few rebuild steps of a single-domain run
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/sd_tl
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$OUT" -o t -- python3 "$ROOT/bench.py" --steps 100 --warmup 20 --equil 600 --no-cpu-baseline > "$OUT/run.json" 2> "$OUT/run.err"
cd "$ROOT"
python3 - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
rows = []
for f in glob.glob(out + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:50]))
for f in glob.glob(out + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")))
rows.sort()
# the last rebuild: find the last nlist build kernel and print the 40 entries around it
idx = max(i for i, r in enumerate(rows) if "nlist_pref" in r[2] or "nlist_b" in r[2])
lo = max(0, idx - 60)
t0 = rows[lo][0]
prev_end = t0
for s, e, n in rows[lo: idx + 6]:
    print(f"{(s - t0) / 1e3:9.1f} us  +gap {(s - prev_end) / 1e3:7.1f}  dur {(e - s) / 1e3:7.1f}  {n}")
    prev_end = max(prev_end, e)
PY
