"""Per-kernel durations of the TIMED region of a bench run from a rocprofv3 --kernel-trace file.

    python scripts/trace_summary.py <dir with *_kernel_trace.csv> <timed steps> [out.json]

The profiler's own --stats table averages over the whole process: the equilibration from the lattice (other neighbor
counts, thermostat steps that run the unfused kernels) and the pair-force launches that a stale neighbor list
predicates off (one per rebuild, a fraction of the normal duration).  The timed region of bench.py is the end of the
process: this script takes, for the pair-force kernel, the last <timed steps> launches that ran in full (longer than
half the median), and for every other kernel the launches that started after the first of those."""
import collections
import csv
import glob
import json
import statistics
import sys

d, steps = sys.argv[1], int(sys.argv[2])
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:100]


pf = [r for r in rows if "pair_force" in r["Kernel_Name"]]
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
med = statistics.median(dur(r) for r in pf)
full = [r for r in pf if dur(r) > 0.5 * med][-steps:]
t0 = int(full[0]["Start_Timestamp"])
acc = collections.defaultdict(list)
for r in rows:
    if int(r["Start_Timestamp"]) >= t0:
        acc[short(r["Kernel_Name"])].append(dur(r))
out = {"timed_steps": steps, "window_us": (int(rows[-1]["End_Timestamp"]) - t0) / 1e3, "kernels": {}}
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    vv = [x for x in v if x > 0.5 * statistics.median(v)] if "pair_force" in k else v
    out["kernels"][k] = {"launches": len(v), "launches_in_full": len(vv), "avg_us": sum(vv) / len(vv), "min_us": min(vv), "max_us": max(vv),
                         "stddev_us": statistics.pstdev(vv), "total_us_per_step": sum(v) / steps}
print(json.dumps(out, indent=1))
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)
