"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; kilobytes per dispatch).

FETCH_SIZE on gfx950 tallies 128-byte requests at 64 bytes for wide coalesced loads; other widths are
uncalibrated.  The stand-alone second half step has a known, purely streaming byte count per particle
(reads vel 32 + net force 32 = 64 B, writes vel 32 + accel 24 = 56 B), so its counters calibrate the read
and write scale in this very run; the same scale is applied to the other kernels.  (Cross-check: the fused
integration pass then comes out at its own count, reads 132 B and writes 96 B per particle.)"""
import collections
import csv
import glob
import json
import re
import sys

out = sys.argv[1]
N = 1048576


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


fetch = per_kernel(out + "/fetch", "FETCH_SIZE")
write = per_kernel(out + "/write", "WRITE_SIZE")


def short(n):
    # the tile force kernel with the integration pass riding on it (template argument STEP, the last but one before the
    # number of waves) is a row of its own
    if "pair_force_tiles_kernel" in n and re.search(r"true, \d+>\(", n.replace(" >", ">")):
        return "pair_force_tiles_kernel_step"
    for k in ("pair_force_tiles_kernel", "pair_force_kernel", "nlist_pref_kernel", "nlist_binned_kernel", "tile_block_groups_kernel", "fused_step_kernel", "cell_list_kernel", "step_one_kernel",
              "step_two_kernel", "nlist_check_kernel"):
        if k in n:
            return k
    return None


rows = {}
for name in set(fetch) | set(write):
    k = short(name)
    if k is None:
        continue
    f = fetch.get(name, [])
    w = write.get(name, [])
    if k.startswith("pair_force"):
        # launches predicated off by a stale list move no data: leave them out of the mean
        f = [x for x in f if x > 1000.0]
    rows[k] = {"launches": len(f), "fetch_kb": sum(f) / max(len(f), 1), "write_kb": sum(w) / max(len(w), 1)}

cal = {"read_scale": None, "write_scale": None}
if "step_two_kernel" in rows:
    r = rows["step_two_kernel"]
    cal["read_scale"] = 64.0 * N / (r["fetch_kb"] * 1024.0)
    cal["write_scale"] = 56.0 * N / (r["write_kb"] * 1024.0)
for k, r in rows.items():
    rs = cal["read_scale"] or 2.0
    ws = cal["write_scale"] or 1.0
    r["hbm_bytes_per_launch"] = (r["fetch_kb"] * rs + r["write_kb"] * ws) * 1024.0
    r["hbm_bytes_per_particle"] = r["hbm_bytes_per_launch"] / N
summary = {"N": N, "workload": "lj", "calibration": cal, "kernels": rows,
           "note": "FETCH_SIZE / WRITE_SIZE in KB per dispatch, separate --pmc passes; scales calibrated on the stand-alone second half step "
                   "(known streaming byte count) in the same runs"}
json.dump(summary, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(summary, indent=1))
