#!/bin/bash
# usage: pmc.sh <tag> <script args...> ; one counter pass per rocprofv3 run
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_INSTS_VALU --output-format csv -d $out/p1 -o p -- python3 $GRAFT_REPO_ROOT/tests/dbg/tile_prof.py "$@" > $out.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_LDS_ADDR_CONFLICT SQ_WAVES --output-format csv -d $out/p2 -o p -- python3 $GRAFT_REPO_ROOT/tests/dbg/tile_prof.py "$@" >> $out.log 2>&1 || exit 1
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob, collections
for p in ("p1", "p2"):
    for f in glob.glob("$out/%s/**/*counter_collection.csv" % p, recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        for k, v in agg.items():
            if "pair_force" in k or "nlist" in k:
                print(p, k, {c: "%.3g" % (x / 3) for c, x in v.items()})
PY
