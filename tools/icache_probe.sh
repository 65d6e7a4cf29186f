#!/bin/bash
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_icache; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rm -rf $OUT/p1
CPU_BASELINE=0 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH --output-format csv -d $OUT/p1 -- python3 $REPO/tools/bench_rotation.py $1 $2 > $OUT/p1.log 2>&1
cd $REPO
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/p1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "rotate_modes_resident" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print({c: f"{sum(v)/len(v):.4g}" for c, v in agg.items()})
PY
tail -3 $OUT/p1.log | cut -c1-300
