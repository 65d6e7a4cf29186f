#!/bin/bash
# SQ counters of synthesis_eval_kernel (default build, opt-in route) and of the old synthesis kernel for comparison
export SCRI_AMD_SYNTHESIS_EVAL=1
bash tools/pmc_kernel_probe.sh syneval synthesis_eval -- python3 $PWD/tools/boost_free_rate.py 100000 2>&1 | tail -1
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_syneval_icache; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH --output-format csv -d $OUT/p1 -- python3 $REPO/tools/boost_free_rate.py 100000 > $OUT/p1.log 2>&1
cd $REPO
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/p1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "synthesis_eval" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print({c: f"{sum(v)/len(v):.4g}" for c, v in agg.items()})
PY
unset SCRI_AMD_SYNTHESIS_EVAL
bash tools/pmc_kernel_probe.sh synsplit synthesis_split -- python3 $PWD/tools/boost_free_rate.py 100000 2>&1 | tail -1
