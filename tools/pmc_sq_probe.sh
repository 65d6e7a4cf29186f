#!/bin/bash
# SQ counters of the bench's kernels (LDS conflicts, wait buckets): bash tools/pmc_sq_probe.sh  (through gpurun, repo root)
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES"; do
  tag=$(echo $set | cut -d' ' -f1)
  rm -rf $OUT/sq_$tag
  rocprofv3 --pmc $set --output-format csv -d $OUT/sq_$tag -- python3 $REPO/bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-live-pmc > $OUT/sq_$tag.log 2>&1
done
cd $REPO
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/sq_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
        if "bms::" in n:
            agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    if any(sum(x) / len(x) > 1e6 for x in v.values()):
        print(k, {c: f"{sum(x)/len(x):.3g}" for c, x in sorted(v.items())})
PY
