#!/bin/bash
# SQ counters of one kernel under a command: bash tools/pmc_kernel_probe.sh <tag> <kernel-substring> -- python3 tools/x.py args
# (through gpurun, repo root).  One rocprofv3 --pmc pass per counter group; prints per-kernel means.
set -u
TAG=$1; KSUB=$2; shift 3
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU_MFMA_F64" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" "FETCH_SIZE" "WRITE_SIZE" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  rm -rf $OUT/p$i
  CPU_BASELINE=0 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- "$@" > $OUT/p$i.log 2>&1
done
cd $REPO
python3 - "$OUT" "$KSUB" <<'PY'
import csv, glob, collections, sys, json
out, ksub = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if ksub in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {c: sum(v) / len(v) for c, v in sorted(agg.items())}
res["_launches_seen"] = {c: len(v) for c, v in agg.items()}
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps({k: (f"{v:.4g}" if isinstance(v, float) else v) for k, v in res.items() if k != "_launches_seen"}))
PY
