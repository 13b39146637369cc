#!/bin/bash
# Counters of the assignment kernels (k_predict_rows_*) over bench.py's C2 step: scratch/pmc_predict.sh [config] ; env passes through
# (SITATOR_PREDICT_REC=0 for the split-array kernel).  Two or three counters of a block per pass (more make rocprofv3 abort).
cd /tmp && export TMPDIR=/tmp
export SITATOR_FILL_AUTOTUNE=0
R=$GRAFT_REPO_ROOT
cfg=${1:-C2}
for set in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS" "SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  rm -rf /tmp/pp; timeout -k 5 150 rocprofv3 --pmc $set --output-format csv -d /tmp/pp -o run -- python3 $R/bench.py --config $cfg --steps 2 --warmup 1 --cpu-frames 0 --no-scale-ref > /tmp/pp.log 2>&1 || { echo "pass failed: $set"; tail -3 /tmp/pp.log; continue; }
  python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob("/tmp/pp/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_predict_rows" in k:
            key = (k.split("(")[0][-40:], r["Counter_Name"])
            agg[key] += float(r["Counter_Value"]); n[key] += 1
for key in sorted(agg): print(key[0], key[1], "per launch %.4g" % (agg[key] / n[key]), "launches", n[key])
PY
done
