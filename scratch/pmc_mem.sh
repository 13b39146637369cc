#!/bin/bash
# Vector-memory-path counters of the fill kernel (TA / TCP / TD busy and stall cycles): scratch/pmc_mem.sh <outdir> [frames] [config]
out=$1; F=${2:-100000}; cfg=${3:-C2}
cd /tmp && export TMPDIR=/tmp
export SITATOR_FILL_AUTOTUNE=0
R=$GRAFT_REPO_ROOT
mkdir -p $R/$out
i=0
for set in "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TCP_GATE_EN1_sum TD_TD_BUSY_sum" \
           "TA_TOTAL_WAVEFRONTS_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TD_TC_STALL_sum" \
           "TA_FLAT_READ_LDS_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TD_LOAD_WAVEFRONT_sum" \
           "TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_READ_sum" \
           "SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  echo pass $i: $set; timeout -k 5 90 rocprofv3 --pmc $set --output-format csv -d $R/$out/p$i -o run -- python3 $R/scratch/prof_fill_raw.py $F $cfg > $R/$out/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $R/$out/p$i.log; }
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob("$R/$out/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_fill3" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for c in sorted(agg):
    print("%-44s %16.1f per launch (%d launches)" % (c, agg[c] / n[c], n[c]))
PY
rm -rf $R/$out/p*/
