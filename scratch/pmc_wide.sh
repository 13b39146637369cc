#!/bin/bash
# Wide counter sweep of the fill kernel (separate passes): scratch/pmc_wide.sh <outdir> [frames] [config]
out=$1; F=${2:-100000}; cfg=${3:-C2}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/$out
i=0
for set in "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64" \
           "TA_TA_BUSY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_WAIT_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $R/$out/set$i -o run -- python3 $R/scratch/prof_fill_raw.py $F $cfg > $R/$out/set$i.log 2>&1 || { echo "set $i failed"; tail -3 $R/$out/set$i.log; }
done
python3 - <<PY
import csv, glob, collections
ions = $F * {"C2": 64, "C3": 448, "C4": 256, "C5": 160, "C1": 4}["$cfg"]
for f in sorted(glob.glob("$R/$out/set*/**/*counter_collection.csv", recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if "k_fill" in r["Kernel_Name"]]
    if not rows: continue
    big = max(int(r["Grid_Size"]) for r in rows)
    agg = collections.defaultdict(float); n = collections.Counter()
    for r in rows:
        if int(r["Grid_Size"]) == big:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    print({c: round(x / n[c] / ions, 3) for c, x in agg.items()}, "dispatches", max(n.values()))
PY
