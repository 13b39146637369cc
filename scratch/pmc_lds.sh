#!/bin/bash
# LDS counters of the fill kernel: scratch/pmc_lds.sh <outdir> [frames] [config]
out=$1; F=${2:-20000}; cfg=${3:-C2}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/$out
rocprofv3 --pmc SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES --output-format csv -d $R/$out/lds -o run -- python3 $R/scratch/prof_fill_raw.py $F $cfg > $R/$out/lds.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INST_LEVEL_LDS --output-format csv -d $R/$out/act -o run -- python3 $R/scratch/prof_fill_raw.py $F $cfg > $R/$out/act.log 2>&1 || exit 1
python3 - <<PY
import csv, glob, collections
for sub in ("lds", "act"):
    for f in glob.glob("$R/$out/%s/**/*counter_collection.csv" % sub, recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "k_fill" in r["Kernel_Name"]]
        big = max(int(r["Grid_Size"]) for r in rows)
        agg = collections.defaultdict(float); n = collections.Counter()
        for r in rows:
            if int(r["Grid_Size"]) == big:
                agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
        ions = $F * {"C2": 64, "C3": 448, "C4": 256, "C5": 160, "C1": 4}["$cfg"]
        print(sub, {c: round(x / n[c] / ions, 2) for c, x in agg.items()})
PY
