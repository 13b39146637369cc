#!/bin/bash
# VALU / LDS / wave-cycle counters of the fill kernel per ablation stage: scratch/pmc_stages.sh <outdir> [frames] [config]
out=$1; F=${2:-20000}; cfg=${3:-C2}
cd /tmp && export TMPDIR=/tmp
export SITATOR_FILL_AUTOTUNE=0      # the 60-us shape trials would be averaged in with the launches proper
R=$GRAFT_REPO_ROOT
mkdir -p $R/$out
for stop in 1 2 3 4 5 0; do
  export SITATOR_DEBUG_STOP=$stop
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $R/$out/stop$stop -o run -- python3 $R/scratch/prof_fill_raw.py $F $cfg > $R/$out/stop$stop.log 2>&1 || exit 1
done
python3 - <<PY
import csv, glob, collections
for stop in (1, 2, 3, 4, 5, 0):
    for f in glob.glob("$R/$out/stop%d/**/*counter_collection.csv" % stop, recursive=True):
        agg = collections.defaultdict(float); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            if "k_fill" in r["Kernel_Name"]:
                agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
        ions = $F * {"C2": 64, "C3": 448, "C4": 256, "C5": 160, "C1": 4}["$cfg"]
        # (these passes see 0.05 waves per ion at C2 where 1 / 16 run: scale a row by 0.0625 / SQ_WAVES to compare it with
        # the counter passes of bench.py)
        print("stop", stop, {c: round(x / n[c] / ions, 2) for c, x in agg.items()})
PY
