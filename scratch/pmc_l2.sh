#!/bin/bash
# L2 counters of the fill kernel, frames from HBM vs frames re-read from the L2 (SITATOR_F3_FRAME_MOD): scratch/pmc_l2.sh <outdir>
# (two counters per pass: more TCC counters at once exceed what the hardware collects and the profiler aborts)
out=$1
cd /tmp && export TMPDIR=/tmp
export SITATOR_FILL_AUTOTUNE=0
R=$GRAFT_REPO_ROOT
mkdir -p $R/$out
for mod in 0 1024; do
  export SITATOR_F3_FRAME_MOD=$mod
  p=0
  for ctrs in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
    p=$((p+1))
    timeout -k 10 150 rocprofv3 --pmc $ctrs --output-format csv -d $R/$out/mod${mod}_$p -o run -- python3 $R/scratch/prof_fill_raw.py 100000 C2 > $R/$out/mod${mod}_$p.log 2>&1 || { echo "pass failed: $ctrs"; tail -3 $R/$out/mod${mod}_$p.log; exit 1; }
    echo "pass $mod $p done"
  done
done
python3 - <<PY
import csv, glob, collections
for mod in (0, 1024):
    agg = collections.defaultdict(float); n = collections.Counter()
    for f in glob.glob("$R/$out/mod%d_*/**/*counter_collection.csv" % mod, recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_fill3" in r["Kernel_Name"]:
                agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    print("frame_mod", mod, {c: round(x / n[c] / 6.4e6, 2) for c, x in agg.items()}, "per ion")
PY
