#!/bin/bash
# VALU / SALU wave-instructions per ion and kernel time of k_fill3 for several builds of the library: scratch/pmc_libs.sh lib1.so lib2.so ...
cd /tmp && export TMPDIR=/tmp
export SITATOR_FILL_AUTOTUNE=0
R=$GRAFT_REPO_ROOT
for lib in "$@"; do
  export SITATOR_LIB=$R/$lib
  rm -rf /tmp/pl; timeout -k 5 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_ACTIVE_INST_VALU --output-format csv -d /tmp/pl -o run -- python3 $R/scratch/prof_fill_raw.py 100000 C2 > /tmp/pl.log 2>&1
  python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob("/tmp/pl/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_fill3" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
print("$lib", {c: round(x / n[c] / 6.4e6, 2) for c, x in agg.items()})
PY
done
unset SITATOR_LIB
