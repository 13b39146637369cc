#!/bin/bash
# PMC passes of the fill kernel: scratch/pmc_fill.sh <outdir> [frames] [config]
# (rocprofv3 must get the python program itself after --; counters in separate passes)
out=$1; F=${2:-20000}; cfg=${3:-C2}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/$out
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE"; do
  tag=$(echo $set | cut -d' ' -f2)
  rocprofv3 --pmc $set --output-format csv -d $R/$out/pmc_$tag -o run -- python3 $R/scratch/prof_fill_raw.py $F $cfg > $R/$out/pmc_$tag.log 2>&1 || exit 1
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$R/$out/pmc_*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:28]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[(k, r["Counter_Name"])] += 1
        for k, v in agg.items():
            if "fill" in k:
                print(k, {c: round(x / n[(k, c)]) for c, x in v.items()})
PY
