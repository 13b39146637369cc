#!/bin/bash
# Which launch shape does the fill's measured choice settle on, process after process?  scratch/autotune_stability.sh <outfile> [config] [frames] [runs]
out=$1; cfg=${2:-C2}; F=${3:-20000}; n=${4:-12}
: > $out
for i in $(seq 1 $n); do
  SITATOR_DEBUG_SHAPE=1 python3 scratch/prof_fill_raw.py $F $cfg 2>&1 | grep "k_fill3 shape\|fill ms" | tail -2 | tr '\n' ' ' >> $out; echo >> $out
done
sort $out | cut -c1-60 | uniq -c
