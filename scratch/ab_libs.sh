#!/bin/bash
# Same-box A/B of two builds of the library on the fill + assignment pass (separate processes, alternating):
#   scratch/ab_libs.sh <config> <frames> <libA.so> <libB.so> [rounds]
cfg=$1; F=$2; A=$3; B=$4; R=${5:-2}
cd $GRAFT_REPO_ROOT
for r in $(seq 1 $R); do
  for lib in $A $B; do
    echo "== $lib"
    SITATOR_LIB=$PWD/$lib SWEEP_STEPS=20 python3 scratch/sweep_env.py $cfg $F "" 2>&1 | tail -1
  done
done
