#!/bin/bash
# fill-kernel time of bench.py under a list of environment settings: scratch/ab_fill_env.sh <outfile> "<VAR=val ...>" ...
out=$1; shift
: > $out
for cfg in "$@"; do
  for rep in 1 2; do
    env $cfg SITATOR_DEBUG_SHAPE=1 python3 bench.py --steps 10 --warmup 2 --cpu-frames 0 2> /tmp/ab_err.txt | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$cfg', 'fill ms', round(d['roofline']['kernel_ms'],4), 'step', round(d['ms_per_step'],4), flush=True)" >> $out
  done
  grep "k_fill3 shape" /tmp/ab_err.txt | tail -1 >> $out
done
cat $out
