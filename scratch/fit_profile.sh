#!/bin/bash
# fit tests, then kernel traces of the end-to-end runs: scratch/fit_profile.sh <tag>
tag=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag
mkdir -p $O
cd $R && timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "fit" > $O/tests.txt 2>&1 || { tail -30 $O/tests.txt; exit 1; }
tail -1 $O/tests.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 250 rocprofv3 --kernel-trace --stats --output-format csv -d $O/e2e_c2 -o run -- python3 $R/scratch/e2e_full.py C2 100000 dotprod > $O/e2e_c2.log 2>&1 || exit 1
timeout -k 10 250 rocprofv3 --kernel-trace --stats --output-format csv -d $O/e2e_c5 -o run -- python3 $R/scratch/e2e_c5.py > $O/e2e_c5.log 2>&1 || exit 1
head -1 $O/e2e_c2.log; grep dotprod $O/e2e_c5.log
