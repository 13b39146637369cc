#!/bin/bash
# Collects the judged profile set of a build on the GPU box: scratch/profile_round.sh <tag>
# (outputs under gpurun_out/<tag>/; scratch/archive_profiles.py copies the summaries into profiles/)
tag=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
T="timeout -k 10 240"
B="--steps 20 --warmup 5"
$T python3 $R/bench.py $B > $O/bench.json 2> $O/bench.err || exit 1
for c in C3 C4 C5 C2h C2t; do $T python3 $R/bench.py $B --config $c --cpu-frames 0 > $O/bench_$c.json 2> $O/bench_$c.err || exit 1; done
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o run -- python3 $R/bench.py $B --cpu-frames 0 --no-scale-ref > $O/prof.log 2>&1 || exit 1
# counter passes with the default survivor / task-table sizes: the 60-us shape trial is distorted by the counter
# collection and picked another pair in some sets (54 instead of 50 VALU per ion at the same kernel time)
export SITATOR_FILL_AUTOTUNE=0
$T rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o run -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-frames 0 --no-scale-ref > $O/fetch.log 2>&1 || exit 1
$T rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o run -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-frames 0 --no-scale-ref > $O/write.log 2>&1 || exit 1
$T rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O/sq -o run -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-frames 0 --no-scale-ref > $O/sq.log 2>&1 || exit 1
$T rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --output-format csv -d $O/f64 -o run -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-frames 0 --no-scale-ref > $O/f64.log 2>&1 || exit 1
unset SITATOR_FILL_AUTOTUNE
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/e2e_c2 -o run -- python3 $R/scratch/e2e_full.py C2 100000 dotprod > $O/e2e_c2.log 2>&1 || exit 1
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/e2e_c5 -o run -- python3 $R/scratch/e2e_c5.py > $O/e2e_c5.log 2>&1 || exit 1
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/dyn -o run -- python3 $R/scratch/time_dynamics.py > $O/dyn.log 2>&1 || exit 1
# round 4: the two forms of the pass side by side (fused / not), the per-stage counters of the fill kernel, the 2-rank rehearsal
$T python3 $R/scratch/ab_fill.py C2 100000 20 > $O/ab_fill_c2.txt 2>&1 || exit 1
SWEEP_STEPS=10 $T python3 $R/scratch/sweep_env.py C5 62500 "SITATOR_FUSE=0" "SITATOR_FUSE=1" > $O/ab_fill_c5.txt 2>&1 || exit 1
timeout -k 10 400 bash $R/scratch/pmc_stages.sh gpurun_out/$tag/stages 20000 C2 > $O/stages.txt 2>&1 || exit 1
rm -rf $O/stages
SITATOR_BENCH_BACKEND=tcp $T python3 $R/bench.py --gpus 2 --config C2 --frames 30000 --steps 5 --warmup 2 --no-scale-ref > $O/bench_tcp2.json 2> $O/bench_tcp2.err || exit 1
# round 5: the 8-GPU configurations rehearsed with two ranks on the one GPU (configs[3] per GPU; configs[4] with the mcl plugin)
SITATOR_BENCH_BACKEND=tcp $T python3 $R/bench.py --gpus 2 --config C4 --steps 5 --warmup 2 --cpu-frames 0 > $O/bench_tcp2_C4.json 2> $O/bench_tcp2_C4.err || exit 1
SITATOR_BENCH_BACKEND=tcp $T python3 $R/bench.py --gpus 2 --config C5 --algo mcl --steps 5 --warmup 2 --cpu-frames 0 > $O/bench_tcp2_C5.json 2> $O/bench_tcp2_C5.err || exit 1
timeout -k 10 200 python3 $R/scratch/phase_times.py C2 100000 > $O/phase_times.txt 2>&1 || exit 1
timeout -k 10 500 bash $R/scratch/pmc_mem.sh gpurun_out/$tag/mem 100000 C2 > $O/mem_counters.txt 2>&1 || exit 1
$T python3 $R/scratch/e2e_walls.py C2 100000 4 > $O/e2e_walls_c2.txt 2>&1 || exit 1
# round 5: the assignment kernel's counters, packed columns and (for comparison) the split arrays of round 4
timeout -k 10 500 bash $R/scratch/pmc_predict.sh C2 > $O/predict_counters.txt 2>&1 || exit 1
echo "---- SITATOR_PREDICT_REC=0 (the split-array kernel of round 4) ----" >> $O/predict_counters.txt
SITATOR_PREDICT_REC=0 timeout -k 10 500 bash $R/scratch/pmc_predict.sh C2 >> $O/predict_counters.txt 2>&1 || exit 1
SWEEP_FIT_FRAMES=100000 $T python3 $R/scratch/sweep_env.py C2 100000 "" "SITATOR_PREDICT_REC=0" "SITATOR_PREDICT_LDS=0" > $O/ab_predict_c2.txt 2>&1 || exit 1
sha256sum $R/sitator_amd/lib/libsitator_hip.so | cut -c1-16 > $O/lib_sha16
find $O -name "*.db" -delete
tail -n 3 $O/e2e_c2.log $O/e2e_c5.log $O/dyn.log
echo profile set $tag done
