cd $GRAFT_REPO_ROOT
for r in 1 2; do
echo "== default lib"; python3 scratch/sweep_env.py C2 100000 "" "SITATOR_FILL_RCAP=48,SITATOR_FILL_TCAP=64" 2>&1 | tail -2
echo "== wpe8 lib"; SITATOR_LIB=$PWD/scratch/_bin/libsitator_hip_wpe8.so SITATOR_DEBUG_SHAPE=1 python3 scratch/sweep_env.py C2 100000 "" "SITATOR_FILL_RCAP=48,SITATOR_FILL_TCAP=64" "SITATOR_FILL_RCAP=40,SITATOR_FILL_TCAP=64" 2>&1 | grep -v "^predict" | sort -u | tail -8
done
