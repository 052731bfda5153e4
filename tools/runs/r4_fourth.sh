ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 600 python tools/probes/stats_epilogue_probe.py 2>&1 | grep -v amdgpu.ids > $OUT/probe_r04_stats_epilogue.log
cat $OUT/probe_r04_stats_epilogue.log
timeout 1200 python -m pytest tests -m gpu -q -k "cfg2_batch8 or train_step_replay" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" > $OUT/r4_fourth_tests.log
grep -n "median relative\|worst deviations" $OUT/r4_fourth_tests.log; tail -4 $OUT/r4_fourth_tests.log
