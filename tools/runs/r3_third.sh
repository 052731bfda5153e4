ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 1200 python -m pytest tests -m gpu -q -x -k "fullsize_sampled or upsample_folded or depthmix_unlabeled or conv_random or test_conv or real_model_two_ranks or selfspawn" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -40 > $OUT/r3_tests3.log
bash tools/runs/trace.sh r3b
tail -8 $OUT/r3_tests3.log; head -45 $OUT/trace_r3b.txt; tail -3 $OUT/trace_r3b.log | cut -c1-400
