ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_models_gpu.py -m gpu -q -x -k "depthmix_unlabeled_step_vs_oracle" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -60 > $OUT/r3_test_depthmix.log
bash tools/runs/trace.sh r3a
tail -30 $OUT/r3_test_depthmix.log; head -40 $OUT/trace_r3a.txt
