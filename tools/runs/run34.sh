ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 1200 python -m pytest tests/test_models_gpu.py tests/test_fullsize_gpu.py -q -x -k "unlabeled or cfg5 or reducer or depthmix" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -4 > $OUT/r34_tests.log
python bench.py --workload cfg5 --steps 6 --warmup 2 --no-cpu-baseline > $OUT/r34_cfg5.json 2> $OUT/r34_cfg5.err
SEGSDE_FORCE_REDUCER=1 python bench.py --workload cfg5 --steps 4 --warmup 2 --no-cpu-baseline > $OUT/r34_cfg5_reducer.json 2> $OUT/r34_cfg5_reducer.err
cat $OUT/r34_tests.log; tail -1 $OUT/r34_cfg5.json | cut -c1-220; tail -1 $OUT/r34_cfg5_reducer.json | cut -c1-220
