ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py -q -x -k "conv or elementwise or resize or misc" 2>&1 | tail -4 > $OUT/r33_tests.log
python bench.py --workload cfg5 --steps 6 --warmup 2 --no-cpu-baseline > $OUT/r33_cfg5.json 2> $OUT/r33_cfg5.err
python bench.py --workload cfg3pad --steps 6 --warmup 2 --no-cpu-baseline > $OUT/r33_cfg3pad.json 2> $OUT/r33_cfg3pad.err
cat $OUT/r33_tests.log; tail -1 $OUT/r33_cfg5.json | cut -c1-220; tail -1 $OUT/r33_cfg3pad.json | cut -c1-220
