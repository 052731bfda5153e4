ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 1200 python -m pytest tests/test_models_gpu.py -q -x -k "pack or headline or unlabeled or reducer" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -4 > $OUT/r35_tests.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/r35_cfg3.json 2> $OUT/r35_cfg3.err
python bench.py --workload cfg5 --steps 6 --warmup 2 --no-cpu-baseline > $OUT/r35_cfg5.json 2> $OUT/r35_cfg5.err
cat $OUT/r35_tests.log; tail -1 $OUT/r35_cfg3.json | cut -c1-200; tail -1 $OUT/r35_cfg5.json | cut -c1-220; tail -3 $OUT/r35_cfg3.err
