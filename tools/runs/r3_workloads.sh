# all workloads of BASELINE.json re-measured (cfg1 / cfg2 with their CPU baselines on this box's host cores)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -x -k "two_steps" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -15 > $OUT/r3_tests_pad.log
timeout 600 python bench.py --workload cfg1 --steps 20 --warmup 5 --cpu-baseline-timeout 200 > $OUT/bench_r03_cfg1.json 2> $OUT/bench_r03_cfg1.err
timeout 600 python bench.py --workload cfg2 --steps 20 --warmup 5 --cpu-baseline-timeout 200 > $OUT/bench_r03_cfg2.json 2> $OUT/bench_r03_cfg2.err
timeout 600 python bench.py --workload cfg3pad --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_r03_cfg3pad.json 2> $OUT/bench_r03_cfg3pad.err
timeout 600 python bench.py --workload cfg5 --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench_r03_cfg5.json 2> $OUT/bench_r03_cfg5.err
tail -4 $OUT/r3_tests_pad.log
for w in cfg1 cfg2 cfg3pad cfg5; do tail -1 $OUT/bench_r03_$w.json | cut -c1-260; done
