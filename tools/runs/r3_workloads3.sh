# end-of-round re-measurement of the other workloads (GPU side only; the CPU baselines of cfg1 / cfg2 were measured earlier in the
# round on the same host type and are carried over by tools/merge_cpu_baseline.py) + the random-geometry stress of the conv kernels
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 200 python tools/stress_conv.py 150 > $OUT/r3_stress.log 2>&1
timeout 300 python bench.py --workload cfg1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_r03_cfg1.json 2> $OUT/bench_r03_cfg1.err
timeout 300 python bench.py --workload cfg2 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_r03_cfg2.json 2> $OUT/bench_r03_cfg2.err
timeout 300 python bench.py --workload cfg3pad --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_r03_cfg3pad.json 2> $OUT/bench_r03_cfg3pad.err
timeout 300 python bench.py --workload cfg5 --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench_r03_cfg5.json 2> $OUT/bench_r03_cfg5.err
SEGSDE_FORCE_REDUCER=1 timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > $OUT/bench_r03_cfg3_forced_rccl_reducer_1rank.json 2> $OUT/bench_r03_forced.err
tail -3 $OUT/r3_stress.log
for w in cfg1 cfg2 cfg3pad cfg5 cfg3_forced_rccl_reducer_1rank; do tail -1 $OUT/bench_r03_$w.json | cut -c1-230; done
