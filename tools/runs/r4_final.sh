# round-4 evidence run: the whole GPU suite, smoke, the driver's default bench command (with CPU baseline), per-layer table,
# kernel trace, the PMC passes and the other workloads -- all on one box, in this order
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -8 > $OUT/r4_final_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $OUT/r4_final_smoke.log 2>&1
SEGSDE_BENCH_LAYERS=$OUT/layers_r04_latest.txt python bench.py > $OUT/bench_r04_cfg3_default_run.json 2> $OUT/bench_r04_cfg3_default_run.err
bash tools/runs/trace.sh r04_final
PMC_TAG=r04 bash tools/gpu_pmc.sh > $OUT/r4_pmc.log 2>&1
cd $ROOT
timeout 600 python bench.py --workload cfg1 --steps 20 --warmup 5 --cpu-baseline-timeout 120 > $OUT/bench_r04_cfg1.json 2> $OUT/bench_r04_cfg1.err
timeout 600 python bench.py --workload cfg2 --steps 20 --warmup 5 --cpu-baseline-timeout 120 > $OUT/bench_r04_cfg2.json 2> $OUT/bench_r04_cfg2.err
timeout 600 python bench.py --workload cfg3pad --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_r04_cfg3pad.json 2> $OUT/bench_r04_cfg3pad.err
timeout 600 python bench.py --workload cfg5 --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench_r04_cfg5.json 2> $OUT/bench_r04_cfg5.err
SEGSDE_FORCE_REDUCER=1 timeout 600 python bench.py --no-cpu-baseline > $OUT/bench_r04_cfg3_forced_rccl_reducer_1rank.json 2> $OUT/bench_r04_forced.err
tail -3 $OUT/r4_final_tests.log; tail -1 $OUT/r4_final_smoke.log
for f in bench_r04_cfg3_default_run bench_r04_cfg1 bench_r04_cfg2 bench_r04_cfg3pad bench_r04_cfg5 bench_r04_cfg3_forced_rccl_reducer_1rank; do tail -1 $OUT/$f.json | cut -c1-230; done
ls $OUT | grep "r04_pmc\|traffic_r04\|trace_r04"
