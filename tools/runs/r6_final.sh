# round-6 evidence run: the whole GPU suite, smoke, the driver's default bench command (with CPU baseline), per-layer table,
# kernel trace, the PMC passes and the other workloads -- all on one box, in this order
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 2700 python -m pytest tests -m gpu -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -8 > $OUT/r6_final_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $OUT/r6_final_smoke.log 2>&1
PMC_TAG=r06 bash tools/gpu_pmc.sh > $OUT/r6_pmc.log 2>&1
cd $ROOT
# the PMC summaries are what the bench line's traffic / mfma_busy fields read: put them where bench.py looks before the headline run
for f in r06_pmc_mfma r06_pmc_waits r06_pmc_fetch r06_pmc_write; do :; done
cp $OUT/r06_pmc_mfma.txt profiles/pmc_r06_mfma_busy.txt 2>/dev/null
cp $OUT/r06_pmc_waits.txt profiles/pmc_r06_sq_waits.txt 2>/dev/null
cp $OUT/r06_pmc_fetch.txt profiles/pmc_r06_fetch_size.txt 2>/dev/null
cp $OUT/r06_pmc_write.txt profiles/pmc_r06_write_size.txt 2>/dev/null
cp $OUT/traffic_r06.json profiles/traffic_r06.json 2>/dev/null
SEGSDE_BENCH_LAYERS=$OUT/layers_r06_latest.txt python bench.py > $OUT/bench_r06_cfg3_default_run.json 2> $OUT/bench_r06_cfg3_default_run.err
bash tools/runs/trace.sh r06_final
bash tools/runs/trace.sh r06_reference_step --step reference
bash tools/runs/trace.sh r06_amp_step --step amp
cd $ROOT
timeout 600 python bench.py --workload cfg1 --steps 20 --warmup 5 --cpu-baseline-timeout 120 > $OUT/bench_r06_cfg1.json 2> $OUT/bench_r06_cfg1.err
timeout 600 python bench.py --workload cfg2 --steps 20 --warmup 5 --cpu-baseline-timeout 120 > $OUT/bench_r06_cfg2.json 2> $OUT/bench_r06_cfg2.err
timeout 600 python bench.py --workload cfg3pad --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_r06_cfg3pad.json 2> $OUT/bench_r06_cfg3pad.err
timeout 600 python bench.py --workload cfg5 --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench_r06_cfg5.json 2> $OUT/bench_r06_cfg5.err
SEGSDE_FORCE_REDUCER=1 timeout 600 python bench.py --no-cpu-baseline 2> $OUT/bench_r06_forced.err | tail -1 > $OUT/bench_r06_cfg3_forced_rccl_reducer_1rank.json
tail -4 $OUT/r6_final_tests.log; tail -1 $OUT/r6_final_smoke.log
for f in bench_r06_cfg3_default_run bench_r06_cfg1 bench_r06_cfg2 bench_r06_cfg3pad bench_r06_cfg5 bench_r06_cfg3_forced_rccl_reducer_1rank; do tail -1 $OUT/$f.json | cut -c1-230; done
ls $OUT | grep "r06_pmc\|traffic_r06\|trace_r06"
