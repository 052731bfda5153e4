# after a kernel change late in the round: the affected GPU tests, the driver's default bench command, the kernel trace and the
# PMC passes (the full suite / the other workloads ran on the previous sources: tools/runs/r4_final.sh)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests -m gpu -q -x -k "winograd" 2>&1 | tail -2 > $OUT/r4_refresh_tests.log
cat $OUT/r4_refresh_tests.log
SEGSDE_BENCH_LAYERS=$OUT/layers_r04_latest.txt python bench.py > $OUT/bench_r04_cfg3_default_run.json 2> $OUT/bench_r04_cfg3_default_run.err
bash tools/runs/trace.sh r04_final
PMC_TAG=r04 bash tools/gpu_pmc.sh > $OUT/r4_pmc.log 2>&1
cd $ROOT
tail -1 $OUT/bench_r04_cfg3_default_run.json | cut -c1-230
ls $OUT | grep "r04_pmc\|traffic_r04\|trace_r04"
