# round-3 evidence re-run after the test-criterion fix (kernel sources unchanged since the PMC passes of r3_final.sh):
# the whole GPU suite, smoke, the driver's default bench command (with CPU baseline), per-layer table, kernel trace
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -8 > $OUT/final_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $OUT/final_smoke.log 2>&1
SEGSDE_BENCH_LAYERS=$OUT/layers_r03_latest.txt python bench.py > $OUT/bench_r03_cfg3_default_run.json 2> $OUT/bench_r03_cfg3_default_run.err
bash tools/runs/trace.sh r03_final
tail -3 $OUT/final_tests.log; tail -1 $OUT/final_smoke.log; tail -1 $OUT/bench_r03_cfg3_default_run.json | cut -c1-400
