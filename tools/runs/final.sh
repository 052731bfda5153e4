# full GPU verification: tests, smoke, the driver's default bench line
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -3 > $OUT/final_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $OUT/final_smoke.log 2>&1
python bench.py > $OUT/final_bench_default.json 2> $OUT/final_bench_default.err
cat $OUT/final_tests.log; tail -1 $OUT/final_smoke.log; tail -1 $OUT/final_bench_default.json | cut -c1-250
