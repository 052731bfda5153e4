ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -3 > $OUT/r37_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $OUT/r37_smoke.log 2>&1
python bench.py > $OUT/r37_bench_default.json 2> $OUT/r37_bench_default.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof37 -o t -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing > $OUT/r37_trace.log 2>&1
db=$(ls $OUT/prof37/*.db 2>/dev/null | head -1)
[ -n "$db" ] && python $ROOT/tools/prof_summary.py $db $OUT/r37_trace.txt > /dev/null
rm -rf $OUT/prof37
cat $OUT/r37_tests.log; tail -1 $OUT/r37_smoke.log; tail -1 $OUT/r37_bench_default.json | cut -c1-250
