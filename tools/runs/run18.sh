ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $OUT/r18_tests.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/r18_bench.json 2> $OUT/r18_bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof18 -o t -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing > $OUT/r18_trace.log 2>&1
db=$(ls $OUT/prof18/*.db 2>/dev/null | head -1)
[ -n "$db" ] && python $ROOT/tools/prof_summary.py $db $OUT/r18_trace.txt > /dev/null
rm -rf $OUT/prof18
cat $OUT/r18_tests.log; cat $OUT/r18_bench.json | cut -c1-300
