ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof38 -o t -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing > $OUT/r38_trace.log 2>&1
db=$(ls $OUT/prof38/*.db 2>/dev/null | head -1)
[ -n "$db" ] && python $ROOT/tools/prof_summary.py $db $OUT/r38_trace.txt > /dev/null
rm -rf $OUT/prof38
grep "pack_weight\|total kernel" $OUT/r38_trace.txt | cut -c1-120; tail -1 $OUT/r38_trace.log | cut -c1-200
