ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof32 -o t -- python $ROOT/bench.py --workload cfg5 --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing > $OUT/r32_trace.log 2>&1
db=$(ls $OUT/prof32/*.db 2>/dev/null | head -1)
[ -n "$db" ] && python $ROOT/tools/prof_summary.py $db $OUT/r32_cfg5_trace.txt > /dev/null
rm -rf $OUT/prof32
tail -1 $OUT/r32_trace.log | cut -c1-200
