ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "disp or actgrad or head or conv" 2>&1 | tail -2 > $OUT/r39_tests.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof39 -o t -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing > $OUT/r39_trace.log 2>&1
db=$(ls $OUT/prof39/*.db 2>/dev/null | head -1)
[ -n "$db" ] && python $ROOT/tools/prof_summary.py $db $OUT/r39_trace.txt > /dev/null
rm -rf $OUT/prof39
cat $OUT/r39_tests.log; grep "c1s_\|total kernel" $OUT/r39_trace.txt | cut -c1-110
