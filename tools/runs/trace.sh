# usage: bash tools/runs/trace.sh <tag> [bench args]  -> gpurun_out/trace_<tag>.txt (rocprofv3 kernel trace summary of bench.py)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
tag=$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof_$tag -o t -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-reference-step --no-amp-step "$@" > $OUT/trace_$tag.log 2>&1
db=$(ls $OUT/prof_$tag/*.db 2>/dev/null | head -1)
[ -n "$db" ] && python $ROOT/tools/prof_summary.py $db $OUT/trace_$tag.txt > /dev/null
rm -rf $OUT/prof_$tag
