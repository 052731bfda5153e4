# kernel trace of one probe section: which launches the time goes to  (PROBE_ARGS="dgrad" ...)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_probe
rocprofv3 --kernel-trace --stats -d $OUT/prof_probe -o probe -- python $ROOT/tools/probes/r5_winograd_probe.py ${PROBE_ARGS:-dgrad} > $OUT/prof_probe.log 2>&1
db=$(ls $OUT/prof_probe/*.db 2>/dev/null | head -1)
[ -n "$db" ] && python $ROOT/tools/prof_summary.py $db $OUT/prof_probe_summary.txt > /dev/null
rm -rf $OUT/prof_probe
head -40 $OUT/prof_probe_summary.txt
