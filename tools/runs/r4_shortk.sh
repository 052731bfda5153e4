ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
for t in "skn=0" "skn=1" "skn=2"; do
  echo "=== SEGSDE_TUNE=$t"
  SEGSDE_TUNE=$t python tools/probes/stats_epilogue_probe.py 2>&1 | grep -v amdgpu.ids | grep "256->1024\|1024->256\|64->256\|128->512\|per step"
  SEGSDE_TUNE=$t python tools/probes/winograd_route_probe.py 2>&1 | grep -v amdgpu.ids | head -12
done > $OUT/probe_r04_short_k.log 2>&1
cat $OUT/probe_r04_short_k.log
