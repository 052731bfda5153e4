ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "winograd" 2>&1 | tail -2
rm -f $OUT/probe_r05_winograd_fused_ublk.log
for u in 1 0; do
  echo "# SEGSDE_WINO_FUSED_UBLK=$u" | tee -a $OUT/probe_r05_winograd_fused_ublk.log
  SEGSDE_WINO_FUSED_UBLK=$u timeout 300 python tools/probes/winograd_fused_probe.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/probe_r05_winograd_fused_ublk.log
  SEGSDE_WINO_FUSED_UBLK=$u timeout 300 python tools/probes/r5_winograd_probe.py fwd2 dgrad 2>&1 | grep -v amdgpu.ids | tee -a $OUT/probe_r05_winograd_fused_ublk.log
done
