ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "winograd" 2>&1 | tail -3
for v in ${VARS:-0 1}; do
  SEGSDE_WGRAD_FUSED_VAR=$v timeout 300 python tools/probes/r5_winograd_probe.py wgrad 2>&1 | grep -v amdgpu.ids | tee $OUT/probe_r05_winograd_wgrad_${TAG:-b}_var$v.log
done
