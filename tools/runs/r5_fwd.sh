ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "winograd" 2>&1 | tail -3
timeout 300 python tools/probes/winograd_fused_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/probe_r05_winograd_fused_${TAG:-b}.log
timeout 300 python tools/probes/r5_winograd_probe.py dgrad fwd2 2>&1 | grep -v amdgpu.ids | tee $OUT/probe_r05_winograd_dgrad_fwd2_${TAG:-b}.log
