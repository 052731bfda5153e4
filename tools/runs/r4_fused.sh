ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "winograd_fused" 2>&1 | tail -3
timeout 600 python tools/probes/winograd_fused_probe.py 2>&1 | grep -v amdgpu.ids | tee $OUT/probe_r04_winograd_fused.log
