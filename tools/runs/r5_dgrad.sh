ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "winograd" 2>&1 | tail -3
timeout 300 python tools/probes/r5_winograd_probe.py dgrad 2>&1 | grep -v amdgpu.ids | tee $OUT/probe_r05_winograd_dgrad_${TAG:-b}.log
timeout 1200 python -m pytest tests/test_fullsize_gpu.py -q -x -k "winograd or conv_fullsize" 2>&1 | tail -5
