ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 1500 python tests/diag/winograd_step_sensitivity.py 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" > $OUT/diag_r04_winograd_step_sensitivity.log
cat $OUT/diag_r04_winograd_step_sensitivity.log
