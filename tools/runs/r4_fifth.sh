ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 600 python tools/probes/winograd_gate_probe.py 2>&1 | grep -v amdgpu.ids > $OUT/probe_r04_winograd_gate.log
cat $OUT/probe_r04_winograd_gate.log
timeout 1500 python -m pytest tests -m gpu -q -k "cfg2_batch8 or train_step_replay" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" > $OUT/r4_fifth_tests.log
grep -n "per-parameter relative\|worst deviations" $OUT/r4_fifth_tests.log; tail -4 $OUT/r4_fifth_tests.log
