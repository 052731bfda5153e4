# round 5, first GPU call: the new Winograd directions -- parity tests on the real library, then the probe (every staging variant
# of the weight gradient), then a short bench with the new routes on
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "winograd or layer_callables" 2>&1 | tail -5 | tee $OUT/r5_first_tests.log
timeout 300 python tools/probes/r5_winograd_probe.py dgrad fwd2 2>&1 | grep -v amdgpu.ids | tee $OUT/probe_r05_winograd_dgrad_fwd2.log
for v in 1 0 2; do
  SEGSDE_WGRAD_FUSED_VAR=$v timeout 300 python tools/probes/r5_winograd_probe.py wgrad 2>&1 | grep -v amdgpu.ids | tee $OUT/probe_r05_winograd_wgrad_var$v.log
done
SEGSDE_WGRAD_FUSED_WGS=1024 timeout 300 python tools/probes/r5_winograd_probe.py wgrad 2>&1 | grep -v amdgpu.ids | tee $OUT/probe_r05_winograd_wgrad_var1_wgs1024.log
SEGSDE_BENCH_LAYERS=$OUT/layers_r05_first.txt timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_r05_first.json 2> $OUT/bench_r05_first.err
tail -c 600 $OUT/bench_r05_first.json
