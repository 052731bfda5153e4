# round 4, sixth box: the Winograd route as built -- GPU test, per-layer probe, whole-step A/B (SEGSDE_WINOGRAD=0 / 1)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -q -k "winograd or cfg2_batch8 or train_step_replay" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -30 > $OUT/r4_sixth_tests.log
tail -5 $OUT/r4_sixth_tests.log
timeout 600 python tools/probes/winograd_route_probe.py 2>&1 | grep -v amdgpu.ids > $OUT/probe_r04_winograd_route.log
cat $OUT/probe_r04_winograd_route.log
SEGSDE_WINOGRAD=0 timeout 900 python bench.py --no-cpu-baseline > $OUT/bench_r04_wino_off.json 2> $OUT/bench_r04_wino_off.err
SEGSDE_BENCH_LAYERS=$OUT/layers_r04_wino.txt timeout 900 python bench.py --no-cpu-baseline > $OUT/bench_r04_wino_on.json 2> $OUT/bench_r04_wino_on.err
for f in bench_r04_wino_off bench_r04_wino_on; do tail -1 $OUT/$f.json | cut -c1-260; done
grep wino $OUT/layers_r04_wino.txt | head
