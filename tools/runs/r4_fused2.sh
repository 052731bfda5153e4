# the one-kernel Winograd route wired into the model: kernel + model parity tests, then the step with / without it
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -q -x -k "winograd or reference_vectors or big_models or headline or blocks or decoders or encoder" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -5
SEGSDE_BENCH_LAYERS=$OUT/layers_r04_fused.txt timeout 600 python bench.py --no-cpu-baseline > $OUT/bench_r04_wino_fused_on.json 2> $OUT/bench_r04_wino_fused.err
SEGSDE_WINO_FUSED=0 timeout 600 python bench.py --no-cpu-baseline > $OUT/bench_r04_wino_fused_off.json 2>> $OUT/bench_r04_wino_fused.err
for f in bench_r04_wino_fused_on bench_r04_wino_fused_off; do tail -1 $OUT/$f.json | cut -c1-200; done
grep "wino-fused" $OUT/layers_r04_fused.txt | cut -c1-150 | head -30
