ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
cd $ROOT
SEGSDE_WINO_FUSED_MAX_CH=512 timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing > $OUT/bench_r04_fused_max512.json 2> $OUT/bench_r04_fused_max512.err
timeout 300 python bench.py --no-cpu-baseline --no-kernel-timing > $OUT/bench_r04_fused_max256.json 2>> $OUT/bench_r04_fused_max512.err
for f in bench_r04_fused_max512 bench_r04_fused_max256; do tail -1 $OUT/$f.json | cut -c1-210; done
