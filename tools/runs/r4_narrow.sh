ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -q -x -k "dead_tap or aspp or blocks or decoders" 2>&1 | tail -2
SEGSDE_BENCH_LAYERS=$OUT/layers_r04_narrow.txt timeout 600 python bench.py --no-cpu-baseline > $OUT/bench_r04_narrow_d12.json 2> $OUT/bench_r04_narrow.err
SEGSDE_TUNE=tsbn=-1 timeout 600 python bench.py --no-cpu-baseline > $OUT/bench_r04_narrow_off.json 2>> $OUT/bench_r04_narrow.err
for f in bench_r04_narrow_d12 bench_r04_narrow_off; do tail -1 $OUT/$f.json | cut -c1-200; done
grep "2048+0->256 k3" $OUT/layers_r04_narrow.txt | cut -c1-150
SEGSDE_WARP_BLOCKS=512 bash tools/runs/trace.sh r04_warp512
bash tools/runs/trace.sh r04_warp2048
grep -h "warp_fwd" $OUT/trace_r04_warp512.txt $OUT/trace_r04_warp2048.txt | cut -c1-100
