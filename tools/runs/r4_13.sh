ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
for c in 256 128 64; do
  for w in cfg1 cfg2; do
    SEGSDE_WINOGRAD_MIN_CH=$c timeout 600 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_r04_${w}_ch$c.json 2> /dev/null
    echo "$w min_ch=$c: $(tail -1 $OUT/bench_r04_${w}_ch$c.json | cut -c1-150)"
  done
done
