ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
for w in cfg1 cfg2 cfg3pad cfg5; do
  python bench.py --workload $w --steps 8 --warmup 3 --no-cpu-baseline > $OUT/r29_$w.json 2> $OUT/r29_$w.err
done
python bench.py --steps 10 --warmup 3 > $OUT/r29_cfg3_default.json 2> $OUT/r29_cfg3_default.err
SEGSDE_FORCE_REDUCER=1 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > $OUT/r29_cfg3_reducer.json 2> $OUT/r29_cfg3_reducer.err
for f in $OUT/r29_*.json; do tail -1 $f | cut -c1-260; done
