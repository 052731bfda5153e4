ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
SEGSDE_BENCH_ATEN_OPS=$OUT/aten_ops_r04.txt timeout 600 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing > /dev/null 2> $OUT/aten_ops_r04.err
sed -n '/ATen operators/,$p' $OUT/aten_ops_r04.txt | head -40
