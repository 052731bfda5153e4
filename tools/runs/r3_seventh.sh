ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/r3_bench8_fusedopt.json 2> $OUT/r3_bench8.err
SEGSDE_BENCH_FUSED_OPT=0 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/r3_bench8_foreach.json 2>> $OUT/r3_bench8.err
SEGSDE_BENCH_ATEN_OPS=$OUT/r3_aten_ops.txt timeout 600 python bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-kernel-timing > /dev/null 2>> $OUT/r3_bench8.err
for f in r3_bench8_fusedopt r3_bench8_foreach; do tail -1 $OUT/$f.json | cut -c1-200; done
grep "^aten::" $OUT/r3_aten_ops.txt | head -40
