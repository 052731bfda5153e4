ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -8 > $OUT/r3_tests9.log
SEGSDE_BENCH_LAYERS=$OUT/r3_layers9_adjb1.txt timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/r3_bench9_adjb1.json 2> $OUT/r3_bench9.err
SEGSDE_TUNE=adjb=0 SEGSDE_BENCH_LAYERS=$OUT/r3_layers9_adjb0.txt timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/r3_bench9_adjb0.json 2>> $OUT/r3_bench9.err
tail -3 $OUT/r3_tests9.log
for f in r3_bench9_adjb1 r3_bench9_adjb0; do tail -1 $OUT/$f.json | cut -c1-200; done
for f in adjb1 adjb0; do echo "== $f"; grep "conv_dgrad" $OUT/r3_layers9_$f.txt | grep "refl" | head -12; done
