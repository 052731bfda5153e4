# round-3 first GPU call: whole -m gpu suite (all failures listed), then the default bench with the per-layer table,
# once with the upsample-folded route and once without (A/B in separate processes on the same box)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -40 > $OUT/r3_tests.log
SEGSDE_BENCH_LAYERS=$OUT/r3_layers_fold.txt timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > $OUT/r3_bench_fold.json 2> $OUT/r3_bench_fold.err
SEGSDE_UPFOLD=0 SEGSDE_BENCH_LAYERS=$OUT/r3_layers_nofold.txt timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > $OUT/r3_bench_nofold.json 2> $OUT/r3_bench_nofold.err
tail -5 $OUT/r3_tests.log; tail -1 $OUT/r3_bench_fold.json | cut -c1-300; tail -1 $OUT/r3_bench_nofold.json | cut -c1-300
