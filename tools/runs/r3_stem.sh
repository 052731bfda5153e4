# stem A/B: the dedicated 7x7 / stride-2 stem path (SEGSDE_STEM=1, default) against the generic gather (SEGSDE_STEM=0)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "stems" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -8 > $OUT/r3_stem_tests.log
for v in 1 0 1 0; do
  SEGSDE_STEM=$v SEGSDE_BENCH_LAYERS=$OUT/r3_layers_stem$v.txt timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/r3_bench_stem$v.json 2>> $OUT/r3_stem.err
  echo "stem=$v $(tail -1 $OUT/r3_bench_stem$v.json | cut -c1-160)"
done
tail -3 $OUT/r3_stem_tests.log
for v in 1 0; do echo "== stem=$v"; grep -i "k7\|stem" $OUT/r3_layers_stem$v.txt | head -12; done
