ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
SEGSDE_TUNE=tsbn=1 timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "aspp or dil or dead or stress" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -5 > $OUT/r3_tsbn_tests.log
tail -2 $OUT/r3_tsbn_tests.log
grep -q "passed" $OUT/r3_tsbn_tests.log && ! grep -q "failed" $OUT/r3_tsbn_tests.log || exit 0
for v in 1 0 1 0; do
  SEGSDE_TUNE=tsbn=$v SEGSDE_BENCH_LAYERS=$OUT/r3_layers_tsbn$v.txt timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/r3_bench_tsbn$v.json 2>> $OUT/r3_tsbn.err
  echo "tsbn=$v $(tail -1 $OUT/r3_bench_tsbn$v.json | cut -c1-160)"
done
for v in 1 0; do echo "== tsbn=$v"; grep " d6 \| d12 \| d18 \| d2 " $OUT/r3_layers_tsbn$v.txt | grep conv_fwd | head -16; done
