# dead tap rows of dilated zero-padded windows (ConvP::tapskip): parity of the touched kernels + A/B against SEGSDE_TUNE=tskip=0
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 1200 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -8 > $OUT/r3_tskip_tests.log
for v in 1 0 1 0; do
  SEGSDE_TUNE=tskip=$v SEGSDE_BENCH_LAYERS=$OUT/r3_layers_tskip$v.txt timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/r3_bench_tskip$v.json 2>> $OUT/r3_tskip.err
  echo "tskip=$v $(tail -1 $OUT/r3_bench_tskip$v.json | cut -c1-160)"
done
tail -3 $OUT/r3_tskip_tests.log
for v in 1 0; do echo "== tskip=$v"; grep " d6 \| d12 \| d18 \| d2 " $OUT/r3_layers_tskip$v.txt | head -16; done
