# ASPP forward branch convolutions on side streams (SEGSDE_ASPP_STREAMS=1) vs default, two processes each
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
for v in 1 0 1 0; do
  SEGSDE_ASPP_STREAMS=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing > $OUT/r3_bench_streams$v.json 2>> $OUT/r3_streams.err
  echo "streams=$v $(tail -1 $OUT/r3_bench_streams$v.json | cut -c1-330 | sed 's/.*"warmup"//')"
done
SEGSDE_ASPP_STREAMS=1 timeout 600 python -m pytest tests/test_models_gpu.py -m gpu -q -x -k "jsd or joint or r101 or oracle" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -3
