ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 1200 python -m pytest tests -m gpu -q -x -k "batchnorm or bn_ or folded_random or dropout or blocks or encoder or cross_entropy_and_bn or photometric" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -6 > $OUT/r3_tests7.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/r3_bench7.json 2> $OUT/r3_bench7.err
bash tools/runs/trace.sh r3e
tail -3 $OUT/r3_tests7.log
tail -1 $OUT/r3_bench7.json | cut -c1-200
grep -i "photometric\|bn_\|colreduce" $OUT/trace_r3e.txt | head -12
