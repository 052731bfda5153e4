ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -30 > $OUT/r3_tests5.log
SEGSDE_BENCH_LAYERS=$OUT/r3_layers5.txt timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/r3_bench5.json 2> $OUT/r3_bench5.err
bash tools/runs/trace.sh r3d
tail -4 $OUT/r3_tests5.log
tail -1 $OUT/r3_bench5.json | cut -c1-200
grep -i "photometric\|c1s_\|warp_fwd\|colreduce\|wgrad\|bn_\|pair_fin" $OUT/trace_r3d.txt | head -30
