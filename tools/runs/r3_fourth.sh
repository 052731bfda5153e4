# whole GPU suite + default bench (layers) + A/B of the in-kernel reductions
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -40 > $OUT/r3_tests4.log
SEGSDE_BENCH_LAYERS=$OUT/r3_layers4.txt timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/r3_bench4.json 2> $OUT/r3_bench4.err
SEGSDE_TUNE="wred=0,cfin=0" timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/r3_bench4_noredfuse.json 2> $OUT/r3_bench4_noredfuse.err
timeout 300 python bench.py --workload cfg1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/r3_bench4_cfg1.json 2>/dev/null
bash tools/runs/trace.sh r3c
tail -6 $OUT/r3_tests4.log
for f in r3_bench4 r3_bench4_noredfuse r3_bench4_cfg1; do tail -1 $OUT/$f.json | cut -c1-200; done
head -50 $OUT/trace_r3c.txt
