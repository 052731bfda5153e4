# last check of the round: default bench command first (fresh box), then the whole GPU suite
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
SEGSDE_BENCH_LAYERS=$OUT/layers_r03_latest.txt python bench.py > $OUT/bench_r03_cfg3_default_run.json 2> $OUT/bench_r03_cfg3_default_run.err
tail -1 $OUT/bench_r03_cfg3_default_run.json | cut -c1-420
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -8 > $OUT/final_tests.log
tail -3 $OUT/final_tests.log
