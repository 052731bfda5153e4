# the other workloads on the final tree (one box): cfg1 eager + hipGraph, cfg2, cfg3 with the PAD decoder, cfg5, cfg3 through the
# one-rank RCCL reducer with the plain run next to it
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 200 python bench.py --workload cfg1 --steps 30 --warmup 5 --no-cpu-baseline > $OUT/bench_r05_cfg1.json 2> $OUT/bench_r05_cfg1.err
timeout 200 python bench.py --workload cfg1 --steps 30 --warmup 5 --no-cpu-baseline --hip-graph > $OUT/bench_r05_cfg1_hipgraph.json 2> $OUT/bench_r05_cfg1_hipgraph.err
timeout 300 python bench.py --workload cfg2 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_r05_cfg2.json 2> $OUT/bench_r05_cfg2.err
timeout 300 python bench.py --workload cfg3pad --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_r05_cfg3pad.json 2> $OUT/bench_r05_cfg3pad.err
timeout 300 python bench.py --workload cfg5 --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench_r05_cfg5.json 2> $OUT/bench_r05_cfg5.err
timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > $OUT/bench_r05_cfg3_same_box_as_forced.json 2> $OUT/bench_r05_cfg3_same_box_as_forced.err
SEGSDE_FORCE_REDUCER=1 timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 2> $OUT/bench_r05_forced.err | tail -1 > $OUT/bench_r05_cfg3_forced_rccl_reducer_1rank.json
for f in bench_r05_cfg1 bench_r05_cfg1_hipgraph bench_r05_cfg2 bench_r05_cfg3pad bench_r05_cfg5 bench_r05_cfg3_same_box_as_forced bench_r05_cfg3_forced_rccl_reducer_1rank; do tail -1 $OUT/$f.json | cut -c1-200; done
