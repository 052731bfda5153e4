# cfg1 (R18, 256x512, batch 2: a step of ~900 launches that follows the host) eager against the whole step replayed as a hipGraph
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 300 python bench.py --workload cfg1 --steps 30 --warmup 5 --no-cpu-baseline > $OUT/bench_r05_cfg1_eager.json 2> $OUT/bench_r05_cfg1_eager.err
timeout 300 python bench.py --workload cfg1 --steps 30 --warmup 5 --no-cpu-baseline --hip-graph > $OUT/bench_r05_cfg1_hipgraph.json 2> $OUT/bench_r05_cfg1_hipgraph.err
for f in bench_r05_cfg1_eager bench_r05_cfg1_hipgraph; do tail -1 $OUT/$f.json | cut -c1-330; tail -3 $OUT/$f.err | cut -c1-300; done
