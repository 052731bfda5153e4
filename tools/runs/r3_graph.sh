# whole-step hipGraph capture (bench.py --hip-graph): parity test + cfg1 eager vs graph
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 300 python -m pytest tests/test_models_gpu.py -m gpu -q -x -k "hip_graph" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -25 > $OUT/r3_graph_test.log
tail -25 $OUT/r3_graph_test.log
timeout 200 python bench.py --workload cfg1 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing > $OUT/bench_r03_cfg1_eager_no_timing.json 2> $OUT/r3_graph.err
timeout 200 python bench.py --workload cfg1 --steps 20 --warmup 5 --no-cpu-baseline --hip-graph > $OUT/bench_r03_cfg1_hip_graph.json 2>> $OUT/r3_graph.err
for f in bench_r03_cfg1_eager_no_timing bench_r03_cfg1_hip_graph; do tail -1 $OUT/$f.json | cut -c1-300; done
tail -5 $OUT/r3_graph.err | cut -c1-300
