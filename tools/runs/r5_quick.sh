# quick checkpoint: kernel-level GPU tests + short bench with the per-layer table
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
TAG=${TAG:-q}
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x 2>&1 | tail -3
timeout 900 python -m pytest tests/test_models_gpu.py -q -x -k "not selfspawn and not two_ranks" 2>&1 | tail -3
SEGSDE_BENCH_LAYERS=$OUT/layers_r05_${TAG}.txt python bench.py --no-cpu-baseline --steps 10 --warmup 3 > $OUT/bench_r05_${TAG}.json 2> $OUT/bench_r05_${TAG}.err
python - <<PY
import json
r = json.load(open("$OUT/bench_r05_${TAG}.json"))
print("bench:", r["value"], "img/s", r["ms_per_step"], "ms/step  peak GB", r["config"].get("peak_memory_gb"))
print({k: (round(v["seconds"] / r["steps"] * 1e3, 1), round(v["executed_tflops"], 1)) for k, v in r["kernels"].items()})
print(r["fusions_per_step"])
PY
