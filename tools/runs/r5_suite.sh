# round 5 checkpoint: the whole GPU suite, smoke, a short bench with the per-layer table and a kernel trace
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
TAG=${TAG:-ckpt}
mkdir -p $OUT
cd $ROOT
timeout 2700 python -m pytest tests -m gpu -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -15 > $OUT/r5_${TAG}_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $OUT/r5_${TAG}_smoke.log 2>&1
SEGSDE_BENCH_LAYERS=$OUT/layers_r05_${TAG}.txt python bench.py --no-cpu-baseline > $OUT/bench_r05_${TAG}.json 2> $OUT/bench_r05_${TAG}.err
bash tools/runs/trace.sh r05_${TAG}
tail -6 $OUT/r5_${TAG}_tests.log; tail -1 $OUT/r5_${TAG}_smoke.log
python - <<PY
import json
r = json.load(open("$OUT/bench_r05_${TAG}.json"))
print("bench:", r["value"], "img/s", r["ms_per_step"], "ms/step  peak GB", r["config"].get("peak_memory_gb"))
PY
