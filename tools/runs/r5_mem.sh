ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/bench_r05_mem.json
python - <<PY
import json
r = json.load(open("gpurun_out/bench_r05_mem.json")); print("bench:", r["value"], r["ms_per_step"], "peak GB", r["config"]["peak_memory_gb"])
PY
