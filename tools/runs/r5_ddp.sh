ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests/test_models_gpu.py -q -x -k "reducer or two_ranks or selfspawn" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -5
SEGSDE_FORCE_REDUCER=1 timeout 600 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > $OUT/bench_r05_forced_reducer.json 2> $OUT/bench_r05_forced_reducer.err
python - <<PY
import json
r = json.load(open("$OUT/bench_r05_forced_reducer.json"))
print("forced 1-rank reducer:", r["value"], "img/s", r["ms_per_step"], "ms/step")
print(json.dumps(r.get("comm"), indent=0)[:1500])
PY
timeout 600 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > $OUT/bench_r05_plain.json 2> $OUT/bench_r05_plain.err
python - <<PY
import json
r = json.load(open("$OUT/bench_r05_plain.json"))
print("plain:", r["value"], "img/s", r["ms_per_step"], "ms/step")
PY
