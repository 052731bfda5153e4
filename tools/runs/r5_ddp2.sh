ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
SEGSDE_FORCE_REDUCER=1 timeout 600 python bench.py --no-cpu-baseline --steps 10 --warmup 3 2> $OUT/bench_r05_forced_reducer.err | tail -1 > $OUT/bench_r05_forced_reducer.json
python - <<PY
import json
r = json.load(open("$OUT/bench_r05_forced_reducer.json"))
print("forced 1-rank reducer:", r["value"], "img/s", r["ms_per_step"], "ms/step")
c = r.get("comm"); c.pop("how", None); print(c)
PY
