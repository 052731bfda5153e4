ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
cd $ROOT
for w in cfg1 cfg2 cfg3pad cfg5; do
  timeout 600 python bench.py --workload $w --steps 6 --warmup 2 --no-cpu-baseline 2> $OUT/pre_$w.err | tail -1 > $OUT/pre_$w.json
  python - <<PY
import json
try:
    r = json.load(open("$OUT/pre_$w.json")); print("$w", r["value"], r["unit"], r["ms_per_step"], "ms/step", r["config"].get("peak_memory_gb"))
except Exception as ex:
    print("$w FAILED", ex); print(open("$OUT/pre_$w.err").read()[-1500:])
PY
done
