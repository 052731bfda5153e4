ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
cd $ROOT
SEGSDE_BENCH_LAYERS=$OUT/layers_r05_latest.txt python bench.py > $OUT/bench_r05_cfg3_default_run.json 2> $OUT/bench_r05_cfg3_default_run.err
python - <<PY
import json
r = json.loads(open("$OUT/bench_r05_cfg3_default_run.json").read().strip().splitlines()[-1])
ro = r["roofline"]
print(r["value"], r["ms_per_step"], "peak", r["config"]["peak_memory_gb"], "frac", ro["frac"], "cpu", r["cpu_baseline"]["value"])
print(ro.get("mfma_busy_source")); print(ro.get("traffic_source")); print({k[:44]: v for k, v in (ro.get("mfma_busy") or {}).items()})
print({k: v.get("valu_issue_frac") for k, v in r["hbm_kernels"].items() if v.get("valu_issue_frac")})
PY
