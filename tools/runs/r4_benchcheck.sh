ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
python bench.py > $OUT/bench_r04_cfg3_default_run.json 2> $OUT/bench_r04_cfg3_default_run.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/bench_r04_cfg3_default_run.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"].get("traffic"), d["roofline"].get("traffic_source"))
for k,v in d["hbm_kernels"].items():
    if "bound" in v: print(k, v["bound"], round(v["valu_issue_frac"],3), v["valu_wave_instructions_per_launch"], v["ms_per_step"])
print(d["cpu_baseline"]["value"])
PY
