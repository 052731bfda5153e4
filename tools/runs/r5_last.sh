# last check of the round: the whole GPU suite, smoke, the driver's default bench command on the final tree
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout 2700 python -m pytest tests -m gpu -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -8 > $OUT/r5_last_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $OUT/r5_last_smoke.log 2>&1
SEGSDE_BENCH_LAYERS=$OUT/layers_r05_latest.txt python bench.py > $OUT/bench_r05_cfg3_default_run.json 2> $OUT/bench_r05_cfg3_default_run.err
tail -4 $OUT/r5_last_tests.log; tail -1 $OUT/r5_last_smoke.log
tail -1 $OUT/bench_r05_cfg3_default_run.json | cut -c1-400
python - <<PY
import json
r = json.loads(open("$OUT/bench_r05_cfg3_default_run.json").read().strip().splitlines()[-1])
print("peak GB", r["config"]["peak_memory_gb"], "frac", r["roofline"]["frac"], "traffic_source", r["roofline"].get("traffic_source"))
PY
