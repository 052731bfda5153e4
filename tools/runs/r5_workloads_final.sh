# the other workloads + the forced one-rank reducer on the final tree (after r5_last.sh)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout 600 python bench.py --workload cfg1 --steps 20 --warmup 5 --cpu-baseline-timeout 120 2> $OUT/bench_r05_cfg1.err | tail -1 > $OUT/bench_r05_cfg1.json
timeout 600 python bench.py --workload cfg2 --steps 20 --warmup 5 --cpu-baseline-timeout 120 2> $OUT/bench_r05_cfg2.err | tail -1 > $OUT/bench_r05_cfg2.json
timeout 600 python bench.py --workload cfg3pad --steps 10 --warmup 3 --no-cpu-baseline 2> $OUT/bench_r05_cfg3pad.err | tail -1 > $OUT/bench_r05_cfg3pad.json
timeout 600 python bench.py --workload cfg5 --steps 6 --warmup 2 --no-cpu-baseline 2> $OUT/bench_r05_cfg5.err | tail -1 > $OUT/bench_r05_cfg5.json
SEGSDE_FORCE_REDUCER=1 timeout 600 python bench.py --no-cpu-baseline 2> $OUT/bench_r05_forced.err | tail -1 > $OUT/bench_r05_cfg3_forced_rccl_reducer_1rank.json
timeout 600 python bench.py --no-cpu-baseline 2> /dev/null | tail -1 > $OUT/bench_r05_cfg3_same_box_as_forced.json
python - <<PY
import json
for f in ("cfg1", "cfg2", "cfg3pad", "cfg5", "cfg3_forced_rccl_reducer_1rank", "cfg3_same_box_as_forced"):
    r = json.load(open("$OUT/bench_r05_%s.json" % f))
    print("%-34s %8.2f %s %8.2f ms/step  peak %.1f GB  cpu %s  comm %s" % (f, r["value"], r["unit"], r["ms_per_step"], r["config"]["peak_memory_gb"],
          (r.get("cpu_baseline") or {}).get("value"), (r.get("comm") or {}).get("exposed_comm_ms")))
PY
