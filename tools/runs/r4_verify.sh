# after the collective-timing fix: forced 1-rank RCCL reducer bench, the comm-related GPU tests, the new full-size Winograd test,
# and the default line once more (roofline.traffic / mfma_busy must load from the round-4 PMC files)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
SEGSDE_FORCE_REDUCER=1 timeout 600 python bench.py --no-cpu-baseline > $OUT/bench_r04_cfg3_forced_rccl_reducer_1rank.json 2> $OUT/bench_r04_forced.err
timeout 1200 python -m pytest tests -m gpu -q -k "selfspawn or winograd_fullsize or two_ranks or reducer or rccl" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -8 > $OUT/r4_verify_tests.log
timeout 600 python bench.py --no-cpu-baseline > $OUT/bench_r04_cfg3_after_pmc.json 2> $OUT/bench_r04_after_pmc.err
timeout 600 python bench.py --allreduce-only > $OUT/bench_r04_allreduce_only_1rank.json 2> $OUT/bench_r04_allreduce_only.err
tail -3 $OUT/r4_verify_tests.log
for f in bench_r04_cfg3_forced_rccl_reducer_1rank bench_r04_cfg3_after_pmc bench_r04_allreduce_only_1rank; do tail -1 $OUT/$f.json | cut -c1-300; done
python - <<'PY'
import json
for f in ("bench_r04_cfg3_forced_rccl_reducer_1rank","bench_r04_cfg3_after_pmc"):
    d=json.loads(open("gpurun_out/%s.json"%f).read().strip().splitlines()[-1])
    print(f, d["ms_per_step"], json.dumps(d.get("comm"))[:600])
    r=d["roofline"]; print({k:r.get(k) for k in ("traffic","traffic_source","mfma_busy","mfma_busy_source")})
PY
