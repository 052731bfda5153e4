ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
cd $ROOT
run() {
  env "$@" timeout 300 python bench.py --workload ${WL:-cfg1} --steps 20 --warmup 5 --no-cpu-baseline 2> /dev/null | tail -1 > $OUT/ab_tmp.json
  python - "$*" <<PY
import json, sys
r = json.load(open("$OUT/ab_tmp.json")); print("%-60s %7.2f img/s %7.2f ms/step" % (sys.argv[1], r["value"], r["ms_per_step"]))
PY
}
run A=0
run SEGSDE_WINO_FUSED_WGRAD=0
run SEGSDE_WINO_FUSED_DGRAD_EXT=0
run SEGSDE_WINO_FUSED2=0
run SEGSDE_WINO_FUSED_DGRAD2=0
run SEGSDE_WINO_FUSED_WGRAD=0 SEGSDE_WINO_FUSED_DGRAD_EXT=0 SEGSDE_WINO_FUSED2=0 SEGSDE_WINO_FUSED_DGRAD2=0
run SEGSDE_BN_PARTIALS_WIDE_MAX=0
run SEGSDE_WGRAD_FUSED_WGS=256
