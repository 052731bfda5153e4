# round 4, third box: weight-gradient loop A/B (barrier at the chunk end vs inside the chunk), the two tests that failed on box 2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
for v in wlds=1 wlds=2 wlds=1 wlds=2; do echo "=== $v"; SEGSDE_TUNE=$v BENCH_B=16 BENCH_ONLY_CONV=1 timeout 600 python tools/bench_kernels.py 2>&1 | grep " TF"; done > $OUT/ab_r04_wgrad_pipe.log 2>&1
timeout 1200 python -m pytest tests -m gpu -q -k "cfg2_batch8 or train_step_replay" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" > $OUT/r4_third_tests.log
grep -n "worst parameters" -A8 $OUT/r4_third_tests.log; tail -5 $OUT/r4_third_tests.log
python - <<'PY'
import re,collections
rows=collections.defaultdict(dict)
cur=None
for l in open("gpurun_out/ab_r04_wgrad_pipe.log"):
    if l.startswith("==="): cur=l.split()[1]; continue
    m=re.match(r"(.{44}).*wgrad\s+([\d.]+) ms\s+([\d.]+) TF", l)
    if m: rows[m.group(1).strip()].setdefault(cur,[]).append(float(m.group(3)))
for k,v in rows.items():
    a=max(v.get("wlds=1",[0])); b=max(v.get("wlds=2",[0]))
    print("%-46s wgrad TF  end-barrier %6.1f  in-chunk %6.1f  %+5.1f %%" % (k,a,b,100*(b/a-1) if a else 0))
PY
