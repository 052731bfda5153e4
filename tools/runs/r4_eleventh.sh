# round 4, eleventh box: step with V reuse + more accumulating consumers; the affected tests
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
SEGSDE_BENCH_LAYERS=$OUT/layers_r04_v2.txt timeout 900 python bench.py --no-cpu-baseline > $OUT/bench_r04_v2.json 2> $OUT/bench_r04_v2.err
tail -1 $OUT/bench_r04_v2.json | cut -c1-200
python -c "
import json; r=json.loads(open('$OUT/bench_r04_v2.json').read().strip().splitlines()[-1]); print(r['fusions_per_step']['fanout_grad_accumulate'], r['config']['peak_memory_gb'])"
timeout 1500 python -m pytest tests -m gpu -q -k "winograd or skip_gradient or misc or test_conv or full_model or residual or reducer" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -12 > $OUT/r4_eleventh_tests.log
tail -4 $OUT/r4_eleventh_tests.log
