# round 4, tenth box: step + layer table + ATen-operator listing with the gradient collector, then the WHOLE -m gpu suite
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
SEGSDE_BENCH_LAYERS=$OUT/layers_r04_collector.txt SEGSDE_BENCH_ATEN_OPS=$OUT/aten_ops_r04.txt timeout 900 python bench.py --no-cpu-baseline > $OUT/bench_r04_collector.json 2> $OUT/bench_r04_collector.err
tail -1 $OUT/bench_r04_collector.json | cut -c1-260
grep "aten::add\b\|aten::add " $OUT/aten_ops_r04.txt | head -5
python -c "
import json; r=json.loads(open('$OUT/bench_r04_collector.json').read().strip().splitlines()[-1]); print(r['fusions_per_step'])"
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -40 > $OUT/r4_tenth_tests.log
tail -8 $OUT/r4_tenth_tests.log
