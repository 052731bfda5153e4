# round 4, eighth box: Winograd weight gradient: tests, step, layer table
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
SEGSDE_BENCH_LAYERS=$OUT/layers_r04_wino3.txt timeout 900 python bench.py --no-cpu-baseline > $OUT/bench_r04_wino3.json 2> $OUT/bench_r04_wino3.err
tail -1 $OUT/bench_r04_wino3.json | cut -c1-260
grep "wgrad.*wino" $OUT/layers_r04_wino3.txt | head -20
timeout 2400 python -m pytest tests -m gpu -q -k "winograd or cfg2_batch8 or train_step_replay or r50_mono" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -40 > $OUT/r4_eighth_tests.log
tail -6 $OUT/r4_eighth_tests.log
