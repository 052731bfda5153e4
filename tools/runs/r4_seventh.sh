# round 4, seventh box: Winograd route extended (dilation, decoder convs, multi pack): tests + step A/B + layer table
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -q -k "winograd or cfg2_batch8 or train_step_replay or headline or full_model or decoders" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -40 > $OUT/r4_seventh_tests.log
tail -6 $OUT/r4_seventh_tests.log
SEGSDE_BENCH_LAYERS=$OUT/layers_r04_wino2.txt timeout 900 python bench.py --no-cpu-baseline > $OUT/bench_r04_wino2.json 2> $OUT/bench_r04_wino2.err
tail -1 $OUT/bench_r04_wino2.json | cut -c1-260
grep wino $OUT/layers_r04_wino2.txt | head -20
