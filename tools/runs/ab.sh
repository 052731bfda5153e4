# usage: bash tools/runs/ab.sh <tag> <tuneA> <tuneB> [grep-pattern]   (layer-table A/B, two repetitions each, one process per run)
mkdir -p gpurun_out
pat=${4:-TF}
for v in "$2" "$3" "$2" "$3"; do echo "=== $v"; SEGSDE_TUNE=$v BENCH_B=16 BENCH_ONLY_CONV=1 timeout 600 python tools/bench_kernels.py 2>&1 | grep " TF" | grep -- "$pat"; done > gpurun_out/ab_$1.log 2>&1
