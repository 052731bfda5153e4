mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py -q -x -k "conv or actgrad" 2>&1 | tail -3 > gpurun_out/r28_tests.log
for v in var=0 var=8 var=0 var=8; do echo "=== $v"; SEGSDE_TUNE=$v BENCH_B=16 BENCH_ONLY_CONV=1 timeout 600 python tools/bench_kernels.py 2>&1 | grep " TF" | grep "refl" ; done > gpurun_out/r28_ab_xtab.log 2>&1
tail -2 gpurun_out/r28_tests.log
