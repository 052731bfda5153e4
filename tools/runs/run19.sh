mkdir -p gpurun_out
for v in adjl=1 adjl=0 adjl=1 adjl=0; do echo "=== $v"; SEGSDE_TUNE=$v BENCH_B=16 BENCH_ONLY_CONV=1 timeout 600 python tools/bench_kernels.py 2>&1 | grep " TF" | grep refl ; done > gpurun_out/r19_ab_adjl.log 2>&1
