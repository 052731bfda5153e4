mkdir -p gpurun_out
for v in stag=0 stag=2 stag=3 "stag=3,var=6"; do echo "=== $v"; SEGSDE_TUNE=$v BENCH_B=16 BENCH_ONLY_CONV=1 timeout 600 python tools/bench_kernels.py 2>&1 | grep " TF" ; done > gpurun_out/r14_ab_stag.log 2>&1
