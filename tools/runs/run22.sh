mkdir -p gpurun_out
for v in noadj=0 noadj=1; do echo "=== $v"; SEGSDE_TUNE=$v BENCH_B=16 BENCH_ONLY_CONV=1 timeout 600 python tools/bench_kernels.py 2>&1 | grep " TF" | grep "refl\|zero pad" ; done > gpurun_out/r22_noadj.log 2>&1
