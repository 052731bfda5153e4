mkdir -p gpurun_out
SEGSDE_BENCH_ATEN_OPS=gpurun_out/r30_aten_ops.txt python bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-kernel-timing > gpurun_out/r30.json 2> gpurun_out/r30.err
grep -c . gpurun_out/r30_aten_ops.txt
