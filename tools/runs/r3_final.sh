# round-3 evidence run: the whole GPU suite, smoke, the driver's default bench command (with CPU baseline), per-layer table,
# kernel trace and the PMC passes -- all on one box, in this order
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -8 > $OUT/final_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $OUT/final_smoke.log 2>&1
SEGSDE_BENCH_LAYERS=$OUT/layers_r03_latest.txt python bench.py > $OUT/bench_r03_cfg3_default_run.json 2> $OUT/bench_r03_cfg3_default_run.err
# the same command with EVERY launch bracketed (round-2 behaviour) and with none: what the event pairs cost
python bench.py --kernel-timing-period 1 --no-cpu-baseline > $OUT/bench_r03_cfg3_every_launch_bracketed.json 2>> $OUT/bench_r03_cfg3_default_run.err
python bench.py --no-kernel-timing --no-cpu-baseline > $OUT/bench_r03_cfg3_no_kernel_timing.json 2>> $OUT/bench_r03_cfg3_default_run.err
bash tools/runs/trace.sh r03_final
bash tools/gpu_pmc.sh > $OUT/r3_pmc.log 2>&1
tail -3 $OUT/final_tests.log; tail -1 $OUT/final_smoke.log
for f in bench_r03_cfg3_default_run bench_r03_cfg3_every_launch_bracketed bench_r03_cfg3_no_kernel_timing; do tail -1 $OUT/$f.json | cut -c1-330; done
