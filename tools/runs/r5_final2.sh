# round-5 evidence run after the last kernel change (patch-load offsets of the one-kernel Winograd forward): the whole GPU suite,
# smoke, the PMC passes, the driver's default bench command with the per-layer table, and the kernel trace -- one box, this order
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -8 > $OUT/r5_final_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > $OUT/r5_final_smoke.log 2>&1
PMC_TAG=r05 bash tools/gpu_pmc.sh > $OUT/r5_pmc.log 2>&1
cd $ROOT
cp $OUT/r05_pmc_mfma.txt profiles/pmc_r05_mfma_busy.txt 2>/dev/null
cp $OUT/r05_pmc_waits.txt profiles/pmc_r05_sq_waits.txt 2>/dev/null
cp $OUT/r05_pmc_fetch.txt profiles/pmc_r05_fetch_size.txt 2>/dev/null
cp $OUT/r05_pmc_write.txt profiles/pmc_r05_write_size.txt 2>/dev/null
cp $OUT/traffic_r05.json profiles/traffic_r05.json 2>/dev/null
SEGSDE_BENCH_LAYERS=$OUT/layers_r05_latest.txt python bench.py > $OUT/bench_r05_cfg3_default_run.json 2> $OUT/bench_r05_cfg3_default_run.err
bash tools/runs/trace.sh r05_final
cd $ROOT
tail -4 $OUT/r5_final_tests.log; tail -1 $OUT/r5_final_smoke.log
tail -1 $OUT/bench_r05_cfg3_default_run.json | cut -c1-260
ls $OUT | grep "r05_pmc\|traffic_r05\|trace_r05"
