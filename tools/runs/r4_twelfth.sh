# round 4, twelfth box: Winograd multiply-add floor on the small workloads (cfg1 / cfg2), cfg3 unchanged?
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
for m in 4e9 1e8; do
  for w in cfg1 cfg2; do
    SEGSDE_WINOGRAD_MIN_MACS=$m timeout 600 python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_r04_${w}_macs$m.json 2> $OUT/bench_r04_${w}_macs$m.err
    echo "$w min_macs=$m: $(tail -1 $OUT/bench_r04_${w}_macs$m.json | cut -c1-150)"
  done
done
SEGSDE_WINOGRAD=0 timeout 600 python bench.py --workload cfg1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_r04_cfg1_wino_off.json 2> /dev/null
echo "cfg1 winograd off: $(tail -1 $OUT/bench_r04_cfg1_wino_off.json | cut -c1-150)"
SEGSDE_WINOGRAD=0 timeout 600 python bench.py --workload cfg2 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_r04_cfg2_wino_off.json 2> /dev/null
echo "cfg2 winograd off: $(tail -1 $OUT/bench_r04_cfg2_wino_off.json | cut -c1-150)"
SEGSDE_WINOGRAD_MIN_MACS=1e8 timeout 600 python bench.py --no-cpu-baseline > $OUT/bench_r04_cfg3_macs1e8.json 2> /dev/null
echo "cfg3 min_macs=1e8: $(tail -1 $OUT/bench_r04_cfg3_macs1e8.json | cut -c1-150)"
