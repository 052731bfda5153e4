ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" -d $OUT/pmc_$name -o pmc -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $OUT/r2_pmc_$name.log 2>&1
  local db=$(ls $OUT/pmc_$name/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python $ROOT/tools/pmc_table.py $db $OUT/r2_pmc_$name.txt > /dev/null
  rm -rf $OUT/pmc_$name
}
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_WAIT_INST_LDS SQ_WAVE_CYCLES
run act SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES
