ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" -d $OUT/pmcp_$name -o pmc -- python $ROOT/tools/probes/r5_winograd_probe.py ${PROBE_ARGS:-wgrad} > $OUT/pmc_probe_$name.log 2>&1
  local db=$(ls $OUT/pmcp_$name/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python $ROOT/tools/pmc_table.py $db $OUT/pmc_probe_$name.txt > /dev/null
  rm -rf $OUT/pmcp_$name
}
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_UNALIGNED_STALL
run mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA
grep -h "wino\|^kernel" $OUT/pmc_probe_lds.txt $OUT/pmc_probe_mfma.txt | cut -c1-280
tail -3 $OUT/pmc_probe_lds.log
