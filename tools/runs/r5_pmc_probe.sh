# PMC passes over one probe section (PROBE_ARGS), summaries under gpurun_out/pmc_probe_<set>.txt
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
run mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
run waits SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS
run busy SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU
grep -h "wino\|kernel " $OUT/pmc_probe_mfma.txt $OUT/pmc_probe_waits.txt $OUT/pmc_probe_busy.txt | cut -c1-260
