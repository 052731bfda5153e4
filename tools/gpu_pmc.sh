#!/bin/bash
# PMC passes of bench.py on the GPU box (each counter set in its own run; --pmc only with --kernel-trace, never with
# the sys / runtime / hip / hsa trace domains).  Results under gpurun_out/, summaries under gpurun_out/*.txt.
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
TAG=${PMC_TAG:-r04}
rocprofv3 -L > $OUT/${TAG}_counters_list.txt 2>&1 || true
run() {  # name, counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" -d $OUT/pmc_$name -o pmc -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-reference-step > $OUT/${TAG}_pmc_$name.log 2>&1
  local db=$(ls $OUT/pmc_$name/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && python $ROOT/tools/pmc_table.py $db $OUT/${TAG}_pmc_$name.txt > /dev/null
  rm -rf $OUT/pmc_$name      # the databases are tens of MB each: only the summaries travel back
}
run mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
run waits SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS
run fetch FETCH_SIZE
run write WRITE_SIZE
ls $OUT | head -40
python $ROOT/tools/traffic_json.py $OUT/${TAG}_pmc_fetch.txt $OUT/${TAG}_pmc_write.txt $OUT/traffic_${TAG}.json > /dev/null
