# final-state evidence: failing-test recheck, full default bench line (with CPU baseline), kernel trace, PMC passes
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests/test_fullsize_gpu.py -m gpu -q -x -k "fused_photometric or cfg5_pieces" 2>&1 | tail -5 > $OUT/r3_tests6.log
bash tools/gpu_pmc.sh > $OUT/r3_pmc.log 2>&1
tail -3 $OUT/r3_tests6.log; ls $OUT | grep "r3_pmc\|traffic_r03"; head -12 $OUT/r3_pmc_mfma.txt
