ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 200 python -m pytest tests/test_kernels_gpu.py -q -x 2>&1 | tail -2 > $OUT/r4_lastcheck_tests.log
timeout 200 python -m pytest tests/test_models_gpu.py -q -x -k "reference_vectors or big_models or fanout or blocks or decoders" 2>&1 | tail -2 >> $OUT/r4_lastcheck_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -1 >> $OUT/r4_lastcheck_tests.log
cat $OUT/r4_lastcheck_tests.log
