mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -4 > gpurun_out/r31_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > gpurun_out/r31_smoke.log 2>&1
cat gpurun_out/r31_tests.log; tail -2 gpurun_out/r31_smoke.log
