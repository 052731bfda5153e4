mkdir -p gpurun_out
timeout 600 python tools/probes/two_stream_probe.py > gpurun_out/r36_two_stream.log 2>&1
cat gpurun_out/r36_two_stream.log | tail -8
