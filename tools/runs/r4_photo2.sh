# packed photometric backward, variants: parity, probe per knob setting, kernel trace of the step
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 900 python -m pytest tests -m gpu -q -x -k "photometric" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -4 > $OUT/r4_photo2_tests.log
SEGSDE_PHOTO_SPLIT=0 timeout 900 python -m pytest tests -m gpu -q -x -k "photometric" 2>&1 | tail -2 >> $OUT/r4_photo2_tests.log
cat $OUT/r4_photo2_tests.log
( python tools/probes/photometric_probe.py
  SEGSDE_PHOTO_SPLIT=0 python tools/probes/photometric_probe.py
  SEGSDE_PHOTO_PACKED=0 python tools/probes/photometric_probe.py ) 2>&1 | grep -v amdgpu.ids > $OUT/probe_r04_photometric.log
cat $OUT/probe_r04_photometric.log
bash tools/runs/trace.sh r04_photo_packed
grep -i "photometric\|warp_fwd" $OUT/trace_r04_photo_packed.txt | cut -c1-150
