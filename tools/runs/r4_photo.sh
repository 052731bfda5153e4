# packed photometric kernels: parity tests first, then the loss-chain probe per knob setting, then the step
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -q -x -k "photometric or loss or mono or golden or reference_vectors" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -8 > $OUT/r4_photo_tests.log
tail -3 $OUT/r4_photo_tests.log
( python tools/probes/photometric_probe.py
  SEGSDE_PHOTO_PACKED=0 python tools/probes/photometric_probe.py
  SEGSDE_PHOTO_PACKED=0 SEGSDE_WARP_BLOCKS=512 python tools/probes/photometric_probe.py ) > $OUT/probe_r04_photometric.log 2>&1
cat $OUT/probe_r04_photometric.log | grep -v amdgpu.ids
timeout 600 python bench.py --no-cpu-baseline > $OUT/bench_r04_photo_packed.json 2> $OUT/bench_r04_photo_packed.err
SEGSDE_PHOTO_PACKED=0 SEGSDE_WARP_BLOCKS=512 timeout 600 python bench.py --no-cpu-baseline > $OUT/bench_r04_photo_old.json 2> $OUT/bench_r04_photo_old.err
timeout 600 python bench.py --workload cfg2 --no-cpu-baseline > $OUT/bench_r04_cfg2_photo_packed.json 2> $OUT/bench_r04_cfg2_photo.err
for f in bench_r04_photo_packed bench_r04_photo_old bench_r04_cfg2_photo_packed; do tail -1 $OUT/$f.json | cut -c1-260; done
