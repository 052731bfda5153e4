# round 4, first box: HBM-bound || MFMA-bound overlap probe + this round's baseline line
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 600 python tools/probes/bn_wgrad_overlap_probe.py > $OUT/probe_r04_bn_wgrad_overlap.log 2>&1
timeout 900 python bench.py --no-cpu-baseline > $OUT/bench_r04_baseline.json 2> $OUT/bench_r04_baseline.err
cat $OUT/probe_r04_bn_wgrad_overlap.log | grep -v amdgpu.ids
tail -1 $OUT/bench_r04_baseline.json | cut -c1-400
