# A/B of the Winograd kernels' LDS reads as single ds_read_b32 with immediate offsets (new) against the previous library
# (libsegsde_prev.so: HEAD's sources linked the same way): weight gradient, forward on two sources, mirrored data-gradient
# libsegsde_prev.so is built by hand before the call (not kept in the tree): `git show HEAD:<csrc file> > /tmp/prev/<file>`,
# hipcc -c it with __graft_entry__.FLAGS and link it with the other objects of build/obj into the package directory.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k winograd 2>&1 | tail -3
for rep in 1 2; do
  SEGSDE_LIB=$ROOT/improving_segmentation_with_selfsupervised_depth_amd/libsegsde_prev.so timeout 300 python tools/probes/r5_winograd_probe.py wgrad fwd2 dgrad > $OUT/probe_r05_ldsimm_prev_$rep.log 2>&1
  timeout 300 python tools/probes/r5_winograd_probe.py wgrad fwd2 dgrad > $OUT/probe_r05_ldsimm_new_$rep.log 2>&1
done
python - <<PY
import re
for rep in (1, 2):
    a = open("$OUT/probe_r05_ldsimm_prev_%d.log" % rep).read().splitlines()
    b = open("$OUT/probe_r05_ldsimm_new_%d.log" % rep).read().splitlines()
    for la, lb in zip(a, b):
        fa = re.findall(r"(?:fused|Winograd)\s+([\d.]+) us", la); fb = re.findall(r"(?:fused|Winograd)\s+([\d.]+) us", lb)
        if fa and fb:
            print(rep, la[:40], " ".join("%s -> %s (%.3fx)" % (x, y, float(x) / float(y)) for x, y in zip(fa, fb)))
PY
