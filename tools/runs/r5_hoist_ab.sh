# A/B of the one-kernel Winograd forward / data-gradient: patch-load offsets hoisted out of the fill loop (new) against
# the previous kernel (libsegsde_prev.so = HEAD's winograd_fused.hip linked with the same other objects)
# libsegsde_prev.so is built by hand before the call (not kept in the tree): `git show HEAD:<csrc file> > /tmp/prev/<file>`,
# hipcc -c it with __graft_entry__.FLAGS and link it with the other objects of build/obj into the package directory.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k winograd 2>&1 | tail -3
for rep in 1 2; do
  SEGSDE_LIB=$ROOT/improving_segmentation_with_selfsupervised_depth_amd/libsegsde_prev.so timeout 300 python tools/probes/winograd_fused_probe.py > $OUT/probe_r05_hoist_prev_$rep.log 2>&1
  timeout 300 python tools/probes/winograd_fused_probe.py > $OUT/probe_r05_hoist_new_$rep.log 2>&1
done
python - <<PY
import re
for rep in (1, 2):
    a = open("$OUT/probe_r05_hoist_prev_%d.log" % rep).read().splitlines()
    b = open("$OUT/probe_r05_hoist_new_%d.log" % rep).read().splitlines()
    for la, lb in zip(a, b):
        fa = re.findall(r"fused\s+([\d.]+) us", la); fb = re.findall(r"fused\s+([\d.]+) us", lb)
        if len(fa) == 2 and len(fb) == 2:
            print(rep, la[:34], "fwd %s -> %s (%.3fx)  dgrad %s -> %s (%.3fx)" % (fa[0], fb[0], float(fa[0]) / float(fb[0]), fa[1], fb[1], float(fa[1]) / float(fb[1])))
PY
