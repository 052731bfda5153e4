ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
for k in 0 1 2 4; do echo "== SEGSDE_WINO_SKEW=$k"; SEGSDE_WINO_SKEW=$k timeout 600 python tools/probes/winograd_fused_probe.py 2>&1 | grep -v amdgpu.ids | cut -c1-150; done
