ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
bash tools/runs/trace.sh r05_plain
SEGSDE_FORCE_REDUCER=1 bash tools/runs/trace.sh r05_forced
head -4 gpurun_out/trace_r05_plain.txt; head -4 gpurun_out/trace_r05_forced.txt
