ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
run() { # name, env...
  local name=$1; shift
  env "$@" timeout 600 python -m pytest tests/test_models_gpu.py -m gpu -q -k "reference_vectors" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" > $OUT/r3_flaky_$name.log
  echo "== $name: $(tail -1 $OUT/r3_flaky_$name.log)"; grep "^FAILED\|^E  " $OUT/r3_flaky_$name.log | head -12
}
run a1 A=1
run a2 A=1
run nostem SEGSDE_STEM=0
run notskip SEGSDE_TUNE=tskip=0
run neither SEGSDE_STEM=0 SEGSDE_TUNE=tskip=0
python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step_all'], d['kernel_timing'])"
