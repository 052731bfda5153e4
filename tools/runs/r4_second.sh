# round 4, second box: the new tests (drop-in replay, stand-alone layers, cfg2 / cfg5 at their own sizes, bench self-spawn with
# the communication diagnostics) + the fusion / ddp-touching model tests
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
timeout 2400 python -m pytest tests -m gpu -q -k "cfg2_batch8 or train_step_replay or monodepth_layer or cfg5 or selfspawn or two_ranks or reducer or fusion or decoders or unlabeled or loss_kernels or full_model or jitter_blur" | tail -60 > $OUT/r4_second_tests.log
cat $OUT/r4_second_tests.log
