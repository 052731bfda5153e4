#!/usr/bin/env python
"""Drop-in check under the reference's OWN caller (build container only).

Imports /root/reference/train.py and calls the reference's ``Trainer.train_step`` (train.py:442-549) -- and through it
``train_step_segmentation_unlabeled`` (train.py:653-746), ``get_train_params``, ``create_ema_model`` /
``update_ema_variables``, ``generate_mix_mask``, ``calc_pseudo_label_loss`` -- for two iterations on a synthetic loader
batch, in two scenarios ("joint": monodepth + segmentation, three parameter groups, gradient clipping; "depthmix": plus the
unlabeled DepthMix step with the EMA teacher, online depth, depthcomp mask and mix_use_gt), with

  --impl reference   ``models`` / ``loss`` / ``loader`` = the reference's own packages -> tests/golden/trainstep.npz
  --impl package     ``models`` / ``loss`` / ``loader`` = improving_segmentation_with_selfsupervised_depth_amd.{models,loss,
                     loader} (kernels on the host interpreter, tests/hipemu) -> compared with trainstep.npz
                     (tests/trainstep_case.py::compare), log in tests/golden/trainstep_package_run.json

Same train.py source in both runs (it is imported, never copied); what differs is only what the names it imports resolve
to.  The Trainer object is created with ``object.__new__`` (its __init__ builds datasets and TensorBoard writers) and given
exactly the attributes the two step methods read.  Out-of-scope names train.py imports (build_loader, DepthEstimator,
SummaryWriter) are stubbed identically in both runs.  Only the recorded numbers travel to the GPU box.
"""
import argparse
import copy
import json
import os
import sys
import time
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.dont_write_bytecode = True
for p_ in (REPO, os.path.join(REPO, "tests"), HERE):
    if p_ not in sys.path:
        sys.path.insert(0, p_)

ap = argparse.ArgumentParser()
ap.add_argument("--impl", choices=["reference", "package"], required=True)
ap.add_argument("--scenarios", default="joint,depthmix")
ap.add_argument("--compare-only", action="store_true",
                help="package impl: re-compare the numbers of the last package run (build/trainstep_package_out.npz) with trainstep.npz")
args = ap.parse_args()
torch.set_num_threads(8)

_tb = types.ModuleType("torch.utils.tensorboard")
_tb.SummaryWriter = object
sys.modules["torch.utils.tensorboard"] = _tb
if args.impl == "reference":
    import _tv_standin
    _tv_standin.install()
    sys.path.insert(0, REF)
    _loader = types.ModuleType("loader")                 # the real loader/__init__ pulls the PIL dataset classes
    _loader.__path__ = [os.path.join(REF, "loader")]
    sys.modules["loader"] = _loader
    _de = types.ModuleType("loader.depth_estimator")
    sys.modules["loader.depth_estimator"] = _de
else:
    import emu
    emu.install()
    import improving_segmentation_with_selfsupervised_depth_amd as pkg
    from improving_segmentation_with_selfsupervised_depth_amd import models as _m, loss as _l, loader as _loader
    import improving_segmentation_with_selfsupervised_depth_amd.models.joint_segmentation_depth_decoder as _jsd
    import improving_segmentation_with_selfsupervised_depth_amd.loss.loss as _ll
    import improving_segmentation_with_selfsupervised_depth_amd.loader.depth_estimator as _de
    for name, mod in (("models", _m), ("models.joint_segmentation_depth_decoder", _jsd), ("loss", _l), ("loss.loss", _ll),
                      ("loader", _loader), ("loader.transformsgpu", _loader.transformsgpu),
                      ("loader.transformmasks", _loader.transformmasks), ("loader.depth_estimator", _de)):
        sys.modules[name] = mod
    sys.path.append(REF)      # behind everything else: configs/, evaluation/, utils/ (not on the path) come from the reference
sys.modules["loader"].build_loader = None
sys.modules["loader.depth_estimator"].DepthEstimator = None

import train as ref_train  # noqa: E402  (the reference's train.py in BOTH runs)
import trainstep_case as TC  # noqa: E402

assert os.path.realpath(ref_train.__file__) == os.path.join(REF, "train.py"), ref_train.__file__
get_model = sys.modules["models"].get_model
print("train.py:", ref_train.__file__, "| models:", sys.modules["models"].__file__, "| loss:", sys.modules["loss"].__file__)


class _UnlabeledLoader:
    ignore_index = 250


def make_trainer(scenario, noise, ulps=0):
    cfg = TC.full_cfg(scenario)
    t = object.__new__(ref_train.Trainer)
    t.cfg, t.device, t.mIoU = cfg, torch.device("cpu"), 0
    t.setup_segmentation_unlabeled()
    t.n_classes = TC.NCLS
    t.model = get_model(cfg["model"], TC.NCLS)
    t.model.load_state_dict(TC.state_dict(scenario), strict=True)
    TC.no_dropout(t.model)
    if ulps:         # yardstick runs: the stem weights moved by a few units in the last place
        with torch.no_grad():
            t.model.models["encoder"].encoder.conv1.weight.mul_(1.0 + ulps * 2.0 ** -23)
    t.ema_model = None
    if t.enable_unlabled_segmentation:
        t.ema_model = t.create_ema_model(t.model)
        TC.no_dropout(t.ema_model)
        t.unlabeled_loader = _UnlabeledLoader()
        batches = [TC.batch(200 + i, labeled=False, onehot=True) for i in range(TC.ITERS)]
        t.unlabeled_data_loader = iter(batches)
    opt_cls = ref_train.get_optimizer(cfg)
    opt_params = {k: v for k, v in cfg["training"]["optimizer"].items()
                  if k not in ["name", "backbone_lr", "pose_lr", "depth_lr", "segmentation_lr"]}
    t.optimizer = opt_cls(ref_train.get_train_params(t.model, cfg), **opt_params)       # train.py:296-301
    t.scheduler = ref_train.get_scheduler(t.optimizer, cfg["training"]["lr_schedule"])
    t.scaler = ref_train.GradScaler(enabled=False)
    t.loss_fn = sys.modules["loss"].get_segmentation_loss_function(cfg)
    t.monodepth_loss_calculator_train = sys.modules["loss"].get_monodepth_loss(cfg, is_train=True)
    if args.impl == "package":
        t.monodepth_loss_calculator_train.tiebreak_noise = noise
    return t


def run(scenario, d, ulps=0):
    noise = TC.noise()
    t = make_trainer(scenario, noise, ulps)
    groups = [len(g["params"]) for g in t.optimizer.param_groups]
    d[scenario + "_param_group_sizes"] = np.array(groups)
    d[scenario + "_param_group_lrs"] = np.array([g["lr"] for g in t.optimizer.param_groups])
    real_randn = torch.randn
    calls = [0]

    def fixed_randn(*a, **k):          # monodepth_loss.py:163-164 draws its tie-break noise scale by scale
        n = noise[calls[0] % 4]
        calls[0] += 1
        assert tuple(a[0] if isinstance(a[0], (tuple, list, torch.Size)) else a) == tuple(n.shape), (a, n.shape)
        return n.clone()

    for it in range(TC.ITERS):
        before = {k: p.detach().clone() for k, p in t.model.named_parameters()}
        inputs = TC.batch(100 + it)
        t0 = time.time()
        if args.impl == "reference":
            torch.randn = fixed_randn
        try:
            losses = t.train_step(inputs, it)
        finally:
            torch.randn = real_randn
        print("%s it %d: %.0f s  %s" % (scenario, it, time.time() - t0, {k: float(v) for k, v in losses.items()}), flush=True)
        TC.record(d, scenario, it, losses, t.model, t.ema_model, before)
    if args.impl == "reference":
        assert calls[0] % 4 == 0 and calls[0] >= 4 * TC.ITERS


out = {}
keep = os.path.join(REPO, "build", "trainstep_package_out.npz")
if args.compare_only:
    out = dict(np.load(keep, allow_pickle=False))
else:
    for sc in args.scenarios.split(","):
        run(sc, out)
    if args.impl == "package":
        os.makedirs(os.path.dirname(keep), exist_ok=True)
        np.savez_compressed(keep, **out)      # ~40 min of interpretation: kept so that the comparison can be redone
path = os.path.join(HERE, "trainstep.npz")
if args.impl == "reference":
    # yardstick: what fp32 round-off alone does to every recorded per-parameter figure -- the same reference run with the stem
    # weights scaled by 1 +- 1..2 ulp (tests/diag/gradient_norm_sensitivity.py: a ReLU / BatchNorm network over a few dozen
    # samples amplifies that to per cent level in single parameters, more in the second iteration)
    for sc in args.scenarios.split(","):
        spread = {}
        for ulps in (1, -1, 2, -2):
            alt = {}
            run(sc, alt, ulps)
            for k, v in alt.items():
                if k.endswith(("grad_norms", "update_norms", "param_sums", "bn_running_mean_sums")):
                    dlt = np.abs(np.asarray(v) - np.asarray(out[k]))
                    spread[k] = np.maximum(spread[k], dlt) if k in spread else dlt
                elif "_it" in k and np.ndim(v) == 0:
                    spread[k] = max(spread.get(k, 0.0), abs(float(v) - float(out[k])))
        out.update({k + "_spread": np.asarray(v) for k, v in spread.items()})
    np.savez_compressed(path, **out)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))
else:
    ref = dict(np.load(path, allow_pickle=False))
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from trainstep_case import package_fingerprint
    log = {"impl": "package on the kernel interpreter, under /root/reference/train.py", "scenarios": {},
           # what this run executed: the test that reads this log refuses it once the kernels / host code have changed
           "package_sha256": package_fingerprint()}
    for sc in args.scenarios.split(","):
        log["scenarios"][sc] = TC.compare(out, ref, sc)
        for it in range(TC.ITERS):
            log["scenarios"][sc]["it%d" % it] = {k.split("_it%d_" % it)[1]: float(v) for k, v in out.items()
                                                 if k.startswith("%s_it%d_" % (sc, it)) and np.ndim(v) == 0}
    with open(os.path.join(HERE, "trainstep_package_run.json"), "w") as f:
        json.dump(log, f, indent=1, sort_keys=True)
    print("package == reference under the reference's train.py: OK")
