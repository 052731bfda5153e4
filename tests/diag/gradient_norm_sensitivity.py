"""How sensitive is the whole-model gradient-norm criterion of tests/model_cases.py::run_full_model to rounding noise?
The product (stem path on / off) on the golden inputs with the stem weights perturbed by +-1..2 ulp; error of the
per-parameter gradient norms against the fp64 oracle.  (GPU: python tests/diag/gradient_norm_sensitivity.py r18_mono)"""
import os, sys
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _ROOT); sys.path.insert(0, os.path.join(_ROOT, 'tests'))
import numpy as np, torch
DEV = "cuda"
from conftest import load_golden
import model_cases as MC
from oracle import nets as N, photometric as P, segmix as S
from improving_segmentation_with_selfsupervised_depth_amd import hipops as H
from improving_segmentation_with_selfsupervised_depth_amd.models import get_model
from improving_segmentation_with_selfsupervised_depth_amd.loss import get_monodepth_loss
from improving_segmentation_with_selfsupervised_depth_amd.loss.loss import cross_entropy2d
name = sys.argv[1] if len(sys.argv) > 1 else "r18_mono"
g = load_golden("nets")
cfg = MC.contract_cfgs()["cfgs"][name]
sd = N.build_state_dict(cfg, 19, seed=1234, randomize_bn=True)
inputs = {}
for k, v in g.items():
    if k.startswith(name + "_in_color"):
        parts = k[len(name) + 4:].rsplit("_", 2)
        inputs[("color", int(parts[1]), int(parts[2]))] = v
inputs[("K", 0)], inputs[("inv_K", 0)] = g[name + "_in_K_0"], g[name + "_in_inv_K_0"]
for f in (0, -1, 1):
    inputs[("color_aug", f, 0)] = inputs[("color", f, 0)]
B, _, Hh, W = inputs[("color", 0, 0)].shape
ml = dict(num_scales=4, frame_ids=[0, -1, 1], height=Hh, width=W, min_depth=0.1, max_depth=100, test_min_depth=1e-3,
          test_max_depth=80, disparity_smoothness=1e-3, no_ssim=False, avg_reprojection=False, disable_automasking=False)
def truth():
    cast = lambda v: v.double() if v.is_floating_point() else v
    sdo = {k: (cast(v.clone()).requires_grad_(True) if v.is_floating_point() and "running" not in k else cast(v.clone())) for k, v in sd.items()}
    inp = {k: cast(v) for k, v in inputs.items()}
    out = N.model_forward(sdo, cfg, inp, train=True, dropout=False)
    lo = P.MonodepthLossOracle(**ml, batch_size=B)
    lo.generate_images_pred(inp, out)
    tot = lo.compute_losses(inp, out, tiebreak_noise={s: g[name + "_noise_%d" % s].double() for s in range(4)})["loss"]
    if "semantics" in out:
        tot = tot + S.cross_entropy2d(out["semantics"], g[name + "_lbl"])
    tot.backward()
    return {k: float(v.grad.norm()) for k, v in sdo.items() if v.requires_grad and v.grad is not None}
import time; t0=time.time(); T = truth(); print("truth", time.time()-t0, flush=True)
names = [str(x) for x in g[name + "_grad_names"]]
def prod(scale, stem):
    H.STEM = stem
    sd2 = {k: v.clone() for k, v in sd.items()}
    sd2["models.encoder.encoder.conv1.weight"] = sd2["models.encoder.encoder.conv1.weight"] * scale
    model = get_model(cfg, 19); model.load_state_dict(sd2, strict=True); model.to(DEV).train(); MC.dropout_eval(model)
    inp = {k: v.to(DEV) for k, v in inputs.items()}
    out = model(inp)
    lo = get_monodepth_loss({"training": {"batch_size": B, "monodepth_loss": ml}}, is_train=True)
    lo.tiebreak_noise = {s: g[name + "_noise_%d" % s].to(DEV) for s in range(4)}
    lo.generate_images_pred(inp, out)
    tot = lo.compute_losses(inp, out)["loss"]
    if "semantics" in out:
        tot = tot + cross_entropy2d(out["semantics"], g[name + "_lbl"].to(DEV))
    tot.backward()
    p = dict(model.named_parameters())
    e = np.array([abs(float(p[k].grad.norm()) - T[k]) / (T[k] + 1e-30) for k in names if k in T and p[k].grad is not None])
    return np.median(e), e.max()
for stem in (True, False):
    for sc in (1.0, 1 + 1.2e-7, 1 - 1.2e-7, 1 + 2.4e-7, 1 - 2.4e-7):
        m, mx = prod(sc, stem)
        print("product stem=%d conv1 x %.7f: median %.2e max %.2e" % (stem, sc, m, mx), flush=True)
