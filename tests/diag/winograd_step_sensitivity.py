"""Diagnostic (GPU box): does the Winograd route move the cfg2 (ResNet-50 monodepth, 512x1024, batch 2) training step more than
the direct route does?  Per-parameter gradient error of step 0 against the float64 oracle, and the gradient norm of step 1
(after one SGD step), with SEGSDE_WINOGRAD on and off."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import model_cases as MC  # noqa: E402
from oracle import nets as N, photometric as P  # noqa: E402
from improving_segmentation_with_selfsupervised_depth_amd import hipops as H  # noqa: E402
from improving_segmentation_with_selfsupervised_depth_amd.models import get_model  # noqa: E402
from improving_segmentation_with_selfsupervised_depth_amd.loss import get_monodepth_loss  # noqa: E402

B, Hh, W = 2, 512, 1024
cfg = bench.model_cfg("cfg2", Hh, W)
sd = N.build_state_dict(cfg, 19, seed=11, randomize_bn=False)
inp = bench.synthetic_inputs(B, Hh, W, "cpu", 1234)
gen = torch.Generator().manual_seed(12)
noise = {s: torch.randn(B, 2, Hh, W, generator=gen) for s in range(4)}
torch.set_num_threads(max(1, min(64, len(os.sched_getaffinity(0)))))


def groups(named):
    enc = [p for k, p in named if k.startswith("models.encoder.")]
    rest = [p for k, p in named if not k.startswith("models.encoder.")]
    return torch.optim.SGD([{"params": enc, "lr": 1e-3}, {"params": rest}], lr=1e-2, momentum=0.9, weight_decay=5e-4)


def oracle(dt):
    cast = lambda v: v.to(dt) if torch.is_tensor(v) and v.is_floating_point() else v
    sdo = {k: (cast(v.clone()).requires_grad_(True) if v.is_floating_point() and "running" not in k else cast(v.clone())) for k, v in sd.items()}
    leaves = [(k, v) for k, v in sdo.items() if v.is_floating_point() and v.requires_grad]
    opt = groups(leaves)
    i_ = {k: cast(v) for k, v in inp.items()}
    lo = P.MonodepthLossOracle(**bench.loss_cfg(B, Hh, W)["training"]["monodepth_loss"], batch_size=B)
    res = []
    for step in range(2):
        opt.zero_grad(set_to_none=True)
        out = N.model_forward(sdo, cfg, i_, train=True, dropout=False)
        lo.generate_images_pred(i_, out)
        L = lo.compute_losses(i_, out, tiebreak_noise={s: cast(n) for s, n in noise.items()})["loss"]
        L.backward()
        g = {k: v.grad.detach().clone() for k, v in leaves if v.grad is not None}
        gn = float(torch.nn.utils.clip_grad_norm_([v for _, v in leaves if v.grad is not None], 10.0))
        opt.step()
        res.append((float(L), gn, g))
    return res


def product(wino):
    H.WINOGRAD = wino
    model = get_model(cfg, 19)
    model.load_state_dict(sd, strict=True)
    model.cuda().train()
    MC.dropout_eval(model)
    opt = groups(list(model.named_parameters()))
    lo = get_monodepth_loss(bench.loss_cfg(B, Hh, W), True)
    lo.tiebreak_noise = noise
    i_ = {k: v.cuda() for k, v in inp.items()}
    res = []
    for step in range(2):
        opt.zero_grad(set_to_none=True)
        out = model(i_)
        lo.generate_images_pred(i_, out)
        L = lo.compute_losses(i_, out)["loss"]
        L.backward()
        g = {k: p.grad.detach().double().cpu() for k, p in model.named_parameters() if p.grad is not None}
        gn = float(torch.nn.utils.clip_grad_norm_([p for p in model.parameters() if p.grad is not None], 10.0))
        opt.step()
        res.append((float(L), gn, g))
    return res


t64 = oracle(torch.float64)
o32 = oracle(torch.float32)
runs = {"fp32 oracle": o32, "product, direct": product(False), "product, Winograd": product(True)}
for step in range(2):
    print("step %d: float64 loss %.8f grad norm %.5f" % (step, t64[step][0], t64[step][1]))
    for name, r in runs.items():
        errs = sorted(((float((r[step][2][k].double() - t64[step][2][k]).norm()) / (float(t64[step][2][k].norm()) + 1e-30), k)
                       for k in t64[step][2]), reverse=True)
        e = np.array([x[0] for x in errs])
        print("   %-18s loss %.8f grad norm %.5f | per-parameter error vs float64: median %.2e max %.2e (%s)" % (
            name, r[step][0], r[step][1], np.median(e), e[0], errs[0][1]))
