"""cProfile of the host side of cfg1's eager step on the GPU box (R18 monodepth, 256x512, batch 2: ~900 launches in a 20-23 ms step
whose kernels take 14.9 ms).  python tools/probes/cfg1_host_profile.py [steps]"""
import cProfile
import os
import pstats
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench as Bn  # noqa: E402
from improving_segmentation_with_selfsupervised_depth_amd.models import get_model  # noqa: E402
from improving_segmentation_with_selfsupervised_depth_amd.loss import get_monodepth_loss  # noqa: E402
from improving_segmentation_with_selfsupervised_depth_amd.models.layers import weight_pack_scope  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
Hh, W, B, opt_name, _ = Bn.WORKLOADS["cfg1"]
dev = torch.device("cuda", 0)
torch.manual_seed(42)
cfg = Bn.model_cfg("cfg1", Hh, W)
model = get_model(cfg, 19).to(dev).train()
opt = Bn.param_groups(model, opt_name)
loss_obj = get_monodepth_loss(Bn.loss_cfg(B, Hh, W), is_train=True)
inputs = Bn.synthetic_inputs(B, Hh, W, dev, 1234, with_labels=False)


def step():
    with weight_pack_scope(model):
        opt.zero_grad(set_to_none=True)
        out = model(inputs)
        loss_obj.generate_images_pred(inputs, out)
        total = loss_obj.compute_losses(inputs, out)["loss"]
        total.backward()
        opt.step()


for _ in range(5):
    step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(steps):
    step()
torch.cuda.synchronize()
print("eager: %.2f ms/step" % ((time.perf_counter() - t0) / steps * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
