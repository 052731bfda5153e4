"""Which ingredient of a captured training step breaks hipGraph instantiation on this stack?  One variant per process:
python tests/diag/graph_capture_bisect.py <variant>   with variant in base | randn | dropout | adam | big | all"""
import copy
import sys
import os
_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, _ROOT)
sys.path.insert(0, os.path.join(_ROOT, "tests"))
import torch
import bench
import model_cases as MC
from improving_segmentation_with_selfsupervised_depth_amd.models import get_model
from improving_segmentation_with_selfsupervised_depth_amd.models import layers as L_
from improving_segmentation_with_selfsupervised_depth_amd.models.layers import weight_pack_scope
from improving_segmentation_with_selfsupervised_depth_amd.loss import get_monodepth_loss

v = sys.argv[1]
B, Hh, W = (2, 256, 512) if v in ("big", "all") else (2, 64, 128)
dev = torch.device("cuda")
torch.manual_seed(3)
model = get_model(bench.model_cfg("cfg1", Hh, W), 19).to(dev).train()
if v not in ("dropout", "all"):
    MC.dropout_eval(model)
inputs = bench.synthetic_inputs(B, Hh, W, dev, 5, with_labels=False)
loss_obj = get_monodepth_loss(bench.loss_cfg(B, Hh, W), is_train=True)
if v not in ("randn", "all"):
    gen = torch.Generator().manual_seed(9)
    loss_obj.tiebreak_noise = {s: torch.randn(B, 2, Hh, W, generator=gen).to(dev) for s in range(4)}
if v in ("adam", "all"):
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=True, capturable=True)
else:
    opt = torch.optim.SGD(model.parameters(), lr=1e-2, momentum=0.9, fused=True)


def step():
    with weight_pack_scope(model):
        opt.zero_grad(set_to_none=True)
        out = model(inputs)
        loss_obj.generate_images_pred(inputs, out)
        total = loss_obj.compute_losses(inputs, out)["loss"]
        total.backward()
        opt.step()
        return total


L_.GRAPH_SAFE_DROPOUT[0] = True
side = torch.cuda.Stream(dev)
side.wait_stream(torch.cuda.current_stream(dev))
with torch.cuda.stream(side):
    step()
    step()
torch.cuda.current_stream(dev).wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    loss = step()
g.replay()
torch.cuda.synchronize()
print("VARIANT", v, "OK loss", float(loss))
