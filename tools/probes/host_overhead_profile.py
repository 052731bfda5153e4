"""Where the host time of a launch-bound step goes (cfg1: ~900 launches in a 20-25 ms step, kernels 14.9 ms).  Runs the package's
train step on CPU tensors with every C-ABI entry replaced by a function that returns at once -- what is left is exactly the Python /
autograd / ctypes / allocator work the GPU run pays per step -- under cProfile.  Results are host-side only (tensor VALUES are
garbage; control flow that depends on them does not exist in the step).  `python tools/probes/host_overhead_profile.py [workload] [steps]`"""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench as Bn  # noqa: E402
from improving_segmentation_with_selfsupervised_depth_amd import _lib  # noqa: E402


class _Stub:
    """every entry point: returns 0 (success); *_workspace / *_ok style queries return a small positive size"""
    def __init__(self):
        self.calls = 0

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        ret = 4096 if name.endswith("_workspace") else (1 if name.endswith("_ok") or name.endswith("_supported") else 0)
        if name == "segsde_abi_version":
            ret = _lib.ABI_VERSION

        def f(*a, _r=ret):
            self.calls += 1
            return _r
        setattr(self, name, f)
        return f


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "cfg1"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    stub = _Stub()
    _lib._LIB = stub
    _lib.HOST_POINTERS_OK = True
    from improving_segmentation_with_selfsupervised_depth_amd.models import get_model
    from improving_segmentation_with_selfsupervised_depth_amd.loss import get_monodepth_loss
    from improving_segmentation_with_selfsupervised_depth_amd.loss.loss import cross_entropy2d
    from improving_segmentation_with_selfsupervised_depth_amd.models.layers import weight_pack_scope
    Hh, W, B, opt_name, _ = Bn.WORKLOADS[wl]
    torch.manual_seed(42)
    cfg = Bn.model_cfg(wl, Hh, W)
    model = get_model(cfg, 19).train()
    opt = Bn.param_groups(model, opt_name)
    loss_obj = get_monodepth_loss(Bn.loss_cfg(B, Hh, W), is_train=True)
    inputs = Bn.synthetic_inputs(B, Hh, W, "cpu", 1234, with_labels=cfg.get("segmentation_name") is not None)

    def step():
        with weight_pack_scope(model):
            opt.zero_grad(set_to_none=True)
            out = model(inputs)
            loss_obj.generate_images_pred(inputs, out)
            total = loss_obj.compute_losses(inputs, out)["loss"]
            if "semantics" in out:
                total = total + cross_entropy2d(out["semantics"], inputs["lbl"])
            total.backward()
        return total

    for _ in range(2):
        step()
    n0 = stub.calls
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    dt = (time.perf_counter() - t0) / steps
    print("%s: %.2f ms of host time per step without the optimizer, %d C-ABI calls per step (%.1f us per call)"
          % (wl, dt * 1e3, (stub.calls - n0) // steps, dt * 1e6 / max(1, (stub.calls - n0) // steps)))
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(steps):
        step()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(25)
    st.sort_stats("cumtime").print_stats("improving_segmentation|autograd/function", 25)


if __name__ == "__main__":
    main()
