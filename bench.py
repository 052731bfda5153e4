#!/usr/bin/env python
"""Training-throughput benchmark of the hot path.

    python bench.py --gpus N --steps K --warmup W

N>1 runs one rank per GPU over RCCL: either the caller launches the ranks (``python -m torch.distributed.run
--nproc-per-node N ... bench.py --gpus N``: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from the environment), or --
when WORLD_SIZE is not set -- this script re-executes itself under torch.distributed.run with N ranks on 127.0.0.1.

A "step" re-enacts the reference's Trainer.train_step (train.py:442-549) on synthetic, device-resident inputs:
model forward -> monodepth loss -> segmentation loss -> backward -> gradient all-reduce (N>1) -> clip_grad_norm ->
optimiser step.  Default workload = BASELINE.json configs[2]: cityscapes_joint.yml + decoder_variant(dec=6) ResNet-101
joint_seg_depth_dec, 512x1024, per-GPU batch 16, SGD (experiments.py:32-48,139-145).  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import re
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic GFLOP per image (conv + matmul, 2*MAC), measured on the reference model
GFLOP_PER_IMG = {"cfg1": 216.8, "cfg2": 1423.6, "cfg3": 2467.2, "cfg3pad": 2569.4,
                 # cfg5 (per LABELED image of the step): conv work scales with the pixel count (x4 vs 512x1024); the step runs
                 # the mtl_pad student three times fwd+bwd (labeled, unlabeled unmixed, unlabeled mixed; the mixed pass
                 # without a monodepth loss still runs the pose nets like the reference) and the teacher once forward
                 # (no pose nets): 4 * (3 * 2569.4 + (2569.4 / 3 - 83.4)) GFLOP
                 "cfg5": 4 * (3 * 2569.4 + (2569.4 / 3.0 - 83.4))}
PEAK_FP32_MATRIX_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, no TF32 on gfx950
# measured on the GPU box with rocprof-compute's roofline micro-benchmark (/opt/rocm/bin/roofline-ubuntu22_04-rocm7,
# profiles/roofline_r02_microbench.txt): the empirical roofs SURVEY.md 8d asks to quote next to the nominal ones
MEASURED_PEAKS = {"mfma_f32_tflops": 155.06, "hbm_gbs": 6362.5, "mall_gbs": 8711.6, "l2_gbs": 34711.8, "lds_gbs": 68629.9,
                  "source": "profiles/roofline_r02_microbench.txt"}


def model_cfg(workload, H, W):
    mono = dict(frame_ids=[0, -1, 1], num_scales=4, height=H, width=W)
    common = dict(arch="joint_segmentation_depth", pose_model_input="pairs", provide_uncropped_for_pose=False,
                  backbone_pretraining="none", depth_pretraining="none", pose_pretraining="none", freeze_backbone=False,
                  freeze_depth=False, freeze_pose=False, freeze_segmentation=False, disable_monodepth=False,
                  disable_pose=False, enable_imnet_encoder=False, **mono)
    dec = dict(intermediate_aspp=True, aspp_rates=[6, 12, 18], num_ch_dec=[64, 128, 128, 256, 256], max_scale_size=[H, W])
    jsd = dict(weights="none", layers=[9], head_inter_channels=64, layer_out_channels=64, head_dropout=0.1,
               layer_dropout=0, head_inter=False, output_stride=1)
    pad = dict(weights="none", output_stride=1, distillation_layer=7, side_output=True, final_layer=9)
    if workload == "cfg1":
        return dict(common, backbone_name="resnet18", replace_stride_with_dilation=None, segmentation_name=None,
                    segmentation_args=None, depth_args=dec)
    if workload == "cfg2":
        return dict(common, backbone_name="resnet50", replace_stride_with_dilation=[False, False, True],
                    segmentation_name=None, segmentation_args=None, depth_args=dec)
    if workload == "cfg3":
        return dict(common, backbone_name="resnet101", replace_stride_with_dilation=[False, False, True],
                    segmentation_name="joint_seg_depth_dec", segmentation_args=jsd, depth_args=dec)
    if workload in ("cfg3pad", "cfg5"):
        return dict(common, backbone_name="resnet101", replace_stride_with_dilation=[False, False, True],
                    segmentation_name="mtl_pad", segmentation_args=pad, depth_args=dec)
    raise KeyError(workload)


WORKLOADS = {  # name -> (H, W, per-GPU batch, optimiser, description)
    "cfg1": (256, 512, 2, "adam", "ResNet-18 monodepth dec5, 256x512, batch 2 (BASELINE configs[0])"),
    "cfg2": (512, 1024, 8, "adam", "ResNet-50 monodepth dec5, 512x1024, batch 8 (BASELINE configs[1])"),
    "cfg3": (512, 1024, 16, "sgd", "ResNet-101 joint_seg_depth_dec seg+depth, 512x1024, batch 16/GPU (BASELINE configs[2])"),
    "cfg3pad": (512, 1024, 16, "sgd", "ResNet-101 mtl_pad seg+depth, 512x1024, batch 16/GPU"),
    "cfg5": (1024, 2048, 2, "sgd", "ResNet-101 mtl_pad seg+depth + DepthMix unlabeled step (teacher fwd, online-depth depthcomp "
                                   "mask, mix + colour jitter + blur, 2 student fwd/bwd, EMA), 1024x2048 crops, batch 2 labeled + 2 unlabeled per GPU "
                                   "(BASELINE configs[4]; reference asserts batch 2 for depthcomp, train.py:586)"),
}


def synthetic_inputs(B, H, W, device, seed, with_labels=True):
    """SURVEY.md 8d 'Synthetic inputs': U[0,1) frames, Cityscapes intrinsics (not rescaled, as the reference), labels
    with ~5 % ignore."""
    g = torch.Generator().manual_seed(seed)
    inp = {}
    for f in (0, -1, 1):
        inp[("color", f, 0)] = torch.rand(B, 3, H, W, generator=g)
        inp[("color_aug", f, 0)] = inp[("color", f, 0)]
    for s in range(1, 4):
        inp[("color", 0, s)] = torch.rand(B, 3, H // 2 ** s, W // 2 ** s, generator=g)
    K = np.array([[2262.52, 0, 1096.98, 0], [0, 2265.3017905988554, 513.137, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float32)
    inp[("K", 0)] = torch.from_numpy(K).unsqueeze(0).repeat(B, 1, 1)
    inp[("inv_K", 0)] = torch.from_numpy(np.linalg.pinv(K)).unsqueeze(0).repeat(B, 1, 1)
    if with_labels:
        lbl = torch.randint(0, 19, (B, H, W), generator=g)
        lbl[torch.rand(B, H, W, generator=g) < 0.05] = 250
        inp["lbl"] = lbl
    return {k: v.to(device) for k, v in inp.items()}


def loss_cfg(B, H, W):
    return {"training": {"batch_size": B, "monodepth_loss": dict(
        num_scales=4, frame_ids=[0, -1, 1], height=H, width=W, min_depth=0.1, max_depth=100, test_min_depth=1e-3,
        test_max_depth=80, disparity_smoothness=1e-3, no_ssim=False, avg_reprojection=False, disable_automasking=False)}}


def param_groups(model, opt, capturable=False):
    """train.py:67-101 with experiments.py:32-48: backbone lr 1e-3, everything else 1e-2 (sgd); adam 1e-4"""
    # the stock torch optimisers; on the GPU their fused single-kernel implementation (same update rule, one pass over
    # parameter / gradient / momentum instead of ~5 multi-tensor passes: ~0.7 ms of a cfg3 step).  SEGSDE_BENCH_FUSED_OPT=0: foreach
    dev_ok = next(model.parameters()).is_cuda and os.environ.get("SEGSDE_BENCH_FUSED_OPT", "1") != "0"
    extra = {"fused": True} if dev_ok else {}
    if opt == "adam":
        if capturable:
            extra = dict(extra, capturable=True)      # step counters on the device: the update can be captured in a hipGraph
        return torch.optim.Adam(model.parameters(), lr=1e-4, **extra)
    enc = list(model.models["encoder"].parameters())
    ids = {id(p) for p in enc}
    rest = [p for p in model.parameters() if id(p) not in ids]
    return torch.optim.SGD([{"params": enc, "lr": 1e-3}, {"params": rest}], lr=1e-2, momentum=0.9, weight_decay=5e-4, **extra)


def csrc_sha256():
    """fingerprint of the kernel sources: committed PMC summaries carry it, a summary of other sources is refused"""
    import glob
    import hashlib
    h = hashlib.sha256()
    base = os.path.join(ROOT, "improving_segmentation_with_selfsupervised_depth_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(base, "*.hip")) + glob.glob(os.path.join(base, "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def newest_profile(pattern):
    import glob
    c = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    return c[-1] if c else None


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def cpu_baseline(workload, H, W, budget_s):
    """The oracle (CPU torch restatement of the reference path, validated against the reference) timed on this box's host
    cores: one un-timed warm-up step (first-call costs: oneDNN primitive creation, allocator growth), then
    whole train steps (fwd + mono loss + seg loss + bwd) until >= 3 steps (>= 10 for cfg1) are done or the budget is spent;
    batch 2 (cfg1: its own batch 2 = the configuration in full)."""
    from oracle import nets as N, photometric as P, segmix as S
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 64))
    torch.set_num_threads(cores)
    if workload == "cfg5":
        workload_model, H, W = "cfg3pad", 512, 1024     # the labeled part of the step at the cfg3 shape (CPU budget)
    else:
        workload_model = workload
    cfg = model_cfg(workload_model, H, W)
    B = 2
    sd = N.build_state_dict(cfg, 19, seed=0)
    sd = {k: (v.requires_grad_(True) if v.is_floating_point() and "running" not in k else v) for k, v in sd.items()}
    inp = synthetic_inputs(B, H, W, "cpu", 99, with_labels=cfg.get("segmentation_name") is not None)
    lo = P.MonodepthLossOracle(**loss_cfg(B, H, W)["training"]["monodepth_loss"], batch_size=B)
    # the same tail as the GPU step: clip_grad_norm_ (sgd workloads, train.py:516-524) + the stock optimizer's step
    leaves = [v for v in sd.values() if v.is_floating_point() and v.requires_grad]
    opt_name = WORKLOADS[workload][3]
    opt = torch.optim.SGD(leaves, lr=1e-2, momentum=0.9, weight_decay=5e-4) if opt_name == "sgd" else torch.optim.Adam(leaves, lr=1e-4)

    def step():
        for v in sd.values():
            if v.is_floating_point() and v.grad is not None:
                v.grad = None
        out = N.model_forward(sd, cfg, inp, train=True, dropout=True)
        lo.generate_images_pred(inp, out)
        total = lo.compute_losses(inp, out)["loss"]
        if "semantics" in out:
            seg = S.cross_entropy2d(out["semantics"], inp["lbl"])
            if "intermediate_semantics" in out:
                seg = (seg + S.cross_entropy2d(out["intermediate_semantics"], inp["lbl"])) / 2
            total = total + seg
        total.backward()
        if opt_name == "sgd":
            torch.nn.utils.clip_grad_norm_([v for v in leaves if v.grad is not None], 10.0)
        opt.step()

    t0 = time.time()
    step()
    warm = time.time() - t0
    want = 10 if workload == "cfg1" else 3
    times = []
    while len(times) < want and (len(times) == 0 or time.time() - t0 + 1.2 * max(times) < budget_s):
        t1 = time.time()
        step()
        times.append(time.time() - t1)
    med = float(np.median(times))
    return {"value": B / med, "unit": "img/s", "cores": cores, "kind": "port",
            "sample": "oracle (CPU PyTorch restatement, validated against the reference): %s at batch %d, 1 warm-up step "
                      "(%.1f s) + %d timed train steps fwd + losses + ONE backward() of their sum + clip_grad_norm_ + optimizer step, "
                      "like the GPU's `summed` step (median %.2f s, min %.2f s), %d threads on %s, torch %s"
                      % (workload_model, B, warm, len(times), med, min(times), cores, cpu_model_name(), torch.__version__)}


def rccl_env():
    """the documented way to steer RCCL here is its own environment (NCCL_MIN_NCHANNELS / NCCL_MAX_NCHANNELS for the number of
    channels = CUs the collective kernels occupy next to the MFMA kernels, NCCL_ALGO, NCCL_PROTO, ...): the line records what was set"""
    return {k: v for k, v in sorted(os.environ.items()) if k.startswith(("NCCL_", "RCCL_", "HSA_ENABLE_IPC", "TORCH_NCCL_"))}


def _flush_c_stdio():
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)    # median over >= 20 timed steps after >= 5 warm-up steps
    ap.add_argument("--warmup", type=int, default=5)    # (~8 s of device time for cfg3; the allocator settles in step 3)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the workload's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--kernel-timing-period", type=int, default=0,
                    help="bracket launch q of step i with HIP events iff (q + i) %% P == 0 (1: every launch of every step; "
                         "default: min(steps, 8) -- every launch site once per P steps, 1/P of the event overhead per step)")
    ap.add_argument("--hip-graph", action="store_true",
                    help="capture one whole training step (forward, loss, backward, clip, optimiser) as a hipGraph after the warm-up and "
                         "replay it for the timed steps: the launch-bound regime (cfg1: ~1 500 launches of 5-20 us per step through "
                         "Python / ctypes).  Single GPU, labeled step only; implies --no-kernel-timing")
    ap.add_argument("--step", choices=["summed", "reference", "amp"], default="summed",
                    help="what the timed step is.  summed (the headline): the losses of a forward are added and back-propagated with one "
                         "backward() call.  reference: trainer.train_step, the reference's own call sequence (train.py:486,510: one "
                         "backward() per loss, the shared encoder back-propagated once through the deferred trunk backward).  The "
                         "default run times `summed` and reports the reference sequence next to it as `reference_step`")
    ap.add_argument("--no-reference-step", action="store_true", help="skip the `reference_step` block of the default run")
    ap.add_argument("--no-amp-step", action="store_true", help="skip the `amp_step` block (amp: True with f16 convolution operands)")
    ap.add_argument("--bucket-mb", type=float, default=32.0, help="gradient all-reduce bucket size (N > 1)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="N > 1: start every gradient all-reduce after the last backward instead of from the gradient hooks")
    ap.add_argument("--allreduce-only", action="store_true",
                    help="N > 1: after one training step (which builds the buckets) time the gradient payload's all-reduce alone, "
                         "K = --steps times; prints its own JSON line")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-baseline-timeout", type=float, default=150.0)
    args = ap.parse_args()
    if args.cpu_baseline_only:
        Hh, W = WORKLOADS[args.workload][:2]
        print(json.dumps(cpu_baseline(args.workload, Hh, W, args.cpu_baseline_timeout - 25.0)))
        return

    if args.hip_graph:
        if args.gpus != 1 or args.workload == "cfg5":
            raise SystemExit("--hip-graph: single GPU, labeled step only")
        args.no_kernel_timing = True
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N`: become the launcher -- N ranks of this very script over RCCL on 127.0.0.1
        import socket
        import subprocess
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        sys.exit(subprocess.call(cmd, env=env))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch one rank per GPU (or drop WORLD_SIZE and let bench.py spawn them)"
                         % (args.gpus, world))
    # test knobs (a 1-GPU box cannot run two RCCL ranks): SEGSDE_BENCH_ONE_DEVICE=1 puts every rank on device 0,
    # SEGSDE_BENCH_BACKEND=gloo swaps the collective backend -- together they exercise the whole N>1 path (self-spawn,
    # rendezvous, parameter broadcast, hook-driven bucketed all-reduce, max-over-ranks timing) on one GPU
    if os.environ.get("SEGSDE_BENCH_ONE_DEVICE"):
        local_rank = 0
    backend = os.environ.get("SEGSDE_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)     # before constructing anything (SURVEY.md 8b device quirk)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    force_reducer = bool(os.environ.get("SEGSDE_FORCE_REDUCER"))   # exercise the RCCL all-reduce path on one GPU
    if world > 1 or force_reducer:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import __graft_entry__ as ge
    if not os.path.exists(ge.LIB):
        ge.build()
    from improving_segmentation_with_selfsupervised_depth_amd import hipops as H
    from improving_segmentation_with_selfsupervised_depth_amd import trainer as T
    from improving_segmentation_with_selfsupervised_depth_amd.models import get_model
    from improving_segmentation_with_selfsupervised_depth_amd.loss import get_monodepth_loss
    from improving_segmentation_with_selfsupervised_depth_amd.loss.loss import cross_entropy2d
    from improving_segmentation_with_selfsupervised_depth_amd.ddp import GradAllReducer

    Hh, W, B, opt_name, desc = WORKLOADS[args.workload]
    if args.batch:
        B = args.batch
    torch.manual_seed(42)
    cfg = model_cfg(args.workload, Hh, W)
    model = get_model(cfg, 19).to(dev).train()
    optimizer = param_groups(model, opt_name, capturable=args.hip_graph)
    loss_obj = get_monodepth_loss(loss_cfg(B, Hh, W), is_train=True)
    reducer = GradAllReducer(model, bucket_mb=args.bucket_mb, overlap=not args.no_overlap, always=force_reducer,
                             timing=True) if (world > 1 or force_reducer) else None
    use_reducer = [True]      # False: the same step with every backward under no_sync() and no finish() (exposed-communication probe)
    # identical initial parameters (seed 42 above + the reducer's broadcast), then INDEPENDENT dropout / tie-break
    # / augmentation streams per replica
    from improving_segmentation_with_selfsupervised_depth_amd.ddp import seed_per_rank
    rank_seed = seed_per_rank(42, rank)
    inputs = synthetic_inputs(B, Hh, W, dev, 1234 + rank, with_labels=cfg.get("segmentation_name") is not None)
    clip = 10.0 if opt_name == "sgd" else None
    unlabeled = args.workload == "cfg5"
    if unlabeled:
        # train.py:280-288, 328-344: the EMA teacher is a second full copy of the model with detached parameters
        ema_model = get_model(cfg, 19).to(dev).train()
        ema_model.load_state_dict(model.state_dict())
        for p_ in ema_model.parameters():
            p_.detach_()
        unlabeled_inputs = synthetic_inputs(B, Hh, W, dev, 4321 + rank, with_labels=False)
        # exp-212 (experiments.py:343-357) sets mix_use_gt: samples of the unlabeled loader that carry a label hand their
        # one-hot ground truth to the mix instead of the teacher's softmax (train.py:667-672).  Half of the batch is
        # flagged, with one-hot planes in the loader's dtype / layout (int64 [19,H,W], all zero on ignored pixels).
        g_ = torch.Generator().manual_seed(777 + rank)
        lbl_u = torch.randint(0, 19, (B, Hh, W), generator=g_)
        lbl_u[torch.rand(B, Hh, W, generator=g_) < 0.05] = 19
        unlabeled_inputs["onehot_lbl"] = torch.nn.functional.one_hot(lbl_u, 21)[..., :19].permute(0, 3, 1, 2).contiguous().to(dev)
        unlabeled_inputs["is_labeled"] = (torch.arange(B) % 2 == 0).to(dev)
        del lbl_u
    it = [0]

    def nosync():
        import contextlib
        return reducer.no_sync() if reducer is not None else contextlib.nullcontext()

    def last_backward():
        import contextlib
        return reducer.no_sync() if (reducer is not None and not use_reducer[0]) else contextlib.nullcontext()

    from improving_segmentation_with_selfsupervised_depth_amd.models.layers import weight_pack_scope

    def step():
        with weight_pack_scope(model):  # weights change only in optimizer.step(): packed once per step, in one launch
            return step_body()

    def step_body():
        optimizer.zero_grad(set_to_none=True)
        out = model(inputs)
        loss_obj.generate_images_pred(inputs, out)
        total = loss_obj.compute_losses(inputs, out)["loss"]
        if "semantics" in out:
            seg = cross_entropy2d(out["semantics"], inputs["lbl"])
            if "intermediate_semantics" in out:
                seg = (seg + cross_entropy2d(out["intermediate_semantics"], inputs["lbl"])) / 2
            total = total + seg
        if unlabeled:
            # train.py:511-514: labeled backward, then the unlabeled step accumulates onto the same gradients; only the
            # last backward of the step may start the gradient all-reduce
            with nosync():
                total.backward()
            del out
            with last_backward():
                L_u, mono_u = T.train_step_segmentation_unlabeled(
                    model, ema_model, loss_obj, unlabeled_inputs, mix_mask="depthcomp", depthmix_online_depth=True,
                    monodepth_lambda=1.0, consistency_weight=1.0, backward_first_pseudo_label=False, depthcomp_margin=0.03,
                    depthcomp_foreground_threshold=0.0, color_jitter=True, blur=True, reducer=reducer, mix_use_gt=True)
            total = total.detach() + L_u.detach() + mono_u.detach()
        else:
            with last_backward():
                total.backward()
        if reducer is not None and use_reducer[0]:
            reducer.finish()
        if clip is not None:
            torch.nn.utils.clip_grad_norm_(model.parameters(), clip)
        optimizer.step()
        if unlabeled:
            T.update_ema_variables(ema_model, model, 0.99, it[0], segmentation_name=cfg["segmentation_name"])
        it[0] += 1
        return total

    # The reference's own call sequence for the same work: trainer.train_step is Trainer.train_step
    # (train.py:442-549) -- one backward() per loss (486: retain_graph=True, 510), clip_grad_norm_, optimizer.step(), for cfg5 the
    # unlabeled step and the EMA update inside.  ``defer``: the package's deferred trunk backward (functional.defer_trunk; the
    # default of trainer.train_step when the segmentation loss is on) -- off, the encoder is back-propagated once per call.
    has_seg = cfg.get("segmentation_name") is not None
    ref_cfg = {"model": dict(cfg), "training": {
        "amp": False, "monodepth_lambda": 1.0, "segmentation_lambda": 1.0 if has_seg else 0.0, "pseudo_depth_lambda": 0.0,
        "feat_dist_lambda": 0.0, "clip_grad_norm": clip, "save_monodepth_ema": False, "defer_trunk_backward": True,
        "unlabeled_segmentation": None if not unlabeled else dict(
            mix_mask="depthcomp", depthmix_online_depth=True, consistency_weight=1.0, backward_first_pseudo_label=False,
            depthcomp_margin=0.03, depthcomp_foreground_threshold=0.0, color_jitter=True, blur=True, mix_use_gt=True)}}
    backward_calls = {"summed": (1 if not unlabeled else 3), "reference": (1 + int(has_seg)) + (2 if unlabeled else 0)}
    backward_calls["amp"] = backward_calls["reference"]

    def step_reference():
        with weight_pack_scope(model):
            out = T.train_step(model, optimizer, inputs, it[0], ref_cfg, lambda input, target: cross_entropy2d(input, target),
                               loss_obj, ema_model=ema_model if unlabeled else None,
                               unlabeled_inputs=dict(unlabeled_inputs) if unlabeled else None,
                               reducer=reducer if use_reducer[0] else None)
        it[0] += 1
        return out["total_loss"]

    step_summed = step
    if args.step == "reference":
        step = step_reference
    if args.step == "amp":
        # profiling aid: the `amp_step` workload (reference sequence, amp: True, f16 convolution operands) as the timed step --
        # its line says so in `metric` / `dtype` and is never the judged number
        from improving_segmentation_with_selfsupervised_depth_amd import functional as Fn_amp
        Fn_amp.AMP_COMPUTE[0] = "f16"
        ref_cfg["training"]["amp"] = True
        step = step_reference
        args.no_reference_step = True

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # the last warm-up step runs with the launch bracketing on (results discarded): on a fresh box the first use of the timed-event
    # path (hipEventCreate with timing, the record packets) pages cold library code in -- a ~60 ms host stall that otherwise lands
    # in the first TIMED step (measured: 372 vs 311 ms, only in the first bench process of a box)
    import contextlib
    # --hip-graph: the warm-up runs on a side stream, like the capture -- autograd's AccumulateGrad nodes remember the stream of the
    # first backward; nodes born on the default stream make the captured backward wait on it, which invalidates the capture
    # (hipStreamEndCapture then crashed the process)
    warm_stream = torch.cuda.Stream(dev) if args.hip_graph else None
    if warm_stream is not None:
        warm_stream.wait_stream(torch.cuda.current_stream(dev))
    with (torch.cuda.stream(warm_stream) if warm_stream is not None else contextlib.nullcontext()):
        for w in range(args.warmup):
            if w == args.warmup - 1 and not args.no_kernel_timing:
                H.PROFILE, H.PROFILE_PERIOD = [], max(1, min(args.steps, 16 if args.workload == "cfg1" else 8) if args.kernel_timing_period <= 0 else args.kernel_timing_period)
                H.profile_step(0)
            step()
    if warm_stream is not None:
        torch.cuda.current_stream(dev).wait_stream(warm_stream)
    barrier()
    if H.PROFILE:
        _ = [s_.elapsed_time(e_) for _, _, s_, e_, _, _ in H.PROFILE if s_ is not None][:4]
    H.PROFILE = None
    if os.environ.get("SEGSDE_BENCH_ATEN_OPS") and rank == 0:
        # diagnosis: which ATen operators (with shapes) still run inside a step -- everything on the hot path should be a
        # kernel of this package
        from torch.profiler import profile, ProfilerActivity
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as tp:
            step()
            barrier()
        with open(os.environ["SEGSDE_BENCH_ATEN_OPS"], "w") as f:
            ka = tp.key_averages(group_by_input_shape=True)
            f.write(ka.table(sort_by="cuda_time_total", row_limit=80, max_name_column_width=60, max_shapes_column_width=90))
            f.write("\n\nATen operators (device time total us, calls, input shapes)\n")
            for e in sorted((e for e in ka if e.key.startswith("aten::")), key=lambda e: -e.device_time_total):
                if e.device_time_total > 0:
                    f.write("%-28s %10.1f %6d  %s\n" % (e.key, e.device_time_total, e.count, str(e.input_shapes)[:150]))
    graph = None
    if args.hip_graph:
        # whole-step capture (torch.cuda.graphs drives hipStreamBeginCapture / hipGraphInstantiate; the package's kernels are
        # launched on torch's current stream, i.e. the capturing one; every allocation comes from the graph's private pool)
        from improving_segmentation_with_selfsupervised_depth_amd.models import layers as L_
        L_.GRAPH_SAFE_DROPOUT[0] = True
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            step()                      # once more on the side stream, with the graph-safe dropout route
        torch.cuda.current_stream(dev).wait_stream(side)
        barrier()
        graph = torch.cuda.CUDAGraph()
        optimizer.zero_grad(set_to_none=True)
        with torch.cuda.graph(graph):
            graph_loss = step()
        barrier()

        def step():
            graph.replay()
            return graph_loss
        for _ in range(2):
            step()
        barrier()
    if args.allreduce_only:
        if reducer is None or reducer.buckets is None:
            raise SystemExit("--allreduce-only needs --gpus N > 1 (or SEGSDE_FORCE_REDUCER=1) and at least one warm-up step")
        # the gradient payload alone: every bucket launched back to back, then waited for -- what a step's communication costs
        # when nothing overlaps it.  Bus bandwidth by the usual all-reduce convention 2 (n - 1) / n * bytes / time.
        nbytes = sum(b.flat.numel() * 4 for b in reducer.buckets)
        for _ in range(2):
            for b in reducer.buckets:
                reducer._launch(b)
            for b in reducer.buckets:
                reducer._wait(b)
        barrier()
        reducer.timing_ms()
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        marks[0].record()
        for i in range(args.steps):
            for b in reducer.buckets:
                reducer._launch(b)
            for b in reducer.buckets:
                reducer._wait(b)
            marks[i + 1].record()
        barrier()
        ms_all = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
        t = torch.tensor([float(np.median(ms_all))], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        med = float(t[0])
        busy, ncoll, _ = reducer.timing_ms()
        if rank == 0:
            print(json.dumps({"metric": "gradient all-reduce of one training step alone (%s)" % args.workload, "value": med,
                              "unit": "ms", "higher_is_better": False, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "payload_mb": nbytes / 1e6, "buckets": len(reducer.buckets), "bucket_mb": args.bucket_mb,
                              "bus_gbs": (2.0 * (world - 1) / world) * nbytes / (med * 1e-3) / 1e9 if world > 1 else None,
                              "collective_stream_busy_ms": busy / args.steps, "ms_all": [round(x, 3) for x in ms_all],
                              "backend": dist.get_backend() if dist.is_initialized() else None, "rccl_env": rccl_env()}), flush=True)
        if world > 1 or force_reducer:
            dist.barrier()
            dist.destroy_process_group()
        return
    torch.cuda.reset_peak_memory_stats(dev)
    from improving_segmentation_with_selfsupervised_depth_amd import functional as Fn
    Fn.fusion_report(reset=True)
    if reducer is not None:
        reducer.timing_ms()          # drop the warm-up steps' events
    prof = None if args.no_kernel_timing else []
    H.PROFILE = prof
    # (launch-bound workloads: every bracketed launch costs the host two event records, ~8 us -- one launch in 16 instead of 8)
    period = args.kernel_timing_period if args.kernel_timing_period > 0 else max(1, min(args.steps, 16 if args.workload == "cfg1" else 8))
    H.PROFILE_PERIOD = period
    # per-step times without a host synchronisation inside the timed region: one HIP event per step boundary on the
    # launch stream (the device executes the steps back to back; the host runs ahead)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        H.profile_step(i)
        last = step()
        marks[i + 1].record()
    barrier()
    dt = time.perf_counter() - t0
    H.PROFILE = None
    step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
    med_ms = float(np.median(step_ms))
    comm = None
    if reducer is not None:
        # why the N-GPU number is what it is, from the same run: how long the collectives occupied their stream per step
        # (event pairs on the reducer's side stream), how much of that the step could NOT hide (this step minus the same step
        # with every backward under no_sync() and no finish(): identical compute, no collective), and every rank's own median
        busy_ms, ncoll, nbytes = reducer.timing_ms()
        n2 = max(2, min(args.steps, 5))
        use_reducer[0] = False
        step()
        marks2 = [torch.cuda.Event(enable_timing=True) for _ in range(n2 + 1)]
        marks2[0].record()
        for i in range(n2):
            step()
            marks2[i + 1].record()
        barrier()
        use_reducer[0] = True
        quiet_ms = float(np.median([marks2[i].elapsed_time(marks2[i + 1]) for i in range(n2)]))
        per_rank = torch.zeros(world, device=dev, dtype=torch.float64)
        per_rank[rank] = med_ms
        q = torch.tensor([quiet_ms, busy_ms / args.steps], device=dev, dtype=torch.float64)
        if world > 1:
            dist.all_reduce(per_rank, op=dist.ReduceOp.SUM)
            dist.all_reduce(q, op=dist.ReduceOp.MAX)
        comm = {"allreduce_ms_per_step": float(q[1]), "allreduce_launches_per_step": ncoll / args.steps,
                "allreduce_mb_per_step": nbytes / args.steps / 1e6, "buckets": len(reducer.buckets) if reducer.buckets else 0,
                "bucket_mb": args.bucket_mb, "overlap": bool(reducer.overlap),
                "zero_copy_members": dict(__import__("improving_segmentation_with_selfsupervised_depth_amd.ddp", fromlist=["ZERO_COPY"]).ZERO_COPY),
                "ms_per_step_without_allreduce": float(q[0]), "per_rank_ms_per_step": [round(float(x), 3) for x in per_rank],
                "rccl_env": rccl_env(),
                "how": "allreduce_ms_per_step: union of the [start, end] event pairs of the buckets' collectives on the reducer's side "
                       "stream (max over ranks); exposed_comm_ms = median step - median of %d steps of the same work with every backward "
                       "under no_sync() and no finish() (max over ranks each)" % n2}
    # the other call sequence on the same model, inputs and optimizer state, right after the timed region (not part of `value`)
    ref_block = None
    if not (args.no_reference_step or args.hip_graph):
        from improving_segmentation_with_selfsupervised_depth_amd import functional as Fn_

        def time_steps(fn, n, warm):
            for _ in range(warm):
                fn()
            mk = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
            mk[0].record()
            for i in range(n):
                fn()
                mk[i + 1].record()
            barrier()
            return [mk[i].elapsed_time(mk[i + 1]) for i in range(n)]

        other = step_summed if args.step == "reference" else step_reference
        n_ref = max(3, min(args.steps, 10))
        try:
            g0 = (Fn_.TrunkGateFn.trunk_backwards, Fn_.TrunkGateFn.parked_passes)
            other_ms = time_steps(other, n_ref, 2)
            g1 = (Fn_.TrunkGateFn.trunk_backwards, Fn_.TrunkGateFn.parked_passes)
            t = torch.tensor([float(np.median(other_ms))], device=dev, dtype=torch.float64)
            if world > 1:
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
            other_med = float(t[0])
            nodefer_med = None
            if has_seg and args.step != "reference":
                ref_cfg["training"]["defer_trunk_backward"] = False         # the same sequence with the encoder walked by every call
                t = torch.tensor([float(np.median(time_steps(step_reference, 3, 1)))], device=dev, dtype=torch.float64)
                ref_cfg["training"]["defer_trunk_backward"] = True
                if world > 1:
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                nodefer_med = float(t[0])
            which = "summed" if args.step == "reference" else "reference"
            ref_block = {"step": which, "ms_per_step": other_med, "img_s": B * world / (other_med * 1e-3), "steps": n_ref,
                         "backward_calls": backward_calls[which],
                         "ms_per_step_all": [round(x, 2) for x in other_ms]}
            if which == "reference":
                ref_block.update(
                    sequence="trainer.train_step = Trainer.train_step (train.py:442-549): forward, mono_total_loss.backward(retain_graph=True), "
                             "segmentation_total_loss.backward(), clip_grad_norm_, optimizer.step()" + (" + unlabeled step + EMA" if unlabeled else ""),
                    deferred_trunk_backward=bool(has_seg),
                    gate_flushes_per_step=(g1[0] - g0[0]) / (n_ref + 2),       # one per gate and forward (PAD: two gates; cfg5: three forwards)
                    parked_passes_per_step=(g1[1] - g0[1]) / (n_ref + 2),
                    ms_per_step_encoder_walked_by_every_call=nodefer_med)
        except Exception as ex:       # the second sequence must never cost the headline its line
            ref_block = {"step": "reference" if args.step != "reference" else "summed", "error": repr(ex)}
            model.defer_trunk_backward = False
            Fn_.flush_deferred_trunks()
    # `amp: True` with the reference's reduced-precision arithmetic (train.py:300,468-528 + torch autocast): the same
    # trainer.train_step under autocast / GradScaler with the convolutions' operands rounded to fp16 in the kernels
    # (SEGSDE_AMP_COMPUTE=f16, functional.AMP_COMPUTE).  A workload of its own -- never `value`: the metric is the fp32 line.
    amp_block = None
    if ref_block is not None and "error" not in ref_block and not args.no_amp_step:
        try:
            old_amp = Fn_.AMP_COMPUTE[0]
            Fn_.AMP_COMPUTE[0] = "f16"
            ref_cfg["training"]["amp"] = True
            n_amp = max(3, min(args.steps, 8))
            amp_ms = time_steps(step_reference, n_amp, 3)
            t = torch.tensor([float(np.median(amp_ms))], device=dev, dtype=torch.float64)
            if world > 1:
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
            amp_med = float(t[0])
            amp_loss = float(step_reference().detach())
            sc = getattr(optimizer, "_segsde_scaler", None)
            amp_block = {"ms_per_step": amp_med, "img_s": B * world / (amp_med * 1e-3), "steps": n_amp,
                         "dtype": "convolution operands rounded to f16 in the kernels (v_mfma_f32_32x32x16_f16), f32 accumulation; "
                                  "activations, weights, BatchNorm, losses, optimizer f32",
                         "sequence": "trainer.train_step with amp: True -- autocast around forward and segmentation loss, GradScaler around "
                                     "every backward / clip / step, like train.py:468-528",
                         "algorithmic_tflops": B / (amp_med * 1e-3) * GFLOP_PER_IMG[args.workload] / 1e3,
                         "frac_of_dense_f16_peak_2500": B / (amp_med * 1e-3) * GFLOP_PER_IMG[args.workload] / 1e3 / 2500.0,
                         "loss_scale": float(sc.get_scale()) if sc is not None and sc.is_enabled() else None,
                         "loss_after": amp_loss, "speedup_over_f32_headline": med_ms / amp_med,
                         "ms_per_step_all": [round(x, 2) for x in amp_ms]}
        except Exception as ex:
            amp_block = {"error": repr(ex)}
        finally:
            Fn_.AMP_COMPUTE[0] = old_amp
            ref_cfg["training"]["amp"] = False
    if world > 1:
        t = torch.tensor([dt, med_ms], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, med_ms = float(t[0]), float(t[1])
    if comm is not None:
        comm["exposed_comm_ms"] = med_ms - comm["ms_per_step_without_allreduce"]
    loss_val = float(last.detach())

    if rank == 0:
        # the metric is global batch / MEDIAN step time (the reference's Time/Image convention, robust against
        # a stray slow step); the mean over the barrier-bracketed region is reported next to it
        ms_mean = dt / args.steps * 1e3
        ms = med_ms
        value = B * world / (med_ms * 1e-3)
        res = {"metric": "train images/sec, ResNet-101 joint seg+depth @512x1024" if args.workload.startswith("cfg3")
               else ("train labeled images/sec, ResNet-101 joint seg+depth + DepthMix @1024x2048" if unlabeled
                     else "train images/sec (%s)" % args.workload),
               "value": value, "unit": "img/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": ms, "ms_per_step_mean": ms_mean, "value_from_mean": B * world * args.steps / dt,
               "ms_per_step_min": min(step_ms), "ms_per_step_max": max(step_ms), "ms_per_step_all": [round(t, 2) for t in step_ms], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32" if args.step != "amp" else "f16 convolution operands, f32 accumulation (amp: True; NOT the judged fp32 metric)",
               "data": "synthetic", "config": {"workload": desc, "per_gpu_batch": B, "global_batch": B * world,
                                               "height": Hh, "width": W,
                                               "optimizer": opt_name + (" (torch fused)" if getattr(optimizer, "defaults", {}).get("fused") else ""),
                                               "parallelism": "dp%d" % world, "final_loss": loss_val, "hip_graph": bool(args.hip_graph),
                                               "step": args.step, "backward_calls": backward_calls[args.step],
                                               "ranks": dist.get_world_size() if dist.is_initialized() else 1,
                                               "backend": (dist.get_backend() + (" (RCCL)" if dist.get_backend() == "nccl" else "")) if dist.is_initialized() else None,
                                               "allreduce_launches": reducer.collectives if reducer is not None else 0,
                                               "rng": "weights seed 42 on every rank, then seed %d = 42 + 1000 * (rank + 1) per rank" % rank_seed,
                                               "peak_memory_gb": torch.cuda.max_memory_allocated(dev) / 2 ** 30}}
        if unlabeled:
            res["config"]["images_per_step"] = {"labeled": B * world, "unlabeled": B * world}
        if comm is not None:
            res["comm"] = comm
        if ref_block is not None:
            res["reference_step" if ref_block["step"] == "reference" else "summed_step"] = ref_block
        if amp_block is not None:
            res["amp_step"] = amp_block
        # fusion hand-offs of the timed steps, per step (functional.FUSIONS): a hand-off that stopped working shows up as "missed"
        res["fusions_per_step"] = {k: {kk: vv / args.steps for kk, vv in v.items()} for k, v in Fn.fusion_report().items()}
        res["fusions_per_step"]["upsample_folded_launches"] = {k: v / (args.steps + args.warmup) for k, v in H.UPFOLD_TAKEN.items()}
        gflop_img = GFLOP_PER_IMG[args.workload]
        res["step_tflops"] = value * gflop_img / 1e3 / world
        roof = {"bound": "mfma", "peak": PEAK_FP32_MATRIX_TFLOPS, "unit": "TFLOP/s", "traffic": None,
                "kernel": "CLASS: every forward + data-gradient call of the C ABI's convolution entries, whatever kernel it runs -- "
                          "conv_igemm_kernel (direct / upsample-folded; its own figure: roofline.direct), wino_fused_kernel "
                          "(roofline.winograd_fused), grouped Winograd calls (roofline.winograd); v_mfma_f32_32x32x2_f32 throughout",
                "recompute": "frac = launches x avg_launch_executed_gflop / (launches x avg_launch_ms) / peak; every sub-block carries the "
                             "same three numbers for its own kernel (launches, avg_launch_ms, avg_launch_executed_gflop), to be checked "
                             "against that kernel's line of the rocprofv3 trace in profiles/"}
        if prof:
            agg = {}
            layers = {}
            hbm = {}
            # sampled timing (hipops.PROFILE_PERIOD): a launch that was not bracketed takes the mean duration of the bracketed
            # launches of its own (kind, geometry) -- every launch site is bracketed in one step out of `period`
            tsum = {}
            for kind, flops, s, e, tag, executed in prof:
                if s is not None:
                    a = tsum.setdefault((kind, tag), [0.0, 0])
                    a[0] += s.elapsed_time(e) * 1e-3
                    a[1] += 1
            n_timed = sum(a[1] for a in tsum.values())
            n_unsampled = sum(1 for r in prof if r[2] is None and (r[0], r[4]) not in tsum)
            res["kernel_timing"] = {"period": period, "launches": len(prof), "bracketed_launches": n_timed,
                                    "launches_without_a_sample": n_unsampled,
                                    "rule": "launch q of step i is bracketed by HIP events iff (q + i) % period == 0; the others take the "
                                            "mean of the bracketed launches of the same kind and geometry"}
            for kind, flops, s, e, tag, executed in prof:
                if s is not None:
                    t = s.elapsed_time(e) * 1e-3
                elif (kind, tag) in tsum:
                    t = tsum[(kind, tag)][0] / tsum[(kind, tag)][1]
                else:
                    continue
                if kind.startswith("hbm_"):       # `flops` holds the launch's algorithmic BYTES (operands once each)
                    a = hbm.setdefault(kind[4:], [0.0, 0.0, 0])
                    a[0] += flops
                    a[1] += t
                    a[2] += 1
                    continue
                a = agg.setdefault(kind, [0.0, 0.0, 0, 0.0])
                a[0] += flops
                a[1] += t
                a[2] += 1
                a[3] += executed
                b = layers.setdefault((kind, tag), [0.0, 0.0, 0, 0.0])
                b[0] += flops
                b[1] += t
                b[2] += 1
                b[3] += executed
            if os.environ.get("SEGSDE_BENCH_LAYERS"):
                with open(os.environ["SEGSDE_BENCH_LAYERS"], "w") as f:
                    f.write("# per-layer conv launches over %d timed steps (HIP events on the launch stream)\n" % args.steps)
                    for (kind, tag), v in sorted(layers.items(), key=lambda kv: -kv[1][1]):
                        f.write("%-11s %-45s n=%4d  %8.2f ms/step  %6.1f TF  %5.2f%% of step%s\n" % (
                            kind, tag, v[2], v[1] / args.steps * 1e3, v[0] / v[1] / 1e12, 100 * v[1] / dt,
                            "" if v[3] == v[0] else "  (executed %.1f TF)" % (v[3] / v[1] / 1e12)))
            # algorithmic bytes of a conv launch: its input(s), weights and output once each (fp32), from the geometry tag
            def tag_bytes(tag):
                m = re.match(r"(\d+)\+(\d+)->(\d+) k(\d+) s(\d+) d(\d+) (\d+)x(\d+)( up)?", tag)
                if not m:
                    return 0.0
                c0, c1, co, k, st, _, hh, ww = (int(x) for x in m.groups()[:8])
                up = 2 if m.group(9) else 1
                ho, wo = -(-hh // st), -(-ww // st)
                return 4.0 * (B * (hh // up) * (ww // up) * c0 + B * hh * ww * c1 + B * ho * wo * co + co * (c0 + c1) * k * k)
            abytes = sum(tag_bytes(tag) * v[2] for (kind, tag), v in layers.items() if kind in ("conv_fwd", "conv_dgrad"))
            fl = sum(agg[k][0] for k in ("conv_fwd", "conv_dgrad") if k in agg)
            tt = sum(agg[k][1] for k in ("conv_fwd", "conv_dgrad") if k in agg)
            nl = sum(agg[k][2] for k in ("conv_fwd", "conv_dgrad") if k in agg)
            ex = sum(agg[k][3] for k in ("conv_fwd", "conv_dgrad") if k in agg)
            # `achieved` / `frac` count the multiply-adds the launches really ISSUE to the matrix pipe (a roofline fraction:
            # <= 1 by construction).  The upsample-folded decoder convolutions (4 instead of 9 taps on the nearest-upsampled
            # channels), the dead-tap-skipping dilated windows and the Winograd routes (16 instead of 36 per 2x2 outputs) issue
            # fewer than the ALGORITHMIC count (2 * MAC of the convolution as the reference defines it), which
            # `algorithmic_achieved` / `algorithmic_frac` report -- a speed figure that can exceed 1, not a utilisation.
            roof.update(executed_achieved=ex / tt / 1e12, executed_frac=ex / tt / 1e12 / PEAK_FP32_MATRIX_TFLOPS,
                        executed_over_algorithmic=ex / fl,
                        frac_is="executed: the multiply-adds the launches issue / time / fp32 matrix peak (never above 1).  "
                                "algorithmic_frac divides the reference-defined 2 * MAC (SURVEY.md 8d) by the same time and peak; the "
                                "upsample-folded, dead-tap-skipping and Winograd routes issue fewer multiply-adds than that, so it can "
                                "exceed 1 and is a speed figure, not a utilisation")
            # the same split by route: the direct implicit-GEMM launches (what the matrix pipe's utilisation figure is about) and
            # the Winograd calls, whose time includes their HBM-side transforms while `executed` counts the position GEMMs only
            wl = [v for (kind, tag), v in layers.items() if kind in ("conv_fwd", "conv_dgrad") and (tag.endswith(" wino") or tag.endswith(" wino-fused"))]
            wfl = [v for (kind, tag), v in layers.items() if kind in ("conv_fwd", "conv_dgrad") and tag.endswith(" wino-fused")]
            if wfl:
                ff, ft, fn_, fx = (sum(v[i] for v in wfl) for i in range(4))
                roof["winograd_fused"] = {"kernel": "wino_fused_kernel (both transforms inside the kernel, 64..256 channels)",
                                          "launches": fn_, "ms_per_step": ft / args.steps * 1e3, "achieved": ff / ft / 1e12,
                                          "avg_launch_ms": ft / fn_ * 1e3, "avg_launch_executed_gflop": fx / fn_ / 1e9,
                                          "executed_achieved": fx / ft / 1e12,
                                          "executed_frac": fx / ft / 1e12 / PEAK_FP32_MATRIX_TFLOPS, "share_of_step": ft / dt}
            if wl:
                wf, wt, wn, wx = (sum(v[i] for v in wl) for i in range(4))
                roof["winograd"] = {"launches": wn, "ms_per_step": wt / args.steps * 1e3, "achieved": wf / wt / 1e12,
                                    "executed_achieved": wx / wt / 1e12, "share_of_step": wt / dt,
                                    "note": "both Winograd routes (grouped: one call = input transform + 16 position GEMMs + output "
                                            "transform; one-kernel: see winograd_fused); executed = 16/36 of the algorithmic "
                                            "multiply-adds over the whole call's time"}
                roof["direct"] = {"kernel": "conv_igemm_kernel (implicit-GEMM forward + data-gradient, direct and upsample-folded launches)",
                                  "launches": nl - wn, "achieved": (fl - wf) / (tt - wt) / 1e12,
                                  "executed_achieved": (ex - wx) / (tt - wt) / 1e12,
                                  "executed_frac": (ex - wx) / (tt - wt) / 1e12 / PEAK_FP32_MATRIX_TFLOPS,
                                  "ms_per_step": (tt - wt) / args.steps * 1e3, "avg_launch_ms": (tt - wt) / (nl - wn) * 1e3,
                                  "avg_launch_executed_gflop": (ex - wx) / (nl - wn) / 1e9, "share_of_step": (tt - wt) / dt}
            wgf = [v for (kind, tag), v in layers.items() if kind == "conv_wgrad" and tag.endswith(" wino-fused")]
            if wgf:
                gf_, gt_, gn_, gx_ = (sum(v[i] for v in wgf) for i in range(4))
                roof["winograd_wgrad_fused"] = {"kernel": "wino_wgrad_fused_kernel (csrc/winograd_wgrad.hip: both operand transforms inside the "
                                                          "kernel, LDS-DMA staging) + wino_wgrad_finish_kernel",
                                                "launches": gn_, "ms_per_step": gt_ / args.steps * 1e3, "achieved": gf_ / gt_ / 1e12,
                                                "avg_launch_ms": gt_ / gn_ * 1e3, "avg_launch_executed_gflop": gx_ / gn_ / 1e9,
                                                "executed_achieved": gx_ / gt_ / 1e12,
                                                "executed_frac": gx_ / gt_ / 1e12 / PEAK_FP32_MATRIX_TFLOPS, "share_of_step": gt_ / dt}
            roof.update(achieved=ex / tt / 1e12, frac=ex / tt / 1e12 / PEAK_FP32_MATRIX_TFLOPS,
                        algorithmic_achieved=fl / tt / 1e12, algorithmic_frac=fl / tt / 1e12 / PEAK_FP32_MATRIX_TFLOPS, launches=nl,
                        avg_launch_ms=tt / nl * 1e3, avg_launch_gflop=fl / nl / 1e9, avg_launch_executed_gflop=ex / nl / 1e9,
                        share_of_step=tt / dt, algorithmic_bytes_per_launch=abytes / nl)
            # the whole step against the same roof: every convolution launch's issued multiply-adds (forward, data-gradient,
            # weight gradient) over the step time, and the reference-defined count over the same time
            ex_step = sum(v[3] for v in agg.values()) / args.steps
            roof["step"] = {"executed_tflop": ex_step / 1e12, "executed_frac": ex_step / (dt / args.steps) / 1e12 / PEAK_FP32_MATRIX_TFLOPS,
                            "algorithmic_tflop": gflop_img * B / 1e3,
                            "algorithmic_frac": res["step_tflops"] / PEAK_FP32_MATRIX_TFLOPS}
            # HBM-side traffic per launch and matrix-pipe busy fraction: PMC counters cannot be read live, so these are the
            # committed results of the rocprofv3 --pmc passes of this very command (tools/gpu_pmc.sh -> profiles/traffic_rNN.json,
            # pmc_rNN_mfma_busy.txt; FETCH_SIZE x2 + WRITE_SIZE, the gfx950 rule of MI355X_MICROARCH.md).  A summary collected
            # from OTHER kernel sources (fingerprint mismatch) is refused: null + the reason, never a stale number.
            sha = csrc_sha256()
            wg_traffic = None
            tpath = newest_profile("traffic_r*.json")
            if args.workload == "cfg3" and tpath:
                rel = os.path.relpath(tpath, ROOT)
                try:
                    tj = json.load(open(tpath))
                    if tj.get("csrc_sha256") != sha:
                        roof["traffic_source"] = "%s refused: collected from other kernel sources (csrc_sha256 %s != %s); re-run tools/gpu_pmc.sh" % (
                            rel, str(tj.get("csrc_sha256"))[:12], sha[:12])
                    else:
                        # per LAUNCH of the C ABI, like `achieved` (one call = 1.27 kernels on average: class / border
                        # launches of the folded routes): the counters' bytes per step / this run's calls per step
                        kk = [k for k in tj["kernels"] if k.startswith("conv_igemm_kernel")][0]
                        sp = float(tj.get("steps_profiled", 3))
                        per_step = lambda e: e["bytes_per_launch"] * e["launches"] / sp
                        roof["traffic"] = per_step(tj["kernels"][kk]) / (nl / args.steps)
                        roof["traffic_per_kernel_launch"] = tj["kernels"][kk]["bytes_per_launch"]
                        roof["traffic_over_algorithmic"] = roof["traffic"] / (abytes / nl)
                        roof["traffic_gbs"] = per_step(tj["kernels"][kk]) / (tt / args.steps) / 1e9
                        roof["traffic_source"] = rel + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py)"
                        wg_traffic = per_step(tj["kernels"]["conv_wgrad_kernel"]) / (agg["conv_wgrad"][2] / args.steps) if "conv_wgrad" in agg else None
                        res["pmc_traffic_bytes_per_launch"] = {k: v["bytes_per_launch"] for k, v in tj["kernels"].items()}
                except Exception as ex:
                    roof["traffic_source"] = "%s unreadable: %r" % (rel, ex)
            mpath = newest_profile("pmc_r*_mfma_busy.txt")
            if args.workload == "cfg3" and mpath:
                rel = os.path.relpath(mpath, ROOT)
                lines = open(mpath).read().splitlines()
                if not any(l.startswith("# csrc_sha256 " + sha) for l in lines):
                    roof["mfma_busy_source"] = rel + " refused: collected from other kernel sources; re-run tools/gpu_pmc.sh"
                else:
                    busy = {}
                    for line in lines:
                        if line.startswith("conv_") or line.startswith("wino_fused") or line.startswith("wino_wgrad_fused"):
                            f = line.split()
                            busy[line[:72].strip()] = float(f[-1])
                    roof["mfma_busy"] = busy
                    roof["mfma_busy_source"] = rel + " (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE)"
            roof["measured_peaks"] = MEASURED_PEAKS
            roof["frac_of_measured_peak"] = roof["achieved"] / MEASURED_PEAKS["mfma_f32_tflops"]      # executed, like frac
            if "conv_wgrad" in agg:
                wb = sum(tag_bytes(tag) * v[2] for (kind, tag), v in layers.items() if kind == "conv_wgrad")
                res["wgrad"] = {"kernel": "conv_wgrad_kernel (pixel-reduction GEMM, split + deterministic reduce)",
                                "achieved": agg["conv_wgrad"][3] / agg["conv_wgrad"][1] / 1e12,
                                "frac": agg["conv_wgrad"][3] / agg["conv_wgrad"][1] / 1e12 / PEAK_FP32_MATRIX_TFLOPS,
                                "algorithmic_achieved": agg["conv_wgrad"][0] / agg["conv_wgrad"][1] / 1e12,
                                "algorithmic_frac": agg["conv_wgrad"][0] / agg["conv_wgrad"][1] / 1e12 / PEAK_FP32_MATRIX_TFLOPS,
                                "executed_achieved": agg["conv_wgrad"][3] / agg["conv_wgrad"][1] / 1e12,
                                "executed_frac": agg["conv_wgrad"][3] / agg["conv_wgrad"][1] / 1e12 / PEAK_FP32_MATRIX_TFLOPS,
                                "launches": agg["conv_wgrad"][2],
                                "algorithmic_bytes_per_launch": wb / agg["conv_wgrad"][2], "traffic": wg_traffic}
            res["kernels"] = {k: {"tflops": v[0] / v[1] / 1e12, "executed_tflops": v[3] / v[1] / 1e12, "seconds": v[1],
                                  "launches": v[2], "share_of_step": v[1] / dt} for k, v in agg.items()}
            # every HBM-bound kernel family against the HBM roof -- algorithmic bytes (each operand of the
            # launch once) / HIP-event time, as a fraction of the nominal 8 TB/s and of the measured 6.36 TB/s
            res["hbm_kernels"] = {k: {"algorithmic_mb_per_step": v[0] / args.steps / 1e6, "ms_per_step": v[1] / args.steps * 1e3,
                                      "launches_per_step": v[2] / args.steps, "achieved_gbs": v[0] / v[1] / 1e9,
                                      "frac_of_8tbs": v[0] / v[1] / 8e12,
                                      "frac_of_measured_hbm": v[0] / v[1] / (MEASURED_PEAKS["hbm_gbs"] * 1e9),
                                      "share_of_step": v[1] / dt}
                                  for k, v in sorted(hbm.items(), key=lambda kv: -kv[1][1])}
            # The photometric family is bound by vector-instruction issue, not by HBM (DESIGN.md 3.3 "Round 4"): next to the HBM
            # fractions, the share of the VALU issue slots its launches use -- SQ_INSTS_VALU of the committed PMC pass of this
            # command x 4 cycles per wave64 instruction / (256 CUs x 4 SIMDs x 2.4 GHz x the live HIP-event time of the launch)
            wpath = newest_profile("pmc_r*_sq_waits.txt")
            if args.workload == "cfg3" and wpath:
                lines = open(wpath).read().splitlines()
                if any(l.startswith("# csrc_sha256 " + sha) for l in lines):
                    hdr = next((l.split() for l in lines if l.startswith("kernel ")), None)
                    fam = {"photometric_bwd": "photometric_bwd2_kernel<true>", "photometric_fwd": "photometric_fwd2_kernel<false>",
                           "photometric_identity": "photometric_fwd2_kernel<true>", "warp_fwd": "warp_fwd_kernel"}
                    for k, kname in fam.items():
                        row = next((l.split() for l in lines if l.startswith(kname + " ")), None)
                        if not (hdr and row and k in res["hbm_kernels"] and "SQ_INSTS_VALU" in hdr):
                            continue
                        ncol = len(hdr) - 1                       # numeric columns: launches + the counters
                        nums = row[-ncol:]
                        insts = float(nums[hdr.index("SQ_INSTS_VALU") - 1]) / float(nums[0])
                        e = res["hbm_kernels"][k]
                        sec = e["ms_per_step"] / e["launches_per_step"] * 1e-3
                        e["bound"] = "valu"
                        e["valu_wave_instructions_per_launch"] = insts
                        e["valu_issue_frac"] = insts * 4.0 / (256 * 4 * 2.4e9 * sec)
                        e["valu_source"] = os.path.relpath(wpath, ROOT)
        else:
            # no kernel timing in this run: only the step-level algorithmic figure exists (a speed figure, see frac_is above)
            roof.update(achieved=None, frac=None, algorithmic_achieved=res["step_tflops"],
                        algorithmic_frac=res["step_tflops"] / PEAK_FP32_MATRIX_TFLOPS)
        res["roofline"] = roof
        print("gpu result:", json.dumps(res), file=sys.stderr, flush=True)
        if not args.no_cpu_baseline and world == 1:
            # separate process + hard time limit: the baseline is reported, never allowed to stall the GPU number
            import subprocess
            try:
                cp = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--workload",
                                     args.workload], capture_output=True, text=True, timeout=args.cpu_baseline_timeout,
                                    env=dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES=""))
                res["cpu_baseline"] = json.loads(cp.stdout.strip().splitlines()[-1])
            except Exception as ex:
                res["cpu_baseline"] = {"value": None, "unit": "img/s", "cores": os.cpu_count(), "kind": "port",
                                       "sample": "not measured: %r" % (ex,)}
    # RCCL prints a version banner through C stdio, which (stdout being a pipe) would surface at process exit, AFTER the
    # result: tear the process group down and flush C stdio on every rank first, so that rank 0's JSON line is the last
    # line on stdout
    if world > 1 or force_reducer:
        dist.barrier()
        dist.destroy_process_group()
    _flush_c_stdio()
    if rank == 0:
        if world > 1:
            time.sleep(0.5)          # the other ranks' flushes
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
