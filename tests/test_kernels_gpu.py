"""GPU (-m gpu): every kernel through the C ABI of the real libsegsde_hip.so vs plain PyTorch fp32 / golden vectors."""
import os

import pytest
import torch

import kernel_cases as KC
from improving_segmentation_with_selfsupervised_depth_amd import _lib

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _real_lib():
    assert torch.cuda.is_available()
    L = _lib.lib()   # raises if libsegsde_hip.so is missing: there is no fallback
    assert not _lib.HOST_POINTERS_OK
    maps = open("/proc/self/maps").read()
    assert "libsegsde_hip.so" in maps, "native library not loaded"
    return L


@pytest.mark.parametrize("case", KC.CONV_CASES, ids=[c[0] for c in KC.CONV_CASES])
def test_conv(case):
    KC.run_conv_case(case, "cuda")


def test_bn_random_shapes():
    """BatchNorm statistics / apply / backward at random sizes (channel counts that are no multiple of 4 or 64, pixel counts
    that leave partial row blocks, tens of thousands of rows per channel) against torch autograd on the CPU.  No ReLU here:
    an output within rounding distance of zero may get the other sign from two correct implementations, and the mask of
    the gradient check is taken from ours (the fixed-size cases cover ReLU)."""
    import numpy as np
    rng = np.random.RandomState(11)
    for i in range(24):
        C = int(rng.choice([3, 19, 24, 64, 100, 128, 256, 513, 1024]))
        shape = (int(rng.randint(1, 5)), int(rng.randint(3, 70)), int(rng.randint(3, 90)))
        KC.run_bn_case("cuda", C=C, act=str(rng.choice(["none", "elu"])), residual=bool(rng.rand() < 0.5),
                       train=bool(rng.rand() < 0.8), seed=i, shape=shape)


def test_conv_random_geometries():
    """60 random geometries (fixed seed; tools/stress_conv.py runs more): forward, fused statistics, both data gradients and
    the weight gradient against torch autograd on the CPU"""
    import numpy as np
    rng = np.random.RandomState(2024)
    for i in range(60):
        case = KC.random_conv_case(rng, i)
        KC.run_conv_case(case, "cuda", seed=i)
        KC.run_dgrad_epilogue_variants(case, "cuda", seed=i)


BIG_CONV = [
    ("big_refl_up_cat", 2, 64, 96, 64, 32, True, 64, 3, 1, 1, 1, True, True, "elu"),
    ("big_dil6", 2, 32, 64, 128, 0, False, 256, 3, 1, 6, 6, False, False, "none"),
    ("big_1x1", 2, 32, 64, 512, 0, False, 256, 1, 1, 1, 0, False, False, "none"),
    ("big_3x3_s2", 2, 64, 96, 64, 0, False, 128, 3, 2, 1, 1, False, False, "none"),
    ("big_7x7", 2, 64, 128, 3, 0, False, 64, 7, 2, 1, 3, False, False, "none"),
    ("big_cout19", 2, 64, 128, 64, 0, False, 19, 1, 1, 1, 0, False, True, "none"),
    ("big_disp", 2, 64, 128, 64, 0, False, 1, 3, 1, 1, 1, True, True, "sigmoid"),
    # many pixel tiles with Cout = 64 (128x64 tile, three workgroups per CU)
    ("big_n64", 2, 256, 256, 64, 0, False, 64, 3, 1, 1, 1, False, True, "relu"),
    ("big_n64_refl_up", 2, 256, 256, 64, 0, True, 64, 3, 1, 1, 1, True, True, "elu"),
]


@pytest.mark.parametrize("case", BIG_CONV, ids=[c[0] for c in BIG_CONV])
def test_conv_big(case):
    KC.run_conv_case(case, "cuda", seed=1)


@pytest.mark.parametrize("cfg", [dict(C=24, act="relu", residual=True, train=True),
                                 dict(C=70, act="none", residual=False, train=True),
                                 dict(C=8, act="elu", residual=False, train=True),
                                 dict(C=16, act="relu", residual=True, train=False),
                                 dict(C=32, act="relu", residual=False, train=True),
                                 dict(C=12, act="relu", residual=False, train=False)])
def test_batchnorm(cfg):
    KC.run_bn_case("cuda", **cfg)


def test_dropout():
    KC.run_dropout_case("cuda")


def test_misc_kernels():
    KC.run_misc_cases("cuda")


def test_pose(golden):
    KC.run_pose_case("cuda", golden)


def test_segmix(golden):
    KC.run_segmix_cases("cuda", golden)


def test_trainer_rows(golden):
    KC.run_trainer_cases("cuda", golden)


def test_mix_use_gt_vs_reference(golden):
    KC.run_mix_use_gt_cases("cuda", golden)


def test_dead_tap_rows_are_skipped():
    KC.run_dead_tap_rows_case("cuda")


def test_network_stems():
    KC.run_stem_cases("cuda")


def test_upsample_folded_convolutions():
    KC.run_upfold_cases("cuda")


def test_upsample_folded_random_geometries():
    KC.run_upfold_random("cuda", n=40)


def test_depthmix_teacher_kernels():
    KC.run_depthmix_teacher_cases("cuda")


def test_validation_tail_kernels(golden):
    KC.run_valtail_kernel_cases("cuda", golden)


def test_conv_dgrad_fused_activation_backward():
    KC.run_conv_actgrad_cases("cuda")


def test_fused_photometric_vs_stage_kernels():
    KC.run_fused_photometric_vs_stage("cuda")


@pytest.mark.parametrize("knobs", [{"SEGSDE_PHOTO_SPLIT": "0"}, {"SEGSDE_PHOTO_PACKED": "0"}], ids=["unsplit_walkers", "round3_kernels"])
def test_fused_photometric_knob_variants(knobs):
    """the A/B variants of the photometric kernels (knobs are read once per process: a child process each)"""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import kernel_cases as KC; KC.run_fused_photometric_vs_stage(%r); print('VARIANT OK')" % (
        os.path.dirname(here), here, "cuda")
    cp = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **knobs), capture_output=True, text=True, timeout=900)
    assert cp.returncode == 0 and "VARIANT OK" in cp.stdout, cp.stdout[-2000:] + cp.stderr[-2000:]


def test_strong_transform_jitter_blur():
    KC.run_augment_cases("cuda")


def test_validation_metric(golden):
    KC.run_metric_cases("cuda", golden)


def test_residual_gradient_fusion():
    KC.run_residual_fusion_case("cuda")


def test_loss_kernels(golden):
    KC.run_loss_kernel_cases("cuda", golden)


def test_cpu_tensor_is_rejected():
    from improving_segmentation_with_selfsupervised_depth_amd import hipops as H
    with pytest.raises(RuntimeError):
        H.colsum(torch.zeros(4, 4))


def test_monodepth_layer_callables(golden):
    KC.run_monodepth_layer_callables("cuda", golden)


def test_torch_ops_namespace(golden):
    KC.run_torch_ops("cuda", golden)


def test_jitter_blur_properties():
    KC.run_jitter_blur_properties("cuda")


def test_f16_operand_convolutions():
    KC.run_f16_operand_convolutions("cuda")


def test_winograd_route():
    KC.run_winograd_cases("cuda")


def test_winograd_fused_kernel():
    KC.run_winograd_fused_cases("cuda")



def test_winograd_fused_wgrad_kernel():
    KC.run_winograd_fused_wgrad_cases("cuda")


def test_dropout_layer():
    KC.run_dropout_case("cuda")


def test_bn_stats_from_partials_plans():
    KC.run_bn_partials_case("cuda")
