"""CPU: run the real kernel sources through the fiber interpreter and compare with plain PyTorch."""
import os
import sys

import pytest
import torch

import emu
import kernel_cases as KC


@pytest.fixture(scope="module", autouse=True)
def _emu():
    if torch.cuda.is_available():
        pytest.skip("GPU present: the -m gpu suite exercises the real library instead")
    emu.install()


@pytest.mark.parametrize("case", KC.CONV_CASES, ids=[c[0] for c in KC.CONV_CASES])
def test_conv(case):
    KC.run_conv_case(case, "cpu")


@pytest.mark.parametrize("cfg", [dict(C=24, act="relu", residual=True, train=True),
                                 dict(C=70, act="none", residual=False, train=True),
                                 dict(C=8, act="elu", residual=False, train=True),
                                 dict(C=16, act="relu", residual=True, train=False),
                                 dict(C=32, act="relu", residual=False, train=True),
                                 dict(C=12, act="relu", residual=False, train=False)])
def test_batchnorm(cfg):
    KC.run_bn_case("cpu", **cfg)


def test_dropout():
    KC.run_dropout_case("cpu")


def test_misc_kernels():
    KC.run_misc_cases("cpu")


def test_pose(golden):
    KC.run_pose_case("cpu", golden)


def test_segmix(golden):
    KC.run_segmix_cases("cpu", golden)


def test_trainer_rows(golden):
    KC.run_trainer_cases("cpu", golden)


def test_mix_use_gt_vs_reference(golden):
    KC.run_mix_use_gt_cases("cpu", golden)


def test_dead_tap_rows_are_skipped():
    KC.run_dead_tap_rows_case("cpu")


def test_network_stems():
    KC.run_stem_cases("cpu")


def test_upsample_folded_convolutions():
    KC.run_upfold_cases("cpu")


def test_upsample_folded_random_geometries():
    KC.run_upfold_random("cpu", n=6, seed=3)


def test_depthmix_teacher_kernels():
    KC.run_depthmix_teacher_cases("cpu")


def test_validation_tail_kernels(golden):
    KC.run_valtail_kernel_cases("cpu", golden)


def test_conv_dgrad_fused_activation_backward():
    KC.run_conv_actgrad_cases("cpu")


def test_fused_photometric_vs_stage_kernels():
    KC.run_fused_photometric_vs_stage("cpu")


@pytest.mark.parametrize("knobs", [{"SEGSDE_PHOTO_SPLIT": "0"}, {"SEGSDE_PHOTO_PACKED": "0"}], ids=["unsplit_walkers", "round3_kernels"])
def test_fused_photometric_knob_variants(knobs):
    """the A/B variants of the photometric kernels (knobs are read once per process: a child process each)"""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import emu; emu.install(); import kernel_cases as KC; KC.run_fused_photometric_vs_stage(%r); print('VARIANT OK')" % (
        os.path.dirname(here), here, "cpu")
    cp = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **knobs), capture_output=True, text=True, timeout=900)
    assert cp.returncode == 0 and "VARIANT OK" in cp.stdout, cp.stdout[-2000:] + cp.stderr[-2000:]


def test_strong_transform_jitter_blur():
    KC.run_augment_cases("cpu")


def test_validation_metric(golden):
    KC.run_metric_cases("cpu", golden)


def test_residual_gradient_fusion():
    KC.run_residual_fusion_case("cpu")


def test_loss_kernels(golden):
    KC.run_loss_kernel_cases("cpu", golden)


def test_monodepth_layer_callables(golden):
    KC.run_monodepth_layer_callables("cpu", golden)


def test_torch_ops_namespace(golden):
    KC.run_torch_ops("cpu", golden)


def test_jitter_blur_properties():
    KC.run_jitter_blur_properties("cpu")


def test_f16_operand_convolutions():
    KC.run_f16_operand_convolutions("cpu")


def test_winograd_route():
    KC.run_winograd_cases("cpu")


def test_winograd_fused_kernel():
    KC.run_winograd_fused_cases("cpu")



@pytest.mark.parametrize("variant", ["0", "1", "2"])
def test_winograd_fused_wgrad_kernel(variant):
    """csrc/winograd_wgrad.hip under the interpreter, one process per staging variant (the knob is read once per process)"""
    import subprocess
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import emu; emu.install(); import kernel_cases as KC; "
            "KC.run_winograd_fused_wgrad_cases('cpu'); print('wgrad-ok')" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                            os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=1200,
                       env=dict(os.environ, SEGSDE_WGRAD_FUSED_VAR=variant))
    assert r.returncode == 0 and "wgrad-ok" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_dropout_layer():
    KC.run_dropout_case("cpu")


def test_bn_stats_from_partials_plans():
    KC.run_bn_partials_case("cpu")
