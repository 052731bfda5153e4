#!/usr/bin/env python
"""Record ``inspect.signature`` of every constructor / function of the drop-in boundary as the REFERENCE
defines it (imported from /root/reference; build container only) -> tests/golden/signatures.json.
tests/test_abi_and_hygiene.py::test_signatures_match_reference compares the package's own objects with it on every run."""
import importlib
import inspect
import json
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, HERE)
import _tv_standin  # noqa: E402

_tv_standin.install()
sys.path.insert(0, REF)
_loader = types.ModuleType("loader")
_loader.__path__ = [os.path.join(REF, "loader")]
sys.modules["loader"] = _loader
sys.modules.setdefault("kornia", types.ModuleType("kornia"))

# module -> names; "Class.method" records a method (``self`` included, as inspect prints it)
SURFACE = {
    "models": ["get_model"],
    "models.joint_segmentation_depth": ["joint_segmentation_depth", "JointSegmentationMonodepth.forward",
                                        "JointSegmentationMonodepth.predict_test_disp"],
    "models.resnet_encoder": ["ResnetEncoder", "ResnetEncoder.forward"],
    "models.depth_decoder": ["DepthDecoder", "DepthDecoder.forward"],
    "models.joint_segmentation_depth_decoder": ["JointSegDepthDecoder", "JointSegDepthDecoder.forward", "PAD", "PAD.forward",
                                                "PAD.depth_params", "PAD.segmentation_params"],
    "models.pose_decoder": ["PoseDecoder", "PoseDecoder.forward"],
    "models.model_parts": ["ASPP", "ASPP.forward", "SelfAttention", "SelfAttention.forward"],
    "models.monodepth_layers": ["disp_to_depth", "transformation_from_parameters", "get_translation_matrix",
                                "rot_from_axisangle", "ConvBlock", "ConvBlock.forward", "Conv3x3", "Conv3x3.forward", "BackprojectDepth", "BackprojectDepth.forward",
                                "Project3D", "Project3D.forward", "upsample", "get_smooth_loss", "SSIM", "SSIM.forward"],
    "loss": ["get_segmentation_loss_function", "get_monodepth_loss"],
    "loss.loss": ["cross_entropy2d", "berhu"],
    "loss.monodepth_loss": ["MonodepthLoss", "MonodepthLoss.generate_depth_test_pred", "MonodepthLoss.generate_images_pred",
                            "MonodepthLoss.compute_losses"],
    "loader.transformsgpu": ["mix", "color_jitter", "gaussian_blur"],
    "loader.transformmasks": ["generate_class_mask", "generate_depth_mask"],
    "evaluation.metrics": ["runningScore", "runningScore.update", "runningScore.get_scores", "runningScore.reset"],
}


def signature_of(module, name):
    obj = module
    for part in name.split("."):
        obj = getattr(obj, part)
    return str(inspect.signature(obj))


if __name__ == "__main__":
    out = {}
    for mod, names in SURFACE.items():
        m = importlib.import_module(mod)
        assert os.path.realpath(m.__file__).startswith(REF), m.__file__
        for n in names:
            out["%s:%s" % (mod, n)] = signature_of(m, n)
    with open(os.path.join(HERE, "signatures.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote %d signatures" % len(out))
