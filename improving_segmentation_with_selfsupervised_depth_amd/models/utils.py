"""Mirror of models/utils.py builders (get_resnet_backbone, get_depth_decoder, get_posenet, _get_layer).
Checkpoint download paths of the reference (Google Drive, torchvision model zoo) need a network and raise here:
load weights with ``load_state_dict`` -- key names and shapes are identical to the reference's."""
import re

import torch
from torch import nn

from .depth_decoder import DepthDecoder
from .pose_decoder import PoseDecoder
from .resnet_encoder import ResnetEncoder


def _device():
    return torch.device("cuda" if torch.cuda.is_available() else "cpu")


def get_resnet_backbone(backbone_name, backbone_pretraining="none", replace_stride_with_dilation=None,
                        use_intermediate_layer_getter=False, num_input_images=1):
    if backbone_name not in ["resnet18", "resnet50", "resnet101"]:
        raise NotImplementedError
    n_res = int(re.match(r"([a-z]+)([0-9]+)", backbone_name, re.I).groups()[-1])
    if backbone_pretraining != "none":
        raise RuntimeError("backbone_pretraining=%r downloads weights in the reference (models/utils.py:30-42); "
                           "no network here -- use 'none' and load a state_dict" % (backbone_pretraining,))
    if use_intermediate_layer_getter:
        raise NotImplementedError("IntermediateLayerGetter is not on the training path")
    return ResnetEncoder(n_res, False, num_input_images=num_input_images,
                         replace_stride_with_dilation=replace_stride_with_dilation)


def get_depth_decoder(depth_pretraining, num_ch_enc, scales=range(4), **kwargs):
    dec = DepthDecoder(num_ch_enc, scales, **kwargs).to(_device())
    if depth_pretraining not in (None, "none"):
        raise RuntimeError("depth_pretraining=%r needs a downloaded checkpoint (models/utils.py:64-71)" % (depth_pretraining,))
    return dec


def get_posenet(backbone_name, backbone_pretraining, pose_pretraining, num_pose_frames):
    if "mono" in str(pose_pretraining):
        raise RuntimeError("pose_pretraining=%r needs a downloaded checkpoint (models/utils.py:87-95)" % (pose_pretraining,))
    models = {}
    models["pose_encoder"] = get_resnet_backbone(backbone_name, "none", num_input_images=num_pose_frames)
    models["pose"] = PoseDecoder(models["pose_encoder"].num_ch_enc, num_input_features=1, num_frames_to_predict_for=2)
    return models


def _get_layer(encoder, decoder, layer):
    return encoder[layer] if layer <= 4 else decoder[("upconv", 9 - layer)]
