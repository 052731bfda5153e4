"""Mirror of models/joint_segmentation_depth.py: JointSegmentationMonodepth + the ``joint_segmentation_depth``
factory with the reference's exact keyword signature (joint_segmentation_depth.py:116-123)."""
import torch
from torch import nn

from .. import functional as Fn
from .joint_segmentation_depth_decoder import JointSegDepthDecoder, PAD
from .utils import get_depth_decoder, get_posenet, get_resnet_backbone


class JointSegmentationMonodepth(nn.Module):
    def __init__(self, models, frame_ids, use_pose_net, num_pose_frames, provide_uncropped_for_pose):
        super().__init__()
        self.frame_ids = frame_ids
        self.use_pose_net = use_pose_net
        self.num_pose_frames = num_pose_frames
        self.provide_uncropped_for_pose = provide_uncropped_for_pose
        self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.models = nn.ModuleDict(models)
        # how many decoders read the encoder's features as skip sources: the encoder then hands every feature out as views of one
        # gradient collector (resnet_encoder.ResnetEncoder.forward_nhwc, Fn.fan_feature)
        from .depth_decoder import DepthDecoder
        # (only the decoders that fetch their view: identity skip projections, depth_decoder.forward_nhwc)
        dec = [m for k, v in self.models.items() if k in ("depth", "segmentation", "mtl_decoder")
               for m in v.modules() if isinstance(m, DepthDecoder) and m.use_skips and m.takes_fan_views()]
        if "encoder" in self.models:
            self.models["encoder"].skip_consumers = len(dec)
        # opt-in: one encoder backward per forward however many ``backward()`` calls the training step makes on its losses
        # (train.py:486,499,510 make up to three); see functional.defer_trunk for the contract -- the forward's last backward must
        # release its graph.  trainer.train_step and the INTEGRATION.md shim switch it on per configuration.
        self.defer_trunk_backward = False

    def predict_poses(self, inputs, features):
        """reference :20-70"""
        outputs = {}
        key = "color_full_aug" if self.provide_uncropped_for_pose else "color_aug"
        if self.num_pose_frames == 2:
            pose_feats = {f_i: inputs[key, f_i, 0] for f_i in self.frame_ids}
            for f_i in self.frame_ids[1:]:
                if f_i == "s":
                    continue
                pair = [pose_feats[f_i], pose_feats[0]] if f_i < 0 else [pose_feats[0], pose_feats[f_i]]
                pose_inputs = [self.models["pose_encoder"](torch.cat(pair, 1))]
                axisangle, translation = self.models["pose"](pose_inputs)
                outputs[("axisangle", 0, f_i)] = axisangle
                outputs[("translation", 0, f_i)] = translation
                outputs[("cam_T_cam", 0, f_i)] = Fn.PoseMatrixFn.apply(axisangle, translation, f_i < 0)
        else:
            # all frames go through the pose network together and all poses are predicted at once (reference :52-68)
            pose_inputs = torch.cat([inputs[(key, i, 0)] for i in self.frame_ids if i != "s"], 1)
            pose_inputs = [self.models["pose_encoder"](pose_inputs)]
            axisangle, translation = self.models["pose"](pose_inputs)
            for i, f_i in enumerate(self.frame_ids[1:]):
                if f_i != "s":
                    outputs[("axisangle", 0, f_i)] = axisangle
                    outputs[("translation", 0, f_i)] = translation
                    outputs[("cam_T_cam", 0, f_i)] = Fn.PoseMatrixFn.apply(axisangle[:, i:i + 1], translation[:, i:i + 1], False)
        return outputs

    def predict_test_disp(self, x):
        return self.models["depth"](self.models["encoder"](x[("color", 0, 0)]))

    @Fn.fp32_region
    def forward(self, x):
        from ..loss.monodepth_loss import LazyOutputs
        outputs, inputs = LazyOutputs(), x     # a dict; MonodepthLoss registers its API-visible grids / depths as lazy entries
        self.models["encoder"].defer_backward = bool(self.defer_trunk_backward)
        features = self.models["encoder"](inputs["color_aug", 0, 0])
        outputs["bottleneck"] = features[-1]
        if "mtl_decoder" in self.models:
            outputs.update(self.models["mtl_decoder"](features))
        else:
            if "depth" in self.models:
                outputs.update(self.models["depth"](features))
            if "segmentation" in self.models:
                outputs["semantics"] = self.models["segmentation"](features)
        if "imnet_encoder" in self.models:
            outputs["encoder_features"] = features[-1]
            self.models["imnet_encoder"].eval()
            with torch.no_grad():
                outputs["imnet_features"] = self.models["imnet_encoder"](inputs["color_aug", 0, 0])[-1].detach()
        if self.use_pose_net:
            outputs.update(self.predict_poses(inputs, features))
        return outputs


def get_segmentation_network(segmentation_name, num_ch_enc, segmentation_size, num_classes, segmentation_args,
                             depth_args):
    model_map = {"joint_seg_depth_dec": JointSegDepthDecoder, "mtl_pad": PAD}
    num_ch_dec = depth_args.get("num_ch_dec", [16, 32, 64, 128, 256])
    return model_map[segmentation_name](num_ch_enc, num_ch_dec, num_classes, **segmentation_args, depth_args=depth_args)


def joint_segmentation_depth(name, backbone_name, segmentation_name, segmentation_args, num_classes,
                             backbone_pretraining, depth_pretraining, pose_pretraining, freeze_backbone,
                             freeze_segmentation, freeze_depth, freeze_pose, replace_stride_with_dilation, frame_ids,
                             num_scales, pose_model_input, provide_uncropped_for_pose, height, width, depth_args,
                             disable_monodepth, enable_imnet_encoder, disable_pose, imnet_encoder_dilation=True,
                             **kwargs):
    num_input_frames = len(frame_ids)
    num_pose_frames = 2 if pose_model_input == "pairs" else num_input_frames
    assert frame_ids[0] == 0
    use_pose_net = not (tuple(frame_ids) == (0, "s")) and not disable_pose
    models = {}
    models["encoder"] = get_resnet_backbone(backbone_name, backbone_pretraining, replace_stride_with_dilation)
    num_ch_enc = models["encoder"].num_ch_enc
    if enable_imnet_encoder:
        models["imnet_encoder"] = get_resnet_backbone(
            backbone_name, "imnet", replace_stride_with_dilation if imnet_encoder_dilation else None)
        for p in models["imnet_encoder"].parameters():
            p.requires_grad = False
    if use_pose_net and not disable_monodepth:
        models.update(get_posenet("resnet18", backbone_pretraining, pose_pretraining, num_pose_frames))
    if segmentation_name in ["mtl_pad"]:
        models["mtl_decoder"] = get_segmentation_network(segmentation_name, num_ch_enc, (height, width), num_classes,
                                                         segmentation_args, depth_args)
    else:
        if not disable_monodepth:
            models["depth"] = get_depth_decoder(depth_pretraining, num_ch_enc, range(num_scales), **depth_args)
        if segmentation_name is not None:
            models["segmentation"] = get_segmentation_network(segmentation_name, num_ch_enc, (height, width),
                                                              num_classes, segmentation_args, depth_args)
    if freeze_backbone:
        for p in models["encoder"].parameters():
            p.requires_grad = False
    if not disable_monodepth and freeze_depth:
        for p in models["depth"].parameters():
            p.requires_grad = False
    if not disable_monodepth and freeze_pose:
        if "pose_encoder" in models:
            for p in models["pose_encoder"].parameters():
                p.requires_grad = False
        for p in models["pose"].parameters():
            p.requires_grad = False
    if "segmentation" in models and freeze_segmentation:
        for p in models["segmentation"].parameters():
            p.requires_grad = False
    return JointSegmentationMonodepth(models, frame_ids, use_pose_net, num_pose_frames, provide_uncropped_for_pose)
