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
        import os
        self.defer_trunk_backward = os.environ.get("SEGSDE_DEFER_TRUNK", "0") not in ("", "0")

    def _run_pose_nets(self, frames):
        """channel-concatenated frames -> (axisangle, translation), each [B, n_poses, 1, 3]"""
        return self.models["pose"]([self.models["pose_encoder"](frames)])

    def predict_poses(self, inputs, features):
        """reference :20-70.  Two regimes: "pairs" -- one pass of the pose networks per temporal source frame over
        (earlier frame, later frame), the matrix inverted for frames before the target --, or every frame in one pass with one pose
        per source frame.  The stereo frame "s" has no predicted pose (MonodepthLoss takes ``inputs["stereo_T"]`` for it)."""
        key = "color_full_aug" if self.provide_uncropped_for_pose else "color_aug"
        frame = lambda f: inputs[key, f, 0]                                      # noqa: E731
        temporal = [(slot, f) for slot, f in enumerate(self.frame_ids[1:]) if f != "s"]
        outputs = {}

        def emit(f, axisangle, translation, column, invert):
            outputs[("axisangle", 0, f)] = axisangle
            outputs[("translation", 0, f)] = translation
            outputs[("cam_T_cam", 0, f)] = Fn.PoseMatrixFn.apply(axisangle[:, column:column + 1], translation[:, column:column + 1], invert)

        if self.num_pose_frames == 2:
            for _, f in temporal:
                first, second = (f, 0) if f < 0 else (0, f)                      # always in temporal order
                emit(f, *self._run_pose_nets(torch.cat([frame(first), frame(second)], 1)), 0, f < 0)
        else:
            both = self._run_pose_nets(torch.cat([frame(f) for f in self.frame_ids if f != "s"], 1))
            for slot, f in temporal:
                emit(f, *both, slot, False)
        return outputs

    def predict_test_disp(self, x):
        return self.models["depth"](self.models["encoder"](x[("color", 0, 0)]))

    @Fn.fp32_region
    def forward(self, x):
        from ..loss.monodepth_loss import LazyOutputs
        outputs, inputs = LazyOutputs(), x     # a dict; MonodepthLoss registers its API-visible grids / depths as lazy entries
        self.models["encoder"].defer_backward = bool(self.defer_trunk_backward)
        if "mtl_decoder" in self.models:     # PAD's two decoders cross half-way up: a second gate there (PAD.forward)
            self.models["mtl_decoder"].defer_backward = bool(self.defer_trunk_backward)
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
    # which sub-models a freeze flag covers (reference :158-180).  `must`: the reference indexes these without a test, so asking to
    # freeze a sub-model the configuration does not build (the depth decoder of a PAD model, the pose decoder with disable_pose)
    # is a KeyError there and here; the pose encoder and the segmentation decoder are frozen only if present
    frozen = [(freeze_backbone, "encoder", True), (freeze_depth and not disable_monodepth, "depth", True),
              (freeze_pose and not disable_monodepth, "pose_encoder", False), (freeze_pose and not disable_monodepth, "pose", True),
              (freeze_segmentation, "segmentation", False)]
    for flag, name, must in frozen:
        if flag and (must or name in models):
            models[name].requires_grad_(False)
    return JointSegmentationMonodepth(models, frame_ids, use_pose_net, num_pose_frames, provide_uncropped_for_pose)
