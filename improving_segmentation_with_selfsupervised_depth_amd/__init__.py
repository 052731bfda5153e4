"""MI355X-native (gfx950) implementation of the joint segmentation + self-supervised depth training hot path
of lhoyer/improving_segmentation_with_selfsupervised_depth, behind the reference's Python API.

Sub-packages mirror the reference's import surface (``models``, ``loss``, ``loader``); all arithmetic runs in
hand-written HIP kernels (``csrc/``) reached through the C ABI of ``include/segsde_hip.h``.
"""
__version__ = "0.1.0"

from . import torch_ops  # noqa: E402,F401  (registers torch.ops.segsde.*; loads no library until an operator runs)
