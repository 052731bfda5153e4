"""Stand-in models for the EMA-update vectors (tests/golden/trainer.npz): a ``.models`` ModuleDict like the reference's
JointSegDepth, parameters filled from an integer formula so the test can rebuild the exact inputs anywhere."""
import torch


def _fill(shape, seed):
    n = 1
    for s in shape:
        n *= s
    i = torch.arange(n, dtype=torch.int64)
    v = ((i * 7919 + seed * 104729) % 2003).to(torch.float32) / 2003.0 - 0.5
    return v.reshape(shape)


class Tiny(torch.nn.Module):
    def __init__(self, seed):
        super().__init__()

        def lin(k, *shape):
            m = torch.nn.Module()
            m.w = torch.nn.Parameter(_fill(shape, seed * 10 + k))
            m.b = torch.nn.Parameter(_fill((shape[0],), seed * 10 + k + 5))
            return m
        # odd sizes and one tensor longer than a 65536-element chunk
        self.models = torch.nn.ModuleDict({"encoder": lin(0, 7, 3), "depth": lin(1, 33, 5, 3), "mtl_decoder": lin(2, 1030),
                                           "pose": lin(3, 70001)})
