"""Device part of loader/depth_estimator.py (DepthEstimator.prepare_depth_estimates, :62-93): the depth model's test-time
disparity, min-max normalised per image and quantised to the 8-bit map the reference stores as PNG.  File naming, the
PIL write and the data loader around it are host-side I/O (out of scope): the caller gets the uint8 maps."""
import torch

from .. import hipops as H


def estimate_depth_maps(model, monodepth_loss_calculator, inputs_val):
    """depth_estimator.py:80-91 for one batch: ``model.predict_test_disp`` -> ``generate_depth_test_pred`` -> per image
    clamp / (d - min) / (max - min) / ToPILImage.  Returns (uint8 [B,H,W] on the device, the outputs dict)."""
    with torch.no_grad():
        mono_outputs = model.predict_test_disp(inputs_val)
        monodepth_loss_calculator.generate_depth_test_pred(mono_outputs)
        disp = mono_outputs[("disp", 0)]
        if disp.stride(1) != disp.shape[2] * disp.shape[3] and disp.shape[1] == 1:
            disp = disp.reshape(disp.shape[0], 1, disp.shape[2], disp.shape[3])
        u8 = H.minmax_normalize(disp.float().contiguous(), as_uint8=True)
    return u8[:, 0], mono_outputs
