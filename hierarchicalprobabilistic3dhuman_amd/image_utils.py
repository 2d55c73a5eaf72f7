"""Bounding-box cropping by an affine resample: the part of the reference's utils/image_utils.py the predict front end
uses (batch_crop_pytorch_affine :234-372 with given boxes, convert_bbox_corners_to_centre_hw_torch :25-42).

This is glue AROUND the hot path (it feeds the injected HRNet and the Canny / heat-map kernels); it runs on the device
with torch's own affine_grid / grid_sample, like the reference -- PyTorch as plumbing, no libhps kernel involved.
"""
import torch
import torch.nn.functional as F


def convert_bbox_corners_to_centre_hw_torch(bbox_corners):
    """utils/image_utils.py:25-42: (B,4) corners [v1, h1, v2, h2] (vertical, horizontal) -> centres (B,2), heights, widths."""
    centres = torch.stack([(bbox_corners[:, 0] + bbox_corners[:, 2]) / 2.0, (bbox_corners[:, 1] + bbox_corners[:, 3]) / 2.0], dim=1)
    return centres.float(), bbox_corners[:, 2] - bbox_corners[:, 0], bbox_corners[:, 3] - bbox_corners[:, 1]


def batch_crop_pytorch_affine(input_wh, output_wh, num_to_crop, device, joints2D=None, rgb=None, bbox_centres=None,
                              bbox_heights=None, bbox_widths=None, orig_scale_factor=1.2, **unsupported):
    """utils/image_utils.py:234-372 for the arguments predict_hrnet (:86-95) and the predict harness (:78-87) pass: crop the
    box (centre in (vertical, horizontal) order, height, width; widened to the output aspect ratio, scaled by
    ``orig_scale_factor``) out of ``rgb`` (B,3,H,W) into (B,3,out_h,out_w) by bilinear resampling with zero padding, and
    map ``joints2D`` (B,K,2) = (horizontal, vertical) into the crop.  The bounding-box-from-IUV/seg/joints and the random
    augmentation branches (:275-307, :315-326) belong to training and are not implemented."""
    if unsupported:
        raise NotImplementedError("batch_crop_pytorch_affine: unsupported arguments %s" % sorted(unsupported))
    if bbox_centres is None or bbox_heights is None or bbox_widths is None:
        raise NotImplementedError("batch_crop_pytorch_affine: bounding boxes must be given (inference use)")
    in_wh = torch.tensor(input_wh, device=device, dtype=torch.float32)
    out_wh = torch.tensor(output_wh, device=device, dtype=torch.float32)
    h = bbox_heights.to(device).float().reshape(num_to_crop).clone()
    w = bbox_widths.to(device).float().reshape(num_to_crop).clone()
    centres = bbox_centres.to(device).float().reshape(num_to_crop, 2)
    aspect = (out_wh[1] / out_wh[0]).item()                                        # :310-312 (sequential, like the reference)
    grow_w = h > w * aspect
    w[grow_w] = h[grow_w] / aspect
    grow_h = h < w * aspect
    h[grow_h] = w[grow_h] * aspect
    h, w = h * orig_scale_factor, w * orig_scale_factor                           # :321-322
    box_wh = torch.stack([w, h], dim=-1)
    scale = out_wh / box_wh                                                        # pixels of output per pixel of input
    shift = out_wh * 0.5 - scale * centres[:, [1, 0]]                              # :329-335, (horizontal, vertical)
    out = {}
    if joints2D is not None:                                                       # :365-369
        out["joints2D"] = joints2D.to(device).float() * scale[:, None, :] + shift[:, None, :]
    if rgb is not None:
        # normalised inverse map for grid_sample (:337-352)
        theta = torch.zeros(num_to_crop, 2, 3, device=device, dtype=torch.float32)
        theta[:, 0, 0] = w / in_wh[0]
        theta[:, 1, 1] = h / in_wh[1]
        theta[:, :, 2] = (-shift / scale) / (in_wh * 0.5) + (box_wh / in_wh) - 1
        grid = F.affine_grid(theta, size=[num_to_crop, 1, int(output_wh[1]), int(output_wh[0])], align_corners=False)
        out["rgb"] = F.grid_sample(rgb.to(device).float(), grid, mode="bilinear", padding_mode="zeros", align_corners=False)
    return out
