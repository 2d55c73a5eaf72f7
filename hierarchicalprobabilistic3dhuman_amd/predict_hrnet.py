"""2D keypoint prediction front end with the call surface of the reference's predict/predict_hrnet.py: person box ->
crop to the HRNet input -> injected ``hrnet_model`` -> heat-map arg-max.  The detectors themselves (HRNet, Mask R-CNN)
are out of scope and stay injected objects; this module is the glue between them and the proxy representation."""
import torch

from .image_utils import batch_crop_pytorch_affine, convert_bbox_corners_to_centre_hw_torch

_IMAGENET_MEAN = (0.485, 0.456, 0.406)
_IMAGENET_STD = (0.229, 0.224, 0.225)


def get_kp_locations_confs_from_heatmaps(batch_heatmaps):
    """predict/predict_hrnet.py:7-31: (B,K,H,W) -> keypoints (B,K,2) = (column, row) of the maximum, confidences (B,K);
    keypoints of all-non-positive maps are zeroed."""
    B, K, H, W = batch_heatmaps.shape
    confs, flat = torch.max(batch_heatmaps.reshape(B, K, -1), dim=2)
    kps = torch.stack([(flat % W).float(), torch.floor(flat / float(W)).float()], dim=-1)
    return kps * (confs > 0.0)[:, :, None], confs


def predict_hrnet(hrnet_model, hrnet_config, image, object_detect_model=None, object_detect_threshold=0.8,
                  bbox_scale_factor=1.2):
    """predict/predict_hrnet.py:34-117.  image: (3,H,W) RGB in [0,1] on the device.  Returns the reference's dict: joints2D (K,2)
    in HRNet-input pixels, joints2Dconfs (K,), cropped_image (3, h, w), bbox_centre, bbox_height, bbox_width."""
    H, W = image.shape[1:]
    dev = image.device
    whole = (torch.tensor([H, W], device=dev, dtype=torch.float32) * 0.5, torch.tensor(float(H), device=dev),
             torch.tensor(float(W), device=dev))
    centre, height, width = whole
    if object_detect_model is not None:                                            # :48-73
        pred = object_detect_model(image[None])[0]
        person = pred["labels"] == 1                                                # COCO 'person'
        boxes, scores = pred["boxes"][person], pred["scores"][person]
        boxes = boxes[scores > object_detect_threshold]
        if boxes.shape[0] >= 1:
            centres, heights, widths = convert_bbox_corners_to_centre_hw_torch(boxes[:, [1, 0, 3, 2]])
            k = 0
            if boxes.shape[0] > 1:                                                  # the box nearest the image centre
                k = int(torch.argmin((centres[:, 0] - H / 2.0) ** 2 + (centres[:, 1] - W / 2.0) ** 2))
            centre, height, width = centres[k], heights[k], widths[k]
        else:
            print("Could not find person bounding box - using entire image!")
    in_w, in_h = hrnet_config.MODEL.IMAGE_SIZE[0], hrnet_config.MODEL.IMAGE_SIZE[1]
    aspect = float(in_h) / float(in_w)                                              # :80-84
    if height > width * aspect:
        width = height / aspect
    elif height < width * aspect:
        height = width * aspect
    crop = batch_crop_pytorch_affine(input_wh=(W, H), output_wh=(in_w, in_h), num_to_crop=1, device=dev, rgb=image[None],
                                     bbox_centres=centre[None], bbox_heights=height[None], bbox_widths=width[None],
                                     orig_scale_factor=bbox_scale_factor)["rgb"][0]     # :87-95
    mean = torch.tensor(_IMAGENET_MEAN, device=dev)[:, None, None]
    std = torch.tensor(_IMAGENET_STD, device=dev)[:, None, None]
    heatmaps = hrnet_model(((crop - mean) / std)[None])                             # :98-100  (1,17,96,72)
    joints2D, confs = get_kp_locations_confs_from_heatmaps(heatmaps)
    joints2D = joints2D * (in_w / hrnet_config.MODEL.HEATMAP_SIZE[0])               # :104
    return {"joints2D": joints2D[0], "joints2Dconfs": confs[0], "cropped_image": crop, "bbox_centre": centre,
            "bbox_height": height, "bbox_width": width}
