"""``CannyEdgeDetector`` with the constructor and output dict of the reference's
models/canny_edge_detector.py, executed by one fused HIP kernel (hps_canny_edges): separable Gaussian blur,
Sobel gradients averaged over channels, magnitude, orientation bins, threshold and non-max suppression per
32x32 tile, with the zero padding of the reference's chain of nn.Conv2d calls reproduced at image borders.
"""
import numpy as np
import torch
from torch import nn

from . import _capi


class CannyEdgeDetector(nn.Module):
    def __init__(self, non_max_suppression=True, gaussian_filter_std=1.0, gaussian_filter_size=5, threshold=0.2):
        super().__init__()
        self.threshold = threshold
        self.non_max_suppression = non_max_suppression
        # models/canny_edge_detector.py:23-24: scipy.signal.windows.gaussian(size, std) normalised to sum 1
        n = np.arange(gaussian_filter_size, dtype=np.float64) - (gaussian_filter_size - 1) / 2.0
        g = np.exp(-0.5 * (n / gaussian_filter_std) ** 2)
        self.register_buffer("gaussian_taps", torch.from_numpy(g / g.sum()).float())
        self._taps_host = (np.ascontiguousarray((g / g.sum()).astype(np.float32)))

    def forward(self, img):
        """img (B,C,H,W) on the device -> dict with blurred_img (B,C,H,W) and grad_magnitude, grad_orientation,
        thresholded_grad_magnitude[, thin_edges, thresholded_thin_edges] (B,1,H,W)  (:135-166)."""
        _capi.require_device(img, "img")
        x = _capi.f32c(img)
        B, C, H, W = x.shape
        mk = lambda c: torch.empty(B, c, H, W, device=x.device, dtype=torch.float32)
        blurred, mag, ori, thr = mk(C), mk(1), mk(1), mk(1)
        thin = mk(1) if self.non_max_suppression else None
        thr_thin = mk(1) if self.non_max_suppression else None
        P = _capi.ptr
        taps = self._taps_host
        _capi.call("hps_canny_edges", P(x), _capi._P(taps.ctypes.data), int(taps.shape[0]), P(blurred), P(mag), P(ori),
                   P(thr), P(thin) if thin is not None else None, P(thr_thin) if thr_thin is not None else None,
                   B, C, H, W, float(self.threshold), 1 if self.non_max_suppression else 0, _capi.stream())
        out = {"blurred_img": blurred, "grad_magnitude": mag, "grad_orientation": ori, "thresholded_grad_magnitude": thr}
        if self.non_max_suppression:
            out["thin_edges"] = thin
            out["thresholded_thin_edges"] = thr_thin
        return out

    def edge_map_into(self, img, out, nms=None):
        """The one entry of forward()'s dict the predict front end uses (predict/predict_poseMF_shapeGaussian_net.py:92-93:
        'thresholded_thin_edges' with NMS, else 'thresholded_grad_magnitude'), written straight into channel 0 of ``out``
        (B, K+1, H, W) -- same kernel, same arithmetic, none of the other outputs stored (hps_canny_edge_map).

        ``nms`` selects the ENTRY like the reference's caller does from its config (cfg.DATA.EDGE_NMS), independently of how the
        detector was built: False -> 'thresholded_grad_magnitude' (which exists either way), True -> 'thresholded_thin_edges'
        (KeyError, as in the reference, when the detector was built without non-max suppression).  None = the detector's flag."""
        if nms is None:
            nms = self.non_max_suppression
        if nms and not self.non_max_suppression:
            raise KeyError("thresholded_thin_edges")          # the reference's dict has no such entry without NMS (:160-166)
        _capi.require_device(img, "img")
        x = _capi.f32c(img)
        B, C, H, W = x.shape
        assert out.shape[0] == B and out.shape[2:] == (H, W) and out.is_contiguous() and out.dtype == torch.float32
        taps = self._taps_host
        _capi.call("hps_canny_edge_map", _capi.ptr(x), _capi._P(taps.ctypes.data), int(taps.shape[0]), _capi.ptr(out),
                   int(out.stride(0)), B, C, H, W, float(self.threshold), 1 if nms else 0, _capi.stream())
        return out
