"""The reference's notebook functions for the "Run Disparity" path, with cv2 replaced by the
HIP engine: same names, arguments and return values.

    compute_disparity_map   /root/reference/main.ipynb:632-671   (cell c10)
    reconstruct_3D          /root/reference/main.ipynb:680-701   (cell c11)
    valid_point_mask        /root/reference/main.ipynb:726-730   (mask inside cell c12)
    run_disparity           /root/reference/main.ipynb:780-797   (driver cell c13, "Tab 6")
"""
from __future__ import annotations

import numpy as np

from . import stereo as _cv


def compute_disparity_map(imgL, imgR, ndisp, mindis):
    """float32 (H, W) disparity in pixels; invalid or non-positive disparities become +-0.0."""
    matcher = _cv.StereoSGBM_create(
        minDisparity=mindis,
        numDisparities=ndisp,
        blockSize=11,
        P1=8 * 3 * 11 ** 2,
        P2=32 * 3 * 11 ** 2,
        disp12MaxDiff=1,
        preFilterCap=63,
        uniquenessRatio=10,
        speckleWindowSize=100,
        speckleRange=32,
    )
    d16 = matcher.compute(imgL, imgR)
    if _cv._is_torch(d16):
        import torch
        eng = _cv.get_engine(matcher._p, d16.device.index or 0)
        out = torch.empty(d16.shape, dtype=torch.float32, device=d16.device)
        eng.disp_to_float_device(d16.data_ptr(), d16.numel(), out.data_ptr())
        eng.synchronize()
        return out
    # main.ipynb:668-670, on the GPU: astype(float32) / 16 ; mask = > 0 ; multiply
    return _cv.get_engine(matcher._p).disp_to_float_host(d16)


def reconstruct_3D(disparity_map, Q):
    """(H, W, 3) float32 point image, or None on error (same contract as the notebook)."""
    try:
        return _cv.reprojectImageTo3D(disparity_map, Q)
    except Exception as e:  # noqa: BLE001 - mirrors main.ipynb:699-701
        print(f"Error in reconstruct_3D: {e}")
        return None


def valid_point_mask(points_3D, disparity_map):
    """~isnan(X) & ~isinf(X) & (disparity > 0) -- the mask visualize_point_cloud applies."""
    if _cv._is_torch(points_3D):
        import torch
        n = disparity_map.numel()
        eng = _cv.get_engine(_cv._DEFAULT, points_3D.device.index or 0)
        out = torch.empty(disparity_map.shape, dtype=torch.uint8, device=points_3D.device)
        torch.cuda.current_stream(points_3D.device).synchronize()
        eng.valid_mask_device(points_3D.contiguous().data_ptr(), disparity_map.contiguous().data_ptr(), n, out.data_ptr())
        eng.synchronize()
        return out.bool()
    return _cv.get_engine(_cv._DEFAULT).valid_mask_host(np.asarray(points_3D), np.asarray(disparity_map))


def run_disparity(imgL, imgR, Q, ndisp=16, mindis=0):
    """Cell c13 end to end: disparity map, point image and validity mask."""
    disparity_map = compute_disparity_map(imgL, imgR, ndisp, mindis)
    points_3D = reconstruct_3D(disparity_map, Q)
    mask = valid_point_mask(points_3D, disparity_map) if points_3D is not None else None
    return disparity_map, points_3D, mask


def rectify_pair(imgL, imgR, K0, K1, R1, R2, P1, P2, image_size):
    """The rectify block of main.ipynb cell 7 / gui.py:160-164: two float map pairs from the
    stereoRectify outputs, then a bilinear remap of both views.  Returns (imgL_rect, imgR_rect)."""
    mapL1, mapL2 = _cv.initUndistortRectifyMap(K0, None, R1, P1, image_size, _cv.CV_32F)
    mapR1, mapR2 = _cv.initUndistortRectifyMap(K1, None, R2, P2, image_size, _cv.CV_32F)
    imgL_rect = _cv.remap(imgL, mapL1, mapL2, interpolation=_cv.INTER_LINEAR)
    imgR_rect = _cv.remap(imgR, mapR1, mapR2, interpolation=_cv.INTER_LINEAR)
    return imgL_rect, imgR_rect
