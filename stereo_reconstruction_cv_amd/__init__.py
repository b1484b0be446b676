"""MI355X-native dense stereo disparity engine: drop-in for the reference's
cv2.StereoSGBM_create(...).compute() / cv2.reprojectImageTo3D() path (main.ipynb:655-670, 697).

    import stereo_reconstruction_cv_amd as cv2   # for this path only
"""
from .stereo import (CV_32F, STEREO_SGBM_MODE_HH, STEREO_SGBM_MODE_HH4, STEREO_SGBM_MODE_SGBM,
                     STEREO_SGBM_MODE_SGBM_3WAY, Engine, StereoSGBM, StereoSGBM_create, clear_engine_cache, error,
                     get_device, get_engine, reprojectImageTo3D, set_device, initUndistortRectifyMap, remap, CV_32FC1,
                     INTER_LINEAR, BORDER_CONSTANT)
from .pipeline import compute_disparity_map, reconstruct_3D, rectify_pair, run_disparity, valid_point_mask
from .pointcloud import read_point_cloud, valid_points, write_point_cloud

__all__ = [
    "StereoSGBM_create", "StereoSGBM", "reprojectImageTo3D", "error", "Engine", "get_engine", "set_device",
    "get_device", "clear_engine_cache", "compute_disparity_map", "reconstruct_3D", "valid_point_mask",
    "run_disparity", "valid_points", "write_point_cloud", "read_point_cloud", "STEREO_SGBM_MODE_SGBM", "STEREO_SGBM_MODE_HH", "STEREO_SGBM_MODE_SGBM_3WAY",
    "STEREO_SGBM_MODE_HH4", "CV_32F", "CV_32FC1", "INTER_LINEAR", "BORDER_CONSTANT", "initUndistortRectifyMap", "remap", "rectify_pair",
]
