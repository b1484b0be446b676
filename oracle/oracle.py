"""ctypes front end of the CPU oracle -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(see oracle/sgbm_oracle.h for what is restated and the "parity unpinned" statement).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# ORACLE_SANITIZE=1 (with libasan preloaded into the interpreter: tools/sanitize_oracle.sh) loads the
# AddressSanitizer + UBSan build of the same sources instead
_ASAN = os.environ.get("ORACLE_SANITIZE") == "1"
_SO = os.path.join(_HERE, "liboracle_sgbm_asan.so" if _ASAN else "liboracle_sgbm.so")


class Params(C.Structure):
    # keyword arguments of cv2.StereoSGBM_create, /root/reference/main.ipynb:655-666
    _fields_ = [(n, C.c_int32) for n in (
        "minDisparity", "numDisparities", "blockSize", "P1", "P2", "disp12MaxDiff",
        "preFilterCap", "uniquenessRatio", "speckleWindowSize", "speckleRange", "mode")]


class Taps(C.Structure):
    _fields_ = [("C", C.c_void_p), ("S", C.c_void_p), ("disp_raw", C.c_void_p),
                ("disp_median", C.c_void_p), ("max_cost_plus_p2", C.c_int32),
                ("max_delta", C.c_int32), ("headroom_ok", C.c_int32)]


def build(force: bool = False) -> str:
    src = [os.path.join(_HERE, f) for f in ("sgbm_oracle.c", "sgbm_oracle.h", "rectify_oracle.c", "rectify_oracle.h")]
    stale = (not os.path.exists(_SO)) or any(
        os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(_SO) for s in src)
    if force or stale:
        subprocess.run(["make", "-C", _HERE, "-B" if force else "-s", os.path.basename(_SO)],
                       check=True, capture_output=True)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.oracle_sgbm_compute.restype = C.c_int
        L.oracle_sgbm_compute.argtypes = [C.POINTER(Params), C.c_void_p, C.c_void_p, C.c_int,
                                          C.c_int, C.c_int64, C.c_void_p, C.POINTER(Taps)]
        L.oracle_sgbm_geometry.argtypes = [C.POINTER(Params), C.c_int, C.POINTER(C.c_int),
                                           C.POINTER(C.c_int)]
        L.oracle_median3x3_i16.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.oracle_filter_speckles_i16.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                                 C.c_int]
        L.oracle_disp_to_float.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.oracle_reproject_f32.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                           C.c_void_p]
        L.oracle_valid_mask.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.oracle_invert3x3.restype = C.c_int
        L.oracle_invert3x3.argtypes = [C.c_void_p, C.c_void_p]
        L.oracle_init_undistort_rectify_map.restype = C.c_int
        L.oracle_init_undistort_rectify_map.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                                        C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.oracle_remap_linear_u8.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_void_p,
                                             C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.oracle_bilinear_tab_i16.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def make_params(minDisparity=0, numDisparities=16, blockSize=3, P1=0, P2=0, disp12MaxDiff=0,
                preFilterCap=0, uniquenessRatio=0, speckleWindowSize=0, speckleRange=0,
                mode=0) -> Params:
    """Defaults are those of cv2.StereoSGBM_create (OpenCV 4.11 signature)."""
    return Params(minDisparity, numDisparities, blockSize, P1, P2, disp12MaxDiff, preFilterCap,
                  uniquenessRatio, speckleWindowSize, speckleRange, mode)


def geometry(p: Params, W: int):
    a, b = C.c_int(), C.c_int()
    lib().oracle_sgbm_geometry(C.byref(p), W, C.byref(a), C.byref(b))
    return a.value, b.value  # minX1, W1


def workspace(H: int, W: int, **kw):
    """Volumes a caller keeps between calls of sgbm_compute(..., workspace=ws) on frames of one shape: MODE_HH holds the
    block cost and the aggregated cost of the whole frame (upstream does too), and allocating + first touching them is most
    of a short call (bench.py's frame-parallel CPU baseline runs one such call per thread)."""
    p = kw.pop("params", None) or make_params(**kw)
    _, W1 = geometry(p, W)
    shape = (H, max(W1, 0), p.numDisparities)
    return dict(C=np.empty(shape, np.int16), S=np.empty(shape, np.int16))


def sgbm_compute(left: np.ndarray, right: np.ndarray, taps: bool = False, workspace=None, **kw):
    """stereo.compute(left, right) of main.ipynb:668 -> int16 (H, W).

    With taps=True also returns a dict with C, S, disp_raw, disp_median and the headroom record;
    taps="light" leaves out the two volumes (full-size frames: 4 GB each at 4K, D=256).
    workspace (from workspace()): the two volumes are the caller's, reused call after call; returns the map only.
    """
    p = kw.pop("params", None) or make_params(**kw)
    if workspace is not None:
        left = np.ascontiguousarray(left, dtype=np.uint8)
        right = np.ascontiguousarray(right, dtype=np.uint8)
        H, W = left.shape
        _, W1 = geometry(p, W)
        assert workspace["C"].shape == (H, max(W1, 0), p.numDisparities) == workspace["S"].shape
        disp = np.empty((H, W), np.int16)
        t = Taps()
        if W1 > 0:
            t.C = workspace["C"].ctypes.data
            t.S = workspace["S"].ctypes.data
        rc = lib().oracle_sgbm_compute(C.byref(p), left.ctypes.data, right.ctypes.data, H, W, left.strides[0], disp.ctypes.data, C.byref(t))
        if rc != 0:
            raise ValueError(f"oracle_sgbm_compute failed rc={rc}")
        return disp
    left = np.ascontiguousarray(left, dtype=np.uint8)
    right = np.ascontiguousarray(right, dtype=np.uint8)
    assert left.ndim == 2 and left.shape == right.shape
    H, W = left.shape
    disp = np.empty((H, W), np.int16)
    t = Taps()
    out = {}
    if taps:
        _, W1 = geometry(p, W)
        D = p.numDisparities
        if W1 > 0 and taps != "light":
            out["C"] = np.zeros((H, W1, D), np.int16)
            out["S"] = np.zeros((H, W1, D), np.int16)
            t.C = out["C"].ctypes.data
            t.S = out["S"].ctypes.data
        out["disp_raw"] = np.empty((H, W), np.int16)
        out["disp_median"] = np.empty((H, W), np.int16)
        t.disp_raw = out["disp_raw"].ctypes.data
        t.disp_median = out["disp_median"].ctypes.data
    rc = lib().oracle_sgbm_compute(C.byref(p), left.ctypes.data, right.ctypes.data, H, W,
                                   left.strides[0], disp.ctypes.data, C.byref(t))
    if rc != 0:
        raise ValueError(f"oracle_sgbm_compute failed rc={rc}")
    if taps:
        out.update(max_cost_plus_p2=t.max_cost_plus_p2, max_delta=t.max_delta,
                   headroom_ok=bool(t.headroom_ok))
        return disp, out
    return disp


def headroom_ok(left, right, **kw) -> bool:
    p = kw.pop("params", None) or make_params(**kw)
    left = np.ascontiguousarray(left, dtype=np.uint8)
    right = np.ascontiguousarray(right, dtype=np.uint8)
    H, W = left.shape
    disp = np.empty((H, W), np.int16)
    t = Taps()
    rc = lib().oracle_sgbm_compute(C.byref(p), left.ctypes.data, right.ctypes.data, H, W,
                                   left.strides[0], disp.ctypes.data, C.byref(t))
    assert rc == 0
    return bool(t.headroom_ok)


def median3x3(img: np.ndarray) -> np.ndarray:
    img = np.ascontiguousarray(img, dtype=np.int16)
    out = np.empty_like(img)
    lib().oracle_median3x3_i16(img.ctypes.data, out.ctypes.data, img.shape[0], img.shape[1])
    return out


def filter_speckles(img: np.ndarray, newVal: int, maxSpeckleSize: int, maxDiff: int) -> np.ndarray:
    out = np.array(img, dtype=np.int16, order="C", copy=True)
    lib().oracle_filter_speckles_i16(out.ctypes.data, out.shape[0], out.shape[1], newVal,
                                     maxSpeckleSize, maxDiff)
    return out


def disp_to_float(disp: np.ndarray) -> np.ndarray:
    disp = np.ascontiguousarray(disp, dtype=np.int16)
    out = np.empty(disp.shape, np.float32)
    lib().oracle_disp_to_float(disp.ctypes.data, out.ctypes.data, disp.size)
    return out


def reproject(disp: np.ndarray, Q: np.ndarray, handle_missing: bool = False) -> np.ndarray:
    disp = np.ascontiguousarray(disp, dtype=np.float32)
    Q = np.ascontiguousarray(Q, dtype=np.float64)
    assert Q.shape == (4, 4)
    H, W = disp.shape
    out = np.empty((H, W, 3), np.float32)
    lib().oracle_reproject_f32(disp.ctypes.data, H, W, Q.ctypes.data, int(handle_missing),
                               out.ctypes.data)
    return out


def valid_mask(xyz: np.ndarray, disp: np.ndarray) -> np.ndarray:
    xyz = np.ascontiguousarray(xyz, dtype=np.float32)
    disp = np.ascontiguousarray(disp, dtype=np.float32)
    out = np.empty(disp.shape, np.uint8)
    lib().oracle_valid_mask(xyz.ctypes.data, disp.ctypes.data, disp.size, out.ctypes.data)
    return out.astype(bool)


# ---- rectification step in front of the path (rectify_oracle.h; gui.py:160-164) ----

def invert3x3(m: np.ndarray) -> np.ndarray:
    m = np.ascontiguousarray(m, dtype=np.float64).reshape(3, 3)
    out = np.zeros((3, 3), np.float64)
    lib().oracle_invert3x3(m.ctypes.data, out.ctypes.data)
    return out


def init_undistort_rectify_map(K, dist, R, P, size):
    """cv2.initUndistortRectifyMap(K, dist, R, P, (W, H), cv2.CV_32FC1) -> (map1, map2)."""
    W, H = int(size[0]), int(size[1])
    K = np.ascontiguousarray(K, dtype=np.float64).reshape(3, 3)
    d = None if dist is None else np.ascontiguousarray(dist, dtype=np.float64).ravel()
    Rm = None if R is None else np.ascontiguousarray(R, dtype=np.float64).reshape(3, 3)
    Pm = None if P is None else np.ascontiguousarray(P, dtype=np.float64)
    m1 = np.empty((H, W), np.float32)
    m2 = np.empty((H, W), np.float32)
    rc = lib().oracle_init_undistort_rectify_map(
        K.ctypes.data, None if d is None else d.ctypes.data, 0 if d is None else d.size,
        None if Rm is None else Rm.ctypes.data, None if Pm is None else Pm.ctypes.data,
        0 if Pm is None else Pm.shape[1], W, H, m1.ctypes.data, m2.ctypes.data)
    if rc != 0:
        raise ValueError("oracle_init_undistort_rectify_map: unsupported arguments or singular P*R")
    return m1, m2


def remap_linear(src: np.ndarray, map1: np.ndarray, map2: np.ndarray) -> np.ndarray:
    """cv2.remap(src, map1, map2, cv2.INTER_LINEAR) for uint8 images (H, W) or (H, W, cn)."""
    src = np.ascontiguousarray(src, dtype=np.uint8)
    cn = 1 if src.ndim == 2 else src.shape[2]
    map1 = np.ascontiguousarray(map1, dtype=np.float32)
    map2 = np.ascontiguousarray(map2, dtype=np.float32)
    dH, dW = map1.shape
    out = np.empty((dH, dW) if src.ndim == 2 else (dH, dW, cn), np.uint8)
    lib().oracle_remap_linear_u8(src.ctypes.data, src.shape[0], src.shape[1], src.strides[0], cn,
                                 map1.ctypes.data, map2.ctypes.data, dH, dW, out.ctypes.data)
    return out


def bilinear_tab() -> np.ndarray:
    t = np.empty((32, 32, 4), np.int16)
    lib().oracle_bilinear_tab_i16(t.ctypes.data)
    return t
