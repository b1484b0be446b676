"""Host-side mirror of the cv2 calls on the reference's "Run Disparity" path.

    cv2.StereoSGBM_create(...)            /root/reference/main.ipynb:655-666  -> StereoSGBM_create
    stereo.compute(imgL, imgR)            /root/reference/main.ipynb:668      -> StereoSGBM.compute
    cv2.reprojectImageTo3D(disp, Q)       /root/reference/main.ipynb:697      -> reprojectImageTo3D

Same names, keyword arguments, dtypes and error behaviour (a Python exception, so the
`try/except Exception` of main.ipynb:696-701 still works), routed through the C ABI of
include/sgm_hip.h into hand-written HIP for gfx950.  numpy arrays go through the host-pointer
entry points (blocking, like cv2); torch CUDA tensors go through the device-pointer entry
points and come back as torch tensors on the same device, without touching the host.
"""
from __future__ import annotations

import ctypes as C
import os
import weakref

import numpy as np

from . import _lib

STEREO_SGBM_MODE_SGBM = 0
STEREO_SGBM_MODE_HH = 1
STEREO_SGBM_MODE_SGBM_3WAY = 2
STEREO_SGBM_MODE_HH4 = 3
CV_32F = 5


class error(Exception):
    """Counterpart of cv2.error."""


_PARAM_NAMES = ("minDisparity", "numDisparities", "blockSize", "P1", "P2", "disp12MaxDiff",
                "preFilterCap", "uniquenessRatio", "speckleWindowSize", "speckleRange", "mode")

_DEFAULT = dict(minDisparity=0, numDisparities=16, blockSize=3, P1=0, P2=0, disp12MaxDiff=0, preFilterCap=0,
                uniquenessRatio=0, speckleWindowSize=0, speckleRange=0, mode=0)

_device = None


def set_device(index: int) -> None:
    """Select the GPU used by engines created afterwards (default: $LOCAL_RANK or 0)."""
    global _device
    _device = int(index)


def get_device() -> int:
    if _device is not None:
        return _device
    return int(os.environ.get("SGM_DEVICE", os.environ.get("LOCAL_RANK", "0")))


def _check(rc: int) -> None:
    if rc != 0:
        raise error(f"sgm_hip error {rc}: {_lib.last_error()}")


class Engine:
    """Owns one sgm_engine handle (device buffers + stream) for a fixed parameter set."""

    def __init__(self, params: dict, device: int | None = None, stream: int | None = None):
        self._L = _lib.load()
        params = {**_DEFAULT, **params}     # cv2.StereoSGBM_create defaults for missing keys
        unknown = set(params) - set(_PARAM_NAMES)
        if unknown:
            raise TypeError(f"unknown StereoSGBM parameter(s): {sorted(unknown)}")
        self.params = dict(params)
        self.device = get_device() if device is None else int(device)
        p = _lib.SgmParams(*[int(params[n]) for n in _PARAM_NAMES])
        self._p = p
        h = C.c_void_p()
        _check(self._L.sgm_create(C.byref(p), self.device, C.c_void_p(stream or 0), C.byref(h)))
        self._h = h
        self._fin = weakref.finalize(self, self._L.sgm_destroy, h)

    # -- options / introspection
    def set_option(self, opt: int, value: int) -> None:
        _check(self._L.sgm_set_option(self._h, opt, int(value)))

    def geometry(self, W: int):
        a, b = C.c_int(), C.c_int()
        _check(self._L.sgm_geometry(C.byref(self._p), W, C.byref(a), C.byref(b)))
        return a.value, b.value

    def algorithmic_bytes(self, H: int, W: int, with_reproject: bool = False) -> int:
        return int(self._L.sgm_algorithmic_bytes(C.byref(self._p), H, W, int(with_reproject)))

    def synchronize(self) -> None:
        _check(self._L.sgm_synchronize(self._h))

    def check(self) -> None:
        """Status WITHOUT waiting for the engine's stream (sgm_check): raises if a chained sweep gave up since the last
        check -- for callers that order the engine's stream with events of their own."""
        _check(self._L.sgm_check(self._h))

    def trim(self) -> None:
        """Give back the internal engines (and their device memory) the batch entries created (sgm_trim)."""
        _check(self._L.sgm_trim(self._h))

    def stage_times(self):
        st = _lib.SgmStageTimes()
        _check(self._L.sgm_get_stage_times(self._h, C.byref(st)))
        return [(st.name[i].decode(), float(st.ms[i]), int(st.launches[i])) for i in range(st.n)]

    def headroom(self) -> dict:
        """Regime record of the last compute (sgm_get_headroom): inside the int16 no-overflow regime
        of OpenCV's StereoSGBM (where this engine's output is bit-exact) iff `ok`."""
        a, b, ok = C.c_int(), C.c_int(), C.c_int()
        _check(self._L.sgm_get_headroom(self._h, C.byref(a), C.byref(b), C.byref(ok)))
        return dict(ok=bool(ok.value), max_cost_plus_p2=a.value, max_delta=b.value)

    def tap(self, which: int, H: int, W: int) -> np.ndarray:
        if which in (_lib.SGM_TAP_COST, _lib.SGM_TAP_AGGR):
            _, W1 = self.geometry(W)
            out = np.empty((H, max(W1, 0), self.params["numDisparities"]), np.int16)
        else:
            out = np.empty((H, W), np.int16)
        _check(self._L.sgm_get_tap(self._h, which, out.ctypes.data, out.nbytes))
        return out

    # -- host-pointer path
    def compute_host(self, left: np.ndarray, right: np.ndarray) -> np.ndarray:
        H, W = left.shape
        disp = np.empty((H, W), np.int16)
        if left.strides[0] != right.strides[0]:
            right = np.ascontiguousarray(right)
            left = np.ascontiguousarray(left)
        _check(self._L.sgm_compute(self._h, left.ctypes.data, right.ctypes.data, H, W, left.strides[0],
                                   disp.ctypes.data))
        return disp

    def compute_batch_host(self, lefts: np.ndarray, rights: np.ndarray, Q: np.ndarray | None = None, out=None):
        """N pairs from / to host arrays (sgm_compute_batch).  out: optional (disps int16 [N, H, W][, xyz float32 [N, H, W, 3]])
        arrays to fill -- like the `disparity` argument of cv2's compute(); a caller that runs batch after batch saves the
        page faults of a fresh 17 MB per 4K map (about a millisecond per pair)."""
        N, H, W = lefts.shape
        lefts = np.ascontiguousarray(lefts, np.uint8)
        rights = np.ascontiguousarray(rights, np.uint8)
        disps = xyz = None
        if out is not None:
            disps, xyz = (out if isinstance(out, (tuple, list)) else (out, None))
            if disps.shape != (N, H, W) or disps.dtype != np.int16 or not disps.flags.c_contiguous:
                raise error("compute_batch_host: out[0] must be a C-contiguous int16 array of shape (N, H, W)")
            if Q is not None and (xyz is None or xyz.shape != (N, H, W, 3) or xyz.dtype != np.float32 or not xyz.flags.c_contiguous):
                raise error("compute_batch_host: out[1] must be a C-contiguous float32 array of shape (N, H, W, 3)")
        if disps is None:
            disps = np.empty((N, H, W), np.int16)
        qp = None
        if Q is not None:
            Q = np.ascontiguousarray(Q, np.float64)
            if xyz is None:
                xyz = np.empty((N, H, W, 3), np.float32)
            qp = Q.ctypes.data
        _check(self._L.sgm_compute_batch(self._h, N, lefts.ctypes.data, rights.ctypes.data, H, W, disps.ctypes.data,
                                         xyz.ctypes.data if Q is not None else None, qp))
        return (disps, xyz) if Q is not None else disps

    def disp_to_float_host(self, disp: np.ndarray) -> np.ndarray:
        disp = np.ascontiguousarray(disp, np.int16)
        out = np.empty(disp.shape, np.float32)
        _check(self._L.sgm_disp_to_float(self._h, disp.ctypes.data, disp.size, out.ctypes.data))
        return out

    def reproject_host(self, disp: np.ndarray, Q: np.ndarray, handle_missing: bool) -> np.ndarray:
        H, W = disp.shape
        out = np.empty((H, W, 3), np.float32)
        _check(self._L.sgm_reproject(self._h, disp.ctypes.data, H, W, Q.ctypes.data, int(handle_missing),
                                     out.ctypes.data))
        return out

    def compact_points_host(self, xyz: np.ndarray, disp: np.ndarray, colors: np.ndarray | None = None):
        """(points_3D[mask], colors[mask]) with the mask of main.ipynb:726-730, row-major order kept."""
        xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        disp = np.ascontiguousarray(disp, np.float32).reshape(-1)
        n = disp.size
        pts = np.empty((n, 3), np.float32)
        rgb = None
        cp = None
        if colors is not None:
            colors = np.ascontiguousarray(colors, np.uint8).reshape(-1, 3)
            rgb = np.empty((n, 3), np.uint8)
            cp = colors.ctypes.data
        nv = C.c_int64(0)
        _check(self._L.sgm_compact_points(self._h, xyz.ctypes.data, disp.ctypes.data, cp, n, pts.ctypes.data,
                                          rgb.ctypes.data if rgb is not None else None, C.byref(nv)))
        return (pts[:nv.value].copy(), rgb[:nv.value].copy() if rgb is not None else None)

    def compact_points_device(self, d_xyz: int, d_dispf: int, d_colors: int | None, n: int, d_points: int,
                              d_out_colors: int | None) -> int:
        """Device-resident form (sgm_compact_points_device): valid points of one frame packed to the front of
        d_points, in row-major order; returns their number (synchronises the engine's stream)."""
        nv = C.c_int64(0)
        _check(self._L.sgm_compact_points_device(self._h, d_xyz, d_dispf, d_colors, n, d_points, d_out_colors, C.byref(nv)))
        return int(nv.value)

    def compact_points_device_async(self, d_xyz: int, d_dispf: int, d_colors: int | None, n: int, d_points: int,
                                    d_out_colors: int | None, d_count_i64: int) -> None:
        """The same in stream order (sgm_compact_points_device_async): the count goes to an int64 in DEVICE memory,
        nothing is synchronised."""
        _check(self._L.sgm_compact_points_device_async(self._h, d_xyz, d_dispf, d_colors, n, d_points, d_out_colors, d_count_i64))

    def median3x3_host(self, img: np.ndarray) -> np.ndarray:
        img = np.ascontiguousarray(img, np.int16)
        out = np.empty_like(img)
        _check(self._L.sgm_median3x3(self._h, img.ctypes.data, img.shape[0], img.shape[1], out.ctypes.data))
        return out

    def filter_speckles_host(self, img: np.ndarray, newVal: int, maxSpeckleSize: int, maxDiff: int) -> np.ndarray:
        out = np.array(img, dtype=np.int16, order="C", copy=True)
        _check(self._L.sgm_filter_speckles(self._h, out.ctypes.data, out.shape[0], out.shape[1], int(newVal),
                                           int(maxSpeckleSize), int(maxDiff)))
        return out

    def valid_mask_host(self, xyz: np.ndarray, disp: np.ndarray) -> np.ndarray:
        xyz = np.ascontiguousarray(xyz, np.float32)
        disp = np.ascontiguousarray(disp, np.float32)
        out = np.empty(disp.shape, np.uint8)
        _check(self._L.sgm_valid_mask(self._h, xyz.ctypes.data, disp.ctypes.data, disp.size, out.ctypes.data))
        return out.astype(bool)

    # -- device-pointer path (raw addresses; torch only supplies the memory)
    def compute_device(self, d_left: int, d_right: int, H: int, W: int, stride: int, d_disp: int) -> None:
        _check(self._L.sgm_compute_device(self._h, d_left, d_right, H, W, stride, d_disp))

    def pipeline_device(self, d_left: int, d_right: int, H: int, W: int, stride: int, Q: np.ndarray | None,
                        d_disp: int | None, d_dispf: int | None, d_xyz: int | None) -> None:
        qp = None
        if Q is not None:
            Q = np.ascontiguousarray(Q, np.float64)
            qp = Q.ctypes.data
        _check(self._L.sgm_pipeline_device(self._h, d_left, d_right, H, W, stride, qp, d_disp, d_dispf, d_xyz))

    def pipeline_batch_device(self, d_lefts, d_rights, H: int, W: int, stride: int, Q: np.ndarray | None,
                              d_disps, d_dispfs=None, d_xyzs=None) -> None:
        """N resident pairs in throughput mode (sgm_pipeline_batch_device): sequences of N device addresses.
        With SGM_OPT_SCHEDULE = 2 the pairs share one chained sweep launch per pass.  Asynchronous."""
        n = len(d_lefts)
        arr = lambda xs: (C.c_void_p * n)(*[int(x) for x in xs]) if xs is not None else None
        qp = None
        if Q is not None:
            Q = np.ascontiguousarray(Q, np.float64)
            qp = Q.ctypes.data
        a, b, c, d, f = arr(d_lefts), arr(d_rights), arr(d_disps), arr(d_dispfs), arr(d_xyzs)
        _check(self._L.sgm_pipeline_batch_device(self._h, n, a, b, H, W, stride, qp, c, d, f))

    def disp_to_float_device(self, d_disp: int, n: int, d_out: int) -> None:
        _check(self._L.sgm_disp_to_float_device(self._h, d_disp, n, d_out))

    def reproject_device(self, d_disp: int, H: int, W: int, Q: np.ndarray, handle_missing: bool, d_xyz: int) -> None:
        Q = np.ascontiguousarray(Q, np.float64)
        _check(self._L.sgm_reproject_device(self._h, d_disp, H, W, Q.ctypes.data, int(handle_missing), d_xyz))

    # -- rectification in front of the path (gui.py:160-164) --
    @staticmethod
    def _rectify_args(K, dist, R, P):
        K = np.ascontiguousarray(K, np.float64).reshape(3, 3)
        d = None if dist is None else np.ascontiguousarray(dist, np.float64).ravel()
        Rm = None if R is None else np.ascontiguousarray(R, np.float64).reshape(3, 3)
        Pm = None if P is None else np.ascontiguousarray(P, np.float64)
        if Pm is not None and Pm.shape not in ((3, 3), (3, 4)):
            raise error("initUndistortRectifyMap: newCameraMatrix must be 3x3 or 3x4")
        return K, d, Rm, Pm

    def init_undistort_rectify_map_host(self, K, dist, R, P, W: int, H: int):
        K, d, Rm, Pm = self._rectify_args(K, dist, R, P)
        m1 = np.empty((H, W), np.float32)
        m2 = np.empty((H, W), np.float32)
        _check(self._L.sgm_init_undistort_rectify_map(
            self._h, K.ctypes.data, None if d is None else d.ctypes.data, 0 if d is None else d.size,
            None if Rm is None else Rm.ctypes.data, None if Pm is None else Pm.ctypes.data,
            0 if Pm is None else Pm.shape[1], W, H, m1.ctypes.data, m2.ctypes.data))
        return m1, m2

    def init_undistort_rectify_map_device(self, K, dist, R, P, W: int, H: int, d_map1: int, d_map2: int) -> None:
        K, d, Rm, Pm = self._rectify_args(K, dist, R, P)
        _check(self._L.sgm_init_undistort_rectify_map_device(
            self._h, K.ctypes.data, None if d is None else d.ctypes.data, 0 if d is None else d.size,
            None if Rm is None else Rm.ctypes.data, None if Pm is None else Pm.ctypes.data,
            0 if Pm is None else Pm.shape[1], W, H, d_map1, d_map2))

    def remap_linear_host(self, src: np.ndarray, map1: np.ndarray, map2: np.ndarray) -> np.ndarray:
        cn = 1 if src.ndim == 2 else src.shape[2]
        dH, dW = map1.shape
        out = np.empty((dH, dW) if src.ndim == 2 else (dH, dW, cn), np.uint8)
        _check(self._L.sgm_remap_linear_u8(self._h, src.ctypes.data, src.shape[0], src.shape[1], src.strides[0], cn,
                                           map1.ctypes.data, map2.ctypes.data, dH, dW, out.ctypes.data))
        return out

    def remap_linear_device(self, d_src: int, sH: int, sW: int, sstride: int, cn: int, d_map1: int, d_map2: int,
                            dH: int, dW: int, d_dst: int, dstride: int) -> None:
        _check(self._L.sgm_remap_linear_u8_device(self._h, d_src, sH, sW, sstride, cn, d_map1, d_map2, dH, dW, d_dst, dstride))

    def valid_mask_device(self, d_xyz: int, d_disp: int, n: int, d_mask: int) -> None:
        _check(self._L.sgm_valid_mask_device(self._h, d_xyz, d_disp, n, d_mask))


# The notebook builds a matcher per call and throws it away (main.ipynb:655-668); engines are
# cached per (parameters, device) so device buffers survive between such calls.
_engine_cache: dict = {}
_CACHE_MAX = 4


def get_engine(params: dict, device: int | None = None) -> Engine:
    dev = get_device() if device is None else int(device)
    params = {**_DEFAULT, **params}
    key = (tuple(int(params[n]) for n in _PARAM_NAMES), dev)
    e = _engine_cache.get(key)
    if e is None:
        if len(_engine_cache) >= _CACHE_MAX:
            _engine_cache.pop(next(iter(_engine_cache)))
        e = Engine(params, dev)
        _engine_cache[key] = e
    return e


def clear_engine_cache() -> None:
    _engine_cache.clear()


def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch") and hasattr(x, "data_ptr")


class StereoSGBM:
    """Mirror of cv2.StereoSGBM (the subset of the interface the reference exercises, plus the
    parameter getters/setters of the cv2 class)."""

    def __init__(self, **kw):
        self._p = {n: 0 for n in _PARAM_NAMES}
        self._p.update(numDisparities=16, blockSize=3)
        for k, v in kw.items():
            if k not in self._p:
                raise TypeError(f"StereoSGBM_create() got an unexpected keyword argument '{k}'")
            self._p[k] = int(v)

    # cv2-style accessors
    def __getattr__(self, name):
        if name.startswith(("get", "set")) and len(name) > 3:
            field = name[3].lower() + name[4:]
            field = {"mode": "mode", "p1": "P1", "p2": "P2"}.get(field, field)
            if field in self._p:
                if name.startswith("get"):
                    return lambda: self._p[field]
                return lambda v: self._p.__setitem__(field, int(v))
        raise AttributeError(name)

    def compute(self, left, right):
        """int16 (H, W) disparity * 16, invalid = (minDisparity - 1) * 16  (main.ipynb:668)."""
        if self._p["mode"] not in (STEREO_SGBM_MODE_SGBM, STEREO_SGBM_MODE_HH):
            raise error("StereoSGBM.compute: only MODE_SGBM and MODE_HH are implemented "
                        "(the reference never selects MODE_SGBM_3WAY / MODE_HH4)")
        if _is_torch(left) or _is_torch(right):
            return self._compute_torch(left, right)
        left, right = np.asarray(left), np.asarray(right)
        # upstream: CV_Assert(left.size() == right.size() && left.type() == right.type() && depth == CV_8U)
        if left.shape != right.shape or left.dtype != right.dtype or left.dtype != np.uint8:
            raise error("StereoSGBM.compute: (-215:Assertion failed) left.size() == right.size() && "
                        "left.type() == right.type() && left.depth() == CV_8U")
        if left.ndim == 3 and left.shape[2] == 1:
            left, right = left[:, :, 0], right[:, :, 0]
        if left.ndim != 2:
            raise error("StereoSGBM.compute: only single-channel 8-bit images are supported "
                        "(the reference reads its pairs with IMREAD_GRAYSCALE, main.ipynb:362-363)")
        if left.shape[0] == 0 or left.shape[1] == 0:
            raise error("StereoSGBM.compute: empty image")
        if left.strides[1] != 1 or left.strides[0] < left.shape[1]:
            left = np.ascontiguousarray(left)
        if right.strides[1] != 1 or right.strides[0] < right.shape[1]:
            right = np.ascontiguousarray(right)
        if left.shape[1] < 2:
            raise error("StereoSGBM.compute: image width < 2")
        return get_engine(self._p).compute_host(left, right)

    def _compute_torch(self, left, right):
        import torch
        if not (_is_torch(left) and _is_torch(right)) or not left.is_cuda or not right.is_cuda:
            raise error("StereoSGBM.compute: torch inputs must both be CUDA (HIP) tensors")
        if left.shape != right.shape or left.dtype != torch.uint8 or right.dtype != torch.uint8 or left.dim() != 2:
            raise error("StereoSGBM.compute: (-215:Assertion failed) left.size() == right.size() && "
                        "left.type() == right.type() && left.depth() == CV_8U")
        left, right = left.contiguous(), right.contiguous()
        H, W = left.shape
        dev = left.device.index or 0
        eng = get_engine(self._p, dev)
        out = torch.empty((H, W), dtype=torch.int16, device=left.device)
        # the engine runs on its own stream: order it after torch's current stream and wait for it
        torch.cuda.current_stream(left.device).synchronize()
        eng.compute_device(left.data_ptr(), right.data_ptr(), H, W, W, out.data_ptr())
        eng.synchronize()
        return out


def StereoSGBM_create(minDisparity=0, numDisparities=16, blockSize=3, P1=0, P2=0, disp12MaxDiff=0,
                      preFilterCap=0, uniquenessRatio=0, speckleWindowSize=0, speckleRange=0,
                      mode=STEREO_SGBM_MODE_SGBM) -> StereoSGBM:
    """Same signature and defaults as cv2.StereoSGBM_create (OpenCV 4.11)."""
    return StereoSGBM(minDisparity=minDisparity, numDisparities=numDisparities, blockSize=blockSize, P1=P1, P2=P2,
                      disp12MaxDiff=disp12MaxDiff, preFilterCap=preFilterCap, uniquenessRatio=uniquenessRatio,
                      speckleWindowSize=speckleWindowSize, speckleRange=speckleRange, mode=mode)




def reprojectImageTo3D(disparity, Q, _3dImage=None, handleMissingValues=False, ddepth=-1):
    """float32 (H, W, 3) = Q . [x, y, d, 1] / W   (main.ipynb:697)."""
    if ddepth not in (-1, CV_32F):
        raise error("reprojectImageTo3D: only ddepth=-1 / CV_32F is implemented (the reference uses the default)")
    Q = np.asarray(Q)
    if Q.shape != (4, 4):
        raise error("reprojectImageTo3D: (-215:Assertion failed) Q.size() == Size(4,4)")
    Q = np.ascontiguousarray(Q, np.float64)
    if _is_torch(disparity):
        import torch
        if not disparity.is_cuda or disparity.dim() != 2:
            raise error("reprojectImageTo3D: torch disparity must be a 2-D CUDA tensor")
        d = disparity.to(torch.float32).contiguous()
        H, W = d.shape
        eng = get_engine(_DEFAULT, d.device.index or 0)
        out = torch.empty((H, W, 3), dtype=torch.float32, device=d.device)
        torch.cuda.current_stream(d.device).synchronize()
        eng.reproject_device(d.data_ptr(), H, W, Q, bool(handleMissingValues), out.data_ptr())
        eng.synchronize()
        return out
    d = np.asarray(disparity)
    # upstream accepts CV_8UC1, CV_16SC1, CV_32SC1, CV_32FC1 and converts each row to float
    if d.ndim != 2 or d.dtype not in (np.uint8, np.int16, np.int32, np.float32):
        raise error("reprojectImageTo3D: (-215:Assertion failed) stype == CV_8UC1 || stype == CV_16SC1 || "
                    "stype == CV_32SC1 || stype == CV_32FC1")
    if d.size == 0:
        raise error("reprojectImageTo3D: empty disparity")
    d = np.ascontiguousarray(d, np.float32)
    return get_engine(_DEFAULT).reproject_host(d, Q, bool(handleMissingValues))


# ---- rectification in front of the path (gui.py:160-164, main.ipynb cell 7) ----
CV_32FC1 = 5
CV_16SC2 = 11
INTER_NEAREST, INTER_LINEAR, INTER_CUBIC = 0, 1, 2
BORDER_CONSTANT = 0


def initUndistortRectifyMap(cameraMatrix, distCoeffs, R, newCameraMatrix, size, m1type, map1=None, map2=None):
    """(map1, map2) float32 (H, W) source coordinates for every destination pixel
    (cv2.initUndistortRectifyMap(K0, None, R1, P1, image_size, cv2.CV_32F), gui.py:160)."""
    if m1type not in (CV_32F, CV_32FC1):
        raise error("initUndistortRectifyMap: only m1type=CV_32FC1 (two float maps, what the reference asks for) is implemented")
    W, H = int(size[0]), int(size[1])
    if W <= 0 or H <= 0:
        raise error("initUndistortRectifyMap: (-215:Assertion failed) size.width > 0 && size.height > 0")
    K = np.asarray(cameraMatrix)
    if K.shape != (3, 3):
        raise error("initUndistortRectifyMap: (-215:Assertion failed) A.size() == Size(3,3)")
    if distCoeffs is not None and np.asarray(distCoeffs).size == 0:
        distCoeffs = None
    if distCoeffs is not None and np.asarray(distCoeffs).size not in (4, 5, 8, 12, 14):
        raise error("initUndistortRectifyMap: (-215:Assertion failed) distCoeffs must hold 4, 5, 8, 12 or 14 values")
    if R is not None and np.asarray(R).size == 0:
        R = None
    if R is not None and np.asarray(R).shape != (3, 3):
        raise error("initUndistortRectifyMap: (-215:Assertion failed) R.size() == Size(3,3)")
    if newCameraMatrix is not None and np.asarray(newCameraMatrix).size == 0:
        newCameraMatrix = None
    return get_engine(_DEFAULT).init_undistort_rectify_map_host(K, distCoeffs, R, newCameraMatrix, W, H)


def remap(src, map1, map2, interpolation, dst=None, borderMode=BORDER_CONSTANT, borderValue=0):
    """Bilinear gather of an 8-bit image through a float map pair
    (cv2.remap(imgL, mapL1, mapL2, interpolation=cv2.INTER_LINEAR), gui.py:163)."""
    if interpolation != INTER_LINEAR:
        raise error("remap: only interpolation=INTER_LINEAR (what the reference uses) is implemented")
    if borderMode != BORDER_CONSTANT or np.any(np.asarray(borderValue) != 0):
        raise error("remap: only borderMode=BORDER_CONSTANT with borderValue=0 (the defaults) is implemented")
    s = np.asarray(src)
    if s.dtype != np.uint8 or s.ndim not in (2, 3) or (s.ndim == 3 and not 1 <= s.shape[2] <= 4) or s.size == 0:
        raise error("remap: source must be a non-empty uint8 image with 1..4 channels")
    m1, m2 = np.asarray(map1), np.asarray(map2)
    if m1.dtype != np.float32 or m2.dtype != np.float32 or m1.ndim != 2 or m1.shape != m2.shape or m1.size == 0:
        raise error("remap: (-215:Assertion failed) map1 and map2 must be CV_32FC1 of the same size")
    s = np.ascontiguousarray(s)
    return get_engine(_DEFAULT).remap_linear_host(s, np.ascontiguousarray(m1), np.ascontiguousarray(m2))
