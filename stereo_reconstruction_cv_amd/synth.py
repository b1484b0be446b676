"""Deterministic synthetic rectified stereo pairs.

Integer-only, counter-based (SplitMix64 hash of the pixel coordinate) so the same bytes come
out on every machine without depending on numpy's generator state.  Used by bench.py, the
tests and smoke(); the reference itself ships only three JPEG pairs (dataset/d1..d3), which
cannot travel to the GPU box.
"""
from __future__ import annotations

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(z: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (z + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def _hash2(ix: np.ndarray, iy: np.ndarray, seed: int, salt: int) -> np.ndarray:
    """uint64 hash of integer lattice points."""
    with np.errstate(over="ignore"):
        k = (ix.astype(np.uint64) * np.uint64(0x1000193)
             + iy.astype(np.uint64) * np.uint64(0x9E3779B1)
             + np.uint64((seed * 0x2545F4914F6CDD1D + salt * 0xD6E8FEB86659FD93) & 0xFFFFFFFFFFFFFFFF))
    return _splitmix64(_splitmix64(k))


def _octave(xs: np.ndarray, ys: np.ndarray, cell: int, seed: int, salt: int) -> np.ndarray:
    """Bilinear interpolation (integer weights) of 8-bit lattice noise with the given cell size."""
    ix, tx = np.divmod(xs, cell)
    iy, ty = np.divmod(ys, cell)
    # hash only the lattice points that are touched, then gather
    ix0, iy0 = int(ix.min()), int(iy.min())
    gx = np.arange(ix0, int(ix.max()) + 2, dtype=np.int64)
    gy = np.arange(iy0, int(iy.max()) + 2, dtype=np.int64)
    GY, GX = np.meshgrid(gy, gx, indexing="ij")
    lat = (_hash2(GX, GY, seed, salt) >> np.uint64(56)).astype(np.int64)  # 0..255
    jx, jy = ix - ix0, iy - iy0
    a = lat[jy, jx]
    b = lat[jy, jx + 1]
    c = lat[jy + 1, jx]
    d = lat[jy + 1, jx + 1]
    top = a * (cell - tx) + b * tx
    bot = c * (cell - tx) + d * tx
    return (top * (cell - ty) + bot * ty) // (cell * cell)


def texture(H: int, W: int, seed: int, salt: int = 0, x0: np.ndarray | None = None) -> np.ndarray:
    """int64 texture in [16, 240]; x0 optionally gives per-pixel sample columns (for warping)."""
    ys, xs = np.meshgrid(np.arange(H, dtype=np.int64), np.arange(W, dtype=np.int64), indexing="ij")
    if x0 is not None:
        xs = x0
    xs = xs + 4096  # keep lattice indices positive for negative sample columns
    t = (2 * _octave(xs, ys, 4, seed, salt * 3 + 0)
         + 3 * _octave(xs, ys, 16, seed, salt * 3 + 1)
         + 3 * _octave(xs, ys, 64, seed, salt * 3 + 2)) // 8
    return 16 + t * 224 // 255


def ground_truth(H: int, W: int, D: int) -> np.ndarray:
    """Piecewise-planar integer disparity: sloped background plus three fronto-parallel boxes."""
    ys, xs = np.meshgrid(np.arange(H, dtype=np.int64), np.arange(W, dtype=np.int64), indexing="ij")
    g = (15 * D) // 100 + (10 * D * ys) // (100 * max(H, 1))
    boxes = ((0.40, 0.10, 0.15, 0.30, 0.35), (0.60, 0.45, 0.40, 0.30, 0.25), (0.80, 0.25, 0.70, 0.35, 0.20))
    for frac, fy, fx, fh, fw in boxes:
        y0, x0 = int(fy * H), int(fx * W)
        y1, x1 = y0 + max(int(fh * H), 1), x0 + max(int(fw * W), 1)
        g[y0:y1, x0:x1] = int(frac * D)
    return np.minimum(g, max(D - 2, 0))


def make_pair(H: int, W: int, D: int, seed: int = 1234):
    """Return (left u8 HxW, right u8 HxW, gt int HxW).

    left = textured base; right(x, y) = base(x + g(x, y), y), i.e. the scene point seen at right
    column x sits at left column x + g; independent +-2 sensor noise is added to both views.
    """
    ys, xs = np.meshgrid(np.arange(H, dtype=np.int64), np.arange(W, dtype=np.int64), indexing="ij")
    g = ground_truth(H, W, D)
    base = texture(H, W, seed, 0)
    src = xs + g
    warped = texture(H, W, seed, 0, x0=src)
    right = warped
    out = src >= W                      # scene point outside the left view: unrelated texture
    if out.any():
        right = warped.copy()
        right[out] = texture(H, W, seed, 7)[out]
    nl = (_hash2(xs, ys, seed, 101) >> np.uint64(40)).astype(np.int64) % 5 - 2
    nr = (_hash2(xs, ys, seed, 202) >> np.uint64(40)).astype(np.int64) % 5 - 2
    left = np.clip(base + nl, 0, 255).astype(np.uint8)
    right = np.clip(right + nr, 0, 255).astype(np.uint8)
    return left, right, g


def default_Q(W: int) -> np.ndarray:
    """Q printed by the reference notebook (main.ipynb:600-607), principal point and focal
    length scaled by W/3840 for other frame sizes."""
    s = W / 3840.0
    return np.array([[1.0, 0.0, 0.0, -1909.9754 * s],
                     [0.0, 1.0, 0.0, -1057.74529 * s],
                     [0.0, 0.0, 0.0, 2045.48384 * s],
                     [0.0, 0.0, -1.0, 0.0]], dtype=np.float64)
