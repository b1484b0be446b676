"""Shared inputs for the rectification tests (gui.py:157-164)."""
import numpy as np


def rodrigues(rvec):
    """Rotation matrix of a rotation vector (closed form, float64)."""
    r = np.asarray(rvec, np.float64)
    th = float(np.linalg.norm(r))
    if th == 0:
        return np.eye(3)
    k = r / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * (Kx @ Kx)


def camera(W, H, f=None, cx=None, cy=None):
    f = f if f is not None else 0.9 * W
    return np.array([[f, 0, cx if cx is not None else W / 2 - 0.5], [0, f * 1.01, cy if cy is not None else H / 2 + 0.25],
                     [0, 0, 1]], np.float64)


def cases():
    """(name, K, dist, R, P, (W, H)) -- the shapes stereoRectify hands to initUndistortRectifyMap."""
    out = []
    W, H = 161, 97
    K = camera(W, H)
    out.append(("identity", K, None, None, None, (W, H)))
    out.append(("identity_P3x4", K, None, np.eye(3), np.hstack([K, [[0.], [0.], [0.]]]), (W, H)))
    R = rodrigues([0.01, -0.02, 0.005])
    P = np.hstack([camera(W, H, f=0.8 * W, cx=W / 2 + 3.0), [[-40.0], [0.], [0.]]])
    out.append(("rotated", K, None, R, P, (W, H)))
    out.append(("rotated_dist5", K, np.array([-0.12, 0.03, 0.001, -0.0005, 0.004]), R, P, (W, H)))
    out.append(("rotated_dist8", K, np.array([-0.12, 0.03, 0.001, -0.0005, 0.004, 0.01, -0.002, 0.0003]), R, P[:, :3], (W, H)))
    out.append(("dist12", K, np.array([0.05, -0.01, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.001, -0.0002, 0.0005, 0.0001]), None, None, (W, H)))
    W2, H2 = 640, 361
    K2 = camera(W2, H2)
    out.append(("vga_rot", K2, None, rodrigues([-0.03, 0.015, -0.01]), camera(W2, H2, f=0.85 * W2), (W2, H2)))
    return out


def image(H, W, cn, seed):
    rng = np.random.default_rng(seed)
    shape = (H, W) if cn == 1 else (H, W, cn)
    # smooth-ish structure plus noise, full 0..255 range
    y, x = np.mgrid[0:H, 0:W]
    base = (127 + 90 * np.sin(x / 7.0 + seed) * np.cos(y / 5.0)).astype(np.int64)
    if cn > 1:
        base = base[..., None] + np.arange(cn) * 17
    return np.clip(base + rng.integers(-40, 41, shape), 0, 255).astype(np.uint8)


def wild_maps(dH, dW, sH, sW, seed):
    """Float maps that exercise every border case: inside, straddling each edge, far outside,
    exact integers, exact half-steps of the 1/32 grid (round-half-even), huge, inf and nan."""
    rng = np.random.default_rng(seed)
    m1 = rng.uniform(-3, sW + 2, (dH, dW)).astype(np.float32)
    m2 = rng.uniform(-3, sH + 2, (dH, dW)).astype(np.float32)
    m1[0, :] = np.round(m1[0, :])                      # integer coordinates
    m2[1, :] = np.round(m2[1, :])
    m1[2, :] = (np.round(m1[2, :] * 32) + 0.5) / 32    # ties of the 1/32 rounding
    m2[3, :] = (np.round(m2[3, :] * 32) - 0.5) / 32
    m1[4, :8] = [-1, -0.5, -1.0 / 64, 0, sW - 1, sW - 1 + 1.0 / 64, sW - 0.5, sW]
    m2[4, :8] = [0, 0, 0, 0, 1, 1, 1, 1]
    m1[5, :6] = [1e9, -1e9, 3e38, np.inf, -np.inf, np.nan]
    m2[5, :6] = 2
    m2[6, :6] = [1e9, -1e9, 3e38, np.inf, -np.inf, np.nan]
    m1[6, :6] = 2
    m1[7, :4] = [40000.0, -40000.0, 32767.0, -32768.0]  # beyond / at the int16 coordinate range
    m2[7, :4] = 2
    return m1, m2
