"""Second, formula-level restatement of SURVEY.md Appendix A/B in numpy (tiny inputs only).

Written from the specification, not from oracle/sgbm_oracle.c: full volumes, window sums by
direct summation over clamped indices, one path at a time by direct recursion, connected
components through scipy's graph routine.  Its only purpose is to cross-check the C oracle
(the reference holds no fixtures for this path -- "parity unpinned").
"""
from __future__ import annotations

import numpy as np
from scipy.sparse import coo_matrix
from scipy.sparse.csgraph import connected_components

MAX_COST = 32767


def normalise(minDisparity=0, numDisparities=16, blockSize=3, P1=0, P2=0, disp12MaxDiff=0,
              preFilterCap=0, uniquenessRatio=0, speckleWindowSize=0, speckleRange=0, mode=0):
    dim = blockSize if blockSize > 0 else 5
    P1n = P1 if P1 > 0 else 2
    return dict(minD=minDisparity, D=numDisparities, r=dim // 2, P1=P1n,
                P2=max(P2 if P2 > 0 else 5, P1n + 1),
                uniq=uniquenessRatio if uniquenessRatio >= 0 else 10,
                d12=disp12MaxDiff if disp12MaxDiff > 0 else 1,
                ftzero=max(preFilterCap, 15) | 1, spw=speckleWindowSize, spr=speckleRange, mode=mode)


def _features(img, ftzero):
    """(val, lo, hi) for the gradient channel and the raw channel; A.2 + half-pixel interval."""
    I = img.astype(np.int64)
    H, W = I.shape
    P = np.pad(I, ((1, 1), (0, 0)), mode="edge")
    up, mid, dn = P[:-2], P[1:-1], P[2:]
    grad = np.zeros_like(I)
    grad[:, 1:-1] = (2 * (mid[:, 2:] - mid[:, :-2]) + (up[:, 2:] - up[:, :-2]) + (dn[:, 2:] - dn[:, :-2]))
    pf = np.clip(grad, -ftzero, ftzero) + ftzero
    raw = I.copy()
    pf[:, [0, -1]] = ftzero
    raw[:, [0, -1]] = ftzero
    out = []
    for v in (pf, raw):
        l = v.copy()
        r = v.copy()
        l[:, 1:] = (v[:, 1:] + v[:, :-1]) // 2
        r[:, :-1] = (v[:, :-1] + v[:, 1:]) // 2
        out.append((v, np.minimum(v, np.minimum(l, r)), np.maximum(v, np.maximum(l, r))))
    return out


def pixel_cost(left, right, q):
    """pix[y, xi, k] for xi in the valid-column domain, k = d - minD.  A.3"""
    H, W = left.shape
    minD, D = q["minD"], q["D"]
    minX1, maxX1 = max(minD + D, 0), W + min(minD, 0)
    W1 = maxX1 - minX1
    fl, fr = _features(left, q["ftzero"]), _features(right, q["ftzero"])
    pix = np.zeros((H, W1, D), np.int64)
    xs = np.arange(minX1, maxX1)
    for k in range(D):
        xr = xs - (minD + k)
        for c, sh in ((0, 0), (1, 2)):
            u, u0, u1 = (a[:, xs] for a in fl[c])
            v, v0, v1 = (a[:, xr] for a in fr[c])
            c0 = np.maximum(0, np.maximum(u - v1, v0 - u))
            c1 = np.maximum(0, np.maximum(v - u1, u0 - v))
            pix[:, :, k] += np.minimum(c0, c1) >> sh
    return pix, minX1, W1


def block_cost(pix, r):
    """C_true by direct summation over the clamped (2r+1)^2 window.  A.4"""
    H, W1, D = pix.shape
    C = np.zeros_like(pix)
    ys, xs = np.arange(H), np.arange(W1)
    for j in range(-r, r + 1):
        yy = np.clip(ys + j, 0, H - 1)
        for i in range(-r, r + 1):
            xx = np.clip(xs + i, 0, W1 - 1)
            C += pix[yy][:, xx]
    return C


def aggregate_path(C, rx, ry, P1, P2):
    """L_r for predecessor q = p - (rx, ry); zero state outside the domain.  A.5"""
    H, W1, D = C.shape
    L = np.zeros((H, W1, D), np.int64)
    yr = range(H) if ry >= 0 else range(H - 1, -1, -1)
    xr = range(W1) if rx >= 0 else range(W1 - 1, -1, -1)
    if ry == 0:
        for x in xr:
            xq = x - rx
            Lq = L[:, xq] if 0 <= xq < W1 else np.zeros((H, D), np.int64)
            L[:, x] = _step(C[:, x], Lq, P1, P2)
    else:
        for y in yr:
            yq = y - ry
            if 0 <= yq < H:
                Lq = np.zeros((W1, D), np.int64)
                xs = np.arange(W1) - rx
                ok = (xs >= 0) & (xs < W1)
                Lq[ok] = L[yq, xs[ok]]
            else:
                Lq = np.zeros((W1, D), np.int64)
            L[y] = _step(C[y], Lq, P1, P2)
    return L


def _step(Cp, Lq, P1, P2):
    pad = np.pad(Lq, ((0, 0), (1, 1)), constant_values=MAX_COST)
    m = Lq.min(axis=1, keepdims=True)
    t = np.minimum(np.minimum(Lq, pad[:, :-2] + P1), np.minimum(pad[:, 2:] + P1, m + P2))
    return Cp + t - m


DIRS5 = [(1, 0), (1, 1), (0, 1), (-1, 1), (-1, 0)]
DIRS8 = DIRS5 + [(1, -1), (0, -1), (-1, -1)]


def select_disparity(S, W, minX1, q):
    """WTA, uniqueness, right view, sub-pixel, LR check.  A.6"""
    H, W1, D = S.shape
    minD, uniq, d12 = q["minD"], q["uniq"], q["d12"]
    inv = (minD - 1) * 16
    disp = np.full((H, W), inv, np.int64)
    for y in range(H):
        d2 = np.full(W, inv, np.int64)
        d2c = np.full(W, MAX_COST, np.int64)
        for xi in range(W1 - 1, -1, -1):
            s = S[y, xi]
            best = int(np.argmin(s))  # first minimum
            ms = int(s[best])
            ds = np.arange(D)
            if np.any((s * (100 - uniq) < ms * 100) & (np.abs(best - ds) > 1)):
                continue
            x2 = xi + minX1 - best - minD
            if d2c[x2] > ms:
                d2c[x2] = ms
                d2[x2] = best + minD
            if 0 < best < D - 1:
                den = max(int(s[best - 1] + s[best + 1] - 2 * s[best]), 1)
                num = int(s[best - 1] - s[best + 1]) * 16 + den
                # C integer division truncates toward zero
                quo = num // (den * 2) if num >= 0 else -((-num) // (den * 2))
                dsc = best * 16 + quo
            else:
                dsc = best * 16
            disp[y, xi + minX1] = dsc + minD * 16
        for x in range(minX1, minX1 + W1):
            d1 = int(disp[y, x])
            if d1 == inv:
                continue
            lo, hi = d1 >> 4, (d1 + 15) >> 4
            xa, xb = x - lo, x - hi
            if (0 <= xa < W and d2[xa] >= minD and abs(d2[xa] - lo) > d12 and
                    0 <= xb < W and d2[xb] >= minD and abs(d2[xb] - hi) > d12):
                disp[y, x] = inv
    return disp


def median3(img):
    P = np.pad(img, 1, mode="edge")
    H, W = img.shape
    st = np.stack([P[j:j + H, i:i + W] for j in range(3) for i in range(3)], axis=0)
    return np.sort(st, axis=0)[4]


def speckles(img, newVal, maxSize, maxDiff):
    H, W = img.shape
    idx = np.arange(H * W).reshape(H, W)
    v = img.astype(np.int64)
    ok = v != newVal
    rows, cols = [], []
    e = ok[:, :-1] & ok[:, 1:] & (np.abs(v[:, :-1] - v[:, 1:]) <= maxDiff)
    rows.append(idx[:, :-1][e]); cols.append(idx[:, 1:][e])
    e = ok[:-1] & ok[1:] & (np.abs(v[:-1] - v[1:]) <= maxDiff)
    rows.append(idx[:-1][e]); cols.append(idx[1:][e])
    r, c = np.concatenate(rows), np.concatenate(cols)
    g = coo_matrix((np.ones(r.size), (r, c)), shape=(H * W, H * W))
    _, lab = connected_components(g, directed=False)
    lab = lab.reshape(H, W)
    size = np.bincount(lab[ok].ravel(), minlength=lab.max() + 1)
    out = img.copy()
    out[ok & (size[lab] <= maxSize)] = newVal
    return out


def sgbm(left, right, **kw):
    """Returns dict(C, S, disp_raw, disp_median, disp)."""
    q = normalise(**kw)
    H, W = left.shape
    inv = (q["minD"] - 1) * 16
    pix, minX1, W1 = pixel_cost(left, right, q)
    C = block_cost(pix, q["r"])
    dirs = DIRS8 if q["mode"] == 1 else DIRS5
    S = np.zeros_like(C)
    for rx, ry in dirs:
        S += aggregate_path(C, rx, ry, q["P1"], q["P2"])
    S = np.minimum(S, MAX_COST)
    raw = select_disparity(S, W, minX1, q)
    med = median3(raw)
    out = med
    if q["spw"] > 0:
        out = speckles(med, inv, q["spw"], 16 * q["spr"])
    return dict(C=C, S=S, disp_raw=raw, disp_median=med, disp=out)


def reproject(disp, Q, handle_missing=False):
    """Appendix B in numpy float64 with explicit operation order."""
    H, W = disp.shape
    Q = np.asarray(Q, np.float64)
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    d = disp.astype(np.float64)
    vec = (xs, ys, d, np.ones_like(d))
    h = []
    for i in range(4):
        s = np.zeros_like(d)
        for k in range(4):
            s = s + Q[i, k] * vec[k]
        h.append(s)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        ia = 1.0 / h[3]
        out = np.stack([(h[i].astype(np.float32).astype(np.float64) * ia).astype(np.float32) for i in range(3)], axis=-1)
    if handle_missing:
        out[..., 2][np.abs(d - d.min()) <= np.finfo(np.float32).eps] = 10000.0
    return out
