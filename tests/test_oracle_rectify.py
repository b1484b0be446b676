"""CPU tests of the rectification oracle (oracle/rectify_oracle.c): known answers that do not
depend on any implementation (cv2 is absent: parity unpinned, see the oracle header)."""
import numpy as np
import pytest

from oracle import oracle as O
import rectify_cases as RC


def test_bilinear_table():
    t = O.bilinear_tab().astype(np.int64)
    assert (t.sum(axis=2) == 32768).all()
    assert t[0, 0].tolist() == [32767, 0, 0, 1]          # saturate_cast<short>(32768) + the sum fix-up
    fy, fx = np.mgrid[0:32, 0:32]
    exact = np.stack([(32 - fy) * (32 - fx), (32 - fy) * fx, fy * (32 - fx), fy * fx], axis=2) * 32
    exact[0, 0] = [32767, 0, 0, 1]
    assert np.array_equal(t, exact)


def test_invert3x3_against_numpy():
    rng = np.random.default_rng(3)
    for _ in range(20):
        m = rng.normal(size=(3, 3)) + 3 * np.eye(3)
        assert np.allclose(O.invert3x3(m), np.linalg.inv(m), rtol=1e-12, atol=1e-12)
    assert not O.invert3x3(np.zeros((3, 3))).any()


def test_identity_map_and_remap_are_exact():
    W, H = 161, 97
    K = RC.camera(W, H)
    m1, m2 = O.init_undistort_rectify_map(K, None, None, None, (W, H))
    assert m1.shape == (H, W) and m1.dtype == np.float32
    assert np.abs(m1 - np.arange(W, dtype=np.float32)[None]).max() < 1e-3
    assert np.abs(m2 - np.arange(H, dtype=np.float32)[:, None]).max() < 1e-3
    img = RC.image(H, W, 1, 1)
    assert np.array_equal(O.remap_linear(img, np.round(m1), np.round(m2)), img)


def test_map_matches_closed_form():
    """No distortion: (u, v) = K . normalise(inv(P R) . (j, i, 1)); float64 closed form vs the
    accumulated float32 maps (accumulation error is far below float32 resolution here)."""
    for name, K, dist, R, P, (W, H) in RC.cases():
        if dist is not None:
            continue
        m1, m2 = O.init_undistort_rectify_map(K, dist, R, P, (W, H))
        A = (K if P is None else np.asarray(P)[:, :3]) @ (np.eye(3) if R is None else R)
        iR = np.linalg.inv(A)
        j, i = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
        X = iR @ np.stack([j.ravel(), i.ravel(), np.ones(j.size)])
        u = K[0, 0] * X[0] / X[2] + K[0, 2]
        v = K[1, 1] * X[1] / X[2] + K[1, 2]
        assert np.abs(m1.ravel() - u).max() < 2e-4 * max(1, np.abs(u).max() / 100), name
        assert np.abs(m2.ravel() - v).max() < 2e-4 * max(1, np.abs(v).max() / 100), name


def test_distortion_is_applied():
    W, H = 161, 97
    K = RC.camera(W, H)
    a1, a2 = O.init_undistort_rectify_map(K, None, None, None, (W, H))
    b1, b2 = O.init_undistort_rectify_map(K, [-0.2, 0, 0, 0], None, None, (W, H))
    # barrel term pulls the corners towards the centre, leaves the principal point alone
    cy, cx = int(round(K[1, 2])), int(round(K[0, 2]))
    assert abs(b1[cy, cx] - a1[cy, cx]) < 1e-2
    assert b1[0, 0] > a1[0, 0] + 0.5 and b1[0, -1] < a1[0, -1] - 0.5
    with pytest.raises(ValueError):
        O.init_undistort_rectify_map(K, np.zeros(14), None, None, (W, H))


def test_half_pixel_shift_and_borders():
    H, W = 40, 50
    img = RC.image(H, W, 1, 2)
    m1, m2 = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32))
    out = O.remap_linear(img, m1 + 0.5, m2)
    a = img.astype(np.int64)
    assert np.array_equal(out[:, :-1], (a[:, :-1] + a[:, 1:] + 1) >> 1)
    assert np.array_equal(out[:, -1], (a[:, -1] + 1) >> 1)          # right tap is the zero border
    assert not O.remap_linear(img, m1 + W, m2).any()                 # fully outside -> borderValue 0
    assert not O.remap_linear(img, m1, m2 - H - 1).any()
    out = O.remap_linear(img, m1 - 1, m2)                            # column -1 straddles: tap (x+1) only, weight 0
    assert not out[:, 0].any() and np.array_equal(out[:, 1:], img[:, :-1])


def test_remap_matches_float_bilinear_within_rounding():
    H, W = 61, 83
    for cn in (1, 3):
        img = RC.image(H, W, cn, 5)
        m1, m2 = RC.wild_maps(32, 40, H, W, 9)
        m1 = np.nan_to_num(np.clip(m1, -10, W + 10), nan=0.0).astype(np.float32)
        m2 = np.nan_to_num(np.clip(m2, -10, H + 10), nan=0.0).astype(np.float32)
        out = O.remap_linear(img, m1, m2).astype(np.float64).reshape(32, 40, cn)
        # float64 bilinear on the 1/32-quantised coordinates with a zero border
        qx, qy = np.round(m1.astype(np.float64) * 32) / 32, np.round(m2.astype(np.float64) * 32) / 32
        x0, y0 = np.floor(qx).astype(int), np.floor(qy).astype(int)
        fx, fy = qx - x0, qy - y0
        pad = np.zeros((H + 40, W + 40, cn))
        pad[20:20 + H, 20:20 + W] = img.reshape(H, W, cn)
        g = lambda yy, xx: pad[np.clip(yy + 20, 0, H + 39), np.clip(xx + 20, 0, W + 39)]
        ref = (g(y0, x0) * ((1 - fx) * (1 - fy))[..., None] + g(y0, x0 + 1) * (fx * (1 - fy))[..., None] +
               g(y0 + 1, x0) * ((1 - fx) * fy)[..., None] + g(y0 + 1, x0 + 1) * (fx * fy)[..., None])
        assert np.abs(out - ref).max() <= 0.5 + 1e-6


def test_wild_maps_do_not_crash_and_stay_in_range():
    H, W = 33, 47
    img = RC.image(H, W, 3, 4)
    m1, m2 = RC.wild_maps(16, 24, H, W, 1)
    out = O.remap_linear(img, m1, m2)
    assert out.shape == (16, 24, 3)
    assert not out[5, :6].any() and not out[6, :6].any() and not out[7, :2].any()  # far outside / inf / nan -> 0
