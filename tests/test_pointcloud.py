"""Next rows of SURVEY.md 8(f): ordered valid-point compaction (GPU) and the PLY writer (host)."""
import numpy as np
import pytest

import stereo_reconstruction_cv_amd as cv
from stereo_reconstruction_cv_amd import synth


def test_ply_roundtrip(tmp_path):
    rng = np.random.default_rng(1)
    pts = rng.normal(size=(7, 5, 3)).astype(np.float32)
    pts[0, 0] = np.inf                       # the notebook writes unmasked points, inf/nan included
    pts[1, 2, 1] = np.nan
    col = rng.integers(0, 256, (7, 5, 3), dtype=np.uint8)
    for binary in (True, False):
        f = tmp_path / f"c_{binary}.ply"
        assert cv.write_point_cloud(str(f), pts, col, binary=binary)
        p2, c2 = cv.read_point_cloud(str(f))
        assert p2.shape == (35, 3) and np.array_equal(c2, col.reshape(-1, 3))
        assert np.array_equal(np.isnan(p2), np.isnan(pts.reshape(-1, 3)))
        fin = np.isfinite(p2)
        assert np.array_equal(p2[fin], pts.reshape(-1, 3).astype(np.float64)[fin])
    f = tmp_path / "nocol.ply"
    cv.write_point_cloud(str(f), pts.reshape(-1, 3))
    p3, c3 = cv.read_point_cloud(str(f))
    assert c3 is None and p3.shape == (35, 3)
    head = open(f, "rb").read(200).decode("ascii", "replace")
    assert head.startswith("ply\nformat binary_little_endian 1.0") and "property double x" in head


@pytest.mark.gpu
def test_compaction_equals_numpy_boolean_indexing():
    rng = np.random.default_rng(2)
    H, W = 123, 257
    disp = (rng.integers(-16, 300, (H, W)).astype(np.float32) / 16.0)
    disp[disp < 0] = -0.0
    Q = synth.default_Q(W)
    pts = cv.reprojectImageTo3D(disp, Q)
    pts[5, 7, 0] = np.nan                                  # a NaN X with positive disparity
    col = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    mask = ~np.isnan(pts[:, :, 0]) & ~np.isinf(pts[:, :, 0]) & (disp > 0)      # main.ipynb:726-730
    vp, vc = cv.valid_points(pts, col, disp)
    assert np.array_equal(vp.view(np.uint32), pts[mask].view(np.uint32))
    assert np.array_equal(vc, col[mask])
    vp2, vc2 = cv.valid_points(pts, None, disp)
    assert vc2 is None and np.array_equal(vp2.view(np.uint32), pts[mask].view(np.uint32))
    # nothing valid / everything valid
    z = np.zeros((H, W), np.float32)
    e, _ = cv.valid_points(cv.reprojectImageTo3D(z, Q), None, z)
    assert e.shape == (0, 3)
    ones = np.ones((H, W), np.float32)
    a, _ = cv.valid_points(cv.reprojectImageTo3D(ones, Q), None, ones)
    assert a.shape == (H * W, 3)
