"""GPU parity of the rectification step against oracle/rectify_oracle.c,
through the C ABI (sgm_init_undistort_rectify_map, sgm_remap_linear_u8) and the cv2-shaped mirror
functions.  Bit-exact: the maps are float32 produced by the same f64 operation sequence, the
remap is integer arithmetic.  (The oracle itself is parity-unpinned against cv2, see its header.)"""
import numpy as np
import pytest

import rectify_cases as RC
import stereo_reconstruction_cv_amd as cv
from oracle import oracle as O
from stereo_reconstruction_cv_amd import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", RC.cases(), ids=lambda c: c[0])
def test_maps_bit_exact(case):
    name, K, dist, R, P, size = case
    w1, w2 = O.init_undistort_rectify_map(K, dist, R, P, size)
    g1, g2 = cv.initUndistortRectifyMap(K, dist, R, P, size, cv.CV_32FC1)
    assert g1.dtype == np.float32 and g1.shape == (size[1], size[0])
    assert np.array_equal(g1.view(np.uint32), w1.view(np.uint32))
    assert np.array_equal(g2.view(np.uint32), w2.view(np.uint32))


@pytest.mark.parametrize("cn", [1, 3, 4])
def test_remap_bit_exact_wild_maps(cn):
    H, W = 61, 83
    img = RC.image(H, W, cn, 7)
    for seed, (dH, dW) in enumerate([(32, 40), (70, 97), (8, 300)]):
        m1, m2 = RC.wild_maps(dH, dW, H, W, seed)
        want = O.remap_linear(img, m1, m2)
        got = cv.remap(img, m1, m2, cv.INTER_LINEAR)
        assert got.shape == want.shape and got.dtype == np.uint8
        assert np.array_equal(got, want)


def test_remap_strided_source_and_identity():
    H, W = 50, 64
    big = RC.image(H, W + 13, 1, 3)
    src = big[:, 5:5 + W]                       # non-contiguous rows
    m1, m2 = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32))
    assert np.array_equal(cv.remap(src, m1, m2, cv.INTER_LINEAR), src)
    eng = cv.get_engine({"numDisparities": 16})
    got = eng.remap_linear_host(src, m1 + 0.25, m2 + 0.75)      # the engine takes the row stride as is
    assert np.array_equal(got, O.remap_linear(np.ascontiguousarray(src), m1 + 0.25, m2 + 0.75))


def test_rectify_then_disparity_chain():
    """gui.py:160-164 followed by the disparity path: rectify both views with slightly rotated
    cameras, run SGBM on the result; every stage equals the oracle chain."""
    H, W, D = 96, 256, 32
    l, r, _ = synth.make_pair(H, W, D, 21)
    K = RC.camera(W, H)
    R1, R2 = RC.rodrigues([0.004, -0.006, 0.002]), RC.rodrigues([-0.003, 0.005, -0.001])
    P = RC.camera(W, H, f=0.93 * W)
    p = dict(minDisparity=0, numDisparities=D, blockSize=5, P1=8 * 25, P2=32 * 25, disp12MaxDiff=1, preFilterCap=63,
             uniquenessRatio=10, speckleWindowSize=50, speckleRange=2, mode=1)
    imgs = []
    for img, R in ((l, R1), (r, R2)):
        g1, g2 = cv.initUndistortRectifyMap(K, None, R, P, (W, H), cv.CV_32F)
        w1, w2 = O.init_undistort_rectify_map(K, None, R, P, (W, H))
        assert np.array_equal(g1, w1) and np.array_equal(g2, w2)
        got, want = cv.remap(img, g1, g2, interpolation=cv.INTER_LINEAR), O.remap_linear(img, w1, w2)
        assert np.array_equal(got, want)
        imgs.append(got)
    disp = cv.StereoSGBM_create(**p).compute(imgs[0], imgs[1])
    assert np.array_equal(disp, O.sgbm_compute(imgs[0], imgs[1], **p))
    assert (disp >= 0).mean() > 0.3


def test_device_entry_points_torch():
    torch = pytest.importorskip("torch")
    H, W = 120, 200
    K = RC.camera(W, H)
    R, P = RC.rodrigues([0.01, 0.02, -0.01]), RC.camera(W, H, f=0.8 * W)
    eng = cv.get_engine({"numDisparities": 16})
    m1 = torch.empty((H, W), dtype=torch.float32, device="cuda")
    m2 = torch.empty_like(m1)
    img = RC.image(H, W, 3, 11)
    src = torch.from_numpy(img).cuda()
    dst = torch.empty_like(src)
    torch.cuda.synchronize()
    eng.init_undistort_rectify_map_device(K, None, R, P, W, H, m1.data_ptr(), m2.data_ptr())
    eng.remap_linear_device(src.data_ptr(), H, W, W * 3, 3, m1.data_ptr(), m2.data_ptr(), H, W, dst.data_ptr(), W * 3)
    eng.synchronize()
    w1, w2 = O.init_undistort_rectify_map(K, None, R, P, (W, H))
    assert np.array_equal(m1.cpu().numpy(), w1) and np.array_equal(m2.cpu().numpy(), w2)
    assert np.array_equal(dst.cpu().numpy(), O.remap_linear(img, w1, w2))


def test_errors_mirror_upstream_shapes():
    K = RC.camera(64, 48)
    with pytest.raises(cv.error):
        cv.initUndistortRectifyMap(K, None, None, None, (64, 48), cv.CV_16SC2 if hasattr(cv, "CV_16SC2") else 11)
    with pytest.raises(cv.error):
        cv.initUndistortRectifyMap(K[:2], None, None, None, (64, 48), cv.CV_32FC1)
    with pytest.raises(cv.error):
        cv.initUndistortRectifyMap(K, np.zeros(14), None, None, (64, 48), cv.CV_32FC1)   # tilt terms: unsupported
    with pytest.raises(cv.error):
        cv.initUndistortRectifyMap(K, None, None, np.zeros((3, 3)), (64, 48), cv.CV_32FC1)  # singular
    img = RC.image(48, 64, 1, 0)
    m = np.zeros((48, 64), np.float32)
    with pytest.raises(cv.error):
        cv.remap(img, m, m, 0)                                  # INTER_NEAREST: not the reference's call
    with pytest.raises(cv.error):
        cv.remap(img.astype(np.float32), m, m, cv.INTER_LINEAR)
    with pytest.raises(cv.error):
        cv.remap(img, m, m[:10], cv.INTER_LINEAR)
    with pytest.raises(cv.error):
        cv.remap(img, m, m, cv.INTER_LINEAR, borderValue=5)
