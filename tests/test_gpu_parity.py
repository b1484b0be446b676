"""HIP engine vs CPU oracle, bit-exact, stage by stage, through the C ABI (needs an MI355X).

Parity statement: "bit-exact vs. a restatement of OpenCV 4.11 MODE_SGBM / MODE_HH inside the
int16 no-overflow regime" -- the oracle itself is unpinned against cv2.
"""
import numpy as np
import pytest

import parity_util as U
from oracle import oracle as O
from stereo_reconstruction_cv_amd import synth

pytestmark = pytest.mark.gpu

CASES = [
    # H, W, D, bs, minD, mode, seed
    (40, 120, 16, 3, 0, 0, 1),
    (64, 200, 16, 11, 0, 0, 2),      # the notebook's own setting (bs=11, D=16), 5 paths
    (48, 320, 32, 5, 0, 1, 3),
    (56, 400, 64, 7, 0, 0, 4),
    (40, 300, 48, 5, 0, 1, 5),       # NP=1, idle lanes
    (48, 500, 128, 7, 0, 0, 6),      # NP=1 full wave
    (36, 640, 160, 5, 0, 1, 7),      # NP=2, idle lanes
    (40, 700, 256, 7, 0, 0, 8),      # NP=2 full wave (bench shape in D)
    (32, 700, 256, 7, 0, 1, 9),
    (24, 900, 320, 5, 0, 0, 10),     # NP=4, idle lanes
    (20, 1100, 512, 3, 0, 1, 11),    # NP=4 full wave
    (40, 200, 32, 5, 3, 0, 12),      # positive minDisparity
    (40, 200, 32, 5, -6, 1, 13),     # negative minDisparity
    (3, 150, 32, 7, 0, 1, 14),       # fewer rows than the block radius
    (30, 70, 64, 5, 0, 0, 15),       # only 6 matchable columns
    (70, 333, 64, 9, 0, 0, 16),      # odd sizes, chunk tails
]


@pytest.mark.parametrize("H,W,D,bs,minD,mode,seed", CASES)
def test_every_stage_bit_exact(H, W, D, bs, minD, mode, seed):
    """default schedule: fused 4-direction sweeps (k_sweep) + boundary pre-pass"""
    l, r, _ = synth.make_pair(H, W, D, seed)
    p = U.params(D, bs, minD, mode, speckleWindowSize=30, speckleRange=2)
    rep, t, h = U.compare_stages(l, r, p)
    assert t["headroom_ok"]
    bad = [U.describe_mismatch(k, h[k], t[k]) for k, n in rep.items() if n]
    assert not bad, "\n".join(bad)


@pytest.mark.parametrize("H,W,D,bs,minD,mode,seed", CASES[::2])
def test_every_stage_bit_exact_per_direction_schedule(H, W, D, bs, minD, mode, seed):
    """schedule 0: one k_path launch per direction"""
    l, r, _ = synth.make_pair(H, W, D, seed)
    p = U.params(D, bs, minD, mode, penalty="plain", speckleWindowSize=30, speckleRange=2)   # single-channel 8 / 32 bs^2
    rep, t, h = U.compare_stages(l, r, p, schedule=0)
    bad = [U.describe_mismatch(k, h[k], t[k]) for k, n in rep.items() if n]
    assert not bad, "\n".join(bad)


@pytest.mark.parametrize("rows", [1, 2, 3, 5, 9, 10, 11])
@pytest.mark.parametrize("mode", [0, 1])
def test_sweep_band_heights(rows, mode):
    """band height of the fused sweep must not matter (boundary pre-pass + LDS hand-off)"""
    for (H, W, D, bs, seed) in ((37, 300, 64, 5, 61), (23, 420, 256, 7, 62), (19, 260, 48, 3, 63)):
        l, r, _ = synth.make_pair(H, W, D, seed)
        p = U.params(D, bs, 0, mode, speckleWindowSize=30, speckleRange=2)
        rep, t, h = U.compare_stages(l, r, p, schedule=1, sweep_rows=rows)
        bad = [U.describe_mismatch(k, h[k], t[k]) for k, n in rep.items() if n]
        assert not bad, f"rows={rows} {(H, W, D)}: " + "\n".join(bad)


def test_more_columns_than_disparities_required():
    img = np.zeros((16, 32), np.uint8)
    p = U.params(32, 5)
    h = U.run_hip_with_taps(img, img, p)
    assert (h["disp"] == -16).all()


def test_constant_and_shift_known_answers():
    img = np.full((40, 96), 100, np.uint8)
    h = U.run_hip_with_taps(img, img, dict(numDisparities=16, blockSize=3))
    assert (h["disp"][:, :16] == -16).all() and (h["disp"][:, 16:] == 0).all()
    base = synth.texture(64, 192, 5).astype(np.uint8)
    right = np.roll(base, -7, axis=1)
    d = U.run_hip_with_taps(base, right, U.params(16, 5, P1=200, P2=800))["disp"]
    assert (np.abs(d[12:52, 48:160].astype(int) - 112) <= 1).all()


def test_strided_rows_and_engine_reuse():
    l, r, _ = synth.make_pair(48, 256, 32, 21)
    big_l = np.zeros((48, 300), np.uint8); big_l[:, :256] = l
    big_r = np.zeros((48, 300), np.uint8); big_r[:, :256] = r
    p = U.params(32, 5)
    import stereo_reconstruction_cv_amd as cv
    m = cv.StereoSGBM_create(**p)
    want = O.sgbm_compute(l, r, **p)
    assert np.array_equal(m.compute(big_l[:, :256], big_r[:, :256]), want)      # row stride 300
    # same engine, different shape, then back
    l2, r2, _ = synth.make_pair(30, 180, 32, 22)
    assert np.array_equal(m.compute(l2, r2), O.sgbm_compute(l2, r2, **p))
    assert np.array_equal(m.compute(l, r), want)


def test_error_behaviour_matches_cv2_shape():
    import stereo_reconstruction_cv_amd as cv
    m = cv.StereoSGBM_create(numDisparities=16, blockSize=3)
    with pytest.raises(cv.error):
        m.compute(np.zeros((4, 8), np.uint8), np.zeros((4, 9), np.uint8))
    with pytest.raises(cv.error):
        m.compute(np.zeros((4, 8), np.float32), np.zeros((4, 8), np.float32))
    with pytest.raises(cv.error):
        cv.StereoSGBM_create(numDisparities=16, mode=2).compute(np.zeros((4, 40), np.uint8), np.zeros((4, 40), np.uint8))
    with pytest.raises(cv.error):
        cv.reprojectImageTo3D(np.zeros((4, 4), np.float32), np.eye(3))
    # the notebook's wrapper turns the exception into None (main.ipynb:696-701)
    assert cv.reconstruct_3D(np.zeros((4, 4), np.float32), np.eye(3)) is None


def test_post_stages_on_arbitrary_maps():
    """median / speckle / float / reproject / mask on a random disparity map via the pipeline entry."""
    import stereo_reconstruction_cv_amd as cv
    rng = np.random.default_rng(5)
    d16 = rng.integers(-16, 700, (37, 53)).astype(np.int16)
    eng = cv.get_engine(U.params(16, 3))
    f = eng.disp_to_float_host(d16)
    fo = O.disp_to_float(d16)
    assert np.array_equal(f.view(np.uint32), fo.view(np.uint32))        # incl. the sign of zero
    Q = synth.default_Q(3840)
    for hm in (False, True):
        a, b = cv.reprojectImageTo3D(f, Q, handleMissingValues=hm), O.reproject(f, Q, hm)
        fin = np.isfinite(b)
        assert np.array_equal(np.isfinite(a), fin)
        # north_star tolerance: 1e-4 relative; the kernel is in fact bit-exact on finite values
        assert np.allclose(a[fin], b[fin], rtol=1e-4, atol=0)
        assert np.array_equal(a[fin], b[fin])
    xyz = cv.reprojectImageTo3D(f, Q)
    assert np.array_equal(cv.valid_point_mask(xyz, f), O.valid_mask(xyz, f))
    # int16 disparity input is accepted like upstream (converted to float)
    assert np.array_equal(np.isfinite(cv.reprojectImageTo3D(d16, Q)), np.isfinite(O.reproject(d16.astype(np.float32), Q)))


def test_notebook_functions_end_to_end():
    import stereo_reconstruction_cv_amd as cv
    l, r, _ = synth.make_pair(90, 400, 16, 31)
    Q = synth.default_Q(400)
    disp, pts, mask = cv.run_disparity(l, r, Q, 16, 0)
    p = dict(minDisparity=0, numDisparities=16, blockSize=11, P1=8 * 3 * 121, P2=32 * 3 * 121, **U.NB)
    want16, taps = O.sgbm_compute(l, r, taps=True, **p)
    assert taps["headroom_ok"]
    wantf = O.disp_to_float(want16)
    assert np.array_equal(disp.view(np.uint32), wantf.view(np.uint32))
    wxyz = O.reproject(wantf, Q)
    fin = np.isfinite(wxyz)
    assert np.array_equal(np.isfinite(pts), fin) and np.array_equal(pts[fin], wxyz[fin])
    assert np.array_equal(mask, O.valid_mask(wxyz, wantf))
    assert mask.mean() > 0.3


def test_batch_entry_matches_single():
    import stereo_reconstruction_cv_amd as cv
    p = U.params(32, 5)
    pairs = [synth.make_pair(40, 160, 32, 40 + i)[:2] for i in range(5)]   # odd count: both engines of the
    L = np.stack([a for a, _ in pairs]); R = np.stack([b for _, b in pairs])   # two-in-flight batch path end differently
    Q = synth.default_Q(160)
    eng = cv.get_engine(p)
    disps, xyz = eng.compute_batch_host(L, R, Q)
    one = eng.compute_batch_host(L[:1], R[:1], None)                         # N = 1: no peer engine involved
    assert np.array_equal(one[0], disps[0])
    for i in range(5):
        want = O.sgbm_compute(L[i], R[i], **p)
        assert np.array_equal(disps[i], want)
        w = O.reproject(O.disp_to_float(want), Q)
        fin = np.isfinite(w)
        assert np.array_equal(xyz[i][fin], w[fin])


def test_torch_device_path_zero_copy():
    import torch
    import stereo_reconstruction_cv_amd as cv
    l, r, _ = synth.make_pair(50, 300, 64, 51)
    p = U.params(64, 7)
    tl, tr = torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda()
    out = cv.StereoSGBM_create(**p).compute(tl, tr)
    assert out.is_cuda and out.dtype == torch.int16
    assert np.array_equal(out.cpu().numpy(), O.sgbm_compute(l, r, **p))


def _speckle_maps():
    rng = np.random.default_rng(11)
    H, W = 97, 203
    maps = {}
    maps["noise"] = rng.integers(-16, 300, (H, W)).astype(np.int16)
    smooth = (np.add.outer(np.arange(H), np.arange(W)) // 3).astype(np.int16)
    smooth[rng.random((H, W)) < 0.05] = -16
    maps["smooth_with_holes"] = smooth
    # serpentine: one long thin component that unions many runs in both directions
    serp = np.full((H, W), -16, np.int16)
    for y in range(0, H, 2):
        serp[y, :] = 100
        if y + 1 < H:
            serp[y + 1, (W - 1) if (y // 2) % 2 == 0 else 0] = 100
    maps["serpentine"] = serp
    # spiral-ish nested rectangles joined at one corner each
    sp = np.full((H, W), -16, np.int16)
    for k in range(0, min(H, W) // 2 - 1, 2):
        sp[k, k:W - k] = 50; sp[H - 1 - k, k:W - k] = 50
        sp[k:H - k, k] = 50; sp[k:H - k, W - 1 - k] = 50
        if k + 2 < min(H, W) // 2 - 1:
            sp[k + 1, k + 1] = 50
    maps["rings"] = sp
    # staircase of values: links depend on maxDiff exactly
    maps["staircase"] = (np.arange(W)[None, :] * 7 + np.arange(H)[:, None] * 16).astype(np.int16)
    blobs = np.full((H, W), -16, np.int16)
    for _ in range(150):
        y, x = rng.integers(0, H - 6), rng.integers(0, W - 6)
        h, w = rng.integers(1, 6), rng.integers(1, 6)
        blobs[y:y + h, x:x + w] = rng.integers(0, 500)
    maps["blobs"] = blobs
    maps["single_row"] = rng.integers(-16, 40, (1, 300)).astype(np.int16)
    maps["single_col"] = rng.integers(-16, 40, (300, 1)).astype(np.int16)
    return maps


@pytest.mark.parametrize("name", sorted(_speckle_maps()))
def test_speckle_filter_and_median_on_adversarial_maps(name):
    import stereo_reconstruction_cv_amd as cv
    img = _speckle_maps()[name]
    eng = cv.get_engine(U.params(16, 3))
    for maxSize, maxDiff in ((0, 0), (1, 0), (4, 7), (30, 16), (100, 512), (10 ** 6, 16)):
        got = eng.filter_speckles_host(img, -16, maxSize, maxDiff)
        want = O.filter_speckles(img, -16, maxSize, maxDiff)
        assert np.array_equal(got, want), (name, maxSize, maxDiff, int((got != want).sum()))
    assert np.array_equal(eng.median3x3_host(img), O.median3x3(img))


def test_speckle_filter_large_frame():
    """4K-sized map with huge smooth regions, thin bridges and salt noise (speckle stress)."""
    import stereo_reconstruction_cv_amd as cv
    rng = np.random.default_rng(3)
    H, W = 1080, 1920
    img = (np.add.outer(np.arange(H) // 7, np.arange(W) // 5) % 900).astype(np.int16)
    img[rng.random((H, W)) < 0.02] = -16
    img[::97, :] = -16
    img[:, ::131] = -16
    img[48::97, 65::131] = 5                     # bridges
    eng = cv.get_engine(U.params(16, 3))
    got = eng.filter_speckles_host(img, -16, 100, 16)
    want = O.filter_speckles(img, -16, 100, 16)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("uniq", [0, 1, 55, 100, 150])
def test_uniqueness_ratio_edges(uniq):
    l, r, _ = synth.make_pair(40, 260, 64, 71)
    for mode in (0, 1):
        p = U.params(64, 5, 0, mode, uniquenessRatio=uniq)
        rep, t, h = U.compare_stages(l, r, p)
        bad = [U.describe_mismatch(k, h[k], t[k]) for k, n in rep.items() if n]
        assert not bad, f"uniq={uniq} mode={mode}: " + "\n".join(bad)

@pytest.mark.gpu
@pytest.mark.parametrize("D,W", [(512, 1100), (256, 700), (32, 300), (16, 250)])
def test_fused_sweeps_repeatable(D, W):
    """The fused sweeps must reproduce the per-direction schedule's S and disparity bit for bit,
    also when the same frame is run again.  (Round 1: with D = 512 the 128-bit S stores picked up
    later register contents now and then.  Root cause, round 2: a VMEM-store data hazard hipcc does
    not pad for MUBUF stores with an SGPR soffset -- DESIGN.md 4.3; the guard against it is the ISA
    check in tests/test_abi.py, not repetition; three runs here only keep the symptom in sight.)"""
    H = 20
    l, r, _ = synth.make_pair(H, W, D, 11)
    p = U.params(D, 3, 0, 1, speckleWindowSize=30, speckleRange=2)
    ref = U.run_hip_with_taps(l, r, p, schedule=0)
    for trial in range(3):
        for rows in (0, 1, 3):
            h = U.run_hip_with_taps(l, r, p, schedule=1, sweep_rows=rows)
            assert np.array_equal(h["S"], ref["S"]), (trial, rows)
            assert np.array_equal(h["disp"], ref["disp"]), (trial, rows)


@pytest.mark.gpu
@pytest.mark.parametrize("D,W,mode", [(512, 700, 1), (256, 500, 1), (64, 200, 1), (16, 120, 0), (128, 300, 0)])
def test_both_winner_take_all_forms(D, W, mode):
    """The WTA runs either inside the last path kernel (k_sweep / k_path / k_rows_g, PATH_LAST) or as
    its own pass over S (k_wta_t); the engine picks per mode and D, debug bit 2 forces the fused
    form.  Both must give the oracle's disparities and identical S."""
    H = 18
    l, r, _ = synth.make_pair(H, W, D, 5)
    p = U.params(D, 5, 0, mode, speckleWindowSize=30, speckleRange=2)
    want, t = O.sgbm_compute(l, r, taps=True, **p)
    assert t["headroom_ok"]
    for debug in (0, 2, 2048, 256):   # 2 / 2048: fused / separate WTA everywhere; 256: the int16 cost pipeline (k_hsum + k_vsum_ring) instead of the byte one
        h = U.run_hip_with_taps(l, r, p, schedule=1, debug=debug)
        assert np.array_equal(h["C"], t["C"]), debug
        assert np.array_equal(h["S"], t["S"]), debug
        assert np.array_equal(h["disp_raw"], t["disp_raw"]), debug
        assert np.array_equal(h["disp"], want), debug


@pytest.mark.gpu
@pytest.mark.parametrize("H,W,D,bs,mode", [(203, 640, 64, 5, 0), (197, 700, 256, 7, 1), (181, 900, 512, 3, 1), (211, 560, 160, 9, 0)])
def test_medium_frames_every_stage(H, W, D, bs, mode):
    """Frames tall enough for many bands per sweep (automatic band height), odd heights, every
    lane packing: all stage taps against the oracle."""
    l, r, _ = synth.make_pair(H, W, D, 77)
    p = U.params(D, bs, 0, mode, speckleWindowSize=60, speckleRange=2)
    rep, t, h = U.compare_stages(l, r, p)
    assert t["headroom_ok"]
    bad = [U.describe_mismatch(k, h[k], t[k]) for k, n in rep.items() if n]
    assert not bad, "\n".join(bad)
