"""Reachable-but-rare code paths of the engine, each against the oracle (needs an MI355X):
window sizes 13..31 (generic k_vsum, k_hsum<NP,0>), preFilterCap > 96 (byte cost pipeline off),
the 3-launch pre-pass fallback of frames with rowsz*H >= 2^31 (forced through debug bit 16), the
A/B switches of csrc/sgm_debug.h, the engine's own regime record (sgm_get_headroom) at the edge of
and outside the int16 no-overflow regime, and upstream's condition for the speckle filter."""
import numpy as np
import pytest

import parity_util as U
from oracle import oracle as O
from stereo_reconstruction_cv_amd import synth

pytestmark = pytest.mark.gpu


def _low_contrast_pair(H, W, D, seed):
    l, r, _ = synth.make_pair(H, W, D, seed)
    return (96 + l.astype(int) // 6).astype(np.uint8), (96 + r.astype(int) // 6).astype(np.uint8)


@pytest.mark.parametrize("H,W,D,bs,cap,mode", [
    (40, 200, 32, 13, 100, 0), (36, 260, 64, 15, 127, 1), (44, 300, 128, 21, 97, 0),
    (70, 420, 256, 31, 63, 1), (50, 200, 16, 31, 127, 0), (33, 700, 512, 13, 120, 1)])
def test_large_blocks_and_high_prefilter_caps(H, W, D, bs, cap, mode):
    l, r = _low_contrast_pair(H, W, D, 900 + bs)
    p = dict(minDisparity=0, numDisparities=D, blockSize=bs, P1=2 * bs, P2=9 * bs, disp12MaxDiff=1, preFilterCap=cap,
             uniquenessRatio=10, speckleWindowSize=50, speckleRange=2, mode=mode)
    for schedule in (1, 0):
        rep, t, h = U.compare_stages(l, r, p, schedule=schedule)
        assert t["headroom_ok"]
        bad = [U.describe_mismatch(k, h[k], t[k]) for k, n in rep.items() if n]
        assert not bad, f"schedule {schedule}: " + "\n".join(bad)


@pytest.mark.parametrize("debug", [8, 16, 32, 128, 16 | 32, 4, 4 | 16, 512, 65536, 2, 2048, 4096, 4096 | 16, 8192, 8192 | 16])
@pytest.mark.parametrize("H,W,D,bs,mode", [(45, 420, 256, 7, 1), (38, 300, 64, 5, 1), (41, 200, 16, 11, 0), (29, 640, 160, 5, 0)])
def test_debug_switches_keep_results(debug, H, W, D, bs, mode):
    """8: k_vsum_ring with 4 int16 per thread; 16: the pre-pass as three launches of the single-direction
    kernel (what frames with rowsz*H >= 2^31 take); 32: no auxiliary stream; 128: fork before the
    downward pre-pass; 4: no lane groups; 65536: MODE_SGBM's fifth path after the sweep (S +=) instead of
    beside it into its own volume; 2 / 2048: winner-take-all fused into the last path kernel / always its own pass
    (csrc/sgm_debug.h); 4096: D <= 64, the left-to-right in-row path after the vertical kernel instead of beside it
    into a third volume; 8192: D <= 64 MODE_SGBM, the per-row record + element-wise vertical kernel instead of one volume per
    direction (k_lines3_g).  Results must not change; 256 (int16 cost pipeline) is in
    test_gpu_parity.py::test_both_winner_take_all_forms."""
    l, r, _ = synth.make_pair(H, W, D, 300 + debug)
    p = U.params(D, bs, 0, mode, speckleWindowSize=30, speckleRange=2)
    want, t = O.sgbm_compute(l, r, taps=True, **p)
    assert t["headroom_ok"]
    for dbg in (debug, debug | 256):
        h = U.run_hip_with_taps(l, r, p, schedule=1, debug=dbg)
        for k in ("C", "S", "disp_raw"):
            assert np.array_equal(h[k], t[k]), (dbg, k)
        assert np.array_equal(h["disp"], want), dbg
        assert U.headroom_equal(h, t), (dbg, h["headroom"], t["max_cost_plus_p2"], t["max_delta"])


@pytest.mark.parametrize("H,W,D", [(130, 100, 32), (90, 120, 48), (200, 56, 16), (50, 19, 16), (61, 24, 16), (7, 300, 64),
                                   (1, 90, 16), (2, 90, 32), (95, 150, 64), (300, 81, 16)])
def test_small_d_line_walks(H, W, D):
    """D <= 64, MODE_SGBM: the three directions from the row above are walked along their lines, several lines per
    wave (k_lines3_g: one volume per direction; debug 8192: k_prepass3_g + k_vert3_g through the per-row record).  A
    line that leaves the image re-enters on the other side with a fresh state; the kernels find those steps with
    scalar counters modulo the row width.  Frames taller than wide (several wraps per line), narrower than the
    lines of one wave (W1 = 3, 8 lines per wave), D = 48 (idle lanes inside a group), one and two rows."""
    l, r, _ = synth.make_pair(H, W, D, 4000 + H + W)
    p = U.params(D, 3, 0, 0, speckleWindowSize=30, speckleRange=2)
    want, t = O.sgbm_compute(l, r, taps=True, **p)
    assert t["headroom_ok"]
    for dbg in (0, 8192, 4096):
        h = U.run_hip_with_taps(l, r, p, schedule=1, debug=dbg)
        for k in ("C", "S", "disp_raw"):
            assert np.array_equal(h[k], t[k]), (dbg, k)
        assert np.array_equal(h["disp"], want), dbg
        assert U.headroom_equal(h, t), (dbg, h["headroom"], t["max_cost_plus_p2"], t["max_delta"])


@pytest.mark.parametrize("chunk", [1, 2, 5, 7, 8, 16, 24, 64])
@pytest.mark.parametrize("mode", [0, 1])
def test_prepass_row_chunks(chunk, mode):
    """The boundary pre-pass walks the image in row chunks (one launch each) and hands the state of
    every line from chunk to chunk through a ping-pong buffer, with the lines re-dealt to the waves
    at every chunk (XCD-grouped base columns, kernels_path.h).  Chunk height must not matter --
    also when diagonals wrap around the side border inside or exactly at the end of a chunk
    (frames narrower than they are tall)."""
    for (H, W, D, bs, rows, seed) in ((61, 300, 64, 5, 3, 71), (47, 420, 256, 7, 2, 72), (90, 120, 48, 3, 4, 73),
                                      (33, 1100, 512, 3, 1, 74), (130, 100, 32, 5, 9, 75)):
        l, r, _ = synth.make_pair(H, W, D, seed)
        p = U.params(D, bs, 0, mode, speckleWindowSize=30, speckleRange=2)
        rep, t, h = U.compare_stages(l, r, p, schedule=1, sweep_rows=rows, prepass_rows=chunk)
        assert t["headroom_ok"]
        bad = [U.describe_mismatch(k, h[k], t[k]) for k, n in rep.items() if n]
        assert not bad, f"chunk={chunk} {(H, W, D)}: " + "\n".join(bad)


def test_headroom_record_at_the_edge_of_the_regime():
    """Sawtooth against its mirror image at the notebook's blockSize=11 / P2=11616: the largest
    C + P2 is 32 406 of 32 767 -- still inside the regime, results bit-exact, and the engine's own
    record agrees with the oracle's to the count."""
    H, W, D = 40, 200, 32
    x = np.arange(W)[None, :] + np.zeros((H, 1), int)
    y = np.arange(H)[:, None]
    l = ((x * 8) % 256).astype(np.uint8)
    r = ((-(x * 8) - y * 3) % 256).astype(np.uint8)
    for mode in (0, 1):
        p = dict(minDisparity=0, numDisparities=D, blockSize=11, P1=2904, P2=11616, mode=mode, **U.NB)
        rep, t, h = U.compare_stages(l, r, p)
        assert t["headroom_ok"] and t["max_cost_plus_p2"] > 32000
        bad = [U.describe_mismatch(k, h[k], t[k]) for k, n in rep.items() if n]
        assert not bad, "\n".join(bad)


def test_headroom_record_flags_inputs_outside_the_regime():
    """Constant 0 against constant 255, blockSize=7, P2=30000 (tests/test_oracle_known_answers.py::
    test_headroom_flag): C + P2 does not fit an int16 lane -- the engine says so."""
    import stereo_reconstruction_cv_amd as cv
    a = np.zeros((24, 64), np.uint8)
    b = np.full((24, 64), 255, np.uint8)
    p = dict(numDisparities=16, blockSize=7, P1=10, P2=30000)
    eng = cv.get_engine(p)
    eng.compute_host(a, b)
    hr = eng.headroom()
    assert not hr["ok"] and hr["max_cost_plus_p2"] == 3087 + 441 + 30000
    _, t = O.sgbm_compute(a, b, taps="light", **p)
    assert (t["max_cost_plus_p2"], t["max_delta"]) == (hr["max_cost_plus_p2"], hr["max_delta"]) and not t["headroom_ok"]
    # window 31 on the same pair: the block cost itself (63 * 961 = 60 543) leaves int16
    eng = cv.get_engine(dict(numDisparities=16, blockSize=31, P1=10, P2=100))
    eng.compute_host(np.zeros((40, 80), np.uint8), np.full((40, 80), 255, np.uint8))
    assert not eng.headroom()["ok"]
    # and a harmless pair afterwards: the record is per compute
    l, r, _ = synth.make_pair(40, 80, 16, 3)
    eng.compute_host(l, r)
    assert eng.headroom()["ok"]


@pytest.mark.parametrize("srange", [-1, -5, 0])
def test_speckle_filter_runs_only_for_non_negative_range(srange):
    """upstream: filterSpeckles only if speckleRange >= 0 && speckleWindowSize > 0"""
    l, r, _ = synth.make_pair(40, 220, 32, 8)
    p = U.params(32, 5, 0, 0, speckleWindowSize=50, speckleRange=srange)
    rep, t, h = U.compare_stages(l, r, p)
    bad = [U.describe_mismatch(k, h[k], t[k]) for k, n in rep.items() if n]
    assert not bad, "\n".join(bad)
    if srange < 0:
        assert np.array_equal(h["disp"], h["disp_median"])      # the filter did not run
        assert (h["disp"] >= 0).mean() > 0.3


def test_the_switch_that_makes_results_wrong_must_be_asked_for_twice():
    """debug bit 64 (the sweep's loader skips its loads: a timing experiment) is refused unless
    SGM_ALLOW_WRONG_RESULTS=1 is in the environment; the public header does not list the debug option at all."""
    import os
    import stereo_reconstruction_cv_amd as cv
    from stereo_reconstruction_cv_amd import _lib
    assert os.environ.get("SGM_ALLOW_WRONG_RESULTS") != "1"
    eng = cv.Engine(dict(numDisparities=64))
    with pytest.raises(cv.error, match="SGM_ALLOW_WRONG_RESULTS"):
        eng.set_option(_lib.SGM_OPT_DEBUG, 64)
    eng.set_option(_lib.SGM_OPT_DEBUG, 8)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert "SGM_OPT_DEBUG =" not in open(os.path.join(root, "include", "sgm_hip.h")).read()


@pytest.mark.parametrize("schedule", [1, 0, 2])
def test_every_cost_saturated_whole_map_invalid(schedule):
    """Constant 0 against constant 255, 11x11 window, 8 paths, minDisparity < 0: every S saturates at 32767, upstream's
    first-minimum scan keeps best = -1 and the map is invalid everywhere (tests/test_oracle_known_answers.py::
    test_every_cost_saturated_at_the_right_most_pixel) -- the engine's "minS == MAX_COST" rejection must give the same."""
    a = np.zeros((14, 64), np.uint8)
    b = np.full((14, 64), 255, np.uint8)
    p = dict(minDisparity=-3, numDisparities=16, blockSize=11, P1=10, P2=100, disp12MaxDiff=1, preFilterCap=63,
             uniquenessRatio=0, speckleWindowSize=0, speckleRange=0, mode=1)
    rep, t, h = U.compare_stages(a, b, p, schedule=schedule, sweep_rows=3 if schedule else 0)
    assert t["headroom_ok"] and (t["S"] == 32767).all() and (t["disp"] == -64).all()
    bad = [U.describe_mismatch(k, h[k], t[k]) for k, n in rep.items() if n]
    assert not bad, "\n".join(bad)
