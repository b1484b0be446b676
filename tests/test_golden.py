"""Committed golden vectors (tests/golden/sgbm_golden.npz, made by tests/golden/make_golden.py
with the oracle -- regression anchors, not cv2 outputs: parity is unpinned)."""
import hashlib
import os

import numpy as np
import pytest

from oracle import oracle as O
from stereo_reconstruction_cv_amd import synth

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sgbm_golden.npz"))
NAMES = sorted({k.split("/")[0] for k in G.files})
PKEYS = ("minDisparity", "numDisparities", "blockSize", "P1", "P2", "disp12MaxDiff", "preFilterCap",
         "uniquenessRatio", "speckleWindowSize", "speckleRange", "mode")


def _params(name):
    return dict(zip(PKEYS, (int(v) for v in G[f"{name}/params"])))


@pytest.mark.parametrize("name", NAMES)
def test_oracle_reproduces_golden(name):
    d, t = O.sgbm_compute(G[f"{name}/left"], G[f"{name}/right"], taps=True, **_params(name))
    for k in ("disp_raw", "disp_median"):
        assert np.array_equal(t[k], G[f"{name}/{k}"])
    assert np.array_equal(d, G[f"{name}/disp"])
    assert hashlib.sha256(t["C"].tobytes()).digest() == G[f"{name}/sha_C"].tobytes()
    assert hashlib.sha256(t["S"].tobytes()).digest() == G[f"{name}/sha_S"].tobytes()
    xyz = O.reproject(O.disp_to_float(d), synth.default_Q(d.shape[1]))
    assert np.array_equal(xyz.view(np.uint32), G[f"{name}/xyz"].view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_hip_reproduces_golden(name):
    import parity_util as U
    import stereo_reconstruction_cv_amd as cv
    h = U.run_hip_with_taps(G[f"{name}/left"], G[f"{name}/right"], _params(name))
    for k in ("disp_raw", "disp_median", "disp"):
        assert np.array_equal(h[k], G[f"{name}/{k}"]), k
    assert hashlib.sha256(h["C"].tobytes()).digest() == G[f"{name}/sha_C"].tobytes()
    assert hashlib.sha256(h["S"].tobytes()).digest() == G[f"{name}/sha_S"].tobytes()
    xyz = cv.reprojectImageTo3D(cv.get_engine(_params(name)).disp_to_float_host(h["disp"]), synth.default_Q(h["disp"].shape[1]))
    want = G[f"{name}/xyz"]
    fin = np.isfinite(want)
    assert np.array_equal(np.isfinite(xyz), fin) and np.allclose(xyz[fin], want[fin], rtol=1e-4, atol=0)
