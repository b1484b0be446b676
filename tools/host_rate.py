#!/usr/bin/env python3
"""PCIe-inclusive rate through the host-pointer C ABI (numpy in, numpy out), 4K D=256 (GPU box)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import stereo_reconstruction_cv_amd as cv  # noqa: E402
from stereo_reconstruction_cv_amd import synth  # noqa: E402

H, W, D = 2160, 3840, 256
l, r, _ = synth.make_pair(H, W, D, 1234)
Q = synth.default_Q(W)
for mode in (0, 1):
    m = cv.StereoSGBM_create(**bench.sgbm_params(D, 7, mode))
    m.compute(l, r)
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        d16 = m.compute(l, r)
    t1 = time.perf_counter()
    f = cv.get_engine(bench.sgbm_params(D, 7, mode)).disp_to_float_host(d16)
    t2 = time.perf_counter()
    xyz = cv.reprojectImageTo3D(f, Q)
    t3 = time.perf_counter()
    dt = (t1 - t0) / n
    print(f"mode {mode}: host->host compute() {dt * 1e3:.1f} ms/pair = {H * W * D / dt / 1e6:.0f} Mdisp/s; "
          f"disp_to_float {1e3 * (t2 - t1):.1f} ms; reprojectImageTo3D (99.5 MB back) {1e3 * (t3 - t2):.1f} ms")

# batch entry (sgm_compute_batch): N pairs in host memory, three in flight inside the engine
import numpy as np  # noqa: E402
N = 12
L = np.stack([l] * N)
R = np.stack([r] * N)
for mode in (0, 1):
    eng = cv.get_engine(bench.sgbm_params(D, 7, mode))
    eng.compute_batch_host(L[:3], R[:3], None)   # creates the peers and their buffers
    t0 = time.perf_counter()
    disps = eng.compute_batch_host(L, R, None)
    dt = (time.perf_counter() - t0) / N
    assert all(np.array_equal(disps[i], disps[0]) for i in range(N))
    print(f"mode {mode}: compute_batch of {N} host pairs {dt * 1e3:.1f} ms/pair = {H * W * D / dt / 1e6:.0f} Mdisp/s")

# the same entry in throughput mode (SGM_OPT_SCHEDULE 2: groups of 12 pairs share one chained sweep launch per pass)
from stereo_reconstruction_cv_amd import _lib  # noqa: E402
eng = cv.Engine(bench.sgbm_params(D, 7, 1))
eng.set_option(_lib.SGM_OPT_SCHEDULE, 2)
eng.compute_batch_host(L, R, None)
t0 = time.perf_counter()
disps = eng.compute_batch_host(L, R, None)
dt = (time.perf_counter() - t0) / N
assert all(np.array_equal(disps[i], disps[0]) for i in range(N))
print(f"mode 1, throughput mode: compute_batch of {N} host pairs {dt * 1e3:.1f} ms/pair = {H * W * D / dt / 1e6:.0f} Mdisp/s")
