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
PROBE = len(sys.argv) > 1 and sys.argv[1] == 'probe'
for mode in (() if PROBE else (0, 1)):
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
for mode in (() if PROBE else (0, 1)):
    eng = cv.get_engine(bench.sgbm_params(D, 7, mode))
    eng.compute_batch_host(L[:3], R[:3], None)   # creates the peers and their buffers
    t0 = time.perf_counter()
    disps = eng.compute_batch_host(L, R, None)
    dt = (time.perf_counter() - t0) / N
    assert all(np.array_equal(disps[i], disps[0]) for i in range(N))
    print(f"mode {mode}: compute_batch of {N} host pairs {dt * 1e3:.1f} ms/pair = {H * W * D / dt / 1e6:.0f} Mdisp/s")

# the same entry in throughput mode (SGM_OPT_SCHEDULE 2): chained groups as large as device memory holds (or SGM_OPT_GROUP_MAX),
# two groups in flight -- uploads of the next and downloads of the previous group beside the kernels of the current one
from stereo_reconstruction_cv_amd import _lib  # noqa: E402


def throughput(N, gmax, with_xyz=False, mode=1):
    eng = cv.Engine(bench.sgbm_params(D, 7, mode))
    eng.set_option(_lib.SGM_OPT_SCHEDULE, 2)
    eng.set_option(_lib.SGM_OPT_GROUP_MAX, gmax)
    Ln, Rn = np.stack([l] * N), np.stack([r] * N)
    eng.compute_batch_host(Ln[:min(N, max(gmax, 2) * 2)], Rn[:min(N, max(gmax, 2) * 2)], Q if with_xyz else None)   # engines, buffers, streams
    best, times = 1e9, []
    res = None
    for it in range(4):     # the first call into fresh arrays (page faults on every map), the others into the same arrays again
        t0 = time.perf_counter()
        res = eng.compute_batch_host(Ln, Rn, Q if with_xyz else None, out=res)
        times.append((time.perf_counter() - t0) / N)
        if it:
            best = min(best, times[-1])
    disps = res[0] if with_xyz else res
    assert all(np.array_equal(disps[i], disps[0]) for i in range(N))
    print(f"mode {mode}, throughput mode: compute_batch of {N} host pairs in groups of <= {gmax or 'auto'}"
          f"{' + XYZ (99.5 MB per pair back)' if with_xyz else ''}: {best * 1e3:.2f} ms/pair = {H * W * D / best / 1e6:.0f} Mdisp/s  (calls: {', '.join(f'{t * 1e3:.2f}' for t in times)}; the first into freshly allocated output arrays)", flush=True)
    del eng


if len(sys.argv) > 1 and sys.argv[1] == "probe":
    throughput(48, 24)
    throughput(48, 0)
    throughput(51, 17)
    sys.exit(0)
throughput(12, 12)
throughput(17, 17)
throughput(34, 17)
throughput(48, 12)
throughput(68, 17)
throughput(34, 17, mode=0)
throughput(17, 17, with_xyz=True)
