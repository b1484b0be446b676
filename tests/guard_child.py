"""Child process of tests/test_gpu_guard.py: parity cases with the engine's GUARDED allocation mode on
(SGM_DEBUG_ALLOC=1, sgm_engine.hip: DevBuf::ensure_guarded).  Every device buffer of every engine then ends exactly at
the end of its mapping with unmapped pages behind and in front of it, and lies where bit 31 of the low address half is
set.  A kernel that reads or writes one byte past a buffer, or that puts a 64-bit pointer together from 32-bit halves
with a sign extension, dies here with a GPU memory access fault -- which ends THIS process, not the test session.
Prints one line `GUARD_OK <cases>` when everything ran and matched the oracle."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p_ in (ROOT, os.path.join(ROOT, "tests")):
    if p_ not in sys.path:
        sys.path.insert(0, p_)

import numpy as np  # noqa: E402


def main():
    assert os.environ.get("SGM_DEBUG_ALLOC") == "1"
    import torch

    import parity_util as U
    from oracle import oracle as O
    from stereo_reconstruction_cv_amd import _lib, synth
    from stereo_reconstruction_cv_amd import stereo as cv
    from stereo_reconstruction_cv_amd.stereo import Engine

    ncase = 0

    def stages(H, W, D, bs, mode, seed, schedule, rows=0, minD=0, **kw):
        nonlocal ncase
        l, r, _ = synth.make_pair(H, W, D, seed)
        p = U.params(D, bs, minD, mode, speckleWindowSize=30, speckleRange=2, **kw)
        rep, t, h = U.compare_stages(l, r, p, schedule=schedule, sweep_rows=rows)
        bad = [U.describe_mismatch(k, h[k], t[k]) for k, n in rep.items() if n]
        assert not bad, f"{(H, W, D, bs, mode, schedule)}: " + "\n".join(bad)
        ncase += 1

    # ---- k_pix: the last chunk of a row holds 1, 2, 3, 5 or 127 columns (W1 % 128), for every lane packing: the last
    # row's last left-pixel record is the last 8 bytes of the record buffer, which ends at the end of its mapping
    for NP, D in ((1, 128), (1, 80), (2, 256), (2, 160), (4, 512), (4, 336)):
        for rem in (1, 2, 3, 5, 127):
            W1 = 128 * (2 if NP < 4 else 1) + rem
            stages(9, W1 + D, D, 5, 1, 7000 + 10 * D + rem, schedule=1)
    for rem in (1, 3, 127):       # ... the same behind chained sweeps, and a single chunk shorter than four columns
        stages(26, 256 + rem + 128, 128, 7, 0, 7100 + rem, schedule=2, rows=4)
    for W1 in (1, 2, 3, 4, 6):
        stages(11, W1 + 96, 96, 3, 1, 7200 + W1, schedule=1)
    # ---- the other routes through the engine: small-D kernels, int16 cost pipeline (window 13, preFilterCap 127), D = 48
    # in both schedules, negative / positive minDisparity, one-kernel-per-direction schedule
    for (H, W, D, bs, mode, sched, kw) in ((33, 200, 16, 11, 0, 1, {}), (30, 260, 32, 5, 1, 1, {}), (40, 300, 64, 5, 0, 1, {}),
                                            (40, 300, 64, 5, 0, 2, {}), (31, 333, 48, 3, 1, 2, {}), (28, 400, 128, 13, 1, 1, {}),
                                            (24, 300, 64, 7, 0, 1, dict(preFilterCap=127)), (20, 420, 256, 7, 1, 0, {})):
        stages(H, W, D, bs, mode, 7300 + H + D, schedule=sched, **kw)
    stages(36, 260, 32, 5, 0, 7400, schedule=1, minD=-6)
    stages(36, 260, 32, 5, 1, 7401, schedule=1, minD=3)
    # ---- the test that met round 3's fault, at a smaller size: a chained batch of six on one engine while a second engine
    # runs latency-mode frames beside it (seven engines' worth of buffers, all at addresses with bit 31 set)
    H, W, D = 70, 1500, 128
    p = U.params(D, 7, 0, 1, speckleWindowSize=30, speckleRange=2)
    pairs = [synth.make_pair(H, W, D, 7500 + i)[:2] for i in range(3)]
    wants = [O.sgbm_compute(a, b, **p) for a, b in pairs]
    dev = torch.device("cuda", 0)
    n = 6
    dl = [torch.from_numpy(pairs[i % 3][0]).to(dev) for i in range(n)]
    dr = [torch.from_numpy(pairs[i % 3][1]).to(dev) for i in range(n)]
    dd = [torch.full((H, W), -9, dtype=torch.int16, device=dev) for _ in range(n)]
    other = [torch.full((H, W), -9, dtype=torch.int16, device=dev) for _ in range(4)]
    torch.cuda.synchronize()
    a = Engine(p)
    a.set_option(_lib.SGM_OPT_SCHEDULE, 2)
    b = Engine(p)
    ptr = lambda ts: [t.data_ptr() for t in ts]
    for k in range(2):
        b.compute_device(dl[k].data_ptr(), dr[k].data_ptr(), H, W, W, other[k].data_ptr())
    a.pipeline_batch_device(ptr(dl), ptr(dr), H, W, W, None, ptr(dd))
    for k in range(2, 4):
        b.compute_device(dl[k % 3].data_ptr(), dr[k % 3].data_ptr(), H, W, W, other[k].data_ptr())
    a.synchronize()
    b.synchronize()
    for i in range(n):
        assert np.array_equal(dd[i].cpu().numpy(), wants[i % 3]), i
    for k in range(4):
        assert np.array_equal(other[k].cpu().numpy(), wants[k % 3]), k
    ncase += 1
    # ---- epilogue entry points on guarded buffers: reprojection, mask, compaction with colours, median, speckles, rectification
    rng = np.random.default_rng(5)
    d16 = rng.integers(-16, 700, (37, 53)).astype(np.int16)
    e = cv.get_engine(dict(numDisparities=16))
    f = e.disp_to_float_host(d16)
    Q = synth.default_Q(53)
    xyz = cv.reprojectImageTo3D(f, Q)
    ref = O.reproject(O.disp_to_float(d16), Q)
    fin = np.isfinite(ref)
    assert np.array_equal(np.isfinite(xyz), fin) and np.array_equal(xyz[fin], ref[fin])
    mask = e.valid_mask_host(xyz, f)
    assert np.array_equal(mask, O.valid_mask(ref, O.disp_to_float(d16)))
    rgb = rng.integers(0, 256, (37, 53, 3)).astype(np.uint8)
    pts, cols = e.compact_points_host(xyz, f, rgb)
    assert np.array_equal(pts, xyz[mask]) and np.array_equal(cols, rgb[mask])
    assert np.array_equal(e.median3x3_host(d16), O.median3x3(d16))
    assert np.array_equal(e.filter_speckles_host(d16, -16, 20, 32), O.filter_speckles(d16, -16, 20, 32))
    K = np.array([[50.0, 0, 26.2], [0, 51.0, 18.4], [0, 0, 1]])
    m1, m2 = cv.initUndistortRectifyMap(K, np.array([0.05, -0.01, 0.001, 0.002, 0.0]), None, K, (53, 37), cv.CV_32F)
    w1, w2 = O.init_undistort_rectify_map(K, np.array([0.05, -0.01, 0.001, 0.002, 0.0]), None, K, (53, 37))
    assert np.array_equal(m1, w1) and np.array_equal(m2, w2)
    img = rng.integers(0, 256, (37, 53)).astype(np.uint8)
    assert np.array_equal(cv.remap(img, m1, m2, cv.INTER_LINEAR), O.remap_linear(img, w1, w2))
    ncase += 1
    print(f"GUARD_OK {ncase}", flush=True)


if __name__ == "__main__":
    main()
