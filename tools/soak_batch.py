#!/usr/bin/env python3
"""Randomised soak of the batch entries in throughput mode (round 4's new host code: groups sized from memory or
SGM_OPT_GROUP_MAX, batches larger than a group, the host entry with two groups in flight, headroom over a batch, trim):
random shapes, disparity ranges, modes, band heights, batch sizes and group caps; every map (and XYZ image where asked
for) against the oracle's for its own pair, the headroom record against the maximum over the batch.
  gpurun -- 'python tools/soak_batch.py 150'          exits non-zero on the first mismatch"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from oracle import oracle as O  # noqa: E402
from stereo_reconstruction_cv_amd import _lib, synth  # noqa: E402
from stereo_reconstruction_cv_amd.stereo import Engine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda", 0)
engines = {}
for case in range(n):
    rng = np.random.default_rng(91000 + case)
    D = int(rng.choice([48, 64, 96, 128, 160, 256, 320, 512]))
    bs = int(rng.choice([3, 5, 7, 9, 11]))
    mode = int(rng.integers(0, 2))
    H = int(rng.integers(14, 100))
    W = D + int(rng.integers(40, 700))
    N = int(rng.integers(2, 11))
    gmax = int(rng.choice([0, 0, 2, 3, 5]))
    rows = int(rng.choice([0, 0, 2, 3, 5, 12]))
    entry = str(rng.choice(["device", "host", "host_xyz", "device_xyz"]))
    P1 = int(rng.integers(1, 8 * bs * bs + 2))
    P2 = P1 + int(rng.integers(1, 24 * bs * bs + 2))
    p = dict(minDisparity=int(rng.integers(-4, 5)), numDisparities=D, blockSize=bs, P1=P1, P2=P2, disp12MaxDiff=1, preFilterCap=int(rng.choice([31, 63])),
             uniquenessRatio=int(rng.choice([0, 10, 15])), speckleWindowSize=int(rng.choice([0, 30, 100])), speckleRange=int(rng.choice([1, 2, 32])), mode=mode)
    nu = min(N, 3)
    pairs = [synth.make_pair(H, W, D, 92000 + 10 * case + i)[:2] for i in range(nu)]
    orc = [O.sgbm_compute(a, b, taps="light", **p) for a, b in pairs]
    if not all(t["headroom_ok"] for _, t in orc):
        print(f"case {case}: outside the regime, skipped", flush=True)
        continue
    Q = synth.default_Q(W)
    key = tuple(sorted(p.items()))
    eng = engines.get(key)
    if eng is None:
        if len(engines) > 6:
            engines.clear()
        eng = engines[key] = Engine(p)
    eng.set_option(_lib.SGM_OPT_SCHEDULE, 2)
    eng.set_option(_lib.SGM_OPT_SWEEP_ROWS, rows)
    eng.set_option(_lib.SGM_OPT_GROUP_MAX, gmax)
    with_q = entry.endswith("xyz")
    if entry.startswith("host"):
        L = np.stack([pairs[i % nu][0] for i in range(N)])
        R = np.stack([pairs[i % nu][1] for i in range(N)])
        res = eng.compute_batch_host(L, R, Q if with_q else None)
        disps, xyz = res if with_q else (res, None)
    else:
        dl = [torch.from_numpy(pairs[i % nu][0]).to(dev) for i in range(N)]
        dr = [torch.from_numpy(pairs[i % nu][1]).to(dev) for i in range(N)]
        dd = [torch.full((H, W), -7, dtype=torch.int16, device=dev) for _ in range(N)]
        df = [torch.empty((H, W), dtype=torch.float32, device=dev) for _ in range(N)] if with_q else None
        dx = [torch.empty((H, W, 3), dtype=torch.float32, device=dev) for _ in range(N)] if with_q else None
        torch.cuda.synchronize()
        ptr = lambda ts: [t.data_ptr() for t in ts] if ts is not None else None
        eng.pipeline_batch_device(ptr(dl), ptr(dr), H, W, W, Q if with_q else None, ptr(dd), ptr(df), ptr(dx))
        eng.synchronize()
        disps = [t.cpu().numpy() for t in dd]
        xyz = [t.cpu().numpy() for t in dx] if with_q else None
    bad = 0
    for i in range(N):
        want = orc[i % nu][0]
        bad += int((disps[i] != want).sum())
        if with_q:
            ref = O.reproject(O.disp_to_float(want), Q)
            fin = np.isfinite(ref)
            bad += int(not np.array_equal(np.isfinite(xyz[i]), fin)) + int((xyz[i][fin] != ref[fin]).sum())
    hr = eng.headroom()
    want_hr = dict(ok=True, max_cost_plus_p2=max(t["max_cost_plus_p2"] for _, t in orc), max_delta=max(t["max_delta"] for _, t in orc))
    bad += int(hr != want_hr)
    print(f"case {case}: {N} x {H}x{W} D={D} bs={bs} mode={mode} rows={rows} gmax={gmax} {entry}: {'OK' if not bad else 'MISMATCH ' + str(bad) + ' ' + str(hr) + ' ' + str(want_hr)}", flush=True)
    if bad:
        sys.exit(1)
    if case % 7 == 6:
        eng.trim()
print(f"{n} cases, 0 with mismatches")
