#!/usr/bin/env python3
"""Randomised parity soak on frames wide enough for the large-frame kernels (W1 > 1536: k_prepass3 row chunks,
fused sweeps with a loader wave, bands up to 11 rows; chained sweeps with bands up to 12 rows and 1 .. 40 workgroups in
flight) -- what tests/test_gpu_fuzz.py's tiny frames do not reach.
  gpurun -- 'python tools/soak_medium.py 40'
Every case: all stage taps + final disparity + headroom record against the oracle; exits non-zero on a mismatch."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_util as U  # noqa: E402
from oracle import oracle as O  # noqa: E402
from stereo_reconstruction_cv_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
bad = 0
for seed in range(n):
    rng = np.random.default_rng(7000 + seed)
    D = int(rng.choice([64, 128, 192, 256, 384, 512]))
    bs = int(rng.choice([3, 5, 7, 9]))
    mode = int(rng.integers(0, 2))
    H = int(rng.integers(20, 90))
    W = D + 1540 + int(rng.integers(0, 400))
    P1 = int(rng.integers(1, 8 * bs * bs + 2))
    P2 = P1 + int(rng.integers(1, 30 * bs * bs + 2))
    p = dict(minDisparity=int(rng.integers(-8, 9)), numDisparities=D, blockSize=bs, P1=P1, P2=P2, disp12MaxDiff=1,
             preFilterCap=int(rng.integers(8, 64)), uniquenessRatio=int(rng.choice([0, 10, 15])),
             speckleWindowSize=int(rng.choice([0, 50])), speckleRange=2, mode=mode)
    rows = int(rng.choice([0, 5, 9, 10, 11]))
    chunk = int(rng.choice([0, 8, 16, 24, 40]))
    l, r, _ = synth.make_pair(H, W, D, 100 + seed)
    want, t = O.sgbm_compute(l, r, taps=True, **p)
    if not t["headroom_ok"]:
        print(f"case {seed}: outside the regime, skipped")
        continue
    errs = []
    wgs = int(rng.choice([0, 1, 3, 9, 40]))
    for schedule in (1, 2):     # latency mode (pre-pass + sweeps) and throughput mode (chained sweeps, `wgs` workgroups in flight)
        h = U.run_hip_with_taps(l, r, p, schedule=schedule, sweep_rows=rows if schedule == 1 else int(rng.choice([0, 4, 7, 12])),
                                prepass_rows=chunk if schedule == 1 else 0, chain_wgs=wgs if schedule == 2 else 0)
        errs += [f"s{schedule}:{k}" for k in ("C", "S", "disp_raw", "disp_median") if k in h and k in t and not np.array_equal(h[k], t[k])]
        if not np.array_equal(h["disp"], want):
            errs.append(f"s{schedule}:disp")
        if not U.headroom_equal(h, t):
            errs.append(f"s{schedule}:headroom")
    print(f"case {seed}: {H}x{W} D={D} bs={bs} mode={mode} rows={rows} chunk={chunk} chain_wgs={wgs}: {'OK' if not errs else 'MISMATCH ' + ','.join(errs)}", flush=True)
    bad += bool(errs)
print(f"{n} cases, {bad} with mismatches")
sys.exit(1 if bad else 0)
