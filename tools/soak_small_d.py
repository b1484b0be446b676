#!/usr/bin/env python3
"""Randomised parity soak of the small-D kernels (D = 16 .. 64: kernels_group.h -- k_paths5_g, k_prepass3_g + k_vert3_g,
k_rows_g in lane groups -- and k_box_u8 in lane groups): frames from one row / three columns to a few hundred each way,
taller than wide (line walks that wrap several times) and wider than tall, both modes, the byte and the int16 cost
pipeline, the record form (debug 8192), the in-row path on the main stream (debug 4096), no lane groups (debug 4),
latency and throughput mode.
  gpurun -- 'python tools/soak_small_d.py 200'
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

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
bad = skipped = 0
for seed in range(n):
    rng = np.random.default_rng(9000 + seed)
    D = int(rng.choice([16, 16, 32, 48, 64, 64]))
    bs = int(rng.choice([1, 3, 5, 7, 9, 11, 13]))
    mode = int(rng.integers(0, 2))
    shape = int(rng.integers(0, 4))
    if shape == 0:      # narrow and tall
        H, W = int(rng.integers(40, 400)), D + int(rng.integers(3, 40))
    elif shape == 1:    # wide and low
        H, W = int(rng.integers(1, 12)), D + int(rng.integers(100, 900))
    else:
        H, W = int(rng.integers(12, 200)), D + int(rng.integers(8, 500))
    P1 = int(rng.integers(1, 8 * bs * bs + 2))
    P2 = P1 + int(rng.integers(1, 30 * bs * bs + 2))
    p = dict(minDisparity=int(rng.integers(-8, 9)), numDisparities=D, blockSize=bs, P1=P1, P2=P2, disp12MaxDiff=int(rng.choice([-1, 0, 1, 3])),
             preFilterCap=int(rng.integers(4, 64)), uniquenessRatio=int(rng.choice([0, 10, 15, 100])),
             speckleWindowSize=int(rng.choice([0, 30])), speckleRange=int(rng.choice([1, 2])), mode=mode)
    l, r, _ = synth.make_pair(H, W, D, 500 + seed)
    want, t = O.sgbm_compute(l, r, taps=True, **p)
    if not t["headroom_ok"]:
        skipped += 1
        continue
    errs = []
    dbg = int(rng.choice([0, 0, 8192, 4096, 256, 4, 8192 | 256]))
    for schedule, debug in ((1, 0), (1, dbg), (2, 0)):
        h = U.run_hip_with_taps(l, r, p, schedule=schedule, debug=debug)
        errs += [f"s{schedule}/{debug}:{k}" for k in ("C", "S", "disp_raw", "disp_median") if k in h and k in t and not np.array_equal(h[k], t[k])]
        if not np.array_equal(h["disp"], want):
            errs.append(f"s{schedule}/{debug}:disp")
        if not U.headroom_equal(h, t):
            errs.append(f"s{schedule}/{debug}:headroom")
    print(f"case {seed}: {H}x{W} D={D} bs={bs} mode={mode} minD={p['minDisparity']} debug={dbg}: {'OK' if not errs else 'MISMATCH ' + ','.join(errs)}", flush=True)
    bad += bool(errs)
print(f"{n} cases, {skipped} outside the regime (skipped), {bad} with mismatches")
sys.exit(1 if bad else 0)
