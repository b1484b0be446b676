#!/usr/bin/env python3
"""First-contact diagnostic for the GPU box: per case, per stage mismatch counts vs the oracle
(continues after failures, unlike pytest -x).  Output goes to stdout; redirect under gpurun_out/."""
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

import parity_util as U  # noqa: E402
from stereo_reconstruction_cv_amd import synth  # noqa: E402
from test_gpu_parity import CASES  # noqa: E402

only = [int(a) for a in sys.argv[1:]]
for i, (H, W, D, bs, minD, mode, seed) in enumerate(CASES):
    if only and i not in only:
        continue
    l, r, _ = synth.make_pair(H, W, D, seed)
    p = U.params(D, bs, minD, mode, speckleWindowSize=30, speckleRange=2)
    t0 = time.time()
    try:
        rep, t, h = U.compare_stages(l, r, p)
        msg = " ".join(f"{k}={n}" for k, n in rep.items())
        print(f"case {i} {(H, W, D, bs, minD, mode)}: {msg}  [{time.time() - t0:.2f}s]", flush=True)
        for k, n in rep.items():
            if n:
                print("    " + U.describe_mismatch(k, h[k], t[k]), flush=True)
                break
    except Exception:
        print(f"case {i} {(H, W, D, bs, minD, mode)}: EXCEPTION", flush=True)
        traceback.print_exc()
