#!/usr/bin/env python3
"""Per-stage HIP-event timings of both schedules and both modes at a given shape (GPU box)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import stereo_reconstruction_cv_amd as cv  # noqa: E402
from stereo_reconstruction_cv_amd import _lib, synth  # noqa: E402

H, W, D, bs = (int(a) for a in (sys.argv[1:5] if len(sys.argv) >= 5 else (2160, 3840, 256, 7)))
scheds = [int(a) for a in sys.argv[5].split(",")] if len(sys.argv) > 5 else [0, 1]
l, r, _ = synth.make_pair(H, W, D, 1234)
dl, dr = torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda()
out = torch.empty((H, W), dtype=torch.int16, device="cuda")
for mode in (0, 1):
    for sched in scheds:
        eng = cv.Engine(bench.sgbm_params(D, bs, mode))
        eng.set_option(_lib.SGM_OPT_PROFILE, 1)
        eng.set_option(_lib.SGM_OPT_SCHEDULE, sched)
        for _ in range(4):
            eng.compute_device(dl.data_ptr(), dr.data_ptr(), H, W, W, out.data_ptr())
            st = eng.stage_times()
        print(f"{H}x{W} D={D} mode {mode} sched {sched}: total {sum(m for _, m, _ in st):.2f} ms  "
              + " ".join(f"{n}={m:.2f}" for n, m, _ in st if m > 0.25), flush=True)
        del eng
