#!/usr/bin/env python3
"""Timing ablations of k_sweep (GPU box): which resource bounds the lockstep step?"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import stereo_reconstruction_cv_amd as cv  # noqa: E402
from stereo_reconstruction_cv_amd import _lib, synth  # noqa: E402

H, W, D, bs = 2160, 3840, 256, 7
l, r, _ = synth.make_pair(H, W, D, 1234)
dl, dr = torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda()
out = torch.empty((H, W), dtype=torch.int16, device="cuda")
names = {1: "noSstore", 2: "noLoads", 4: "noLDS", 8: "noBarrier"}
for rows in (9, 5):
    for dbg in (0, 1, 2, 4, 8, 3, 12, 15):
        eng = cv.Engine(bench.sgbm_params(D, bs, 0))
        eng.set_option(_lib.SGM_OPT_PROFILE, 1)
        eng.set_option(_lib.SGM_OPT_DEBUG, dbg)
        eng.set_option(_lib.SGM_OPT_SWEEP_ROWS, rows)
        for _ in range(3):
            eng.compute_device(dl.data_ptr(), dr.data_ptr(), H, W, W, out.data_ptr())
            st = dict((n, m) for n, m, _ in eng.stage_times())
        tag = "+".join(v for k, v in names.items() if dbg & k) or "baseline"
        print(f"rows={rows} dbg={dbg:2d} {tag:32s} sweep_dn={st['sweep_dn']:.3f} ms prepass={st['prepass_dn']:.3f}", flush=True)
        del eng
