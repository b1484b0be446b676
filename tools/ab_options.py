#!/usr/bin/env python3
"""In-process A/B of engine debug variants (same GPU, interleaved rounds): stage times per variant."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import stereo_reconstruction_cv_amd as cv  # noqa: E402
from stereo_reconstruction_cv_amd import _lib, synth  # noqa: E402

mode = int(sys.argv[1]) if len(sys.argv) > 1 else 1
variants = [int(v) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["0", "16"])]
H, W, D, bs = int(os.environ.get('AB_H', 2160)), int(os.environ.get('AB_W', 3840)), int(os.environ.get('AB_D', 256)), 7
l, r, _ = synth.make_pair(H, W, D, 1234)
dl, dr = torch.from_numpy(l).cuda(), torch.from_numpy(r).cuda()
out = torch.empty((H, W), dtype=torch.int16, device="cuda")
engs = {}
for v in variants:
    e = cv.Engine(bench.sgbm_params(D, bs, mode))
    e.set_option(_lib.SGM_OPT_PROFILE, 1)
    e.set_option(_lib.SGM_OPT_DEBUG, v % 1000)       # variant = debug + 1000 * sweep rows
    e.set_option(_lib.SGM_OPT_SWEEP_ROWS, v // 1000)
    engs[v] = e
acc = {v: [] for v in variants}
vnames = {}
for rnd in range(6):
    for v in variants:
        engs[v].compute_device(dl.data_ptr(), dr.data_ptr(), H, W, W, out.data_ptr())
        st = engs[v].stage_times()
        if rnd > 0:
            acc[v].append([m for _, m, _ in st])
        vnames[v] = [n for n, _, _ in st]   # stage lists differ between variants
for v in variants:
    names = vnames[v]
    a = np.median(np.array(acc[v]), axis=0)
    wall = a[names.index("_wall")] if "_wall" in names else a.sum()
    print(f"debug={v:3d} wall {wall:6.2f} ms  " + " ".join(f"{n}={m:.2f}" for n, m in zip(names, a) if m > 0.25 and n != "_wall"), flush=True)
