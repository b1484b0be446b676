#!/usr/bin/env python3
"""Measurement of the rectification row on one GPU: map generation and the
bilinear remap of a 4K pair, inputs resident in HBM, HIP events via torch on the engine's stream.
Prints one JSON line in the shape of bench.py's (metric, roofline, cpu_baseline).

    python tools/rectify_bench.py [--width 3840 --height 2160 --channels 1 --steps 50]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--channels", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()
    import torch
    import rectify_cases as RC
    import stereo_reconstruction_cv_amd as cv
    W, H, cn = a.width, a.height, a.channels
    K = RC.camera(W, H)
    R1, R2 = RC.rodrigues([0.004, -0.006, 0.002]), RC.rodrigues([-0.003, 0.005, -0.001])
    P = RC.camera(W, H, f=0.93 * W)
    eng = cv.Engine({"numDisparities": 16})
    dev = torch.device("cuda", 0)
    imgs = [torch.from_numpy(RC.image(H, W, cn, s)).to(dev) for s in (1, 2)]
    outs = [torch.empty_like(i) for i in imgs]
    maps = [[torch.empty((H, W), dtype=torch.float32, device=dev) for _ in range(2)] for _ in range(2)]
    torch.cuda.synchronize()

    def make_maps():
        for (m1, m2), R in zip(maps, (R1, R2)):
            eng.init_undistort_rectify_map_device(K, None, R, P, W, H, m1.data_ptr(), m2.data_ptr())

    def remap_pair():
        for img, out, (m1, m2) in zip(imgs, outs, maps):
            eng.remap_linear_device(img.data_ptr(), H, W, W * cn, cn, m1.data_ptr(), m2.data_ptr(), H, W, out.data_ptr(), W * cn)

    def timed(fn, steps, warmup):
        for _ in range(warmup):
            fn()
        eng.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        eng.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

    ms_maps = timed(make_maps, max(3, a.steps // 10), 1)
    ms_remap = timed(remap_pair, a.steps, a.warmup)
    npx = 2 * H * W
    # algorithmic bytes per destination pixel: two float map entries + one source pixel + one result
    alg = npx * (8 + 2 * cn)
    out = {
        "metric": "Mpixels/s remapped (cv2.remap INTER_LINEAR of a stereo pair)", "value": npx / ms_remap / 1e3,
        "unit": "Mpixels/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_remap,
        "higher_is_better": True, "dtype": "u8 (1/32-pixel fixed point, int32 accumulate)", "data": "synthetic",
        "config": {"workload": f"rectify: 2 x {W}x{H}x{cn} uint8 through float32 map pairs", "maps_ms_per_pair": ms_maps},
        "roofline": {"bound": "hbm", "kernel": f"k_remap_linear<{cn}>", "achieved": alg / ms_remap / 1e6, "peak": 8000.0,
                     "unit": "GB/s", "frac": alg / ms_remap / 1e6 / 8000.0, "traffic": None,
                     "algorithmic_bytes_per_launch": alg // 2},
    }
    if not a.no_cpu_baseline:
        from oracle import oracle as O
        m1 = maps[0][0].cpu().numpy()
        m2 = maps[0][1].cpu().numpy()
        img = imgs[0].cpu().numpy()
        t0 = time.perf_counter()
        want = O.remap_linear(img, m1, m2)
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": H * W / dt / 1e6, "unit": "Mpixels/s", "cores": 1, "kind": "port",
                               "sample": f"one {W}x{H}x{cn} image, {dt:.2f} s on one host core (C restatement, not cv2)"}
        out["verify"] = {"remap_mismatches": int((outs[0].cpu().numpy() != want).sum()),
                         "maps_equal_oracle": bool(np.array_equal(m1, O.init_undistort_rectify_map(K, None, R1, P, (W, H))[0]))}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
