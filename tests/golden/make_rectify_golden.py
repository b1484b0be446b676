#!/usr/bin/env python3
"""Generates tests/golden/rectify_golden.npz with the CPU oracle (run from the repo root).

Regression anchors for the rectification step (gui.py:160-164): the reference holds no fixtures
for it and cv2 is not importable here ("parity unpinned"), so these are the oracle's own outputs:
camera matrices in, SHA-256 of the two float maps + their first and last rows, an input image and
its remapped result."""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as O  # noqa: E402
import rectify_cases as RC  # noqa: E402


def main():
    out = {}
    for name, K, dist, R, P, (W, H) in RC.cases():
        if name not in ("rotated", "rotated_dist5", "dist12"):
            continue
        m1, m2 = O.init_undistort_rectify_map(K, dist, R, P, (W, H))
        out[f"{name}/K"] = K
        out[f"{name}/dist"] = np.zeros(0) if dist is None else np.asarray(dist, np.float64)
        out[f"{name}/R"] = np.zeros((0, 3)) if R is None else R
        out[f"{name}/P"] = np.zeros((0, 3)) if P is None else np.asarray(P, np.float64)
        out[f"{name}/size"] = np.array([W, H], np.int32)
        out[f"{name}/sha_map1"] = np.frombuffer(hashlib.sha256(m1.tobytes()).digest(), np.uint8)
        out[f"{name}/sha_map2"] = np.frombuffer(hashlib.sha256(m2.tobytes()).digest(), np.uint8)
        out[f"{name}/map1_rows"] = m1[[0, H - 1]]
        out[f"{name}/map2_rows"] = m2[[0, H - 1]]
        for cn in (1, 3):
            img = RC.image(H, W, cn, 40 + cn)
            out[f"{name}/img{cn}"] = img
            out[f"{name}/remap{cn}"] = O.remap_linear(img, m1, m2)
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "rectify_golden.npz"), **out)


if __name__ == "__main__":
    main()
