"""Builds tests/golden/real_pairs.npz from the reference's three stereo pairs (build container only:
/root/reference does not exist on the GPU box).

    python tests/golden/make_real_pairs.py

Inputs: /root/reference/dataset/d{1,2,3}/img{1,2}.jpg (the pairs the notebook reads at
main.ipynb:358-363), decoded with Pillow and converted to 8-bit gray (Pillow's ITU-R 601 rounding,
not OpenCV's imread -- these are inputs, not expected outputs).  To stay near 1 MB the fixture
holds a full-resolution crop of d2 and box-downsampled whole frames of d1 and d3 (downsampling
sharpens the per-pixel contrast, i.e. it stresses the int16 headroom more than the originals).

Expected outputs are the CPU oracle's (oracle/sgbm_oracle.c; parity unpinned against cv2, see its
header): for each pair and each setting the headroom record and a sha256 of the int16 disparity
map -- at the notebook's setting (numDisparities=16, blockSize=11, main.ipynb:655-666, :781) and at
the C2 / C3 parameter sets of BASELINE.json (D=128 / D=256 MODE_HH, blockSize=7).
"""
import hashlib
import json
import os
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

REF = "/root/reference/dataset"
NB = dict(disp12MaxDiff=1, preFilterCap=63, uniquenessRatio=10, speckleWindowSize=100, speckleRange=32)


def settings():
    def p(D, bs, mode):
        return dict(minDisparity=0, numDisparities=D, blockSize=bs, P1=8 * 3 * bs * bs, P2=32 * 3 * bs * bs, mode=mode, **NB)
    return {"notebook_D16_bs11": p(16, 11, 0), "c2_D128_bs7": p(128, 7, 0), "c3_D256_bs7_HH": p(256, 7, 1)}


def gray(path):
    return Image.open(path).convert("L")


def main():
    imgs = {}
    for name, how in (("d1", ("box", 8)), ("d2", ("crop", (440, 600, 200, 704))), ("d3", ("box", 6))):
        for i in (1, 2):
            im = gray(f"{REF}/{name}/img{i}.jpg")
            if how[0] == "box":
                f = how[1]
                im = im.resize((im.size[0] // f, im.size[1] // f), Image.BOX)
                a = np.asarray(im, dtype=np.uint8)
            else:
                y0, x0, h, w = how[1]
                a = np.asarray(im, dtype=np.uint8)[y0:y0 + h, x0:x0 + w]
            imgs[f"{name}_img{i}"] = np.ascontiguousarray(a)
    expected = {}
    for name in ("d1", "d2", "d3"):
        l, r = imgs[f"{name}_img1"], imgs[f"{name}_img2"]
        for sname, p in settings().items():
            d, t = O.sgbm_compute(l, r, taps="light", **p)
            expected[f"{name}/{sname}"] = dict(
                params=p, shape=list(l.shape), headroom_ok=bool(t["headroom_ok"]),
                max_cost_plus_p2=int(t["max_cost_plus_p2"]), max_delta=int(t["max_delta"]),
                disp_sha256=hashlib.sha256(np.ascontiguousarray(d).tobytes()).hexdigest(),
                valid_fraction=float((d >= 0).mean()))
            print(name, sname, l.shape, expected[f"{name}/{sname}"]["headroom_ok"],
                  expected[f"{name}/{sname}"]["max_cost_plus_p2"], expected[f"{name}/{sname}"]["max_delta"],
                  round(expected[f"{name}/{sname}"]["valid_fraction"], 3))
    out = os.path.join(HERE, "real_pairs.npz")
    np.savez_compressed(out, expected=np.frombuffer(json.dumps(expected).encode(), dtype=np.uint8), **imgs)
    print(out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
