"""The one cv2-made artefact the reference holds for the hot path: the figure of cell c13 of
/root/reference/main.ipynb (`plt.imshow(disparity_map, cmap='jet'); plt.colorbar()`, main.ipynb:783-787), a
770x417 PNG rendering of cv2.StereoSGBM's map of dataset/d3 at numDisparities=16, blockSize=11 (main.ipynb:655-666,
:781), embedded in the notebook as display_data.  Build container only (/root/reference is not on the GPU box):

    python tests/golden/make_notebook_figure.py

writes tests/golden/notebook_c13_axes.png -- the image area inside the axes (347 x 619 pixels, RGB), located by the
black axes frame -- which tests/test_oracle_vs_notebook_figure.py compares with the same rendering of the ORACLE's map
of the same pair.  It is data the reference's notebook holds (an output image), not source text.
"""
import base64
import io
import json
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
NOTEBOOK = "/root/reference/main.ipynb"


def figure_png():
    nb = json.load(open(NOTEBOOK))
    cell = nb["cells"][13]
    assert "compute_disparity_map(imgL, imgR, 16, 0)" in "".join(cell["source"])
    for o in cell["outputs"]:
        if o.get("output_type") == "display_data" and "image/png" in o["data"]:
            return Image.open(io.BytesIO(base64.b64decode(o["data"]["image/png"]))).convert("RGB")
    raise SystemExit("cell c13 holds no PNG")


def axes_box(rgb):
    """(top, bottom, left, right) of the image area: inside the long black lines of the axes frame"""
    dark = rgb.astype(int).sum(axis=2) < 60
    rows = np.where(dark.sum(axis=1) > rgb.shape[1] // 2)[0]
    cols = np.where(dark.sum(axis=0) > rgb.shape[0] // 2)[0]
    return rows.min() + 1, rows.max(), cols.min() + 1, cols[cols < rgb.shape[1] * 0.9].max()


def main():
    im = np.asarray(figure_png())
    t, b, l, r = axes_box(im)
    crop = im[t:b, l:r]
    print("figure", im.shape, "axes area rows", (t, b), "cols", (l, r), "->", crop.shape)
    Image.fromarray(crop).save(os.path.join(HERE, "notebook_c13_axes.png"), optimize=True)
    json.dump({"figure_size": [int(im.shape[1]), int(im.shape[0])], "axes_box_tblr": [int(t), int(b), int(l), int(r)]},
              open(os.path.join(HERE, "notebook_c13_axes.json"), "w"))


if __name__ == "__main__":
    main()
