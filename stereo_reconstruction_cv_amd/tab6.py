"""Tab 6 "Run Disparity" for the reference's Tk GUI, backed by the HIP engine.

The mounted snapshot of the reference creates Tabs 1-4 only (/root/reference/gui.py:367-377); its
README (README.md:82-83, 104) describes a Tab 6 "Run Disparity" on other branches.  This module
is that tab in the shape every tab of gui.py has (create_stereo_rect_tab / run_stereo_rect,
gui.py:422-487):

  * a module-level pipeline function taking the pair folder (img1.jpg / img2.jpg, gui.py:96-100),
    returning a dict of display images or an error STRING (gui.py:99-100, 476);
  * a controls frame with a folder entry + Browse button, numeric entries validated with
    messagebox.showerror and a default (gui.py:466-472), and a Run button;
  * a blocking call on the Tk main thread, the result kept on `self` (gui.py:362-365, 474);
  * images shown as tk.PhotoImage(data=<png bytes>) with a reference kept on the label
    (gui.py:483-487) -- PNG-encoded with Pillow here instead of cv2.imencode.

The compute is the notebook's driver cell c13 (main.ipynb:780-797) through
pipeline.run_disparity -> C ABI -> HIP kernels.  tkinter is imported lazily (and can be
injected), so the module imports and is testable on a box without a display.

    from stereo_reconstruction_cv_amd.tab6 import DisparityTab
    app = NotebookGUI(root)                       # the reference's class, gui.py:325
    app.disparity_tab = DisparityTab(app.notebook, owner=app)     # adds "Run Disparity"
"""
from __future__ import annotations

import glob
import io
import os

import numpy as np

from . import pipeline as _pipeline
from . import synth as _synth

DEFAULT_NDISP = 16      # main.ipynb:781: compute_disparity_map(imgL, imgR, 16, 0)
DEFAULT_MINDISP = 0
DISPLAY_SIZE = (640, 360)   # the GUI resizes its display images to 640x360 (gui.py:197-200)


def _jet(v: np.ndarray) -> np.ndarray:
    """matplotlib's 'jet' (the notebook shows the map with cmap='jet', main.ipynb:783-787), v in [0, 1]."""
    v = np.clip(v, 0.0, 1.0)
    r = np.clip(1.5 - np.abs(4.0 * v - 3.0), 0.0, 1.0)
    g = np.clip(1.5 - np.abs(4.0 * v - 2.0), 0.0, 1.0)
    b = np.clip(1.5 - np.abs(4.0 * v - 1.0), 0.0, 1.0)
    return (np.stack([r, g, b], axis=-1) * 255.0 + 0.5).astype(np.uint8)


def disparity_to_rgb(disparity_map: np.ndarray, ndisp: int) -> np.ndarray:
    """uint8 (H, W, 3) rendering of the float disparity map, 0 .. ndisp-1 over the jet colormap."""
    d = np.asarray(disparity_map, dtype=np.float32)
    return _jet(d / float(max(ndisp - 1, 1)))


def _load_gray(path: str) -> np.ndarray:
    from PIL import Image   # Pillow stands in for cv2.imread(..., IMREAD_GRAYSCALE) (gui.py:102-103)
    return np.ascontiguousarray(np.asarray(Image.open(path).convert("L"), dtype=np.uint8))


def _resize_rgb(img: np.ndarray, size) -> np.ndarray:
    from PIL import Image
    return np.asarray(Image.fromarray(img).resize(size, Image.BILINEAR))


def png_bytes(img: np.ndarray) -> bytes:
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(img).save(buf, format="PNG")
    return buf.getvalue()


def run_disparity_folder(stereo_path, ndisp=DEFAULT_NDISP, mindisp=DEFAULT_MINDISP, Q=None, runner=None):
    """The tab's pipeline function (the counterpart of stereo_rect(), gui.py:92-209).

    Returns an error string (gui.py:99-100) or a dict: display images under the keys of
    DisparityTab.IMG_TITLES plus the raw results ("disparity_map" float32 (H, W), "points_3D"
    float32 (H, W, 3) or None, "mask" bool (H, W) or None) for later tabs."""
    left_image = glob.glob(f"{stereo_path}/img1.jpg")
    right_image = glob.glob(f"{stereo_path}/img2.jpg")
    if not left_image or not right_image:
        return "Error: Missing img1.jpg or img2.jpg in the folder."
    imgL = _load_gray(left_image[0])
    imgR = _load_gray(right_image[0])
    if imgL.shape != imgR.shape:
        return "Error: img1.jpg and img2.jpg differ in size."
    if Q is None:
        Q = _synth.default_Q(imgL.shape[1])   # the notebook's Q (main.ipynb:600-607), scaled to the frame width
    try:
        disparity_map, points_3D, mask = (runner or _pipeline.run_disparity)(imgL, imgR, Q, ndisp, mindisp)
    except Exception as e:  # noqa: BLE001 - the GUI shows errors as text, it never raises into Tk's mainloop
        return f"Error: {e}"
    left_rgb = np.repeat(imgL[:, :, None], 3, axis=2)
    return {
        "Left Image": _resize_rgb(left_rgb, DISPLAY_SIZE),
        "Disparity Map": _resize_rgb(disparity_to_rgb(disparity_map, ndisp), DISPLAY_SIZE),
        "disparity_map": disparity_map,
        "points_3D": points_3D,
        "mask": mask,
        "Q": np.asarray(Q, dtype=np.float64),
    }


def _tk_modules():
    import tkinter as tk
    from tkinter import filedialog, messagebox, ttk
    return tk, ttk, filedialog, messagebox


class DisparityTab:
    """One more tab on the reference's ttk.Notebook; same anatomy as its Tabs 2-4."""

    IMG_TITLES = ("Left Image", "Disparity Map")

    def __init__(self, notebook, owner=None, toolkit=None, runner=None):
        """notebook: the ttk.Notebook of NotebookGUI (gui.py:358); owner: that NotebookGUI (results
        are mirrored to owner.disparity_results, as Tabs 1-4 do with their *_results attributes);
        toolkit: (tk, ttk, filedialog, messagebox) -- defaults to tkinter, injected by the tests."""
        self.tk, self.ttk, self.filedialog, self.messagebox = toolkit or _tk_modules()
        self.notebook = notebook
        self.owner = owner
        self.runner = runner
        self.disparity_results = None
        self.create_disparity_tab()

    # -- gui.py:422-454 --
    def create_disparity_tab(self):
        tk, ttk = self.tk, self.ttk
        tab6 = ttk.Frame(self.notebook)
        self.notebook.add(tab6, text="Run Disparity")
        self.tab = tab6

        controls_frame = ttk.Frame(tab6)
        controls_frame.grid(row=0, column=0, columnspan=2, padx=5, pady=5, sticky="ew")

        ttk.Label(controls_frame, text="Stereo Image Pair Folder Path:").grid(row=0, column=0, padx=5, pady=5, sticky="e")
        self.disparity_path_entry = ttk.Entry(controls_frame, width=50)
        self.disparity_path_entry.grid(row=0, column=1, padx=5, pady=5)

        browse_btn = ttk.Button(controls_frame, text="Browse", command=self.disparity_browse_folder)
        browse_btn.grid(row=0, column=2, padx=5, pady=5)

        ttk.Label(controls_frame, text="Number of disparities (multiple of 16):").grid(row=1, column=0, padx=5, pady=5, sticky="e")
        self.ndisp_entry = ttk.Entry(controls_frame, width=10)
        self.ndisp_entry.grid(row=1, column=1, padx=5, pady=5)
        self.ndisp_entry.insert(0, str(DEFAULT_NDISP))

        ttk.Label(controls_frame, text="Minimum disparity:").grid(row=2, column=0, padx=5, pady=5, sticky="e")
        self.mindisp_entry = ttk.Entry(controls_frame, width=10)
        self.mindisp_entry.grid(row=2, column=1, padx=5, pady=5)
        self.mindisp_entry.insert(0, str(DEFAULT_MINDISP))

        run_btn = ttk.Button(controls_frame, text="Run", command=self.run_disparity_tab)
        run_btn.grid(row=3, column=1, pady=10)

        self.disparity_img_labels = {title: tk.Label(tab6) for title in self.IMG_TITLES}
        for i, (title, label) in enumerate(self.disparity_img_labels.items()):
            ttk.Label(tab6, text=title).grid(row=4, column=i, pady=(0, 2), sticky="n")
            label.grid(row=5, column=i, padx=5, pady=5)
        self.status_label = tk.Label(tab6, text="")
        self.status_label.grid(row=7, column=0, columnspan=2)

    def disparity_browse_folder(self):
        folder = self.filedialog.askdirectory(title="Select Folder with the Stereo Image Pair (img1.jpg, img2.jpg)")
        if folder:
            self.disparity_path_entry.delete(0, self.tk.END)
            self.disparity_path_entry.insert(0, folder)

    def _show_text(self, text):
        for label in self.disparity_img_labels.values():
            label.config(image="")
        self.status_label.config(text=text)

    def _validated_int(self, entry, name, default, check):
        s = entry.get().strip()
        try:
            v = int(s)
            check(v)
            return v
        except ValueError as e:   # gui.py:466-472: message box, then carry on with the default
            self.messagebox.showerror("Invalid Input", f"Invalid {name} value: {e}. Using default ({default}).")
            return default

    # -- gui.py:456-487 --
    def run_disparity_tab(self):
        folder_path = self.disparity_path_entry.get()
        if not folder_path:
            self._show_text("Please provide a folder path.")
            return

        def check_ndisp(v):
            if v <= 0 or v % 16:
                raise ValueError("numDisparities must be a positive multiple of 16")

        ndisp = self._validated_int(self.ndisp_entry, "number of disparities", DEFAULT_NDISP, check_ndisp)
        mindisp = self._validated_int(self.mindisp_entry, "minimum disparity", DEFAULT_MINDISP, lambda v: None)

        # blocking call on the Tk main thread, result kept on self (and on the owner, for later tabs)
        self.disparity_results = run_disparity_folder(folder_path, ndisp, mindisp, runner=self.runner)
        if self.owner is not None:
            self.owner.disparity_results = self.disparity_results

        if isinstance(self.disparity_results, str):  # error case
            self._show_text(self.disparity_results)
            return

        valid = self.disparity_results["mask"]
        self.status_label.config(text=f"numDisparities={ndisp} minDisparity={mindisp}: "
                                      f"{int(valid.sum()) if valid is not None else 0} valid 3-D points")
        for title in self.IMG_TITLES:
            img_pil = self.tk.PhotoImage(data=png_bytes(self.disparity_results[title]))
            self.disparity_img_labels[title].config(image=img_pil)
            self.disparity_img_labels[title].image = img_pil  # keep a reference (gui.py:487)
