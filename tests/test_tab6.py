"""Tab 6 "Run Disparity" shim (stereo_reconstruction_cv_amd/tab6.py) driven headlessly with a fake
widget set: same anatomy as the reference's tabs (/root/reference/gui.py:422-487) -- folder entry,
validated numeric entries (message box + default), blocking Run, results kept on self / the owner,
images handed to tk.PhotoImage as PNG bytes."""
import os

import numpy as np
import pytest

from stereo_reconstruction_cv_amd import synth, tab6


# ---- a minimal stand-in for tkinter / ttk / filedialog / messagebox ----
class _Widget:
    def __init__(self, master=None, **kw):
        self.master, self.kw, self.image = master, dict(kw), None

    def grid(self, **kw):
        self.grid_kw = kw

    def config(self, **kw):
        self.kw.update(kw)


class _Entry(_Widget):
    def __init__(self, master=None, **kw):
        super().__init__(master, **kw)
        self.text = ""

    def get(self):
        return self.text

    def insert(self, pos, s):
        self.text = s if pos == 0 else self.text + s

    def delete(self, a, b=None):
        self.text = ""


class _Notebook(_Widget):
    def __init__(self):
        super().__init__()
        self.tabs = []

    def add(self, frame, text=""):
        self.tabs.append(text)


class _Photo:
    def __init__(self, data=None):
        self.data = data


class _Tk:
    END = "end"
    Label = _Widget
    PhotoImage = _Photo


class _Ttk:
    Frame = Label = Button = _Widget
    Entry = _Entry


class _FileDialog:
    folder = ""

    @classmethod
    def askdirectory(cls, title=""):
        return cls.folder


class _MessageBox:
    def __init__(self):
        self.errors = []

    def showerror(self, title, msg):
        self.errors.append((title, msg))


def _toolkit():
    mb = _MessageBox()
    return (_Tk, _Ttk, _FileDialog, mb), mb


def _pair_folder(tmp_path, H=72, W=160, D=16, seed=4):
    from PIL import Image
    l, r, _ = synth.make_pair(H, W, D, seed)
    Image.fromarray(l).save(tmp_path / "img1.jpg", quality=95)
    Image.fromarray(r).save(tmp_path / "img2.jpg", quality=95)
    return str(tmp_path)


def _fake_runner(calls):
    def run(imgL, imgR, Q, ndisp, mindisp):
        calls.append((imgL.shape, ndisp, mindisp, np.asarray(Q).shape))
        H, W = imgL.shape
        d = np.tile(np.linspace(0, ndisp - 1, W, dtype=np.float32), (H, 1))
        return d, np.zeros((H, W, 3), np.float32), d > 1
    return run


def test_module_imports_without_a_display_and_builds_the_tab():
    tk, mb = _toolkit()
    nb = _Notebook()
    tab = tab6.DisparityTab(nb, toolkit=tk, runner=_fake_runner([]))
    assert nb.tabs == ["Run Disparity"]
    assert tab.ndisp_entry.get() == "16" and tab.mindisp_entry.get() == "0"      # the notebook's call, main.ipynb:781
    assert set(tab.disparity_img_labels) == {"Left Image", "Disparity Map"}


def test_run_keeps_results_on_self_and_owner_and_shows_png_images(tmp_path):
    tk, mb = _toolkit()
    calls = []

    class Owner:
        disparity_results = None

    owner = Owner()
    tab = tab6.DisparityTab(_Notebook(), owner=owner, toolkit=tk, runner=_fake_runner(calls))
    _FileDialog.folder = _pair_folder(tmp_path)
    tab.disparity_browse_folder()
    assert tab.disparity_path_entry.get() == str(tmp_path)
    tab.ndisp_entry.delete(0, _Tk.END)
    tab.ndisp_entry.insert(0, "32")
    tab.run_disparity_tab()
    assert calls == [((72, 160), 32, 0, (4, 4))] and not mb.errors
    res = tab.disparity_results
    assert isinstance(res, dict) and owner.disparity_results is res
    assert res["disparity_map"].dtype == np.float32 and res["Disparity Map"].shape == (360, 640, 3)
    for title in tab6.DisparityTab.IMG_TITLES:
        lab = tab.disparity_img_labels[title]
        assert lab.image is lab.kw["image"] and lab.image.data[:8] == b"\x89PNG\r\n\x1a\n"   # reference kept, PNG bytes
    assert "valid 3-D points" in tab.status_label.kw["text"]


def test_invalid_numbers_raise_a_message_box_and_fall_back_to_the_defaults(tmp_path):
    tk, mb = _toolkit()
    calls = []
    tab = tab6.DisparityTab(_Notebook(), toolkit=tk, runner=_fake_runner(calls))
    tab.disparity_path_entry.insert(0, _pair_folder(tmp_path))
    tab.ndisp_entry.insert(0, "24")        # not a multiple of 16
    tab.mindisp_entry.insert(0, "abc")
    tab.run_disparity_tab()
    assert len(mb.errors) == 2 and all(t == "Invalid Input" for t, _ in mb.errors)
    assert "Using default (16)" in mb.errors[0][1] and "Using default (0)" in mb.errors[1][1]
    assert calls[0][1:3] == (16, 0)


def test_error_strings_are_shown_not_raised(tmp_path):
    tk, mb = _toolkit()
    tab = tab6.DisparityTab(_Notebook(), toolkit=tk, runner=_fake_runner([]))
    tab.run_disparity_tab()                                   # no folder
    assert tab.status_label.kw["text"] == "Please provide a folder path." and tab.disparity_results is None
    tab.disparity_path_entry.insert(0, str(tmp_path))          # folder without the pair
    tab.run_disparity_tab()
    assert tab.disparity_results == "Error: Missing img1.jpg or img2.jpg in the folder."
    assert tab.status_label.kw["text"] == tab.disparity_results

    def boom(*a):
        raise RuntimeError("engine exploded")
    tab.runner = boom
    _pair_folder(tmp_path)
    tab.run_disparity_tab()
    assert tab.disparity_results == "Error: engine exploded"


def test_without_a_gpu_the_real_engine_reports_its_error_as_text(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    tk, mb = _toolkit()
    tab = tab6.DisparityTab(_Notebook(), toolkit=tk)          # the real pipeline.run_disparity
    tab.disparity_path_entry.insert(0, _pair_folder(tmp_path))
    tab.run_disparity_tab()
    assert isinstance(tab.disparity_results, str) and "no CPU fallback" in tab.disparity_results


@pytest.mark.gpu
def test_tab_runs_the_hip_engine_and_matches_the_oracle(tmp_path):
    from oracle import oracle as O
    tk, mb = _toolkit()
    tab = tab6.DisparityTab(_Notebook(), toolkit=tk)
    folder = _pair_folder(tmp_path, 90, 400, 16, 31)
    tab.disparity_path_entry.insert(0, folder)
    tab.run_disparity_tab()
    res = tab.disparity_results
    assert isinstance(res, dict), res
    imgL, imgR = tab6._load_gray(os.path.join(folder, "img1.jpg")), tab6._load_gray(os.path.join(folder, "img2.jpg"))
    p = dict(minDisparity=0, numDisparities=16, blockSize=11, P1=8 * 3 * 121, P2=32 * 3 * 121, disp12MaxDiff=1,
             preFilterCap=63, uniquenessRatio=10, speckleWindowSize=100, speckleRange=32)
    want = O.disp_to_float(O.sgbm_compute(imgL, imgR, **p))
    assert np.array_equal(res["disparity_map"].view(np.uint32), want.view(np.uint32))
    wx = O.reproject(want, res["Q"])
    fin = np.isfinite(wx)
    assert np.array_equal(np.isfinite(res["points_3D"]), fin) and np.allclose(res["points_3D"][fin], wx[fin], rtol=1e-4, atol=0)
    assert np.array_equal(res["mask"], O.valid_mask(wx, want))
