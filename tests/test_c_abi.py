"""The drop-in boundary from plain C (tests/c/abi_smoke.c: include/sgm_hip.h + dlopen, nothing else): every declared entry
point resolves and the geometry / validation calls answer without a GPU; on an MI355X one pair goes through the calls the
reference makes (/root/reference/main.ipynb:655-670, 697: create, compute, scale + mask, reprojectImageTo3D) and a batch of
three through sgm_compute_batch in throughput mode, each compared bit for bit with the CPU oracle's C entry points."""
import os
import re
import subprocess

import pytest

from oracle import oracle as O
from stereo_reconstruction_cv_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c", "abi_smoke.c")
EXE = os.path.join(ROOT, "tests", "c", "abi_smoke")


def build():
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < max(os.path.getmtime(SRC), os.path.getmtime(os.path.join(ROOT, "include", "sgm_hip.h"))):
        r = subprocess.run(["gcc", "-O1", "-std=gnu11", "-Wall", "-o", EXE, SRC, "-ldl", "-lm"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
    return EXE


def test_every_entry_point_resolves_from_c_and_the_list_is_the_headers():
    exe = build()
    src = open(SRC).read()
    loaded = set(re.findall(r"LOAD\(h, (sgm_[a-z_0-9]+)\)", src))
    assert loaded == set(_lib.EXPORTS), sorted(loaded ^ set(_lib.EXPORTS))       # (test_abi.py ties EXPORTS to the header)
    r = subprocess.run([exe, _lib.LIB_PATH, "symbols"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert f"{len(_lib.EXPORTS)} entry points" in r.stdout, r.stdout


@pytest.mark.gpu
def test_the_reference_calls_from_c_match_the_oracle():
    exe = build()
    O.lib()
    r = subprocess.run([exe, _lib.LIB_PATH, "parity", O.LIB_PATH if hasattr(O, "LIB_PATH") else os.path.join(ROOT, "oracle", "liboracle_sgbm.so")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 of 34560 disparities differ" in r.stdout and "float map identical" in r.stdout and "0 XYZ values differ" in r.stdout
    assert "batch of 3: 0 differ" in r.stdout and "headroom ok=1" in r.stdout, r.stdout
