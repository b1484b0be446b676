"""Parity cases under the engine's guarded allocation mode (needs an MI355X), in a CHILD process: a GPU memory access
fault ends the process that caused it, and the mode exists to turn every out-of-range access into one.

Why: round 3's memory access fault.  While k_pix's left-pixel records were being moved to scalar loads,
an intermediate build put the 64-bit row address together as `(uint64_t)hi << 32 | readfirstlane(lo)` -- the builtin
returns int, so a low half with bit 31 set was sign-extended over the high half (ISA: s_bfe_i64 ..., 0x200000 in front of
the s_or_b64) and the load went to 0xffffffff'f2c1c000 -> "Memory access fault ... on address 0xfffff2c1c000".  It only
showed when an allocation happened to lie in the upper half of a 4 GiB window: after 300 green tests, in the one test
that keeps seven engines alive.  The guarded mode makes that placement the rule (and the buffer's end the end of its
mapping), so the whole class -- and any read or write past a buffer by loads that have no bounds check: scalar loads,
flat / global accesses -- fails deterministically, on the first frame.
(upstream counterpart of what the kernels compute: /root/reference/main.ipynb:668 -> SURVEY.md A.2-A.8)"""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_parity_cases_with_every_buffer_guarded_and_in_the_upper_half_of_a_4gib_window():
    env = dict(os.environ, SGM_DEBUG_ALLOC="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "guard_child.py")], capture_output=True, text=True, env=env, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    m = re.search(r"GUARD_OK (\d+)", r.stdout)
    assert m and int(m.group(1)) >= 50, tail


def test_k_pix_builds_its_row_address_without_a_sign_extension():
    """ISA of the library build: in every k_pix instantiation the left-pixel records are read by scalar loads whose base
    comes straight from two v_readfirstlane_b32 -- no s_bfe_i64 (the 32 -> 64 bit sign extension of the faulty form)
    anywhere in the kernel -- and the widest of them is s_load_dwordx8: the four 8-byte records of one loop iteration
    (`j + 3 <= j1`), nothing merged beyond them."""
    text = open(os.path.join(ROOT, "stereo_reconstruction_cv_amd", "csrc", "sgm_engine.s")).read()
    seen = 0
    for km in re.finditer(r"^(_ZN3sgm5k_pixILi(\d)EEEv\w+):\s*; @", text, flags=re.M):
        body = text[km.end():text.index(".Lfunc_end", km.end())]
        seen += 1
        assert "s_bfe_i64" not in body, km.group(1)
        widths = [int(w) if w else 1 for w in re.findall(r"\ts_load_dword(?:x(\d+))?\s", body)]
        assert widths and max(widths) <= 8, (km.group(1), widths)
        # the record loads: x2 (one record: first column, tail columns) and x8 (four records); their bases are SGPR pairs fed by v_readfirstlane
        assert re.search(r"v_readfirstlane_b32 s(\d+), v\d+\n(?:.*\n){0,6}?\s*v_readfirstlane_b32 s\d+, v\d+", body), km.group(1)
    assert seen == 3
