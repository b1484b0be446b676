#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of `bench.py --workload W` into
profiles/pmc_traffic.json: HBM traffic per stage of one frame, stamped with the hash of the kernel
sources it was measured on (bench.py ignores the file when the stamp does not match its sources).

    python tools/pmc_to_json.py WORKLOAD FETCH_DIR WRITE_DIR FRAMES BENCH_JSON OUT_JSON

gfx950: FETCH_SIZE reports half of the streamed bytes (MI355X_MICROARCH.md, HBM), WRITE_SIZE is
exact: traffic = 2 * FETCH_SIZE + WRITE_SIZE (KiB -> bytes).  FRAMES = steps + warmup of the
profiled run.  A stage's traffic is the sum over the launches inside its HIP-event bracket."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import source_stamp  # noqa: E402

STAGE_KERNEL = [  # stage -> regex on the demangled kernel name
    ("features", r"k_features"), ("cost_pix", r"k_pix(<|_px)"), ("cost_box", r"k_box_u8"),
    ("cost_hsum", r"k_hsum"), ("cost_vsum", r"k_vsum"),
    ("prepass", r"k_prepass3|k_path<\d+, \w+, 3"),
    ("sweep_dn", r"k_sweep<\d+, \w+, 0|k_rows4_g<\d+, 0"), ("sweep_up", r"k_sweep<\d+, \w+, 1|k_rows4_g<\d+, 1"),
    ("sweep_up_wta", r"k_sweep<\d+, \w+, 2"), ("path_W_wta", r"k_rows_g<\d+, \d+, \w+, 2"), ("path_W", r"k_rows_g<\d+, \d+, \w+, 1"),
    ("wta", r"k_wta_t"), ("select_lr", r"k_select"), ("median3", r"k_median3"), ("speckle", r"k_ccl_"),
    ("to_float", r"k_disp_to_float"), ("reproject", r"k_reproject"), ("post", r"k_post"), ("float_xyz", r"k_float_xyz"),
]


def collect(d, counter):
    tot, calls = defaultdict(float), defaultdict(int)
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"].split("(")[0]
            tot[name] += float(r["Counter_Value"])
            calls[name] += 1
    return tot, calls


def main():
    wl, dfetch, dwrite, frames, bench_json, out = sys.argv[1:7]
    frames = int(frames)
    fetch, calls = collect(dfetch, "FETCH_SIZE")
    write, _ = collect(dwrite, "WRITE_SIZE")
    bench = json.loads([l for l in open(bench_json) if l.startswith("{")][-1])
    stage_ms = bench["stage_ms"]
    stages = {}
    for name in sorted(set(fetch) | set(write)):
        b = (2.0 * fetch.get(name, 0.0) + write.get(name, 0.0)) * 1024.0 / frames
        for st, rx in STAGE_KERNEL:
            if re.search(rx, name):
                targets = [s for s in stage_ms if s.startswith("prepass")] if st == "prepass" else [st]
                for t in targets:
                    rec = stages.setdefault(t, {"kernels": [], "traffic_bytes_per_launch": 0.0, "launches_per_frame": 1,
                                                "kernel_launches": 0})
                    rec["kernels"].append(name)
                    rec["traffic_bytes_per_launch"] += b / len(targets)
                    rec["kernel_launches"] += calls.get(name, 0) / frames / len(targets)
                break
    V = 2 * bench["config"]["height"] * (bench["config"]["width"] - bench["config"]["numDisparities"]) * bench["config"]["numDisparities"]
    for t, rec in stages.items():
        rec["traffic_bytes_per_launch"] = int(rec["traffic_bytes_per_launch"])
        rec["traffic_in_V"] = round(rec["traffic_bytes_per_launch"] / V, 3)
        rec["stage_ms"] = stage_ms.get(t)
        if stage_ms.get(t):
            rec["GBps"] = round(rec["traffic_bytes_per_launch"] / (stage_ms[t] * 1e-3) / 1e9, 1)
    total = sum(r["traffic_bytes_per_launch"] for r in stages.values())
    json.dump({
        "workload": wl, "source_stamp": source_stamp(), "frames_profiled": frames, "V_bytes": V,
        "note": "traffic = 2*FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE counts half of streamed reads), per stage of one "
                "frame = sum over the launches inside the stage's HIP-event bracket; separate --pmc passes; stage_ms/GBps "
                "from the bench line of the FETCH pass (profiled runs are slower than unprofiled ones)",
        "whole_frame_traffic_bytes": int(total), "whole_frame_traffic_in_V": round(total / V, 2),
        "stages": stages}, open(out, "w"), indent=1)
    print(f"{out}: {len(stages)} stages, whole frame {total / V:.2f} V = {total / 1e9:.1f} GB, stamp {source_stamp()}")


if __name__ == "__main__":
    main()
