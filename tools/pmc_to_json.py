#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of `bench.py --workload W` into an entry of
profiles/pmc_traffic.json: HBM traffic per stage and PAIR, stamped with the hash of the kernel sources it was
measured on (bench.py ignores the file when the stamp does not match its sources).

    python tools/pmc_to_json.py WORKLOAD FETCH_DIR WRITE_DIR BENCH_JSON OUT_JSON

gfx950: FETCH_SIZE reports half of the streamed bytes (MI355X_MICROARCH.md, HBM), WRITE_SIZE is
exact: traffic = 2 * FETCH_SIZE + WRITE_SIZE (KiB -> bytes).  The number of pairs the profiled run
processed comes from its own bench line: (steps + warmup) * pairs_per_gpu_per_step (run the passes with
--no-latency-mode --no-cpu-baseline so that nothing else launches kernels).  A stage's traffic is the sum over
the launches of its kernels.  OUT_JSON holds one entry per workload (entries measured on other sources are dropped)."""
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
    ("paths5", r"k_paths5_g"),
    ("chain_dn", r"k_sweep_chain<\d+, \w+, 0"), ("chain_up", r"k_sweep_chain<\d+, \w+, 1"),
    ("sweep_dn", r"k_sweep<\d+, \w+, 0|k_vert3_g<\d+, 0"), ("sweep_up", r"k_sweep<\d+, \w+, 1|k_vert3_g<\d+, 1"),
    ("sweep_up_wta", r"k_sweep<\d+, \w+, 2"), ("path_W_wta", r"k_rows_g<\d+, \d+, \w+, 2"), ("path_W", r"k_rows_g<\d+, \d+, \w+, [01]"),
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


def from_summary(path):
    """the same totals from a pmc_fetch_write_summary.txt (tools/pmc_summary.py: per kernel {counter: (launches, average)})"""
    fetch, write, calls = defaultdict(float), defaultdict(float), defaultdict(int)
    for line in open(path):
        name, _, rest = line.partition(" {")
        if not rest:
            continue
        rec = eval("{" + rest)  # noqa: S307 -- our own summary file
        if "FETCH_SIZE" in rec:
            fetch[name] = rec["FETCH_SIZE"][0] * rec["FETCH_SIZE"][1]
            calls[name] = rec["FETCH_SIZE"][0]
        if "WRITE_SIZE" in rec:
            write[name] = rec["WRITE_SIZE"][0] * rec["WRITE_SIZE"][1]
    return fetch, write, calls


def main():
    wl, dfetch, dwrite, bench_json, out = sys.argv[1:6]
    if dfetch.endswith(".txt"):      # a kept summary instead of the raw passes (FETCH_DIR = WRITE_DIR = the summary file)
        fetch, write, calls = from_summary(dfetch)
    else:
        fetch, calls = collect(dfetch, "FETCH_SIZE")
        write, _ = collect(dwrite, "WRITE_SIZE")
    bench = json.loads([l for l in open(bench_json) if l.startswith("{")][-1])
    cfg = bench["config"]
    pairs = (bench["steps"] + bench["warmup"]) * cfg["pairs_per_gpu_per_step"]
    stage_ms = bench["stage_ms"]
    stages = {}
    for name in sorted(set(fetch) | set(write)):
        b = (2.0 * fetch.get(name, 0.0) + write.get(name, 0.0)) * 1024.0 / pairs
        for st, rx in STAGE_KERNEL:
            if re.search(rx, name):
                # the in-row kernel of the small-D schedule runs inside the sweep stages; both pre-pass stages share one kernel
                targets = [s for s in stage_ms if s.startswith("prepass")] if st == "prepass" else [st]
                if st == "path_W":   # the in-row kernel: both in-row stages of the small-D schedule, else the sweep stages it runs in
                    targets = [s for s in stage_ms if s in ("path_W", "path_E")] or [s for s in stage_ms if s.startswith("sweep")] or [st]
                for t in targets:
                    rec = stages.setdefault(t, {"kernels": [], "bytes_per_pair": 0.0, "kernel_launches_per_pair": 0.0})
                    rec["kernels"].append(name)
                    rec["bytes_per_pair"] += b / len(targets)
                    rec["kernel_launches_per_pair"] += calls.get(name, 0) / pairs / len(targets)
                break
    V = 2 * cfg["height"] * (cfg["width"] - cfg["numDisparities"]) * cfg["numDisparities"]
    for t, rec in stages.items():
        rec["bytes_per_pair"] = int(rec["bytes_per_pair"])
        rec["traffic_in_V"] = round(rec["bytes_per_pair"] / V, 3)
        rec["kernel_launches_per_pair"] = round(rec["kernel_launches_per_pair"], 3)
    total = sum(r["bytes_per_pair"] for r in stages.values())
    stamp = source_stamp()
    doc = {}
    if os.path.exists(out):
        try:
            doc = json.load(open(out))
        except ValueError:
            doc = {}
    if doc.get("source_stamp") != stamp:
        doc = {"source_stamp": stamp, "workloads": {}}
    doc["note"] = ("traffic = 2*FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE counts half of streamed reads), per stage and PAIR = sum "
                   "over the launches of the stage's kernels / pairs the profiled run processed; separate --pmc passes "
                   "(tools/profile_gpu.sh); V = 2*H*W1*D bytes")
    doc["workloads"][wl] = {
        "schedule": {"one kernel per direction": 0, "fused sweeps behind a boundary pre-pass (latency mode)": 1,
                     "chained sweeps, no pre-pass (throughput mode)": 2}[cfg["schedule"]],
        "batch": bool(cfg.get("batch_entry")), "pairs_profiled": pairs, "V_bytes": V,
        "whole_pair_traffic_bytes": int(total), "whole_pair_traffic_in_V": round(total / V, 2),
        "floor_bytes": bench.get("floor_bytes"), "traffic_over_floor": round(total / bench["floor_bytes"], 3) if bench.get("floor_bytes") else None,
        "stages": stages}
    json.dump(doc, open(out, "w"), indent=1)
    print(f"{out}: {wl}: {len(stages)} stages, whole pair {total / V:.2f} V = {total / 1e9:.2f} GB, stamp {stamp}")


if __name__ == "__main__":
    main()
