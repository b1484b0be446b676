#!/usr/bin/env python3
"""Markdown table of a round's bench lines (profiles/<tag>/bench_<workload>.json) joined with the PMC records
(profiles/<tag>/pmc_traffic.json): what BASELINE.md section 3 is pasted from.   python tools/summarize_round.py r03"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
d = os.path.join(ROOT, "profiles", tag)
pmc = json.load(open(os.path.join(d, "pmc_traffic.json")))["workloads"]
order = ["default", "c3c5x24", "c3c5x18", "c3c5x16", "c3c5x14", "c3c5x12", "c3c5x8", "c3c5x6", "c3c5", "c3", "c3c5x2", "c3c5x3", "c5", "c5x17", "c5x12", "c2", "c4", "c4t", "c1", "c1x8", "c1t", "nb"]
print("| workload | pairs / step | schedule | ms / pair | pairs/s | Mdisp/s | floor GB | PMC traffic GB (V) | traffic / floor | time / floor at 6.29 TB/s | "
      "HBM GB/s (frac of 8 TB/s) | dominant kernel: achieved GB/s (frac), basis |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|")
for w in order:
    f = os.path.join(d, f"bench_{w}.json")
    if not os.path.exists(f):
        continue
    b = json.loads(open(f).read().strip().splitlines()[-1])
    name = b["config"]["workload"].split(":")[0]
    p = pmc.get(name)
    tr = p["whole_pair_traffic_bytes"] if p else None
    ms = b["ms_per_pair"]
    r = b["roofline"]
    sched = {"chained": "chained (throughput)", "fused": "pre-pass (latency)", "one": "per direction"}[b["config"]["schedule"].split()[0]]
    print(f"| `{name}`{' (default)' if w == 'default' else ''} | {b['config']['pairs_per_gpu_per_step']} | {sched} | {ms:.3f} | {b['pairs_per_s']:.1f} | {b['value']:,.0f} | "
          f"{b['floor_bytes'] / 1e9:.2f} | " + (f"{tr / 1e9:.2f} ({p['whole_pair_traffic_in_V']:.2f} V) | {tr / b['floor_bytes']:.2f} | " if tr else "- | - | ") +
          f"{b['time_over_floor_at_6.29TBps']:.2f} | " + (f"{tr / ms / 1e6:,.0f} ({tr / ms / 1e6 / 8000:.2f}) | " if tr else "- | ") +
          f"`{r['kernel']}`: {r['achieved']:,.0f} ({r['frac']:.2f}), {r['basis']} |")
