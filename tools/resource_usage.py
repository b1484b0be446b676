#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output (one line per kernel)."""
import re
import subprocess
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "stereo_reconstruction_cv_amd/csrc/resource_usage.txt"
rows, cur = [], None
for line in open(path):
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = m.group(1)
        try:
            name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
        except OSError:
            pass
        cur = {"name": re.sub(r"\(.*", "", name)}
        rows.append(cur)
        continue
    # remark: ../kernels_sweep.h:510:0:     TotalSGPRs: 106 [-Rpass-analysis=...]   /   ScratchSize [bytes/lane]: 0   /   Occupancy [waves/SIMD]: 4
    m = re.search(r"remark: \S+:\s+([A-Za-z][A-Za-z ]*?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
for r in rows:
    print(f"{r['name']:<60} vgpr={r.get('VGPRs', '?'):>3} agpr={r.get('AGPRs', '?'):>3} sgpr={r.get('TotalSGPRs', '?'):>3} "
          f"sgpr_spill={r.get('SGPRs Spill', '?'):>3} vgpr_spill={r.get('VGPRs Spill', '?'):>3} "
          f"scratch={r.get('ScratchSize', '?'):>4} occ={r.get('Occupancy', '?')} lds={r.get('LDS Size', '?')}")
