#!/usr/bin/env python3
"""Basic blocks of one kernel in csrc/sgm_engine.s: instruction count, v_writelane / v_readlane (SGPR spill traffic;
the sweeps' own 4 v_readlane per pixel are in there too), scratch accesses, s_waitcnt vmcnt(0).
usage: tools/isa_blocks.py <mangled-name-substring> [min_instructions]"""
import re
import sys
t = open('stereo_reconstruction_cv_amd/csrc/sgm_engine.s').read() if len(sys.argv) < 4 else open(sys.argv[3]).read()
pat = sys.argv[1]
thr = int(sys.argv[2]) if len(sys.argv) > 2 else 150
for km in re.finditer(r'^(_Z\w+):\s*; @', t, flags=re.M):
    name = km.group(1)
    if pat not in name:
        continue
    body = t[km.end():t.index('.Lfunc_end', km.end())].split('\n')
    blocks, cur = [], ['entry', 0, 0, 0, 0, 0]
    for l in body:
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m:
            blocks.append(cur)
            cur = [m.group(1), 0, 0, 0, 0, 0]
        elif l.startswith('\t') and not l.strip().startswith(('.', ';')):
            cur[1] += 1
            cur[2] += 'v_writelane' in l
            cur[3] += 'v_readlane' in l
            cur[4] += 'scratch_' in l
            cur[5] += bool(re.search(r's_waitcnt vmcnt\(0\)', l))
    blocks.append(cur)
    print(name, len(body), 'lines')
    print('  block, instructions, writelane, readlane, scratch, vmcnt(0)')
    for b in blocks:
        if b[1] >= thr or b[2] or b[4]:
            print('  ', b)
