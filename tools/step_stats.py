#!/usr/bin/env python3
"""Instruction mix of the steady-state step of each k_sweep instantiation, from csrc/sgm_engine.s
(`make -C stereo_reconstruction_cv_amd/csrc sgm_engine.s`): the shortest barrier-to-barrier
segment holding more than 100 packed instructions, plus the s_waitcnt vmcnt histogram."""
import re, sys
from collections import Counter
s = open(sys.argv[1] if len(sys.argv) > 1 else 'stereo_reconstruction_cv_amd/csrc/sgm_engine.s').read()
pat = sys.argv[2] if len(sys.argv) > 2 else r'_ZN3sgm7k_sweepILi2ELb0ELi[02]ELb1E'
for name in re.findall(r'^(' + pat + r'\S*):', s, flags=re.M):
    i = s.index(name + ':'); j = s.index('.Lfunc_end', i)
    lines = [l for l in s[i:j].splitlines() if l.strip() and not l.strip().startswith((';', '.'))]
    idx = [k for k, l in enumerate(lines) if 's_barrier' in l]
    best = None
    for n in range(len(idx) - 1):
        seg = lines[idx[n]:idx[n + 1]]
        c = Counter(l.split()[0] for l in seg)
        pk = sum(v for k, v in c.items() if k.startswith('v_pk'))
        if pk > 100 and (best is None or len(seg) < best[0]): best = (len(seg), c)
    if best is None: continue
    n, c = best
    vm = Counter(re.findall(r's_waitcnt vmcnt\((\d+)\)', s[i:j]))
    print(name[:40], 'step', n, 'valu', sum(v for k, v in c.items() if k.startswith('v_')),
          'salu', sum(v for k, v in c.items() if k.startswith('s_')), 'branch',
          sum(v for k, v in c.items() if 'cbranch' in k), 'nop', c['s_nop'], 'vmcnt', dict(vm))
