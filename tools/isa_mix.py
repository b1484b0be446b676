#!/usr/bin/env python3
"""Instruction mix of the largest basic block (the unrolled steady-state loop) of every kernel in csrc/sgm_engine.s
whose mangled name contains one of the given substrings.
usage: tools/isa_mix.py <substring> [<substring> ...]      (-d <dir> writes the block of each kernel to <dir>)"""
import re
import sys
from collections import Counter
args = sys.argv[1:]
dump = None
if args and args[0] == '-d':
    dump, args = args[1], args[2:]
t = open('stereo_reconstruction_cv_amd/csrc/sgm_engine.s').read()
for km in re.finditer(r'^(_Z\w+):\s*; @', t, flags=re.M):
    name = km.group(1)
    if not any(p in name for p in args):
        continue
    body = t[km.end():t.index('.Lfunc_end', km.end())].split('\n')
    blocks, cur = [], []
    for l in body:
        if re.match(r'^\.LBB', l):
            blocks.append(cur)
            cur = [l]
        else:
            cur.append(l)
    blocks.append(cur)
    big = max(blocks, key=len)
    ins = [l for l in big if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
    c = Counter(l.split()[0] for l in ins)
    valu = sum(n for k, n in c.items() if k.startswith('v_'))
    print(f'{name[:70]}: {len(ins)} instructions in the largest block, {valu} vector ALU, {c["s_nop"]} s_nop')
    print('   ', ', '.join(f'{k} {n}' for k, n in c.most_common(28)))
    if dump:
        open(f'{dump}/{name[:60]}.s', 'w').write('\n'.join(big))
