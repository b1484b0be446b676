#!/usr/bin/env python3
"""Static check of the kernel ISA (csrc/sgm_engine.s, a by-product of every library build).

gfx950: a VMEM store of more than 64 bits reads its upper data registers a cycle or two after
issue; a VALU write to those registers needs 2 wait states behind the store.  hipcc (ROCm 7.2)
pads this hazard for flat/global stores and for MUBUF stores with an immediate soffset, but NOT for
MUBUF stores whose soffset is an SGPR (GCNHazardRecognizer::createsVALUHazard exempts that form) --
the root cause of the wrong S vectors of the D = 512 sweeps in round 1.

Two findings make this script fail (exit status 1):
  * any buffer_store_dwordx3/x4 at all (the engine issues 64-bit MUBUF stores only), and
  * any wide store (buffer/global/flat/scratch, x3 or x4) whose data VGPRs are written by a VALU
    instruction within the next two wait states.
"""
import re
import sys

path = sys.argv[1] if len(sys.argv) > 1 else 'stereo_reconstruction_cv_amd/csrc/sgm_engine.s'
text = open(path).read()
WIDE = re.compile(r'^\s*(buffer|global|flat|scratch)_store_(dwordx3|dwordx4|b96|b128)\s+(.*)$')


def vregs(tok):
    tok = tok.strip().rstrip(',')
    m = re.match(r'v\[(\d+):(\d+)\]$', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'v(\d+)$', tok)
    return {int(m.group(1))} if m else set()


def valu_dst(line):
    """VGPRs written by a VALU instruction line (first operand), else empty."""
    m = re.match(r'^\s*(v_[a-z0-9_]+)\s+([^,]+)', line)
    if not m or m.group(1).startswith(('v_cmp', 'v_readlane', 'v_readfirstlane')):
        return set()
    return vregs(m.group(2))


findings = 0
for km in re.finditer(r'^(_Z\w+):\s*; @', text, flags=re.M):
    name = km.group(1)
    end = text.index('.Lfunc_end', km.end())
    lines = [l for l in text[km.end():end].split('\n')]
    insts = [(i, l) for i, l in enumerate(lines)
             if l.startswith('\t') and not l.lstrip().startswith((';', '.'))]
    for k, (i, l) in enumerate(insts):
        m = WIDE.match(l)
        if not m:
            continue
        ops = m.group(3).split(',')
        # data operand: first for buffer stores, second for global/flat/scratch (after the address)
        data = vregs(ops[0]) if m.group(1) == 'buffer' else vregs(ops[1])
        if m.group(1) == 'buffer':
            print(f'{name}: wide MUBUF store: {l.strip()}')
            findings += 1
        waited = 0
        for _, nl in insts[k + 1:k + 4]:
            nm = re.match(r'^\s*s_nop\s+(\d+)', nl)
            if waited < 2 and valu_dst(nl) & data:
                print(f'{name}: VALU write {nl.strip()!r} only {waited} wait state(s) behind {l.strip()!r}')
                findings += 1
                break
            waited += (int(nm.group(1)) + 1) if nm else 1
            if waited >= 2:
                break
print(f'{findings} finding(s)')
sys.exit(1 if findings else 0)
