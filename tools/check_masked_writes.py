#!/usr/bin/env python3
"""Static check of the kernel ISA (csrc/sgm_engine.s, from `make -C .../csrc sgm_engine.s`).

Flags, per kernel, every VALU instruction that (a) executes while EXEC is narrowed (between an
s_and_saveexec_b64 and the s_or_b64 exec, exec, ... that undoes it) and (b) writes a VGPR that was
the data operand of a 128-bit buffer/global store issued earlier under the full mask in the same
barrier-to-barrier stretch of code and not rewritten since.

Why: on gfx950 the NP = 4 sweeps now and then stored a wrong S vector (dword 0 of lanes 12-15 of
every 16) when hipcc put the winner-take-all's masked selects into the registers of the S vector
whose buffer_store_dwordx4 had been issued ~100 instructions before (round 1; tools/dbg_case.py,
DESIGN.md 4.3).  The kernels now materialise such values under the full mask; this script is the
regression check (exit status 1 on a finding)."""
import re, sys
path = sys.argv[1] if len(sys.argv) > 1 else 'stereo_reconstruction_cv_amd/csrc/sgm_engine.s'
s = open(path).read()


def regs(tok):
    tok = tok.rstrip(',')
    m = re.match(r'v\[(\d+):(\d+)\]$', tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'v(\d+)$', tok)
    return {int(m.group(1))} if m else set()


total = 0
for m in re.finditer(r'^(_Z\w+):\s*; @', s, flags=re.M):
    name = m.group(1)
    end = s.index('.Lfunc_end', m.end())
    stored, depth, hits = set(), 0, 0
    for line in s[m.end():end].splitlines():
        t = line.strip()
        if not t or t.startswith((';', '.')): continue
        f = t.replace(',', ' ').split()
        op = f[0]
        if op == 's_barrier': stored = set()
        elif op in ('buffer_store_dwordx4', 'global_store_dwordx4', 'buffer_store_dwordx3', 'global_store_dwordx3'):
            if not depth: stored |= regs(f[1] if op.startswith('buffer') else f[2])  # stores issued under the full mask
        elif op.startswith('s_and_saveexec') or op.startswith('s_andn2_saveexec'): depth += 1
        elif op.startswith('s_or_b64') and f[1] == 'exec': depth = max(0, depth - 1)
        elif op.startswith(('v_', 'ds_read', 'buffer_load', 'global_load')) and not op.startswith(('v_cmp', 'v_readlane', 'v_readfirstlane', 'v_nop')):
            if not depth:
                stored -= regs(f[1])  # rewritten under the full mask: no longer "the stored vector"
            elif op.startswith('v_') and regs(f[1]) & stored:
                hits += 1
                if hits <= 3: print(name[:60], ':', t[:80])
    if hits: print(hits, 'masked writes to recent wide-store data in', name)
    total += hits
print('findings:', total)
sys.exit(1 if total else 0)
