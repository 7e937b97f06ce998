#!/usr/bin/env python3
"""Static instruction histogram of one kernel of an assembly listing (hipcc -S), split at s_memtime markers (the CRT_POOL_STAMPS build) in layout order.
    python tools/isa_sections.py file.s 'kernel-name-regex'"""
import re, sys
lines = open(sys.argv[1]).read().split('\n'); pat = re.compile(sys.argv[2])
start = end = None
for i, l in enumerate(lines):
    if start is None and l.endswith(':') is False and re.match(r'^(_Z\S+):', l) and pat.search(l): start = i
    elif start is not None and l.strip().startswith('s_endpgm'): end = i; break
def kind(t):
    if t.startswith('v_'): return 'valu'
    if t.startswith('ds_'): return 'lds'
    if t.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): return 'vmem'
    if t.startswith(('s_load', 's_buffer_load')): return 'smem'
    if t.startswith(('s_cbranch', 's_branch')): return 'br'
    if t.startswith('s_waitcnt') or t == 's_nop': return 'wait'
    if t.startswith('s_'): return 'salu'
    return None
sec = []; cur = {}
for l in lines[start:end]:
    t = l.strip().split()[0] if l.strip() else ''
    if not t or t.startswith((';', '.')) or t.endswith(':'): continue
    if t == 's_memtime': sec.append(cur); cur = {}; continue
    k = kind(t)
    if k: cur[k] = cur.get(k, 0) + 1
sec.append(cur)
tot = {}
for i, s in enumerate(sec):
    print(i, dict(sorted(s.items())), sum(s.values()))
    for k, v in s.items(): tot[k] = tot.get(k, 0) + v
print('total', dict(sorted(tot.items())), sum(tot.values()))
