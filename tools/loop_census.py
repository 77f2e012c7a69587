#!/usr/bin/env python3
"""tools/loop_census.py ASM KERNEL_SUBSTR: basic blocks of a kernel with their instruction mix
(VALU / SALU / VMEM / LDS / scratch / branch), so that the common path of the pixel loops can be added up by eye."""
import re, sys
txt = open(sys.argv[1]).read()
want = sys.argv[2]
m = [x for x in re.finditer(r'^(_ZN3nid7k_eval2\S*):.*\n', txt, re.M) if want in x.group(1)]
i = m[0].end(); j = txt.index('.Lfunc_end', i)
blocks = []; cur = ['entry', []]
for l in txt[i:j].split('\n'):
    l = l.strip()
    if not l or l.startswith(';') or l.startswith('.') and not re.match(r'\.LBB\d+_\d+:', l): continue
    mm = re.match(r'(\.LBB\d+_\d+):\s*(;.*)?', l)
    if mm:
        blocks.append(cur); cur = [mm.group(1) + ' ' + (mm.group(2) or ''), []]; continue
    cur[1].append(l.split()[0] + ' ' + ' '.join(l.split()[1:2]))
blocks.append(cur)
def kind(op):
    if op.startswith('v_'): return 'valu'
    if op.startswith('s_cbranch') or op.startswith('s_branch'): return 'br'
    if op.startswith('s_waitcnt'): return 'wait'
    if op.startswith('s_'): return 'salu'
    if op.startswith('ds_'): return 'lds'
    if op.startswith('scratch_'): return 'scr'
    if op.startswith('global_') or op.startswith('buffer_') or op.startswith('flat_'): return 'vmem'
    return 'other'
tot = {}
for name, ins in blocks:
    c = {}
    for x in ins:
        k = kind(x.split()[0]); c[k] = c.get(k, 0) + 1; tot[k] = tot.get(k, 0) + 1
    tgt = [x.split()[1] for x in ins if x.startswith('s_cbranch') or x.startswith('s_branch')]
    print(f"{name[:70]:70s} n={len(ins):4d} " + ' '.join(f"{k}={v}" for k, v in sorted(c.items())) + (" -> " + ','.join(tgt) if tgt else ''))
print("TOTAL", tot)
