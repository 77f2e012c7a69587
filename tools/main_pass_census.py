#!/usr/bin/env python3
"""tools/main_pass_census.py ASM KERNEL_SUBSTR: the instruction mix of the COMMON PATH of the two main-pass pixel loops of a
k_eval2 instantiation (tools/kernel_regs.py --keep DIR writes the assembly): the cost phase's main pass = the loop whose
body holds the twenty ds_add_u64 of a sample and no IEEE division, the Jacobian phase's main pass = the small loop with
the sixteen weight-table reads; blocks that only rare lanes enter are left out (a block is on the common path if it is
reached by fall-through / s_cbranch_execz skipping from the loop header to the back edge)."""
import collections, os, re, sys
txt = open(sys.argv[1]).read()
want = sys.argv[2]
m = [x for x in re.finditer(r'^(_ZN3nid7k_eval2\S*):.*\n', txt, re.M) if want in x.group(1)]
i = m[0].end(); j = txt.index('.Lfunc_end', i)
blocks = collections.OrderedDict(); cur = 'entry'; blocks[cur] = []; hdr = {}
for l in txt[i:j].split('\n'):
    s = l.strip()
    mm = re.match(r'\.L(BB\d+_\d+):\s*(;.*)?', s)
    if mm:
        cur = mm.group(1); blocks[cur] = []; hdr[cur] = mm.group(2) or ''; continue
    if not s or s.startswith(';') or s.startswith('.'): continue
    blocks[cur].append(s)
names = list(blocks)


def cls(op):
    if op in ('v_fma_f64', 'v_fmac_f64_e32', 'v_fmac_f64_e64', 'v_mul_f64', 'v_add_f64'): return 'VALU f64 arithmetic'
    if op.startswith(('v_rcp', 'v_cvt', 'v_floor', 'v_fract', 'v_trunc')): return 'VALU convert / rcp'
    if op.startswith(('v_cmp', 'v_cndmask')): return 'VALU compare / select'
    if op.startswith(('v_mov', 'v_accvgpr', 'v_readlane', 'v_writelane', 'v_readfirstlane')): return 'VALU moves'
    if op.startswith('v_'): return 'VALU integer / address'
    if op.startswith('ds_'): return 'LDS'
    if op.startswith(('global_', 'buffer_')): return 'VMEM'
    if op.startswith('s_waitcnt'): return 's_waitcnt'
    if op.startswith(('s_cbranch', 's_branch')): return 'branch'
    if op.startswith('s_load'): return 'SMEM'
    if op.startswith('s_'): return 'SALU'
    return 'other'


def loop_blocks(h):
    return [n for n in names if n == h or ('Header=' + h + ' ') in hdr.get(n, '') + ' ']


def succ(n):
    """successors of a block: branch targets and, unless it ends in an unconditional branch, the next block"""
    out = []
    for x in blocks[n]:
        if x.startswith(('s_cbranch', 's_branch')): out.append(x.split()[1][2:])
    last = blocks[n][-1] if blocks[n] else ''
    if not last.startswith('s_branch'):
        k = names.index(n)
        if k + 1 < len(names): out.append(names[k + 1])
    return out


def shortest(a, b, inside):
    """fewest instructions from the start of block a to the start of block b along the CFG, staying inside the loop"""
    import heapq
    dist = {a: 0}; prev = {}; pq = [(0, a)]
    while pq:
        d, n = heapq.heappop(pq)
        if n == b and d > 0: break
        if d > dist.get(n, 1 << 30): continue
        for t in succ(n):
            if t not in inside: continue
            nd = d + len(blocks[n])
            if nd < dist.get(t, 1 << 30) or (t == b and t == a and (t, 'loop') not in dist):
                if t == a:  # the back edge
                    dist[(t, 'loop')] = nd; prev[(t, 'loop')] = n
                    continue
                dist[t] = nd; prev[t] = n; heapq.heappush(pq, (nd, t))
    key = (b, 'loop') if a == b else b
    path = []; n = prev.get(key)
    while n is not None and n != a: path.append(n); n = prev.get(n)
    return [a] + path[::-1]


def common_path(h, must):
    """the cheapest way round the loop through block `must` (the sample's work): the path lanes without a rare sample take"""
    inside = set(loop_blocks(h))
    first = shortest(h, must, inside) if must != h else [h]
    # from `must` back to the header
    import heapq
    dist = {must: 0}; prev = {}; pq = [(0, must)]; end = None
    while pq:
        d, n = heapq.heappop(pq)
        if d > dist.get(n, 1 << 30): continue
        for t in succ(n):
            nd = d + len(blocks[n])
            if t == h:
                if end is None or nd < end[0]: end = (nd, n)
                continue
            if t in inside and nd < dist.get(t, 1 << 30): dist[t] = nd; prev[t] = n; heapq.heappush(pq, (nd, t))
    tail = []; n = end[1]
    while n != must: tail.append(n); n = prev[n]
    return first + ([must] if must != h else []) + tail[::-1]


def show(h, must, title):
    path = common_path(h, must)
    c = collections.Counter()
    for n in path:
        for x in blocks[n]: c[cls(x.split()[0])] += 1
    tot = sum(c.values()); valu = sum(v for k, v in c.items() if k.startswith('VALU'))
    print(f"{title}: {tot} instructions per round on the common path ({len(path)} blocks: {' '.join(path)}), {valu} VALU")
    for k, v in sorted(c.items(), key=lambda kv: -kv[1]): print(f"    {k:24s} {v}")


heads = [n for n in names if 'Loop Header' in hdr.get(n, '')]
cost = jac = None
for h in heads:
    lb = loop_blocks(h)
    cnt = lambda n, pre: sum(1 for x in blocks[n] if x.startswith(pre))
    badd = max(lb, key=lambda n: cnt(n, 'ds_add_u64')); bread = max(lb, key=lambda n: cnt(n, 'ds_read'))
    div = sum(cnt(n, 'v_div_') for n in lb)
    strict = 'ELb1ELb1E' in want or os.environ.get('CENSUS_STRICT')  # (STRICT kernels: one pass per phase, IEEE divisions in it)
    if cnt(badd, 'ds_add_u64') >= 20 and cost is None and (div < 8 or strict): cost = (h, badd)
    if cnt(badd, 'ds_add_u64') == 0 and cnt(bread, 'ds_read') >= 6 and len(lb) <= (12 if strict else 6) and jac is None: jac = (h, bread)
if cost: show(cost[0], cost[1], "cost phase, main pass")
if jac: show(jac[0], jac[1], "Jacobian phase, main pass")
