#!/usr/bin/env python3
"""Cheapest path through the step loop of a kernel in a gfx950 assembly listing: the instructions a wavefront issues in a step in which no rare
branch is taken.  Splits the function into basic blocks (labels and branch instructions), finds the largest loop (backward branch) and runs a
shortest-path search from its header to its latch, weighing a block by its VALU instructions.

usage: tools/main_path.py listing.s [substring-of-kernel-name] [--blocks]"""
import heapq
import re
import sys


def kernel_text(path, needle):
    src = open(path).read().split('\n')
    starts = [i for i, l in enumerate(src) if re.match(r'^_Z\w+:', l) and (needle is None or needle in l)]
    if not starts:
        raise SystemExit('kernel not found')
    s0 = starts[0]
    end = next(i for i in range(s0, len(src)) if src[i].startswith('.Lfunc_end'))
    return src[s0].split(':')[0], src[s0 + 1:end]


def blocks_of(lines):
    """[(label, [instructions], [successor labels], falls_through)]; unlabeled blocks after a conditional branch get synthetic labels."""
    out, cur, lab, n = [], [], 'entry', 0
    for l in lines:
        s = l.split(';')[0].strip()
        if not s:
            continue
        if re.match(r'^\.LBB\d+_\d+:', s):
            out.append([lab, cur, [], True])
            lab, cur = s[:-1], []
            continue
        if s.startswith('.') or s.endswith(':'):
            continue
        cur.append(s)
        op = s.split()[0]
        if op.startswith('s_cbranch') or op == 's_branch':
            out.append([lab, cur, [s.split()[1]], op != 's_branch'])
            n += 1
            lab, cur = f'_syn{n}', []
        elif op in ('s_endpgm', 's_setpc_b64'):
            out.append([lab, cur, [], False])
            n += 1
            lab, cur = f'_syn{n}', []
    out.append([lab, cur, [], False])
    return out


def stats(ins):
    v = sum(1 for i in ins if i.startswith('v_'))
    acc = sum(1 for i in ins if i.startswith('v_accvgpr'))
    f64 = sum(1 for i in ins if i.startswith(('v_fma', 'v_mul_f64', 'v_add_f64')))
    lds = sum(1 for i in ins if i.startswith('ds_'))
    vm = sum(1 for i in ins if i.startswith(('global_', 'scratch_', 'buffer_', 'flat_')))
    sal = sum(1 for i in ins if i.startswith('s_') and not i.startswith(('s_waitcnt', 's_nop')))
    return dict(all=len(ins), valu=v, accvgpr=acc, f64=f64, lds=lds, vmem=vm, salu=sal)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    name, lines = kernel_text(args[0], args[1] if len(args) > 1 else None)
    bl = blocks_of(lines)
    idx = {b[0]: i for i, b in enumerate(bl)}
    succ = []
    for i, (lab, ins, tg, ft) in enumerate(bl):
        s = [idx[t] for t in tg if t in idx]
        if ft and i + 1 < len(bl):
            s.append(i + 1)
        succ.append(s)
    # the step loop = the strongly connected component with the most instructions (outlined cold blocks sit anywhere in the listing, so
    # "backward branch" does not identify it); its header = the member that is entered from outside
    sys.setrecursionlimit(100000)
    index, low, onst, st, comps, counter = {}, {}, set(), [], [], [0]

    def strong(v):
        index[v] = low[v] = counter[0]
        counter[0] += 1
        st.append(v)
        onst.add(v)
        for w in succ[v]:
            if w not in index:
                strong(w)
                low[v] = min(low[v], low[w])
            elif w in onst:
                low[v] = min(low[v], index[w])
        if low[v] == index[v]:
            comp = []
            while True:
                w = st.pop()
                onst.discard(w)
                comp.append(w)
                if w == v:
                    break
            comps.append(comp)

    for v in range(len(bl)):
        if v not in index:
            strong(v)
    loop = set(max(comps, key=lambda c: sum(len(bl[k][1]) for k in c) if len(c) > 1 else 0))
    # anchor of the cycle: the block every step passes -- the control law's QR (most DPP moves); the cheapest cycle through it is the plain step
    head = max(loop, key=lambda k: sum(1 for i in bl[k][1] if 'dpp' in i or i.startswith(('v_permlane', 'global_store'))))
    cost = lambda k: stats(bl[k][1])['valu'] + 0.25 * len(bl[k][1])       # noqa: E731  (VALU first, everything else as a tie-breaker)
    dist, prev, pq = {}, {}, []
    for v in succ[head]:
        if v in loop:
            dist[v] = cost(v)
            prev[v] = head
            heapq.heappush(pq, (dist[v], v))
    while pq:
        d, u = heapq.heappop(pq)
        if d > dist.get(u, 1e18) or u == head:
            continue
        for v in succ[u]:
            if v not in loop:
                continue
            nd = d + cost(v)
            if nd < dist.get(v, 1e18):
                dist[v], prev[v] = nd, u
                heapq.heappush(pq, (nd, v))
    path, u = [], prev[head]
    while u != head:
        path.append(u)
        u = prev[u]
    path.append(head)
    path.reverse()
    latch = path[-1]
    tot = {}
    for k in path:
        st = stats(bl[k][1])
        for key, val in st.items():
            tot[key] = tot.get(key, 0) + val
        if '--blocks' in sys.argv:
            print(f'  {bl[k][0]:12s} {st}')
    whole = stats([i for b in bl for i in b[1]])
    print(f'{name[:110]}\n  loop {bl[head][0]} .. {bl[latch][0]}: cheapest step {tot}\n  whole kernel {whole}')


if __name__ == '__main__':
    main()
