#!/usr/bin/env python3
"""Static loop census of a hipcc -S listing: for every loop the assembler comments mark (`Loop Header: Depth=N`), count the
instructions between the header label and the last backward branch to it, by class.  Used to see what sits inside the layer
loops of samsim_step_kernel (scratch traffic, SGPR-spill lane moves, FP64 VALU, memory).
usage: isa_loops.py kern.s <kernel-name-substring> [min_instructions]"""
import re, sys, collections

def main():
    path, kname = sys.argv[1], sys.argv[2]
    min_n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    lines = open(path).read().split('\n')
    start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and kname in l and l.split(':')[0].endswith(l.split(':')[0]) and ':' in l)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
    body = lines[start:end]
    label_at = {}
    for i, l in enumerate(body):
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m: label_at[m.group(1)] = i
    # backward branches
    loops = collections.defaultdict(int)
    for i, l in enumerate(body):
        m = re.match(r'^\s+s_c?branch\S*\s+(\.LBB\d+_\d+)', l)
        if m and m.group(1) in label_at and label_at[m.group(1)] <= i:
            loops[m.group(1)] = max(loops[m.group(1)], i)
    def classify(op):
        if op.startswith('scratch_load'): return 'scr_ld'
        if op.startswith('scratch_store'): return 'scr_st'
        if op.startswith('global_load') or op.startswith('buffer_load'): return 'g_ld'
        if op.startswith('global_store') or op.startswith('buffer_store'): return 'g_st'
        if op.startswith('ds_'): return 'lds'
        if op.startswith('v_readlane') or op.startswith('v_writelane') or op.startswith('v_readfirstlane'): return 'lane'
        if op.startswith('v_') and 'f64' in op: return 'v_f64'
        if op.startswith('v_'): return 'v_other'
        if op.startswith('s_waitcnt'): return 'wait'
        if op.startswith('s_cbranch') or op.startswith('s_branch'): return 'branch'
        if op.startswith('s_'): return 's_alu'
        return 'other'
    rows = []
    for lab, last in loops.items():
        first = label_at[lab]
        cnt = collections.Counter()
        for l in body[first:last + 1]:
            m = re.match(r'^\s+([a-z_0-9]+)', l)
            if not m or l.strip().startswith(';') or l.strip().startswith('.'): continue
            cnt[classify(m.group(1))] += 1
        n = sum(cnt.values())
        hdr = body[first + 1] if first + 1 < len(body) else ''
        depth = re.search(r'Depth=(\d+)', ' '.join(body[first:first + 3]))
        rows.append((first, last, lab, n, cnt, depth.group(1) if depth else '?'))
    rows.sort()
    keys = ['v_f64', 'v_other', 's_alu', 'branch', 'wait', 'g_ld', 'g_st', 'scr_ld', 'scr_st', 'lane', 'lds']
    print('%-12s %7s %7s %3s %6s ' % ('label', 'line', 'end', 'd', 'n') + ' '.join('%7s' % k for k in keys))
    for first, last, lab, n, cnt, d in rows:
        if n < min_n: continue
        print('%-12s %7d %7d %3s %6d ' % (lab, first, last, d, n) + ' '.join('%7d' % cnt[k] for k in keys))
    tot = collections.Counter()
    for l in body:
        m = re.match(r'^\s+([a-z_0-9]+)', l)
        if not m or l.strip().startswith(';') or l.strip().startswith('.'): continue
        tot[classify(m.group(1))] += 1
    print('%-12s %7d %7d %3s %6d ' % ('TOTAL', 0, len(body), '', sum(tot.values())) + ' '.join('%7d' % tot[k] for k in keys))

if __name__ == '__main__':
    main()
