#!/usr/bin/env python3
"""Static instruction counts of the search kernel between its s_memtime stamps (diagnostic -DFIN_STATS build).
usage: hipcc -O3 -std=c++17 --offload-arch=gfx950 -DFIN_STATS --cuda-device-only -S -o /tmp/v2.s fin_kernel_v2.hip; isa_segments.py /tmp/v2.s"""
import sys
lines = open(sys.argv[1]).read().split('\n')
start = [i for i, l in enumerate(lines) if l.startswith('_Z20fin_search_v2_kernel')][0]
end = [i for i, l in enumerate(lines) if i > start and l.strip().startswith('s_endpgm')][0]
names = ["(prologue)", "serve+wait", "head", "ustart_kdrop", "shrink", "kmerrec", "out_res", "base", "exti", "extk", "arrive", "tail", "(epilogue)"]
def classify(op):
    if op.startswith('v_'): return 'valu'
    if op.startswith('s_cbranch') or op.startswith('s_branch'): return 'branch'
    if op.startswith('s_waitcnt') or op.startswith('s_nop'): return 'wait'
    if op.startswith('s_'): return 'salu'
    if op.split('_')[0] in ('global', 'buffer', 'flat', 'scratch'): return 'vmem'
    if op.startswith('ds_'): return 'lds'
    return 'other'
keys = ['valu', 'salu', 'branch', 'wait', 'vmem', 'lds', 'other']
seg = []; cur = dict.fromkeys(keys, 0)
for i in range(start + 1, end + 1):
    l = lines[i].strip()
    if not l or l[0] in ';.' or l.endswith(':'): continue
    op = l.split()[0]
    if op == 's_memtime':
        seg.append(cur); cur = dict.fromkeys(keys, 0); continue
    cur[classify(op)] += 1
seg.append(cur)
tot = dict.fromkeys(keys, 0)
print("%-14s" % "segment", " ".join("%6s" % k for k in keys), "  total")
for i, c in enumerate(seg):
    for k in keys: tot[k] += c[k]
    print("%-14s" % (names[i] if i < len(names) else str(i)), " ".join("%6d" % c[k] for k in keys), "%7d" % sum(c.values()))
print("%-14s" % "all", " ".join("%6d" % tot[k] for k in keys), "%7d" % sum(tot.values()))
