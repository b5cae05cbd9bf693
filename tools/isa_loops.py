#!/usr/bin/env python3
"""local: per-loop instruction statistics of one kernel in a hipcc -S listing (are SGPR spills --
v_readlane / v_writelane -- or scratch accesses INSIDE the traversal loops?).
usage: tools/isa_loops.py <listing.s> <mangled kernel name substring>
A loop = a backward branch to a label; loops nest, a line belongs to every loop that spans it."""
import re, sys
src = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
s = next(i for i, l in enumerate(src) if l.startswith('_Z') and key in l.split(':')[0] and ':' in l)
e = next(i for i in range(s, len(src)) if src[i].startswith('.Lfunc_end'))
lines = src[s:e]
labels = {}
for i, l in enumerate(lines):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: labels[m.group(1)] = i
loops = []
for i, l in enumerate(lines):
    m = re.search(r's_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)', l)
    if m:
        t = m.group(1) or m.group(2)
        if t in labels and labels[t] < i: loops.append((labels[t], i))
loops.sort()
PAT = [('valu', r'^\s+v_'), ('salu', r'^\s+s_'), ('ubyte', 'cvt_f32_ubyte'), ('readlane', 'v_readlane'), ('writelane', 'v_writelane'), ('scratch', 'scratch_'),
       ('gload', 'global_load'), ('gstore', 'global_store'), ('ds', r'\sds_'), ('rcp', 'v_rcp'), ('sload', 's_load')]
print(src[s])
print('%-13s' % 'loop' + ''.join('%10s' % n for n, _ in PAT))
for a, b in [(0, len(lines) - 1)] + loops:
    body = lines[a:b + 1]
    print('%5d-%5d  ' % (a, b) + ''.join('%10d' % sum(1 for l in body if re.search(p, l)) for _, p in PAT))
