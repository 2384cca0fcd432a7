#!/usr/bin/env python3
"""Instruction mix of the long loops of one kernel in a hipcc -S listing (MFMA, LDS, VMEM, VALU by kind, SALU).
usage: isa_mix.py file.s <substring of mangled name>"""
import re, sys, collections
lines = open(sys.argv[1]).read().split('\n')
sub = sys.argv[2]
start = next(i for i, l in enumerate(lines) if re.match(r'_Z\w+:', l) and sub in l)
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith('.Lfunc_end'))
labels = {}
for i in range(start, end):
    m = re.match(r'(\.LBB\d+_\d+):', lines[i])
    if m: labels[m.group(1)] = i
loops = []
for i in range(start, end):
    m = re.search(r's_c?branch\w* (\.LBB\d+_\d+)', lines[i])
    if m and m.group(1) in labels and labels[m.group(1)] < i and i - labels[m.group(1)] > 300:
        loops.append((labels[m.group(1)], i))
for a, b in loops:
    c = collections.Counter(); v = collections.Counter()
    for l in lines[a:b]:
        l = l.strip()
        if not l or l[0] in ';.': continue
        op = l.split()[0]
        if op.startswith('v_mfma'): c['mfma'] += 1
        elif op.startswith('ds_read'): c['ds_read'] += 1
        elif op.startswith('ds_write'): c['ds_write'] += 1
        elif re.match(r'(global|buffer|flat)_load', op): c['vload'] += 1
        elif re.match(r'(global|buffer|flat)_store', op): c['vstore'] += 1
        elif op.startswith('v_'): c['valu'] += 1; v[op] += 1
        elif op.startswith('s_'): c['salu'] += 1
    print("loop @%d..%d: %s" % (a - start, b - start, dict(c)))
    print("   valu:", sorted(((n, k) for k, n in v.items()), reverse=True)[:14])
