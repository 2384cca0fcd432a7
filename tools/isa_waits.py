#!/usr/bin/env python3
"""Control-flow skeleton of one kernel of a hipcc -S listing: labels, branches, barriers, s_setprio and every
s_waitcnt vmcnt(N), with the number of MFMAs between them.  usage: isa_waits.py file.s <substring of mangled name>"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
start = next(i for i, l in enumerate(lines) if re.match(r'_Z\w+:', l) and sys.argv[2] in l)
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith('.Lfunc_end'))
mf = vm = st = 0
for i in range(start, end):
    l = lines[i].strip()
    if 'v_mfma' in l: mf += 1
    if re.match(r'(global|buffer|flat)_load', l): vm += 1
    if re.match(r'(global|buffer|flat)_store', l): st += 1
    m = re.search(r's_waitcnt.*vmcnt\((\d+)\)', l)
    if re.match(r'\.LBB\d+_\d+:', l) or 's_barrier' in l or 's_cbranch' in l or 's_branch' in l or 's_setprio' in l or m:
        print("%6d  [mfma %3d ld %3d st %3d]  %s" % (i - start, mf, vm, st, l.split(';')[0].strip()))
        mf = vm = st = 0
