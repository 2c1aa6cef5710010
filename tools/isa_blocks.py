#!/usr/bin/env python3
"""Per-basic-block table of one kernel in a hipcc -S dump: instructions, scratch stores / loads (spills), global loads, fp64
instructions, LDS instructions.  usage: python tools/isa_blocks.py file.s mangled-name-prefix"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
start = [i for i, l in enumerate(lines) if l.startswith(sys.argv[2])][0]
end = [i for i, l in enumerate(lines) if i > start and 's_endpgm' in l][0]
blk, stats, order = None, {}, []
for i in range(start, end + 1):
    l = lines[i]
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m or blk is None:
        blk = m.group(1) if m else 'entry'
        order.append(blk)
        stats[blk] = dict(n=0, sst=0, sld=0, gl=0, f64=0, ds=0, line=i + 1)
        if m:
            continue
    t = l.strip()
    if not t or t.startswith(';') or t.startswith('.'):
        continue
    s = stats[blk]
    s['n'] += 1
    s['sst'] += t.startswith('scratch_store')
    s['sld'] += t.startswith('scratch_load')
    s['gl'] += t.startswith('global_load')
    s['f64'] += '_f64' in t
    s['ds'] += t.startswith('ds_')
tot = dict(n=0, sst=0, sld=0)
for b in order:
    s = stats[b]
    for k in tot: tot[k] += s[k]
    if s['sst'] or s['sld'] or s['gl'] or '-a' in sys.argv:
        print("%-12s line %5d  n %4d  spill st %3d ld %3d  global_load %2d  f64 %3d  lds %2d" % (b, s['line'], s['n'], s['sst'], s['sld'], s['gl'], s['f64'], s['ds']))
print("total", tot)
