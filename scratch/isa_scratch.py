"""scratch (spill) instructions of a kernel by basic block   usage: isa_scratch.py file.s kernel_regex"""
import re, sys
from collections import Counter
lines = open(sys.argv[1]).read().split('\n')
st = [i for i, l in enumerate(lines) if re.match(r'^_Z\S*' + sys.argv[2] + r'\S*:', l)][0]
en = next(i for i in range(st, len(lines)) if lines[i].startswith('.Lfunc_end'))
name = 'entry'; hd = {}; n = Counter(); out = []
for l in lines[st + 1:en]:
    m = re.match(r'^(\.LBB\d+_\d+):\s*(;.*)?', l)
    if m: name = m.group(1); hd[name] = m.group(2) or ''
    else:
        t = l.strip()
        if t and not t.startswith(('.', ';')):
            n[name] += 1
            if t.startswith('scratch_'): out.append((name, t))
for nm, t in out: print('%-12s %3d %-60s %s' % (nm, n[nm], t[:60], hd.get(nm, '')[:50]))
