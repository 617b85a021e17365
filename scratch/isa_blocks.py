"""instruction mix of the basic blocks of a kernel that hold a given instruction (from a --save-temps .s file)
usage: isa_blocks.py file.s kernel_substring [instr_substring]"""
import re, sys
from collections import Counter
lines = open(sys.argv[1]).read().split('\n')
kname = sys.argv[2]
want = sys.argv[3] if len(sys.argv) > 3 else 'v_mfma'
starts = [i for i, l in enumerate(lines) if re.match(r'^_Z\S*' + re.escape(kname), l) and l.rstrip().split(';')[0].strip().endswith(':')]
for st in starts:
    en = next(i for i in range(st, len(lines)) if lines[i].startswith('.Lfunc_end'))
    print(lines[st].split(':')[0][:150])
    blocks = []; cur = []; name = 'entry'
    for l in lines[st + 1:en]:
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m:
            blocks.append((name, cur)); cur = []; name = m.group(1)
        else:
            t = l.strip()
            if t and not t.startswith(('.', ';')): cur.append(t)
    blocks.append((name, cur))
    for n, b in blocks:
        c = Counter(x.split()[0] for x in b)
        if any(want in k for k in c):
            tot = sum(v for k, v in c.items() if k.startswith('v_'))
            print('  ', n, 'instrs', len(b), 'vector(incl mfma)', tot, dict(c.most_common(16)))
