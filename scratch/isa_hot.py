"""per basic block of a kernel (>= min instrs): VALU / MFMA / s_nop / LDS / SALU counts   usage: isa_hot.py file.s kernel_regex [min]"""
import re, sys
from collections import Counter
lines = open(sys.argv[1]).read().split('\n')
pat = sys.argv[2]; mn = int(sys.argv[3]) if len(sys.argv) > 3 else 40
st = [i for i, l in enumerate(lines) if re.match(r'^_Z\S*' + pat + r'\S*:', l)][0]
en = next(i for i in range(st, len(lines)) if lines[i].startswith('.Lfunc_end'))
blocks = []; cur = []; name = 'entry'
for l in lines[st + 1:en]:
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: blocks.append((name, cur)); cur = []; name = m.group(1)
    else:
        t = l.strip()
        if t and not t.startswith(('.', ';')): cur.append(t)
blocks.append((name, cur))
tot = Counter()
for n, b in blocks:
    c = Counter(x.split()[0] for x in b)
    tot.update(c)
    if len(b) >= mn:
        v = sum(k for kk, k in c.items() if kk.startswith('v_') and not kk.startswith('v_mfma'))
        print('%-12s %4d  valu %3d mfma %2d nop %2d alignbit %3d pk_add %2d ds %2d salu %2d' % (n, len(b), v, sum(k for kk, k in c.items() if kk.startswith('v_mfma')), c['s_nop'], c['v_alignbit_b32'], c['v_pk_add_f32'], sum(k for kk, k in c.items() if kk.startswith('ds_')), sum(k for kk, k in c.items() if kk.startswith('s_') and kk != 's_nop')))
print('kernel total', sum(tot.values()), 'nop', tot['s_nop'])
