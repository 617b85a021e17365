"""Instruction mix of the basic blocks that hold MFMAs in one kernel of a hipcc -S listing.
usage: isa_loop.py file.s <substring of the mangled kernel name> [min mfma per block]"""
import re, sys, collections
lines = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
minm = int(sys.argv[3]) if len(sys.argv) > 3 else 4
start = next(i for i, l in enumerate(lines) if l.startswith('_ZN') and key in l and l.rstrip().endswith(':') or (l.startswith('_ZN') and key in l and ': ' in l and '@' in l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
blocks, cur, name = [], [], 'entry'
for l in lines[start + 1:end]:
    t = l.strip()
    if not t or t.startswith(';') or t.startswith('.'):
        if re.match(r'^\.LBB\d+_\d+:', t):
            blocks.append((name, cur)); cur = []; name = t.split(':')[0]
        continue
    cur.append(t.split()[0])
blocks.append((name, cur))
for name, ins in blocks:
    m = sum(1 for x in ins if x.startswith('v_mfma'))
    if m >= minm:
        c = collections.Counter(ins)
        valu = sum(v for k, v in c.items() if k.startswith('v_') and not k.startswith('v_mfma'))
        salu = sum(v for k, v in c.items() if k.startswith('s_') and not k.startswith('s_waitcnt') and not k.startswith('s_nop'))
        print(f"{name}: {len(ins)} instrs, mfma {m}, valu {valu}, salu {salu}, vmem {sum(v for k,v in c.items() if k.startswith('global_') or k.startswith('buffer_'))}, lds {sum(v for k,v in c.items() if k.startswith('ds_'))}, waitcnt {c.get('s_waitcnt',0)}, nop {c.get('s_nop',0)}")
        print("   ", dict(sorted(((k, v) for k, v in c.items() if v >= 2), key=lambda kv: -kv[1])))
