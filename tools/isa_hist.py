"""Mnemonic histogram of one kernel in a hipcc -S dump: python tools/isa_hist.py file.s <name-substring> [...]"""
import collections, re, sys
txt = open(sys.argv[1]).read()
for m in re.finditer(r'^(\S+):\s*; @\1', txt, re.M):
    name = m.group(1)
    if not all(s in name for s in sys.argv[2:]):
        continue
    end = txt.index('.Lfunc_end', m.start())
    body = txt[m.start():end]
    ins = [l.split()[0] for l in body.split('\n') if l.startswith('\t') and l.strip() and not l.strip().startswith(('.', ';'))]
    c = collections.Counter(ins)
    valu = sum(v for k, v in c.items() if k.startswith('v_'))
    print(name[:90], 'total', len(ins), 'valu', valu)
    print('  ' + ', '.join(f'{k}:{v}' for k, v in c.most_common(40)))
