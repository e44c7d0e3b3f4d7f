"""VALU instructions of the hot region of a kernel in a hipcc -S dump: from the first to the last line that matches <marker> (default v_pk_fma_f32), by mnemonic.
   python tools/isa_hot.py file.s <kernel-name-substring> [marker]"""
import collections, re, sys
txt = open(sys.argv[1]).read()
marker = sys.argv[3] if len(sys.argv) > 3 else "v_pk_fma_f32"
for m in re.finditer(r'^(\S+):\s*; @\1', txt, re.M):
    name = m.group(1)
    if sys.argv[2] not in name:
        continue
    end = txt.index('.Lfunc_end', m.start())
    lines = txt[m.start():end].split('\n')
    idx = [i for i, l in enumerate(lines) if marker in l]
    if not idx:
        continue
    body = lines[idx[0]:idx[-1] + 1]
    ins = [l.split()[0] for l in body if l.startswith('\t') and l.strip() and not l.strip().startswith(('.', ';'))]
    c = collections.Counter(ins)
    valu = sum(v for k, v in c.items() if k.startswith('v_'))
    print(name[:100], 'hot-region instructions', len(ins), 'valu', valu, 'vmem', sum(v for k, v in c.items() if 'load' in k or 'store' in k))
    print('  ' + ', '.join(f'{k}:{v}' for k, v in c.most_common(45)))
