#!/usr/bin/env python3
"""Static instruction mix of one kernel from hipcc's gfx950 assembly (-save-temps).
python scripts/isa_count.py <file.s> <kernel-name-substring>"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
pat = sys.argv[2]
for m in re.finditer(r'^(\S*%s\S*):[^\n]*\n(.*?)s_endpgm' % re.escape(pat), s, re.S | re.M):
    body = m.group(2)
    ins = [l.split()[0] for l in body.split('\n') if l.startswith('\t') and len(l.split()) and l.split()[0][0] not in '.;']
    c = Counter(ins)
    valu = sum(v for k, v in c.items() if k.startswith('v_'))
    print(m.group(1)[:70], 'instr', len(ins), 'valu', valu, 'pk', sum(v for k, v in c.items() if k.startswith('v_pk')),
          'ds', sum(v for k, v in c.items() if k.startswith('ds_')), 'global', sum(v for k, v in c.items() if k.startswith('global_')),
          'barrier', c.get('s_barrier', 0))
    print('   ', c.most_common(14))
