#!/usr/bin/env python
"""Instruction mix per kernel of a hipcc -S listing (development aid): python tools/isa_mix.py file.s [name-filter]"""
import re
import sys
from collections import Counter

s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ''
for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)s_endpgm', s, re.S | re.M):
    name = m.group(1)
    if flt not in name:
        continue
    ins = [l.split()[0] for l in m.group(2).split('\n') if l.startswith('\t') and l.strip() and not l.strip().startswith(('.', ';'))]
    c = Counter(ins)
    tot = lambda pred: sum(v for k, v in c.items() if pred(k))
    print(f"{re.sub(r'_ZN12_GLOBAL__N_1[0-9]+', '', name)[:44]:44s} total {len(ins):5d} valu {tot(lambda k: k.startswith('v_') and 'mfma' not in k):5d} "
          f"mfma {tot(lambda k: 'mfma' in k):4d} mul_lo32 {c.get('v_mul_lo_u32', 0):3d} mul24 {tot(lambda k: 'u32_u24' in k):4d} exp {tot(lambda k: k.startswith('v_exp')):4d} "
          f"accvgpr {tot(lambda k: 'accvgpr' in k):4d} cndmask {tot(lambda k: 'cndmask' in k):4d} pk {tot(lambda k: k.startswith('v_pk_')):4d} "
          f"ds {tot(lambda k: k.startswith('ds_')):4d} salu {tot(lambda k: k.startswith('s_')):5d} nop {c.get('s_nop', 0):4d}")
