"""Instruction-class strings of the MFMA-carrying basic blocks of one kernel in a gfx950 .s file:
M mfma, e transcendental, v other VALU, d LDS, g global, w s_waitcnt, B barrier, n s_nop, J branch.
usage: python tools/asm_shape.py file.s <kernel-name-substring> [min_mfma=8]"""
import re
import sys

s = open(sys.argv[1]).read()
pat = sys.argv[2]
min_mfma = int(sys.argv[3]) if len(sys.argv) > 3 else 8
for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)s_endpgm', s, re.S | re.M):
    if pat not in m.group(1):
        continue
    lines = [l.strip() for l in m.group(2).split('\n') if l.strip() and not l.strip().startswith(';')]
    cur, blocks = ["entry"], []
    for l in lines:
        if l.startswith('.LBB'):
            blocks.append(cur); cur = [l]
        else:
            cur.append(l)
    blocks.append(cur)
    print(m.group(1))
    for b in blocks:
        nm = sum('v_mfma' in l for l in b)
        if nm < min_mfma:
            continue
        seq = ''
        for l in b:
            if 'v_mfma' in l: seq += 'M'
            elif re.match(r'v_(exp|log|rcp|rsq|sqrt|sin|cos)', l): seq += 'e'
            elif l.startswith('v_'): seq += 'v'
            elif l.startswith('ds_'): seq += 'd'
            elif l.startswith(('global_', 'buffer_', 'scratch_')): seq += 'g'
            elif l.startswith('s_waitcnt'): seq += 'w'
            elif l.startswith('s_barrier'): seq += 'B'
            elif l.startswith('s_nop'): seq += 'n'
            elif l.startswith(('s_cbranch', 's_branch')): seq += 'J'
        print(' ', b[0].split()[0], 'mfma', nm, 'instr', len(b)); print('   ', seq)
