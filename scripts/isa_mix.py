#!/usr/bin/env python3
"""static instruction mix per kernel from `hipcc -S --cuda-device-only` output: isa_mix.py file.s [name-substring ...]"""
import re, sys, collections
lines = open(sys.argv[1]).read().split('\n')
pats = sys.argv[2:]
cur, funcs = None, collections.OrderedDict()
for ln in lines:
    m = re.match(r'^(_ZN2zv\S+):', ln)
    if m:
        cur = m.group(1); funcs[cur] = []
        continue
    if ln.startswith('.Lfunc_end'):
        cur = None
    if cur: funcs[cur].append(ln.strip())
for name, body in funcs.items():
    if pats and not any(p in name for p in pats): continue
    cnt = collections.Counter()
    for line in body:
        if not line or line.startswith(('.', ';')) or line.endswith(':'): continue
        op = line.split()[0]
        if op.startswith('v_mfma'): cnt['mfma'] += 1
        elif op.startswith('v_accvgpr'): cnt['accvgpr'] += 1
        elif op.startswith('v_'): cnt['valu'] += 1; cnt['v:' + op] += 1
        elif op.startswith('s_waitcnt'): cnt['waitcnt'] += 1
        elif op.startswith('s_barrier'): cnt['barrier'] += 1
        elif op.startswith('s_nop'): cnt['nop'] += 1
        elif op.startswith('s_'): cnt['salu'] += 1
        elif op.startswith('ds_'): cnt['lds'] += 1; cnt['d:' + op] += 1
        elif op.startswith(('buffer_', 'global_', 'flat_')): cnt['vmem'] += 1
    print(name)
    print('  ', {k: v for k, v in cnt.items() if ':' not in k})
    print('  ', sorted([(v, k[2:]) for k, v in cnt.items() if k.startswith('v:')], reverse=True)[:16])
    print('  ', sorted([(v, k[2:]) for k, v in cnt.items() if k.startswith('d:')], reverse=True)[:6])
