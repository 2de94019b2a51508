#!/usr/bin/env python3
"""Instruction mix per phase of mfcc512_kernel from a -DF512_STAMPS assembly listing (the s_memtime pairs
mark the phase boundaries).  usage: tools/asm_phases.py [listing.s] [kernel-name-substring]"""
import collections
import re
import subprocess
import sys

ROOT = __file__.rsplit('/', 2)[0]
src = sys.argv[1] if len(sys.argv) > 1 else '/tmp/f512_stamps.s'
want = sys.argv[2] if len(sys.argv) > 2 else 'mfcc512_kernelILi25ELi5ELi1ELi6ELi0ELi8ELb0ELi0E'
if len(sys.argv) <= 1:
    subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', f'-I{ROOT}/include',
                    '-I.', '-ffp-contract=fast', '-fno-gpu-rdc', '-fno-slp-vectorize', '-DF512_STAMPS', '-S',
                    '--cuda-device-only', '-o', src, 'dsp_frontend.hip'], cwd=f'{ROOT}/dsp-speech-recognition_amd/csrc',
                   check=True, stderr=subprocess.DEVNULL)
CYC = {'f32': 2.26, 'pk': 3.9, 'dpp': 3.7, 'cnd': 3.7, 'mov': 2.0, 'cmp': 3.6, 'vint': 3.6, 'trans': 7.0}


def cat(op):
    if op.startswith('v_pk_'): return 'pk'
    if 'dpp' in op: return 'dpp'
    if op.startswith('v_cndmask'): return 'cnd'
    if op.startswith('v_mov') or op.startswith('v_accvgpr'): return 'mov'
    if re.match(r'v_(add|sub|mul|fma|fmac|fmamk|fmaak|max|min)_f32', op): return 'f32'
    if re.match(r'v_(log|exp|ldexp|rcp|sqrt|frexp)', op): return 'trans'
    if op.startswith('v_cmp'): return 'cmp'
    if op.startswith('v_'): return 'vint'
    if op.startswith('ds_'): return 'lds'
    if op.startswith(('global_', 'scratch_', 'buffer_')): return 'vmem'
    if op.startswith('s_waitcnt'): return 'wait'
    if op.startswith('s_'): return 'salu'
    return 'other'


inside, seg = False, 0
table = collections.defaultdict(collections.Counter)
for ln in open(src):
    if not inside:
        inside = ln.startswith('_Z') and want in ln and ln.rstrip().endswith(':') or (ln.startswith('_Z') and want in ln.split(':')[0])
        continue
    m = re.match(r'\s+([a-z_0-9]+)', ln)
    if not m:
        continue
    op = m.group(1)
    if op == 's_endpgm':
        break
    if op == 's_memtime':
        seg += 1
        continue
    table[seg][cat(op)] += 1
names = {2: 'prologue', 4: 'stage', 6: 'pass1 FFT32', 8: 'untangle+tw', 10: 'exchange', 12: '2xFFT16', 14: 'power+unit0',
         16: 'ps write', 18: 'mel+log', 20: 'DCT+allred', 22: 'select', 24: 'store', 25: 'exit'}
cols = ['f32', 'pk', 'dpp', 'cnd', 'mov', 'cmp', 'vint', 'trans', 'lds', 'vmem', 'salu', 'wait']
print(f'{"phase":14s}' + ' '.join(f'{c:>5s}' for c in cols) + '  ~VALU cyc')
tot = collections.Counter()
for k in sorted(table):
    if k % 2 and k != 25:
        continue
    est = sum(table[k][c] * CYC.get(c, 0) for c in cols)
    print(f'{names.get(k, str(k)):14s}' + ' '.join(f'{table[k][c]:5d}' for c in cols) + f'  {est:8.0f}')
    if 4 <= k <= 24:
        tot.update(table[k])
est = sum(tot[c] * CYC.get(c, 0) for c in cols)
print(f'{"loop total":14s}' + ' '.join(f'{tot[c]:5d}' for c in cols) + f'  {est:8.0f}')
if len(sys.argv) > 3:      # JSON copy for profiles/
    import json
    arith = ('f32', 'pk', 'trans')
    out = {'_comment': 'static instruction mix per phase of ' + want + ' (diagnostic -DF512_STAMPS build: the s_memtime pairs mark '
                       'the phase boundaries; the staging phase lists all three of its paths, of which a group executes one; '
                       'the pins of that build add register moves the product build does not have). Wave instructions per '
                       '8-frame group; arithmetic = f32 + packed + transcendental, the rest of the vector work is selects, '
                       'compares, integer address math, moves, DPP.',
           'phases': {names.get(k, str(k)): {c: table[k][c] for c in cols if table[k][c]} for k in sorted(table) if not (k % 2 and k != 25)},
           'loop_total': {c: tot[c] for c in cols if tot[c]},
           'loop_vector_arithmetic': sum(tot[c] for c in arith),
           'loop_vector_non_arithmetic': sum(tot[c] for c in ('dpp', 'cnd', 'mov', 'cmp', 'vint'))}
    json.dump(out, open(sys.argv[3], 'w'), indent=1)
