#!/usr/bin/env python3
"""Instruction mix of the PRODUCT build of mfcc512_kernel (no stamps): everything in the persistent loop from the
fast staging path (the global_load_dwordx4 burst) to the loop's back edge.  The staging slow paths (utterance
ends / seams) sit before it in the listing and are not counted.  usage: tools/asm_count.py [kernel-substring]"""
import collections, re, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = [a for a in sys.argv[1:] if not a.startswith('--')]
want = args[0] if args else 'mfcc512_kernelILi25ELi5ELi1ELi6ELi0ELi8ELb0E'
src = os.path.join(ROOT, 'dsp-speech-recognition_amd/lib/asm/dsp_frontend.s')
if not os.path.exists(src) or '--rebuild' in sys.argv:
    subprocess.run(['make', '-C', os.path.join(ROOT, 'dsp-speech-recognition_amd/csrc'), 'asm'], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
CYC = {'f32': 2.26, 'pk': 3.9, 'dpp': 3.7, 'cnd': 3.7, 'mov': 2.0, 'cmp': 3.6, 'vint': 3.6, 'trans': 7.0, 'lane': 3.6}
def cat(op, line):
    if op.startswith('v_pk_'): return 'pk'
    if 'dpp' in op or ' quad_perm' in line or ' row_' in line or ' wave_' in line: return 'dpp'
    if op.startswith('v_cndmask'): return 'cnd'
    if op.startswith(('v_readlane', 'v_writelane', 'v_readfirstlane')): return 'lane'
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
lines, inside = [], False
for ln in open(src):
    if not inside:
        inside = ln.startswith('_Z') and want in ln.split(':')[0]
        continue
    lines.append(ln)
    if 's_endpgm' in ln: break
start = next(i for i, ln in enumerate(lines) if 'F512_FAST_STAGE' in ln)
end = max(i for i, ln in enumerate(lines) if re.search(r's_cbranch|s_branch', ln))
cnt = collections.Counter()
lds = collections.Counter()
for ln in lines[start:end]:
    m = re.match(r'\s+([a-z_0-9]+)', ln)
    if not m: continue
    op = m.group(1)
    c = cat(op, ln)
    cnt[c] += 1
    if c == 'lds': lds[op] += 1
valu = sum(cnt[c] for c in CYC)
cyc = sum(cnt[c] * CYC[c] for c in CYC)
print('kernel', want)
print('vector instructions per 8-frame group:', valu, ' =', valu * 8, 'lane-ops per frame;  ~VALU cycles', round(cyc))
print(' '.join(f'{c}={cnt[c]}' for c in ['f32', 'pk', 'dpp', 'cnd', 'mov', 'cmp', 'vint', 'lane', 'trans', 'lds', 'vmem', 'salu', 'wait']))
print('lds:', dict(lds))
