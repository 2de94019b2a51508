#!/usr/bin/env python3
"""Summarise the round-4 rocprofv3 passes of tools/profile_r4.sh: per-kernel averages of every counter,
per-kernel durations, and the derived files r4_traffic.json / r4_instr.json (copied to profiles/ by hand)."""
import collections
import csv
import glob
import re
import json
import os
import sys

O = sys.argv[1]
B, T = 1024, 99


def short(name):
    m = re.search(r'mfcc512_kernel<([^>]*)>', name)
    if m:   # the last template argument is the fused-delta window (0: the MFCC-only kernel)
        args = [a.strip() for a in m.group(1).split(',')]
        fd = args[7] if len(args) > 7 else '0'      # <NROWS, NI, CAPS, NSTAGE, DTYPE, WAVES, RAGGED, FD>
        ragged = len(args) > 6 and args[6] in ('true', '1')
        return 'mfcc512_fused' if fd not in ('0', 'false') else ('mfcc512_ragged' if ragged else 'mfcc512_kernel')
    for key in ('mfcc1536_kernel', 'delta_rows_kernel', 'delta_tiled_kernel', 'vad_sum_kernel', 'vad_scan_kernel',
                'vad_vec_kernel', 'endpoint_rule_kernel', 'endpoint_layout_kernel', 'trim_scale_kernel',
                'f512_group_prefix_kernel', 'features_generic_kernel', 'prefix_ceil_kernel'):
        if key in name:
            return key
    return None


def counters(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f'{O}/{d}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r['Kernel_Name'])
            if k:
                acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} | {'_n': len(next(iter(cs.values())))} for k, cs in acc.items()}


def stats(d):
    out = {}
    for f in glob.glob(f'{O}/{d}/**/*kernel_stats.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r['Name'])
            if k:
                out[k] = {'calls': int(r['Calls']), 'avg_ns': float(r['AverageNs']), 'min_ns': float(r['MinNs']),
                          'max_ns': float(r['MaxNs']), 'pct': float(r['Percentage'])}
    return out


res = {}
for d in ('stats1', 'stats3', 'stats1536', 'statspipe', 'statspipecopy'):
    res[d] = stats(d)
    print(f'== kernel durations, {d}')
    for k, v in sorted(res[d].items()):
        print(f'   {k:28s} calls {v["calls"]:6d}  avg {v["avg_ns"] / 1e3:8.2f} us  min {v["min_ns"] / 1e3:8.2f}  max {v["max_ns"] / 1e3:8.2f}')
for d in ('pmc_a', 'pmc_b', 'fetch', 'write', 'pmc1536_a', 'pmc1536_b', 'fetch1536', 'write1536'):
    res[d] = counters(d)
    print(f'== counters, {d} (average per dispatch)')
    for k, cs in sorted(res[d].items()):
        print(f'   {k} (n={cs["_n"]})')
        for c, v in sorted(cs.items()):
            if c != '_n':
                print(f'       {c:26s} {v:18.1f}')
json.dump(res, open(f'{O}/summary.json', 'w'), indent=1)

# derived: HBM traffic per launch (FETCH_SIZE is in KB and reports half of wide streaming reads on gfx950)
clock = None
try:
    for ln in open(f'{O}/stamps.log'):
        m = re.search(r'shader clock\s+([0-9.]+) GHz', ln)
        if m:
            clock = float(m.group(1))
except OSError:
    pass
try:
    ff, wf = res['fetch']['mfcc512_fused'], res['write']['mfcc512_fused']
    fm, wm = res['fetch']['mfcc512_kernel'], res['write']['mfcc512_kernel']
    traffic = {
        '_comment': 'HBM traffic per launch at configs[1] (1024 x 1 s -> 101376 frames) from separate rocprofv3 --pmc '
                    'FETCH_SIZE / WRITE_SIZE passes of tools/kbench.py (units KB); FETCH_SIZE doubled (gfx950 reports '
                    'half of wide coalesced streaming reads, MI355X_MICROARCH.md section HBM), WRITE_SIZE as is. '
                    'The step is ONE kernel: the fused MFCC + delta + delta-delta kernel ([T, 39] rows).',
        'frames_per_launch': B * T, 'kernel': 'mfcc512_kernel<..., FD = 2> (fused)',
        'mfcc512_fused': {'FETCH_SIZE_KB': ff['FETCH_SIZE'], 'WRITE_SIZE_KB': wf['WRITE_SIZE']},
        'mfcc512_kernel_mfcc_only': {'FETCH_SIZE_KB': fm['FETCH_SIZE'], 'WRITE_SIZE_KB': wm['WRITE_SIZE']},
        'traffic_bytes_per_launch': int(2 * ff['FETCH_SIZE'] * 1024 + wf['WRITE_SIZE'] * 1024),
        'algorithmic_bytes_per_launch': int((4.0 * 16000 / T + 156) * B * T),
        'step_traffic_bytes': int(2 * ff['FETCH_SIZE'] * 1024 + wf['WRITE_SIZE'] * 1024),
        'step_algorithmic_bytes': int((4.0 * 16000 / T + 156) * B * T),
    }
    if 'mfcc1536_kernel' in res.get('fetch1536', {}) and 'mfcc1536_kernel' in res.get('write1536', {}):
        f15, w15 = res['fetch1536']['mfcc1536_kernel'], res['write1536']['mfcc1536_kernel']
        traffic['mfcc1536_kernel'] = {'FETCH_SIZE_KB': f15['FETCH_SIZE'], 'WRITE_SIZE_KB': w15['WRITE_SIZE'],
                                      'traffic_bytes_per_launch': int(2 * f15['FETCH_SIZE'] * 1024 + w15['WRITE_SIZE'] * 1024),
                                      'algorithmic_bytes_per_launch': int(4 * 512 * 48000 + 4 * 50176 * 13)}
    json.dump(traffic, open(f'{O}/r4_traffic.json', 'w'), indent=1)
    print('traffic:', json.dumps(traffic))
    groups = B * T // 8
    instr = {'_comment': 'per-dispatch SQ counters at configs[1] (tools/kbench.py under rocprofv3 --pmc); SQ_INSTS_* are wave '
                         'instructions summed over the chip; valu_lane_ops_per_frame is of the FUSED kernel (the step)'}
    for key in ('mfcc512_fused', 'mfcc512_kernel'):
        b, a = res['pmc_b'][key], res['pmc_a'][key]
        instr[key] = {
            'valu_wave_instr_per_group': b['SQ_INSTS_VALU'] / groups, 'lds_wave_instr_per_group': b['SQ_INSTS_LDS'] / groups,
            'salu_wave_instr_per_group': b['SQ_INSTS_SALU'] / groups,
            'lds_busy_cycles_per_group': b['SQ_LDS_IDX_ACTIVE'] / groups, 'lds_conflict_cycles_per_group': b['SQ_LDS_BANK_CONFLICT'] / groups,
            'valu_lane_ops_per_frame': b['SQ_INSTS_VALU'] * 64.0 / (B * T),
            'wave_cycles': a['SQ_WAVE_CYCLES'], 'wait_any': a['SQ_WAIT_ANY'], 'wait_inst_any': a['SQ_WAIT_INST_ANY'],
            'active_inst_any': a['SQ_ACTIVE_INST_ANY'], 'active_inst_valu': a['SQ_ACTIVE_INST_VALU'],
            'active_inst_lds': a['SQ_ACTIVE_INST_LDS'], 'busy_cycles': a['SQ_BUSY_CYCLES'],
        }
    instr['valu_lane_ops_per_frame'] = instr['mfcc512_fused']['valu_lane_ops_per_frame']
    if clock:
        instr['shader_clock_ghz_under_load'] = clock
    json.dump(instr, open(f'{O}/r4_instr.json', 'w'), indent=1)
    print('instr:', json.dumps(instr))
except KeyError as e:
    print('derived files skipped, missing', e)
