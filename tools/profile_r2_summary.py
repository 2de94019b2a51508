#!/usr/bin/env python3
"""Summarise the round-2 rocprofv3 passes of tools/profile_r2.sh: per-kernel averages of every counter,
per-kernel durations, and the derived files profiles/r2_traffic.json / r2_instr.json."""
import collections
import csv
import glob
import json
import os
import sys

O = sys.argv[1]
B, T = 1024, 99


def short(name):
    for key in ('mfcc512_kernel', 'mfcc1536_kernel', 'delta_rows_kernel', 'delta_tiled_kernel', 'vad_sum_kernel',
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
for d in ('stats1', 'stats3', 'stats1536', 'statsvad'):
    res[d] = stats(d)
    print(f'== kernel durations, {d}')
    for k, v in sorted(res[d].items()):
        print(f'   {k:28s} calls {v["calls"]:6d}  avg {v["avg_ns"] / 1e3:8.2f} us  min {v["min_ns"] / 1e3:8.2f}  max {v["max_ns"] / 1e3:8.2f}')
for d in ('pmc_a', 'pmc_b', 'fetch', 'write', 'pmc1536_a', 'pmc1536_b', 'pmcvad_b'):
    res[d] = counters(d)
    print(f'== counters, {d} (average per dispatch)')
    for k, cs in sorted(res[d].items()):
        print(f'   {k} (n={cs["_n"]})')
        for c, v in sorted(cs.items()):
            if c != '_n':
                print(f'       {c:26s} {v:18.1f}')
json.dump(res, open(f'{O}/summary.json', 'w'), indent=1)

# derived: HBM traffic per launch (FETCH_SIZE is in KB and reports half of wide streaming reads on gfx950)
try:
    fm, wm = res['fetch']['mfcc512_kernel'], res['write']['mfcc512_kernel']
    fd, wd = res['fetch']['delta_rows_kernel'], res['write']['delta_rows_kernel']
    traffic = {
        '_comment': 'HBM traffic per launch at configs[1] (1024 x 1 s -> 101376 frames) from separate rocprofv3 --pmc '
                    'FETCH_SIZE / WRITE_SIZE passes of tools/kbench.py (units KB); FETCH_SIZE doubled (gfx950 reports '
                    'half of wide coalesced streaming reads, MI355X_MICROARCH.md section HBM), WRITE_SIZE as is. '
                    'step = fused MFCC kernel (dense [T, 13] cepstra) + delta_rows_kernel ([T, 39] rows).',
        'frames_per_launch': B * T,
        'mfcc512_kernel': {'FETCH_SIZE_KB': fm['FETCH_SIZE'], 'WRITE_SIZE_KB': wm['WRITE_SIZE']},
        'delta_rows_kernel': {'FETCH_SIZE_KB': fd['FETCH_SIZE'], 'WRITE_SIZE_KB': wd['WRITE_SIZE']},
        'traffic_bytes_per_launch': int(2 * fm['FETCH_SIZE'] * 1024 + wm['WRITE_SIZE'] * 1024),
        'algorithmic_bytes_per_launch': int((4.0 * 16000 / T + 52) * B * T),
        'step_traffic_bytes': int((2 * fm['FETCH_SIZE'] + wm['WRITE_SIZE'] + 2 * fd['FETCH_SIZE'] + wd['WRITE_SIZE']) * 1024),
        'step_algorithmic_bytes': int((4.0 * 16000 / T + 156) * B * T),
    }
    json.dump(traffic, open(f'{O}/r2_traffic.json', 'w'), indent=1)
    print('traffic:', json.dumps(traffic))
    b = res['pmc_b']['mfcc512_kernel']
    a = res['pmc_a']['mfcc512_kernel']
    groups = B * T // 8        # flat grouping: 8 frames per wave iteration, no idle frame slots
    instr = {
        '_comment': 'per-dispatch SQ counters of mfcc512_kernel at configs[1] (tools/kbench.py under rocprofv3 --pmc); '
                    'SQ_INSTS_* are wave instructions summed over the chip',
        'valu_wave_instr_per_group': b['SQ_INSTS_VALU'] / groups, 'lds_wave_instr_per_group': b['SQ_INSTS_LDS'] / groups,
        'salu_wave_instr_per_group': b['SQ_INSTS_SALU'] / groups,
        'lds_busy_cycles_per_group': b['SQ_LDS_IDX_ACTIVE'] / groups, 'lds_conflict_cycles_per_group': b['SQ_LDS_BANK_CONFLICT'] / groups,
        'valu_lane_ops_per_frame': b['SQ_INSTS_VALU'] * 64.0 / (B * T),
        'wave_cycles': a['SQ_WAVE_CYCLES'], 'wait_any': a['SQ_WAIT_ANY'], 'wait_inst_any': a['SQ_WAIT_INST_ANY'],
        'active_inst_any': a['SQ_ACTIVE_INST_ANY'], 'active_inst_valu': a['SQ_ACTIVE_INST_VALU'],
        'active_inst_lds': a['SQ_ACTIVE_INST_LDS'], 'busy_cycles': a['SQ_BUSY_CYCLES'],
    }
    json.dump(instr, open(f'{O}/r2_instr.json', 'w'), indent=1)
    print('instr:', json.dumps(instr))
except KeyError as e:
    print('derived files skipped, missing', e)
