#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV output per kernel: python tools/pmc_summary.py <dir> [<dir> ...]"""
import collections
import csv
import glob
import sys

for d in sys.argv[1:]:
    for f in glob.glob(f'{d}/**/*_counter_collection.csv', recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r['Kernel_Name'][:48]][r['Counter_Name']].append(float(r['Counter_Value']))
        for k, v in acc.items():
            if not any(t in k for t in ('mfcc', 'delta', 'generic', 'vad', 'endpoint', 'trim')):
                continue
            n = len(next(iter(v.values())))
            print(f'{d} | {k} | n={n}')
            for c, x in sorted(v.items()):
                print(f'    {c:28s} {sum(x) / len(x):16.1f}')
