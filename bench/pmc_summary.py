#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc output (counter_collection.csv files under a directory tree): mean counter
value per dispatch for every dgs:: kernel.  Usage: python bench/pmc_summary.py DIR [DIR...]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(dirs):
    acc = defaultdict(lambda: defaultdict(list))
    for d in dirs:
        for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
            per_dispatch = defaultdict(float)
            names = {}
            dur = {}
            for row in csv.DictReader(open(f)):
                k = row.get('Kernel_Name', '')
                if 'dgs::' not in k:
                    continue
                key = (row['Dispatch_Id'], row['Counter_Name'])
                per_dispatch[key] += float(row['Counter_Value'])
                names[row['Dispatch_Id']] = k.split('(')[0].replace('void ', '') + f"  [grid {row.get('Grid_Size', '?')}]"
                dur[row['Dispatch_Id']] = (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3
            for (did, cn), v in per_dispatch.items():
                acc[names[did]][cn].append(v)
            for did, us in dur.items():
                acc[names[did]]['~duration_us(profiled)'].append(us)
    for k in sorted(acc):
        print(k)
        for cn in sorted(acc[k]):
            v = acc[k][cn]
            print(f'    {cn:32s} mean/dispatch {sum(v) / len(v):16.1f}   (n={len(v)})')


if __name__ == '__main__':
    main(sys.argv[1:] or ['.'])
