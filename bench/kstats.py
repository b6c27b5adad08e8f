#!/usr/bin/env python3
"""Print a rocprofv3 kernel_stats.csv compactly: calls, average us, total ms, short kernel name.
    python bench/kstats.py DIR_OR_CSV [max_rows]"""
import csv
import glob
import os
import re
import sys


def main():
    path = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    if os.path.isdir(path):
        path = sorted(glob.glob(os.path.join(path, '**', '*kernel_stats.csv'), recursive=True))[0]
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: -float(r['TotalDurationNs']))
    for r in rows[:n]:
        name = re.sub(r'\(.*', '', r['Name']).replace('void ', '')
        print(f"{int(r['Calls']):6d} calls  avg {float(r['AverageNs']) / 1e3:10.1f} us  total {float(r['TotalDurationNs']) / 1e6:9.1f} ms  {name[:90]}")


if __name__ == '__main__':
    main()
