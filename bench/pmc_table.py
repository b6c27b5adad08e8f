#!/usr/bin/env python3
"""One line per kernel from the pass summaries bench/collect_pmc_configs.sh writes (pmc_summary_*.txt): duration,
HBM-side bytes = (2*FETCH_SIZE + WRITE_SIZE) KB (FETCH doubled: gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md),
fabric reads, L2 hit rate, share of wave cycles spent waiting.    python bench/pmc_table.py FILE... [--all]"""
import re
import sys
from collections import defaultdict


def main():
    files = [f for f in sys.argv[1:] if not f.startswith('--')]
    show_all = '--all' in sys.argv
    for f in files:
        k = None
        d = defaultdict(dict)
        cnt = defaultdict(dict)
        for line in open(f):
            if line.startswith('dgs::'):
                k = line.strip()
            elif 'mean/dispatch' in line and k:
                p = line.split()
                n = int(re.search(r'n=(\d+)', line).group(1))
                if n >= cnt[k].get(p[0], 0):  # the same kernel name can appear for a one-off launch: keep the timed group
                    cnt[k][p[0]] = n
                    d[k][p[0]] = float(p[2])
        print(f'== {f}')
        for k, v in d.items():
            if not show_all and not re.search(r'spmm_fused|spmm_panel|sddmm_|spmm_small', k):
                continue
            us = v.get('~duration_us(profiled)', 0)
            fe, wr = v.get('FETCH_SIZE', 0), v.get('WRITE_SIZE', 0)
            gb = (2 * fe + wr) * 1024 / 1e9
            hit, req = v.get('TCC_HIT_sum', 0), v.get('TCC_HIT_sum', 0) + v.get('TCC_MISS_sum', 0)
            wc = v.get('SQ_WAVE_CYCLES', 0)
            print(f'  {k[:72]:72s} {us:9.1f} us  hbm-side {gb:7.3f} GB ({gb / us * 1e3 if us else 0:5.2f} TB/s)  '
                  f'rdreq {v.get("TCC_EA0_RDREQ_sum", 0) / 1e6:8.2f} M  L2 hit {100 * hit / req if req else 0:5.1f} %  '
                  f'wait {100 * v.get("SQ_WAIT_ANY", 0) / wc if wc else 0:5.1f} %  valu/wave {v.get("SQ_INSTS_VALU", 0) / max(v.get("SQ_WAVES", 1), 1):8.0f}')


if __name__ == '__main__':
    main()
