#!/usr/bin/env python3
"""profiles/r04_fuzz/emu_campaigns.txt from the logs of the emulation fuzz campaigns (gpurun_out/emu*/): the last progress line of
every process, anything that looks like a report, totals.    python bench/summarize_emu_campaigns.py > profiles/r04_fuzz/emu_campaigns.txt"""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CAMPAIGNS = (
    ('emufuzz', 'plain (before the every-width / single-launch / slice-by-slice work)'),
    ('emuasan', 'AddressSanitizer (same source)'),
    ('emuubsan', 'UndefinedBehaviorSanitizer (same source + the null-offset fix it asked for)'),
    ('emufinal', 'every width + single-launch hub rows (stopped by a session restart after ~14 min: last progress line)'),
    ('emufinal2', '+ DGS_HUB_XCD 0 / 1 random: plain / reversed / random fiber schedules, ASAN, UBSAN'),
    ('emumulti', 'same source, 40 resident workgroups taking turns, random / reversed dispatch and fiber order'),
    ('emufinal3', 'final source (DGS_HUB_XCD 0 / 1 / 2 random): plain, ASAN, UBSAN, and resident workgroups in random / reversed order '
                  '(the multi_* processes re-run after the emulator livelock fix, tests/emu/README.md)'),
)
out, tot = [], {}
for d, label in CAMPAIGNS:
    files = sorted(glob.glob(os.path.join(ROOT, 'gpurun_out', d, '*.log')))
    if not files:
        continue
    out.append(f'# {d}: {label}')
    for f in files:
        text = open(f, errors='replace').read().split('\n')
        lines = [ln for ln in text if 'cases green' in ln]
        last = lines[-1] if lines else '(no output)'
        m = re.search(r'(\d+) cases green', last)
        tot[d] = tot.get(d, 0) + (int(m.group(1)) if m else 0)
        rep = [ln for ln in text if re.search(r'ERROR|runtime error|Traceback|DEADLOCK|AssertionError', ln)]
        out.append(f'{d}/{os.path.basename(f)}: {last[:72]}' + (f'   !! {rep[0][:80]}' if rep else ''))
out.append('')
out.append('cases green per campaign: ' + ', '.join(f'{k} {v}' for k, v in tot.items()) + f'; total {sum(tot.values())}; '
           'sanitizer / deadlock reports / tracebacks: ' + ('none' if not any('!!' in ln for ln in out) else 'SEE ABOVE'))
print('\n'.join(out))
