"""TEST INFRASTRUCTURE ONLY: does the emulation notice a broken barrier protocol?  Each mutant is a copy of dgsparse-lib_amd/csrc
with ONE barrier pair of the hub workgroup (spmm_strict.h, strict_hub_coop: the role-specialised waves share a tile and
two barriers per phase, four in the prologue) removed from BOTH roles - so the workgroup still terminates and only the missing
ordering between the waves is left to find - built into tests/emu/_build_mut/<name>/ and run on a matrix with hub rows under the
fiber schedules DGS_EMU_ORDER = fwd, rev and rand:1..3.  A mutant is KILLED when some schedule gives a result that differs from
the oracle's chain (or the run aborts); the unmutated sources must pass under every schedule.  The table this prints is kept in
profiles/ (python tests/emu/mutation_check.py > profiles/r04_emu_mutants.txt)."""
import os
import re
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, 'dgsparse-lib_amd', 'csrc')
ORDERS = ['fwd', 'rev', 'rand:1', 'rand:2', 'rand:3']


def drop(src, pattern, count):
    """remove `count` lines matching `pattern` (a regex on the whole line) inside strict_hub_coop"""
    a = src.index('__device__ __forceinline__ void strict_hub_coop(')
    b = src.index('strict_hub_coop_v1(')
    body = src[a:b]
    lines = body.split('\n')
    hit = [i for i, ln in enumerate(lines) if re.search(pattern, ln)]
    assert len(hit) == count, (pattern, len(hit), count)
    body = '\n'.join(ln for i, ln in enumerate(lines) if i not in hit)
    return src[:a] + body + src[b:]


def chain_prologue(src, keep):
    """the chain wave matches the gather waves' prologue with four unlabeled barriers: keep `keep` of them"""
    four = ('    __syncthreads();  // P1 .. P4: the gather waves\' prologue (column tiles of phases 0 and 1)\n'
            '    __syncthreads();\n    __syncthreads();\n    __syncthreads();\n')
    assert four in src
    return src.replace(four, '    __syncthreads();\n' * keep)


MUTANTS = {
    'none': lambda s: s,
    'A (tile written -> chained)': lambda s: drop(drop(s, r'__syncthreads\(\);  // A$', 1), r'__syncthreads\(\);  // A: the tile', 1),
    'B (tile chained -> overwritten)': lambda s: drop(drop(s, r'__syncthreads\(\);  // B$', 1), r'__syncthreads\(\);  // B: the chain', 1),
    'P1 (columns of phase 0 written -> read)': lambda s: chain_prologue(drop(s, r'__syncthreads\(\);  // P1: ', 1), 3),
    'P2 (columns of phase 0 read -> overwritten)': lambda s: chain_prologue(drop(s, r'__syncthreads\(\);  // P2: ', 1), 3),
    'P3 (columns of phase 1 written -> read)': lambda s: chain_prologue(drop(s, r'__syncthreads\(\);  // P3: ', 1), 3),
    'P4 (columns of phase 1 read -> overwritten)': lambda s: chain_prologue(drop(s, r'__syncthreads\(\);  // P4: ', 1), 3),
}

RUN = r'''
import os, sys
import numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, %(here)r)
import oracle
import emu_lib as E
rng = np.random.default_rng(3)
M, K = 66000, 4000
deg = rng.integers(0, 2, M)
for r, d in ((5, 3500), (40000, 2100), (65999, 1300), (77, 1025)):
    deg[r] = d
rp = np.zeros(M + 1, np.int32); rp[1:] = np.cumsum(deg)
col = np.concatenate([rng.choice(K, d, replace=False) for d in deg]).astype(np.int32)  # unsorted columns: more tile turnover
val = rng.random(col.size, dtype=np.float32)
E.set_env(DGS_HUB_CHAIN=1024, DGS_NBU=8)
bad = 0
for N in (64, 16):
    X = rng.random((K, N), dtype=np.float32)
    ref, _ = oracle.spmm('sum', rp, col, val, X, fma=True)
    C, _ = E.spmm(E.SUM, rp, col, val, X)
    hub = deg > 1024
    bad += int((C[hub].view(np.int32) != ref[hub].view(np.int32)).sum())
print('BAD', bad, 'GATE', E.lib().dgs_spmm_hub_gate())
'''


def main():
    out = os.path.join(HERE, '_build_mut')
    src0 = open(os.path.join(CSRC, 'spmm_strict.h')).read()
    rows = []
    for k, (name, fn) in enumerate(MUTANTS.items()):
        d = os.path.join(out, f'm{k}')
        shutil.rmtree(d, ignore_errors=True)
        shutil.copytree(CSRC, os.path.join(d, 'csrc'), ignore=shutil.ignore_patterns('*.so', '*.o', 'build'))
        open(os.path.join(d, 'csrc', 'spmm_strict.h'), 'w').write(fn(src0))
        r = subprocess.run(['make', '-C', HERE, '-j8', f'CSRC={d}/csrc', f'B={d}/b'], capture_output=True, text=True)
        if r.returncode != 0:
            sys.exit(f'{name}: build failed\n' + r.stderr[-2000:])
        res = []
        for o in ORDERS:
            env = dict(os.environ, DGS_EMU_LIB=f'{d}/b/libdgs_emu.so', DGS_EMU_ORDER=o)
            p = subprocess.run([sys.executable, '-c', RUN % dict(root=ROOT, here=HERE)], capture_output=True, text=True, env=env,
                               timeout=1800)
            m = re.search(r'BAD (\d+) GATE (-?\d+)', p.stdout)
            if p.returncode != 0 or not m:
                res.append('abort' if 'DEADLOCK' not in p.stderr else 'deadlock')
            else:  # (the device self-test the library is gated on ran when it was loaded: did IT notice?)
                res.append(('ok' if m.group(1) == '0' else f'{m.group(1)} wrong') + ('' if m.group(2) == '1' else ' [self-test fails]'))
        rows.append((name, res))
        print(f'{name:48s} ' + '  '.join(f'{o}: {x}' for o, x in zip(ORDERS, res)), flush=True)
    ok = all(x == 'ok' for x in rows[0][1]) and all(any(x != 'ok' for x in r) for _, r in rows[1:])
    print('unmutated sources pass every schedule, every mutant is killed by at least one' if ok else 'MUTATION CHECK FAILED')
    shutil.rmtree(out, ignore_errors=True)
    return 0 if ok else 1


if __name__ == '__main__':
    sys.exit(main())
