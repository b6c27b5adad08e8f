#!/usr/bin/env python3
"""Full-size checks of tests/test_gpu_strict.py / test_gpu_fullsize.py on the CPU EMULATION (tests/emu), for rounds without a GPU:
the headline graph through the strict-order sum (both modes, feat 64 and 128: every element bit for bit the oracle's chain) and
max / min over a plan that carries hub rows (values and arg ids bit-exact).   python bench/emu_fullsize.py > profiles/r04_emu_fullsize.txt"""
import sys, time, numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests'), os.path.join(ROOT, 'tests', 'emu')):
    sys.path.insert(0, p)
import emu_lib as E, oracle
from bench import graphgen
M=1<<20
rp,col,st=graphgen.powerlaw_csr(M, M*16, alpha=2.1, dmax=1<<16, cols='powerlaw', seed=0)
rng=np.random.default_rng(1)
val=rng.random(col.shape[0],dtype=np.float32)
for N in (64,128):
    X=rng.random((st['K'],N),dtype=np.float32)
    for alg,fma in ((E.ALG_STRICT_SUM,True),(E.ALG_STRICT_NOFMA,False)):
        t=time.time(); C,_=E.spmm(E.SUM,rp,col,val,X,algorithm=alg); dt=time.time()-t
        ref=oracle.spmm('sum',rp,col,val,X,fma=fma,threads=8)[0]
        print(f'N={N} strict fma={fma}: all {C.size} elements bit-exact: {np.array_equal(C.view(np.int32),ref.view(np.int32))}  ({dt:.0f} s emulated)', flush=True)
    if N==64:
        plan=E.spmm_plan(rp,col,st['K'])
        for red in ('max','min'):
            C,Ee=E.spmm(getattr(E,red.upper()),rp,col,val,X,plan=plan)
            Co,Eo=oracle.spmm(red,rp,col,val,X,fma=True,threads=8)
            print(f'N={N} {red} over the plan (n_hub {plan[1].n_hub}): values {np.array_equal(C.view(np.int32),Co.view(np.int32))} E {np.array_equal(Ee,Eo)}', flush=True)
