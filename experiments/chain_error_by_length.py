#!/usr/bin/env python3
"""How far is the reference's OWN result from the exact sum, by row length?  (round 4; CPU only, ~1 minute)

The reference computes every (row, feature) as one sequential fp32 chain in CSR order (include/cuda/spmm_cuda.cuh:27-47,
example/util/sp_util.hpp:73-83).  Its rounding error grows like sqrt(len); a reduction tree is far closer to the exact sum, so
beyond some row length "within 1e-5 of the reference" and "accurate" part ways.  This script measures where, on the headline
workload's shape (2^20 rows, power-law degrees, non-negative uniform values and features as in bench.py; CPU generator, so
the numbers are statistics of that distribution, not of the GPU run's exact tensors): the oracle's sequential chains (mul + add
and fmaf) against a float64 sum, rows longer than 256 nnz, binned by length.  It is what the hub threshold (DGS_HUB_CHAIN =
16384) of the default sum / mean schedule was chosen from: below it the chain is within 6.3e-6 of the exact sum (the tree within
~5e-7; 1e-5 is more than 5 sigma of the chain's error away), above 3 10^4 nnz the chain itself leaves 1e-5.   python experiments/chain_error_by_length.py > profiles/r04_chain_error_by_length.txt
"""
import sys, time, numpy as np, torch
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'dgsparse-lib_amd'))
from bench import graphgen
import oracle
M=1<<20
rp,col,st=graphgen.powerlaw_csr(M, M*16, alpha=2.1, dmax=1<<16, cols='powerlaw', seed=0, device='cpu', as_torch=False)
print(st)
rng=np.random.default_rng(1)
lens=np.diff(rp)
# only rows > 256 nnz: build sub-CSR
sel=np.nonzero(lens>256)[0]
print('rows>256', sel.size)
N=64
K=st['K']
val=rng.random(col.shape[0],dtype=np.float32)
X=rng.random((K,N),dtype=np.float32)
# sub-CSR
rs=rp[sel]; re=rp[sel+1]
idx=np.concatenate([np.arange(a,b) for a,b in zip(rs,re)])
rp2=np.zeros(sel.size+1,dtype=np.int32); rp2[1:]=np.cumsum(re-rs)
col2=col[idx].astype(np.int32); val2=val[idx]
t=time.time()
Cseq=oracle.spmm('sum',rp2,col2,val2,X,threads=8)[0]
Cfma=oracle.spmm('sum',rp2,col2,val2,X,fma=True,threads=8)[0]
C64=oracle.spmm_sum_f64(rp2,col2,val2,X)
print('t',time.time()-t)
l2=lens[sel]
for name,C in (('nofma',Cseq),('fma',Cfma)):
    e=np.abs(C.astype(np.float64)-C64)/np.abs(C64)
    print(name)
    for lo,hi in ((256,512),(512,1024),(1024,2048),(2048,4096),(4096,8192),(8192,16384),(16384,32768),(32768,1<<20)):
        m=(l2>lo)&(l2<=hi)
        if m.any():
            ee=e[m]
            print(f'  {lo:6d}-{hi:6d} rows {m.sum():5d} max {ee.max():.3e} mean {ee.mean():.3e} >5e-6: {(ee>5e-6).sum()} >8e-6: {(ee>8e-6).sum()} >1e-5: {(ee>1e-5).sum()}')
d=np.abs(Cseq.astype(np.float64)-Cfma)/np.abs(Cseq)
print('fma vs nofma max', d.max())
