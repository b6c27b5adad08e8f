import sys, numpy as np, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/dgsparse-lib_amd')
import oracle
from bench import graphgen
from dgsparse import _capi as capi
rp, col, st = graphgen.powerlaw_csr(60000, 900000, alpha=1.9, dmax=15000, seed=21)
M,K=st['M'],st['K']
for F in (128,256):
    D1 = graphgen.features(M, F, 31) - 0.4; D2 = graphgen.features(K, F, 32) - 0.6
    rpt, colt = torch.from_numpy(rp).cuda(), torch.from_numpy(col).cuda()
    plan = capi.spmm_plan(rpt, colt, K, F, force=True)
    got = capi.sddmm(rpt, colt, torch.from_numpy(D1).cuda(), torch.from_numpy(D2).cuda(), plan=plan).cpu().numpy()
    ref = oracle.sddmm(rp, col, D1, D2, threads=8)
    bad = ~np.isclose(got, ref, rtol=1e-4, atol=1e-4)
    deg = np.diff(rp); rowid = np.repeat(np.arange(M), deg)
    print('F',F,'bad',bad.sum(),'nan',np.isnan(got).sum(),'n_units',plan.info.n_units)
    if bad.any():
        br = rowid[bad]; print(' bad rows deg: min',deg[br].min(),'max',deg[br].max(), 'first bad idx', np.argwhere(bad)[:5].ravel(), 'pos in row', (np.argwhere(bad)[:5].ravel()-rp[br[:5]]))
for name in ('synth1m','products'):
    rp, col, st = graphgen.dataset_shaped(name, seed=0, device='cuda', as_torch=True)
    plan = capi.spmm_plan(rp, col, st['K'], 64)
    deg=(rp[1:]-rp[:-1]); print(name,"n_units",plan.info.n_units,"n_pslots",plan.info.n_pslots,"n_long",plan.info.n_long,'nnz',st['nnz'],'nnz in rows>64',int(deg[deg>64].sum()))
