import sys, time, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/dgsparse-lib_amd')
from bench import graphgen
from dgsparse import _capi
import dgsparse
for name in ['arxiv','synth1m','products','reddit']:
    rp,col,st=graphgen.dataset_shaped(name,device='cuda',as_torch=True)
    val=torch.rand(st['nnz'],device='cuda')
    for _ in range(2): _capi.csr2csc(rp,col,val,st['K'])
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(5): _capi.csr2csc(rp,col,val,st['K'])
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/5
    t=time.perf_counter(); A=dgsparse.SparseTensor(rowptr=rp,col=col,values=val,has_value=True); torch.cuda.synchronize(); dt2=time.perf_counter()-t
    tc=torch.sparse_csr_tensor(rp,col,val,size=(st['M'],st['K']))
    torch.cuda.synchronize(); t=time.perf_counter(); tt=tc.to_sparse_csc(); torch.cuda.synchronize(); dt3=time.perf_counter()-t
    print(name, st['nnz'], 'csr2csc ms', round(dt*1e3,3), 'SparseTensor() ms', round(dt2*1e3,3), 'torch to_sparse_csc ms', round(dt3*1e3,3))
