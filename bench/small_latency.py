import sys, time, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/dgsparse-lib_amd')
from bench import graphgen
from dgsparse import _capi
import dgsparse
for name in ['cora','pubmed']:
    rp,col,st=graphgen.dataset_shaped(name,device='cuda',as_torch=True)
    val=torch.ones(st['nnz'],device='cuda'); X=torch.rand(st['K'],64,device='cuda')
    for _ in range(20): _capi.spmm(0,rp,col,val,X)
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(1000): _capi.spmm(0,rp,col,val,X)
    t1=time.perf_counter()-t; torch.cuda.synchronize(); t2=time.perf_counter()-t
    print(name,'host enqueue us/call',t1/1000*1e6,'total us/call',t2/1000*1e6)
    tcsr=torch.sparse_csr_tensor(rp,col,val,size=(st['M'],st['K']))
    A=dgsparse.SparseTensor.from_torch_sparse_csr_tensor(tcsr,True)
    for _ in range(20): dgsparse.spmm_sum(A,X,0)
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(1000): dgsparse.spmm_sum(A,X,0)
    torch.cuda.synchronize(); print(name,'dgsparse.spmm_sum us/call',(time.perf_counter()-t)/1000*1e6)
    t=time.perf_counter()
    for _ in range(1000): torch.sparse.mm(tcsr,X)
    torch.cuda.synchronize(); print(name,'torch.sparse.mm (hipSPARSE) us/call',(time.perf_counter()-t)/1000*1e6)
    st_=A.storage
    args=(st_.rowptr(),st_.col(),st_.values(),st_.colptr(),st_.row(),st_.csr2csc(),X,True,0)
    for _ in range(20): torch.ops.dgsparse_spmm.spmm_sum(*args)
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(1000): torch.ops.dgsparse_spmm.spmm_sum(*args)
    torch.cuda.synchronize(); print(name,'torch.ops.dgsparse_spmm.spmm_sum us/call',(time.perf_counter()-t)/1000*1e6)
    for _ in range(20): torch.ops.dgsparse_spmm.spmm_raw(0,st_.rowptr(),st_.col(),st_.values(),X,True,0)
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(1000): torch.ops.dgsparse_spmm.spmm_raw(0,st_.rowptr(),st_.col(),st_.values(),X,True,0)
    torch.cuda.synchronize(); print(name,'torch.ops.dgsparse_spmm.spmm_raw us/call',(time.perf_counter()-t)/1000*1e6)
    Xg=X.clone().requires_grad_()
    def fb():
        o=dgsparse.spmm_sum(A,Xg,0); o.sum().backward(); Xg.grad=None
    for _ in range(20): fb()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(300): fb()
    torch.cuda.synchronize(); print(name,'fwd+bwd (dX only) us/iter',(time.perf_counter()-t)/300*1e6)
    # HIP-graph replay of the same forward (capture-safe C ABI: no allocation, no sync, caller's stream)
    s_ = torch.cuda.Stream(); s_.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s_):
        for _ in range(3): dgsparse.spmm_sum(A, X, 0)
    torch.cuda.current_stream().wait_stream(s_)
    g_ = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g_):
        out_ = dgsparse.spmm_sum(A, X, 0)
    for _ in range(20): g_.replay()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(1000): g_.replay()
    torch.cuda.synchronize(); print(name,'HIP graph replay us/call',(time.perf_counter()-t)/1000*1e6)
