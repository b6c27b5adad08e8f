import sys; sys.path.insert(0,"dgsparse-lib_amd"); sys.path.insert(0,".")
import torch
from bench import graphgen
from dgsparse import _capi
rp,col,st=graphgen.dataset_shaped("reddit",seed=0,device="cuda",as_torch=True)
val=torch.rand(st["nnz"],device="cuda"); X=torch.rand(st["K"],128,device="cuda")
op=int(sys.argv[1]) if len(sys.argv)>1 else 0
for _ in range(3): _capi.spmm(op,rp,col,val,X)
torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True); e0.record()
for _ in range(10): _capi.spmm(op,rp,col,val,X)
e1.record(); torch.cuda.synchronize(); print(e0.elapsed_time(e1)/10, "ms")
