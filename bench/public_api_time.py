import sys; sys.path.insert(0,"dgsparse-lib_amd"); sys.path.insert(0,".")
import torch, dgsparse
from bench import graphgen
rp,col,st=graphgen.dataset_shaped("synth1m",seed=0,device="cuda",as_torch=True)
val=torch.rand(st["nnz"],device="cuda"); X=torch.rand(st["K"],64,device="cuda")
A=dgsparse.SparseTensor(rowptr=rp,col=col,values=val,has_value=True)
def t(fn,n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n
with torch.no_grad():
    print("public spmm_sum no_grad", t(lambda: dgsparse.spmm_sum(A,X,0)))
    print("public spmm_max no_grad", t(lambda: dgsparse.spmm_max(A,X,0)))
Xg=X.clone().requires_grad_()
def fb():
    out=dgsparse.spmm_sum(A,Xg,0); out.sum().backward(); Xg.grad=None
print("public sum fwd+bwd (dX)", t(fb,20))
vg=val.clone().requires_grad_(); A2=dgsparse.SparseTensor(rowptr=rp,col=col,values=vg,has_value=True)
def fb2():
    out=dgsparse.spmm_sum(A2,Xg,0); out.sum().backward(); Xg.grad=None; vg.grad=None
print("public sum fwd+bwd (dX,dW)", t(fb2,20))
