import numpy as np, time, sys
from collections import OrderedDict
import sys; sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
from bench import graphgen
t=time.time()
rp,col,st=graphgen.powerlaw_csr(1<<20,1<<24,seed=0)
M=st['M']; K=st['K']; nnz=st['nnz']
lens=np.diff(rp)
cnt=np.bincount(col,minlength=K)
rank=np.empty(K,np.int64); rank[np.argsort(-cnt,kind='stable')]=np.arange(K)
print('gen',time.time()-t, flush=True)
# column slices by equal refs
cum=np.cumsum(cnt); bounds=[0]+[int(np.searchsorted(cum, nnz*x/8)) for x in range(1,8)]+[K]
row_of=np.repeat(np.arange(M),lens)
long_nnz = lens[row_of]>64
sliced_nnz = lens[row_of]>256
def lru(stream, cap, hotmask=None):
    """stream: array of column ids; cap in rows. hotmask: bool per col: cold refs bypass (no alloc)."""
    od=OrderedDict(); hit=0
    for c in stream.tolist():
        if c in od:
            od.move_to_end(c); hit+=1
        elif hotmask is None or hotmask[c]:
            od[c]=1
            if len(od)>cap: od.popitem(last=False)
    return hit/len(stream)
CAP=16384
# XCD 0 streams
# (a) plan-free: rows part: XCD gets contiguous eighth of rows; interleave of many waves ~ random order within ~ window; approximate by row order
rows0 = (row_of < M//8) & ~long_nnz
s_rows = col[rows0]
print('rows part refs',len(s_rows), 'LRU hit', lru(s_rows,CAP), flush=True)
for R in [8192,12288,16384]:
    print('  rows part static top',R, lru(s_rows,CAP,rank<R), 'ideal', (rank[s_rows]<R).mean(), flush=True)
# (b) sliced unit stream for slice 0: nnz of sliced rows with col in slice 0, processed in unit order: units sorted by start col; approx: sort nnz by (chunk start col): emulate units: per row segment in slice chunked by 256
sel = sliced_nnz & (col < bounds[1])
idx=np.nonzero(sel)[0]
# unit id: consecutive idx in same row and chunk of 256
r=row_of[idx]; 
first=np.r_[True, r[1:]!=r[:-1]]
segstart=np.maximum.accumulate(np.where(first,np.arange(len(idx)),0))
k=(np.arange(len(idx))-segstart)//256
ukey=r*1000+k
ufirst=np.r_[True, ukey[1:]!=ukey[:-1]]
uid=np.cumsum(ufirst)-1
ustartcol=col[idx][ufirst]
order_units=np.argsort(ustartcol,kind='stable')
# waves: 512 in flight: unit u at sorted position s processed in round s//512; within round interleave nnz round-robin
pos=np.empty(len(order_units),np.int64); pos[order_units]=np.arange(len(order_units))
upos=pos[uid]
within=np.arange(len(idx))-np.maximum.accumulate(np.where(ufirst,np.arange(len(idx)),0))
key=(upos//512)*100000+ (within//4)*600 + (upos%512)
s_units=col[idx][np.argsort(key,kind='stable')]
print('sliced slice0 refs',len(s_units),'units',len(order_units),'LRU hit',lru(s_units,CAP), flush=True)
for R in [65536,98304,131072]:
    print('  sliced static top',R, lru(s_units,CAP,rank<R), 'ideal',(rank[s_units]<R).mean(), flush=True)
# (c) unsliced long units (64<len<=256) hashed to XCD: 1/8 of them
um = long_nnz & ~sliced_nnz & ((row_of*2654435761 % 8)==0)
s_un = col[um]
print('unsliced refs',len(s_un),'LRU',lru(s_un,CAP),'static12k',lru(s_un,CAP,rank<12288), flush=True)
# (d) everything interleaved on XCD0 proportional
def interleave(arrs):
    n=sum(len(a) for a in arrs); keys=np.concatenate([np.arange(len(a))/len(a) for a in arrs]); 
    return np.concatenate(arrs)[np.argsort(keys,kind='stable')]
s_all=interleave([s_rows,s_units,s_un])
print('fused all LRU',lru(s_all,CAP), flush=True)
hm = (rank<6144) | ((rank<65536)&(np.arange(K)<bounds[1]))
print('fused policy g6k+s64k', lru(s_all,CAP,hm), flush=True)
hm = (rank<8192) | ((rank<49152)&(np.arange(K)<bounds[1]))
print('fused policy g8k+s48k', lru(s_all,CAP,hm), flush=True)
hm = (rank<4096) | ((rank<98304)&(np.arange(K)<bounds[1]))
print('fused policy g4k+s96k', lru(s_all,CAP,hm), flush=True)
# plan-free everything in row order for XCD0
s_pf = col[row_of < M//8]
print('plan-free row order LRU', lru(s_pf,CAP), flush=True)
