import numpy as np, time, sys
from collections import OrderedDict
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from bench import graphgen
rp,col,st=graphgen.powerlaw_csr(1<<20,1<<24,seed=0)
M=st['M']; K=st['K']; nnz=st['nnz']
lens=np.diff(rp)
cnt=np.bincount(col,minlength=K)
rank=np.empty(K,np.int64); rank[np.argsort(-cnt,kind='stable')]=np.arange(K)
cum=np.cumsum(cnt); bounds=np.array([0]+[int(np.searchsorted(cum, nnz*x/8)) for x in range(1,8)]+[K])
row_of=np.repeat(np.arange(M),lens)
def lru(stream, cap):
    od=OrderedDict(); hit=0
    for c in stream.tolist():
        if c in od:
            od.move_to_end(c); hit+=1
        else:
            od[c]=1
            if len(od)>cap: od.popitem(last=False)
    return hit/len(stream)
CAP=16384
sl = np.searchsorted(bounds, col, side='right')-1
rng=np.random.default_rng(0)
shortrow = (lens<=64)&(lens>0)
def stream_of(rows_in_order, rpw, nwaves):
    """rows are dealt to waves in chunks of rpw (in order); nwaves waves run concurrently, each walking its rows' nnz in
    order at the same pace; returns the interleaved reference stream."""
    starts = rp[rows_in_order]; ln = lens[rows_in_order]
    # wave id and position of each nnz inside its wave
    chunk = np.arange(len(rows_in_order))//rpw
    wl = np.zeros(chunk.max()+1, np.int64); np.add.at(wl, chunk, ln)
    idx = np.concatenate([np.arange(s0, s0+l) for s0,l in zip(starts.tolist(), ln.tolist())])
    ch_of = np.repeat(chunk, ln)
    first = np.r_[True, ch_of[1:]!=ch_of[:-1]]
    segstart = np.maximum.accumulate(np.where(first, np.arange(len(idx)), 0))
    pos = np.arange(len(idx)) - segstart
    rnd = ch_of // nwaves
    key = rnd*10_000_000 + pos*nwaves + (ch_of % nwaves)
    return col[idx][np.argsort(key, kind='stable')]
for x in (0, 5):
    base_rows = np.nonzero(shortrow & (np.arange(M)*8//M == x))[0]
    for rpw in (64, 16):
        s = stream_of(base_rows, rpw, 900)
        print(f'XCD{x} baseline contiguous eighth, {rpw} rows/wave, 900 waves in flight: refs {len(s)} LRU {lru(s,CAP):.4f}', flush=True)
w = rank[col] >= 16384
cnts = np.zeros((M,8),np.int32); np.add.at(cnts, (row_of[w], sl[w]), 1)
noise = rng.random((M,8))*0.5
home = (cnts + noise).argmax(1)
print('rows per home slice', np.bincount(home[shortrow], minlength=8))
for x in (0, 5):
    rows0 = np.nonzero(shortrow & (home==x))[0]
    coldhome = w & (sl==x)
    key = np.full(M, K, np.int64); np.minimum.at(key, row_of[coldhome], col[coldhome])
    srt = rows0[np.argsort(key[rows0], kind='stable')]
    for rpw in (64, 16):
        print(f'XCD{x} home-slice rows, natural order, {rpw} rows/wave: LRU {lru(stream_of(rows0, rpw, 900),CAP):.4f}', flush=True)
        print(f'XCD{x} home-slice rows, sorted by smallest cold home column, {rpw} rows/wave: LRU {lru(stream_of(srt, rpw, 900),CAP):.4f}', flush=True)
