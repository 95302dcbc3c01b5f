"""Offline PGHI on the rank-bitmap queue (the realtime fast path of round 4) -- would it pay?  CPU simulation on one dense 4-second
clip (690 x 513 magnitudes of complex noise): floods in strict magnitude order, follows every group of tied magnitudes
with the block bookkeeping of pghi_hgi_rt_coop_kernel (T = bins a tied entry and its cascade look at, V = bins they visit)
and counts the groups whose blocks collide, i.e. whose order only the reference's heap knows.  usage: pghi_tie_sim.py [seed]"""
import numpy as np, heapq, sys, time
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv)>1 else 0)
T,F = 690,513
mag = np.abs(rng.randn(T,F)+1j*rng.randn(T,F)).astype(np.float32)
tol=1e-2
N=T*F
flat=mag.ravel()
order=np.lexsort((np.arange(N), -flat))
rank=np.empty(N,int); rank[order]=np.arange(N)
lk=flat[order]
same_prev=np.zeros(N,bool); same_prev[1:]=lk[1:]==lk[:-1]
thr=flat.max()*tol
live=flat>=thr   # reference: < max*tol set to abstol
print("tied ranks", same_prev.sum(), "live frac", live.mean())
unvis=live.copy()
# python heap on ranks (strict order); track groups
heap=[]
ptr=0
def reseed():
    global ptr
    while ptr<N and not unvis[order[ptr]]: ptr+=1
    if ptr>=N: return None
    i=order[ptr]; unvis[i]=False; return rank[i]
cur_key=None; accV=set();accT=set();curV=set();curT=set(); groups=0; clashes=0
def close():
    global curV,curT,accV,accT,clashes
    c = bool((curV&accT) or (curT&accV))
    accV|=curV; accT|=curT; curV=set(); curT=set()
    return c
t0=time.time()
r=reseed(); heapq.heappush(heap,r)
pops=0
clash_in_group=False
while True:
    if not heap:
        r=reseed()
        if r is None: break
        if cur_key is not None: curV.add(order[r]); curT.add(order[r])
        heapq.heappush(heap,r); continue
    r=heapq.heappop(heap); pops+=1
    x=lk[r]
    tied = same_prev[r] or (r+1<N and same_prev[r+1])
    if cur_key is not None:
        if x<cur_key:
            if close(): clash_in_group=True
            if clash_in_group: clashes+=1
            cur_key=None; accV=set(); accT=set(); clash_in_group=False
        elif x==cur_key:
            if close(): clash_in_group=True
    if cur_key is None and tied:
        cur_key=x; groups+=1
    i=order[r]; t,f=divmod(i,F)
    tr = cur_key is not None
    if tr: curT.add(i)
    for (tt,ff) in ((t+1,f),(t-1,f),(t,f+1),(t,f-1)):
        if 0<=tt<T and 0<=ff<F:
            j=tt*F+ff
            if tr: curT.add(j)
            if unvis[j]:
                unvis[j]=False; heapq.heappush(heap,rank[j])
                if tr: curV.add(j)
print("pops",pops,"tie groups popped",groups,"clashing groups",clashes,"time",time.time()-t0)
