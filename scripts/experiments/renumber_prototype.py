import numpy as np, sys
sys.path.insert(0,'/root/repo')
from oarfish_amd import synth
M=64
def analyse(st, glue_pct=50, max_seg=32, min_count=64, share=10, max_chain=48, verbose=True):
    T=st.n_txps; rp=st.row_ptr.astype(np.int64); t=st.tid.astype(np.int64); lens=np.diff(rp); R=len(lens)
    row=np.repeat(np.arange(R),lens)
    # anchor: alignment with most others within M (ties smaller id) -- O(k^2) via pairwise within rows: approximate using sorted rows
    # brute force per read in python is slow; vectorise with padding
    K=int(lens.max()); pad=np.full((R,K),-10**9,dtype=np.int64)
    col=np.arange(len(t))-np.repeat(rp[:-1],lens)
    pad[row,col]=t
    n=np.zeros((R,K),dtype=np.int32)
    for j in range(K):
        n[:,j]=((np.abs(pad-pad[:,j:j+1])<=M)&(pad>=0)).sum(1)
    n[pad<0]=-1
    # ties: smaller id -> sort key
    key=n.astype(np.int64)*(10**9)-np.where(pad>=0,pad,0)
    a=pad[np.arange(R),key.argmax(1)]
    far=np.abs(t-a[row])>M
    loc=~far
    lo=np.full(R,10**9); hi=np.full(R,-1)
    np.minimum.at(lo,row[loc],t[loc]); np.maximum.at(hi,row[loc],t[loc])
    cover=np.zeros(T+1,dtype=np.int64); glue=np.zeros(T+1,dtype=np.int64)
    np.add.at(cover,lo,1); np.add.at(cover,hi+1,-1)
    m=hi>lo; np.add.at(glue,lo[m],1); np.add.at(glue,hi[m],-1)
    cv=np.cumsum(cover)[:T]; gl=np.cumsum(glue)[:T]   # gl[t]: boundary t|t+1
    seg_of=np.zeros(T,dtype=np.int64); seg_len=[]; ln=0
    for x in range(T):
        if x>0:
            lo_=min(cv[x],cv[x-1]); cut = ln>=max_seg or lo_<=0 or gl[x-1]*100<glue_pct*lo_
            if cut: seg_len.append(ln); ln=0
        seg_of[x]=len(seg_len); ln+=1
    seg_len.append(ln); seg_len=np.array(seg_len); S=len(seg_len)
    if verbose:
        g=st.gene_of; true_b=(g[1:]!=g[:-1]); my_b=(seg_of[1:]!=seg_of[:-1])
        print('T',T,'genes',g.max()+1,'segments',S,'true boundaries',true_b.sum(),'found',my_b.sum(),'both',(true_b&my_b).sum(),'far',far.sum(), 'far frac', far.mean())
    sa=seg_of[a[row][far]]; stt=seg_of[t[far]]
    keep=sa!=stt; sa,stt=sa[keep],stt[keep]
    lo2=np.minimum(sa,stt); hi2=np.maximum(sa,stt)
    k=lo2*S+hi2; uk,c=np.unique(k,return_counts=True); pa=uk//S; pb=uk%S
    deg=np.zeros(S,dtype=np.int64); np.add.at(deg,pa,c); np.add.at(deg,pb,c)
    best=np.zeros(S,dtype=np.int64); bestp=np.full(S,-1)
    for arr_a,arr_b in ((pa,pb),(pb,pa)):
        o=np.argsort(c,kind='stable')
        # assign in increasing c so the max wins
        best_c=np.zeros(S,dtype=np.int64)
        for i in o:
            if c[i]>=best[arr_a[i]]: best[arr_a[i]]=c[i]; bestp[arr_a[i]]=arr_b[i]
    links=[]
    for b in range(S):
        if bestp[b]<0 or best[b]<min_count or best[b]*100<deg[b]*share: continue
        links.append((best[b],min(b,bestp[b]),max(b,bestp[b])))
    links=sorted(set(links),key=lambda x:(-x[0],x[1],x[2]))
    nbr=[[] for _ in range(S)]; parent=list(range(S)); ids=list(seg_len)
    def find(x):
        while parent[x]!=x: parent[x]=parent[parent[x]]; x=parent[x]
        return x
    made=0
    for c_,x,y in links:
        if len(nbr[x])>=2 or len(nbr[y])>=2: continue
        fx,fy=find(x),find(y)
        if fx==fy or ids[fx]+ids[fy]>max_chain: continue
        nbr[x].append(y); nbr[y].append(x); parent[fx]=fy; ids[fy]+=ids[fx]; made+=1
    order=[]; seen=[False]*S
    for b in range(S):
        if seen[b]: continue
        prev=-1; cur=b
        while True:
            nx=[z for z in nbr[cur] if z!=prev]
            if not nx: break
            prev,cur=cur,nx[0]
        prev=-1
        while True:
            seen[cur]=True; order.append(cur)
            nx=[z for z in nbr[cur] if z!=prev]
            if not nx: break
            prev,cur=cur,nx[0]
    start=np.concatenate([[0],np.cumsum(seg_len)[:-1]])
    fwd=np.zeros(T,dtype=np.int64); pos=0
    for b in order:
        fwd[start[b]:start[b]+seg_len[b]]=np.arange(pos,pos+seg_len[b]); pos+=seg_len[b]
    # far fraction after renumbering: recompute anchors cheaply: keep same anchors (approx)
    t2=fwd[t]; a2=fwd[a]
    far2=np.abs(t2-a2[row])>M
    print(f'glue {glue_pct} share {share} min {min_count}: candidate links {len(links)} made {made}; far before {far.sum()} after (same anchors) {far2.sum()}')
    return fwd
st=synth.make_store(400_000,30_000,seed=29,far='paralog')
for gp,sh,mc in ((50,10,64),(50,5,16),(30,5,16),(70,5,16),(50,2,8)):
    analyse(st,gp,32,mc,sh,48,verbose=(gp,sh,mc)==(50,10,64))
