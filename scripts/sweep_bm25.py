#!/usr/bin/env python3
"""BM25 kernel timing sweep on a synthetic Zipf corpus (GPU box):

    python scripts/sweep_bm25.py [n_docs]      # 8 192 queries of 8 tokens, k = 10 and k = 80

Prints microseconds per 8 192 queries and checks a sample of the results bit for bit against
the CPU oracle.  Used for the slab / waves-per-block decisions recorded in csrc/bm25.hip (bm_plan).
"""
import sys, time, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import torch
from legal_rag_amd import _native
from oracle import bm25 as OB
_native.load()
rng=np.random.default_rng(5)
n_docs=int(sys.argv[1]) if len(sys.argv)>1 else 1260; vocab=4000
p=1.0/np.arange(1,vocab+1); p/=p.sum()
docs=[[f"w{j}" for j in rng.choice(vocab,size=int(rng.integers(20,300)),p=p)] for _ in range(n_docs)]
ob=OB.BM25Okapi(docs); csr=OB.to_csr(ob)
gi=_native.BM25Index(csr["term_ptr"],csr["post_doc"],csr["post_tf"],csr["idf"],csr["doc_len"],ob.avgdl,ob.k1,ob.b)
nq=int(sys.argv[2]) if len(sys.argv)>2 else 8192
tid=[[int(t) for t in rng.choice(vocab,size=8,p=p)] for _ in range(nq)]
qt,qp=_native.BM25Index.pack_queries(tid)
dev=torch.device('cuda',0)
qtd=torch.from_numpy(qt).to(dev); qpd=torch.from_numpy(qp).to(dev)
for k in (10,80):
    s=torch.empty((nq,k),dtype=torch.float64,device=dev); i=torch.empty((nq,k),dtype=torch.int64,device=dev)
    gi.reserve(nq,k,len(qt))
    st=int(torch.cuda.current_stream().cuda_stream)
    for _ in range(3): gi.search_device(qtd.data_ptr(),qpd.data_ptr(),nq,k,s.data_ptr(),i.data_ptr(),st)
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(20): gi.search_device(qtd.data_ptr(),qpd.data_ptr(),nq,k,s.data_ptr(),i.data_ptr(),st)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/20
    # check vs oracle on a few
    ih=i.cpu().numpy(); sh=s.cpu().numpy()
    words=[f"w{j}" for j in range(vocab)]
    inv={v:k_ for k_,v in csr["vocab"].items()}
    ok=True
    for q in range(0,nq,max(1,nq//8)):
        exp=OB.search(ob,[inv[t] for t in tid[q]],k)
        ok&=(ih[q].tolist()==[e[0] for e in exp]) and (sh[q].tolist()==[e[1] for e in exp])
    print('n_docs',n_docs,'k',k,'us per %d queries %.1f'%(nq,dt*1e6),'bit-exact',ok)
