"""Design study (not part of the product or the tests; round 5): would the flat scan's threshold filter be able to ABANDON
a tile after a 32-subspace slice of the ADC tables?  Replays the bench's generator on the CPU (300 k x 768, PQ-96 by torch Lloyd), forms the
partial sums after one and two slices, bounds the rest by the per-subspace maxima and counts how many (query, candidate) PAIRS and how
many WAVES (64 lanes x 8 candidates x 4 queries: a wave-uniform skip needs all of them) fall provably below the threshold, at the
threshold quantiles of C4 (top-50 of 12.5 M) and of the 10 M flat mode.  Result (this container): pairs 0.85 / 0.999, waves 0.00 / 0.28
(C4 quantile); 0.68 / 0.999 and 0.00 / 0.11 (flat-mode quantile) -> a wave-uniform abandon saves < 10 % and was not built.
usage: python scripts/flat_abandon_study.py"""
import sys, math, numpy as np, torch, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from benchlib import Mixture, train_codebooks
torch.set_num_threads(8)
D,M,N=768,96,300_000
dev=torch.device('cpu')
mix=Mixture(D,seed=5,device=dev)
base=mix.sample(N,seed=5)
q=mix.sample(64,seed=6)
cb=train_codebooks(base,M,seed=4,iters=4,sample=50_000)   # [M*k*size] centroid-major
cb=cb.reshape(M,256,8)
# global centroid? (engine centres for cosine? use none)
X=base.reshape(N,M,8)
codes=torch.empty(N,M,dtype=torch.long)
for m in range(M):
    d=torch.cdist(X[:,m,:],cb[m])
    codes[:,m]=d.argmin(1)
lut=torch.einsum('qmj,mcj->qmc',q.reshape(-1,M,8),cb)      # [Q][M][256]
amag=(cb*cb).sum(-1)                                       # [M][256]
nm=amag[torch.arange(M)[None,:],codes].sum(1)              # [N]
qm=(q*q).sum(1)
ent=lut[:, torch.arange(M)[None,:], codes]                 # [Q][N][M]
raw=ent.sum(-1)                                            # [Q][N]
cos=raw/torch.sqrt(nm[None,:]*qm[:,None])
# threshold: same quantile as top-50 of 12.5M = 4e-6 -> top ~1.2 of 300k; use top-2
for K,label in ((2,'quantile ~ C4 (top-50 of 12.5M)'),(15,'quantile ~ C3 flat (top-50 of 1M)')):
    tau=torch.topk(cos,K,dim=1).values[:,-1]               # [Q]
    # bound after slice s (32 subspaces each): partial + rest_max, as cosine: bound on raw then / sqrt(nm*qm) (nm exact per candidate)
    mx=lut.max(-1).values                                  # [Q][M]
    for s in (1,2):
        part=ent[:,:,:32*s].sum(-1)
        rest=mx[:,32*s:].sum(-1)
        bound=(part+rest[:,None])/torch.sqrt(nm[None,:]*qm[:,None])
        rej=bound<tau[:,None]                              # [Q][N]
        # wave = 64 lanes x R=8 candidates = 512 consecutive candidates x P=4 queries
        nw=N//512
        rw=rej[:, :nw*512].reshape(-1,4,nw,512) if False else None
        r=rej[:, :nw*512].reshape(16,4,nw,512)             # 16 query groups of 4
        wave_all=r.all(dim=3).all(dim=1)                   # [16][nw]
        print(label, f'after slice {s}: pairs rejectable {rej.float().mean():.4f}  waves fully rejectable {wave_all.float().mean():.4f}')
