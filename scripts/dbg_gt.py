import sys, torch, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import jvector_amd as J
from bench import Mixture, ground_truth, train_codebooks, recall_at_k
dev=torch.device('cuda',0)
ctx=J.HipContext(0, stream=torch.cuda.current_stream().cuda_stream)
N,D=2_000_000,768
mix=Mixture(D,5,dev)
base=mix.sample(N,5)
q=mix.sample(300,6)
vs=J.VectorSet(ctx,base)
VSF=J.VectorSimilarityFunction.COSINE
gt=ground_truth(J,ctx,vs,q,VSF,10).cpu().numpy()
sims=q@base.t()
ref=sims.topk(10,dim=1).indices.cpu().numpy()
print('gt[0]',gt[0]); print('ref[0]',ref[0])
print('overlap', recall_at_k(gt,ref))
# single-chunk check
out=vs.scan(q,VSF,first=0,count=N)
ids,sc=J.topk(ctx,out,10); ctx.sync()
print('single', recall_at_k(ids.cpu().numpy(),ref), ids[0], sc[0], sims.topk(10,dim=1).values[0])
