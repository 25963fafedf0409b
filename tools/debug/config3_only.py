"""configs[2] only (5000 x 25000 x 768 matching-head retrieval): a few calls of the similarity GEMM (store mode) and of the fused
similarity + rank pass -- the workload tools/collect_eval_pmc.sh profiles"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from aladin_amd import ops, synth
dev = torch.device('cuda:0')
img, cap = synth.retrieval_embeddings(5000, 768, seed=303, sigma=12.0)
a = torch.from_numpy(img[0::5]).to(dev); b = torch.from_numpy(cap).to(dev)
for _ in range(4):
    sim = ops.sim_matrix(a, b)
    ops.recall_ranks(sim)
    ops.retrieval_ranks(a, b)
torch.cuda.synchronize()
print('ok')
