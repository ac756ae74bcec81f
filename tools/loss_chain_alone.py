#!/usr/bin/env python
"""The step's batched contrastive losses ALONE (no neighbour on the chip): `rocprofv3 --kernel-trace --stats -- python
tools/loss_chain_alone.py N F` -> per-kernel times of the chain (forward + backward), three problems of [N, F], 64 labels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from instascene_amd.contrastive import contrastive_loss_batch

N, F = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (8192, 64)
g = torch.Generator(device="cuda").manual_seed(0)
for it in range(30):
    feats = [torch.randn(N, F, device="cuda", generator=g).requires_grad_(True) for _ in range(3)]
    labs = [torch.randint(0, 65, (N,), device="cuda", generator=g) for _ in range(3)]
    tot, parts = contrastive_loss_batch(feats, labs, [None, None, None], [1.0, 0.5, 2.0], 65)
    tot.backward()
torch.cuda.synchronize()
print("ok", float(tot))
