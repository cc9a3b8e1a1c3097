"""per-launch times of the Gram-Schmidt + loss family at C2 shapes (HIP events around every C-ABI call)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "generative-audio_amd"))
import torch
from nppc_audio import _hip as H
from nppc_audio import pc_ops
B, K, F, T = 32, 5, 128, 251
g = torch.Generator().manual_seed(0)
x = torch.randn(B, K, 2, F, T, generator=g).cuda().requires_grad_(True)
gt = torch.randn(B, 2, F, T, generator=g).cuda()
pred = torch.randn(B, 2, F, T, generator=g).cuda()
def step():
    w = pc_ops.gram_schmidt_to_crm(x)
    out = pc_ops.NPPCLoss.apply(w, gt, pred, 0.5)
    out[1].backward()
for _ in range(3):
    step()
torch.cuda.synchronize()
H.PROFILE = []
NP = 5
for _ in range(NP):
    step()
torch.cuda.synchronize()
calls, H.PROFILE = H.PROFILE, None
tot = {}
order = []
for i, (name, e0, e1) in enumerate(calls):
    k = (i % (len(calls) // NP), name)
    if k not in tot:
        order.append(k)
    tot[k] = tot.get(k, 0.0) + e0.elapsed_time(e1) / NP
for k in order:
    print(f"{k[0]:2d} {k[1]:26s} {tot[k]*1e3:8.1f} us")
print("total", sum(tot.values()) * 1e3, "us")
