"""Stand-alone timing of the NT GEMM at the TCN shapes (C2: 3 branches x 8192 rows): conv1x1 forward (PReLU + stats),
sconv forward (residual), the two data-gradient products of the backward.  NPPC_NT_STAGED=0: residual / mask epilogues in the
accumulator layout instead of the coalesced epilogue through LDS (A/B)."""
import os, sys
root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(root, "generative-audio_amd"))
import torch
from nppc_audio import _hip as H
Z, R, Tp, Tv = 3, 8192, 256, 251
dt = torch.bfloat16
g = torch.Generator().manual_seed(0)
def mk(*shape): return (torch.randn(*shape, generator=g) * 0.1).to(dt).cuda()
def timed(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, epi, N, K in (("conv1x1 fwd  (PReLU+stats) N=512 K=576", 1, 512, 576), ("sconv fwd    (residual)    N=576 K=512", 2, 576, 512),
                        ("dA2 bwd      (plain)       N=512 K=576", 0, 512, 576), ("dXi bwd      (residual)    N=576 K=512", 2, 576, 512)):
    A, W, res = mk(Z, R, K), mk(Z, N, K), mk(Z, R, N)
    out = torch.empty(Z, R, N, dtype=dt, device="cuda")
    slope = torch.full((Z,), 0.25, device="cuda"); stats = torch.zeros(Z, R // Tp, 2, dtype=torch.float64, device="cuda")
    fn = lambda: H.call("nppc_gemm_nt", 0, epi, A, K, R * K, W, K, N * K, out, N, R * N, None, 0, res if epi == 2 else None, N, R * N,
                        slope if epi == 1 else None, 1, stats if epi == 1 else None, (R // Tp) * 2, R, N, K, Tp, Tv, N, 0, Z, 1, H.stream())
    us = timed(fn)
    fl = 2.0 * Z * R * N * K
    print(f"{name}: {us:7.1f} us  {fl / us / 1e6:7.1f} TFLOP/s  operands+output {(A.numel() + W.numel() + out.numel() * (2 if epi == 2 else 1)) * 2 / 1e6:.0f} MB", flush=True)
