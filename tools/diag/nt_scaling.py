"""How the NT GEMM's time scales with the row count at the TCN shapes (bf16, Z = 3 branches): latency- or throughput-bound?"""
import os, sys
root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(root, "generative-audio_amd"))
import torch
from nppc_audio import _hip as H
Z, Tp, Tv = 3, 256, 251
dt = torch.bfloat16
g = torch.Generator().manual_seed(0)
def mk(*shape): return (torch.randn(*shape, generator=g) * 0.1).to(dt).cuda()
def timed(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, epi, N, K in (("conv1x1 fwd (PReLU+stats) N=512 K=576", 1, 512, 576), ("sconv fwd (residual) N=576 K=512", 2, 576, 512), ("dA2 bwd (plain) N=512 K=576", 0, 512, 576)):
    for R in (1024, 2048, 4096, 8192, 16384, 32768):
        A, W, res = mk(Z, R, K), mk(Z, N, K), mk(Z, R, N)
        out = torch.empty(Z, R, N, dtype=dt, device="cuda")
        slope = torch.full((Z,), 0.25, device="cuda"); stats = torch.zeros(Z, R // Tp, 2, dtype=torch.float64, device="cuda")
        fn = lambda: H.call("nppc_gemm_nt", 0, epi, A, K, R * K, W, K, N * K, out, N, R * N, None, 0, res if epi == 2 else None, N, R * N,
                            slope if epi == 1 else None, 1, stats if epi == 1 else None, (R // Tp) * 2, R, N, K, Tp, Tv, N, 0, Z, 1, H.stream())
        us = timed(fn)
        wgs = Z * (R // 128) * (N // 128 if N % 128 == 0 else N // 64)
        print(f"{name}  R={R:6d}  workgroups {wgs:5d}  {us:7.1f} us  {2.0 * Z * R * N * K / us / 1e6:7.1f} TFLOP/s", flush=True)
