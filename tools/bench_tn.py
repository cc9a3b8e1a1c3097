#!/usr/bin/env python3
"""Microbenchmark of the TN split-K weight-gradient GEMM (csrc/tcn.hip) on the LSTM weight-gradient shapes."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "generative-audio_amd"))
from nppc_audio import _hip as H

def run(M, N, R, S, reps=5):
    A = torch.randn(R + 4096, M, device="cuda").to(torch.bfloat16)
    B = torch.randn(R + 4096, N, device="cuda").to(torch.bfloat16)
    C = torch.empty(S * M * N, dtype=torch.float32, device="cuda")
    s = H.stream()
    for _ in range(2):
        H.call("nppc_gemm_tn_splitk", A, M, B, N, C, N, M, N, R, S, s)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        H.call("nppc_gemm_tn_splitk", A, M, B, N, C, N, M, N, R, S, s)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"M={M} N={N} R={R} S={S}: {ms:.3f} ms  {2.0*M*N*R/ms/1e9:.0f} TFLOP/s  operands {(M+N)*R*2/1e9:.2f} GB -> {(M+N)*R*2/ms/1e6:.0f} GB/s min", flush=True)

for S in (32, 64):
    run(1536, 384, 253 * 4096, S)
run(1536, 384, 16 * 4096, 16)
run(1536, 384, 16 * 4096, 64)
run(1536, 64, 253 * 4096, 16)
run(1536, 128, 253 * 4096, 16)
run(512, 512, 253 * 4096, 64)
