"""Inference throughput of NPPCModel.forward (restorer + direction net, no grad) at the C2 shape with the CU-pair LSTM kernels vs
the weight-stationary cluster kernel.  usage (GPU box): NPPC_LSTM_WS=0|1 python tools/diag/infer_ws.py"""
import os, sys, time
root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, root)
import torch
import bench
torch.cuda.set_device(0)
tr, (noisy, clean) = bench.build_trainer("bf16", 0, 1, 32, 64000)
model = tr.nppc_model
model.eval()
with torch.no_grad():
    for _ in range(3):
        w = model(noisy)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        w = model(noisy)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
print(f"NPPC_LSTM_WS={os.environ.get('NPPC_LSTM_WS', '0')}: {1e3 * dt:.3f} ms per forward (B=32 x 4 s, K=5), {32 * 251 / dt / 1e3:.1f} k frames/s, finite={bool(torch.isfinite(w).all())}")
