"""Micro-benchmark of the fused LSTM kernels at BASELINE C2 shapes (GPU box only)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "generative-audio_amd"))
import torch
from nppc_audio import _hip as H
from nppc_audio import ops_lstm
from nppc_audio.ops_lstm import PackedLSTM, lstm2_forward

dev = torch.device("cuda")
I, Hd, Tn = 34, 384, 253
torch.manual_seed(0)
ws = [torch.randn(4 * Hd, I) * .05, torch.randn(4 * Hd, Hd) * .05, torch.zeros(4 * Hd), torch.zeros(4 * Hd),
      torch.randn(4 * Hd, Hd) * .05, torch.randn(4 * Hd, Hd) * .05, torch.zeros(4 * Hd), torch.zeros(4 * Hd)]
ws = [w.to(dev) for w in ws]
precs = ((0, "bf16"),) if "--bf16" in sys.argv else ((0, "bf16"), (1, "f32"))
for prec, name in precs:
    pk = PackedLSTM(I, Hd, prec, dev).pack(*ws)
    for N, train, mts in ((8224, False, ((2, 5), 3)), (4096, True, ((4, 4), (2, 2), 1)), (4096, False, ((4, 4), (2, 2), 1))):
        if prec == 1:
            mts = (1,)
            if N == 8224:
                continue
        x = torch.randn(Tn, N, pk.kx, device=dev).to(H.dtype_of(prec))
        x[:, :, I:] = 0
        for mt in mts:
            for _ in range(2):
                out = lstm2_forward(x, pk, train, mt)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps = 3
            for _ in range(reps):
                out = lstm2_forward(x, pk, train, mt)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            flops = N * Tn * 2 * 1536 * (34 + 384 + 768)
            print(f"{name} N={N} train={train} mtile={mt}: {dt*1e3:.2f} ms  {flops/dt/1e12:.1f} TFLOP/s "
                  f"finite={bool(torch.isfinite(out['h2'].float()).all())} timeouts={ops_lstm.coop_timeouts()}", flush=True)

if "--bwd" in sys.argv:
    from nppc_audio.ops_lstm import PackedLSTMBwd, lstm2_backward
    pk = PackedLSTM(I, Hd, 0, dev).pack(*ws)
    pb = PackedLSTMBwd(I, Hd, 0, dev).pack(ws[0], ws[1], ws[4], ws[5])
    N = 4096
    x = torch.randn(Tn, N, pk.kx, device=dev).to(torch.bfloat16)
    x[:, :, I:] = 0
    saved = lstm2_forward(x, pk, True, None)
    dh2 = (torch.randn(Tn, N, Hd, device=dev) * 0.01).to(torch.bfloat16)
    for _ in range(2):
        out = lstm2_backward(saved, dh2, pb, pk.kx)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        out = lstm2_backward(saved, dh2, pb, pk.kx)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    flops = N * Tn * 2 * 1536 * (34 + 384 + 768)
    print(f"bf16 coop bwd N={N}: {dt*1e3:.2f} ms  {flops/dt/1e12:.1f} TFLOP/s finite={bool(torch.isfinite(out[0].float()).all())} "
          f"dx_sum={float(out[0].float().abs().sum()):.6e} timeouts={ops_lstm.coop_timeouts()}", flush=True)
