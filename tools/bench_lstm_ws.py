"""Weight-stationary cluster forward vs the streaming CU-pair kernels at BASELINE C2 shapes (GPU box only).
Alternating rounds in one process (cdna guide rule 24); prints median / min per variant."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "generative-audio_amd"))
import torch
from nppc_audio import _hip as H
from nppc_audio import ops_lstm
from nppc_audio.ops_lstm import PackedLSTM, lstm2_forward

dev = torch.device("cuda")
I, Hd = 34, 384
Tn = int(os.environ.get("TN", "253"))
torch.manual_seed(0)
ws = [torch.randn(4 * Hd, I) * .05, torch.randn(4 * Hd, Hd) * .05, torch.zeros(4 * Hd), torch.zeros(4 * Hd),
      torch.randn(4 * Hd, Hd) * .05, torch.randn(4 * Hd, Hd) * .05, torch.zeros(4 * Hd), torch.zeros(4 * Hd)]
ws = [w.to(dev) for w in ws]
pk = PackedLSTM(I, Hd, 0, dev).pack(*ws)
rounds = int(os.environ.get("ROUNDS", "5"))
for N, train, O, variants in ((8224, False, 2, ("ws", (2, 5))), (4096, True, 10, ("ws", (2, 2))), (4096, False, 10, ("ws", (2, 2)))):
    x = torch.randn(Tn, N, pk.kx, device=dev).to(torch.bfloat16)
    x[:, :, I:] = 0
    wh = torch.zeros(16, Hd, dtype=torch.bfloat16, device=dev)
    wh[:O] = (torch.randn(O, Hd) * 0.1).to(dev)
    times = {v: [] for v in variants}
    for v in variants:
        lstm2_forward(x, pk, train, v, head=(wh, O))
    torch.cuda.synchronize()
    for r in range(rounds):
        for v in variants:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = lstm2_forward(x, pk, train, v, head=(wh, O))
            e1.record()
            torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1))
    flops = N * Tn * 2 * (1536 * (34 + 384 + 768) + 384 * O)
    for v in variants:
        ts = sorted(times[v])
        med, mn = ts[len(ts) // 2], ts[0]
        print(f"N={N} train={train} {str(v):8s}: median {med:.3f} ms min {mn:.3f} ms  {flops/med/1e9:.0f} TFLOP/s = {flops/med/1e9/2500:.3f} of peak; "
              f"{med*1e3/Tn:.1f} us/step  timeouts={ops_lstm.coop_timeouts()}", flush=True)
