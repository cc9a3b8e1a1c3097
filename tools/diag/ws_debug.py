"""debug: weight-stationary forward vs single-workgroup kernel, per tensor / time step / unit block"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "generative-audio_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from test_lstm_gpu import _weights
from nppc_audio import _hip as H
from nppc_audio import ops_lstm
from nppc_audio.ops_lstm import PackedLSTM, lstm2_forward

N, Tn, I, Hd = int(sys.argv[1]) if len(sys.argv) > 1 else 128, int(sys.argv[2]) if len(sys.argv) > 2 else 4, 34, 384
P = _weights(I, Hd, 7)
pre = "sb_model.sequence_model."
dev = torch.device("cuda")
pk = PackedLSTM(I, Hd, 0, dev).pack(*[P[pre + n].to(dev) for n in (
    "weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0", "weight_ih_l1", "weight_hh_l1", "bias_ih_l1", "bias_hh_l1")])
g = torch.Generator().manual_seed(1)
x = torch.randn(N, Tn, I, generator=g)
xt = torch.zeros(Tn, N, pk.kx, dtype=torch.bfloat16, device=dev)
xt[:, :, :I] = x.permute(1, 0, 2).to(dev)
single = {k: v.clone() for k, v in lstm2_forward(xt, pk, True, 1).items()}
ws = lstm2_forward(xt, pk, True, "ws")
torch.cuda.synchronize()
print("timeouts", ops_lstm.coop_timeouts())
for k in ("g1", "c1", "h1", "g2", "c2", "h2"):
    a, b = ws[k].float(), single[k].float()
    for t in range(Tn):
        d = (a[t] - b[t]).abs()
        print(k, "t", t, "max err %.4f" % d.max().item(), "| per 32-unit block max:",
              " ".join("%.3f" % d.reshape(N, 12, -1).amax(dim=(0, 2))[j].item() for j in range(12)),
              "| per 32-row chunk:", " ".join("%.3f" % d.reshape(N // 32, 32, -1).amax(dim=(1, 2))[j].item() for j in range(min(N // 32, 8))))
    if k == "g1":
        d = (a[0] - b[0]).abs()          # [N][H][4]
        print("   g1 t0 per gate:", d.amax(dim=(0, 1)).tolist(), " per unit%8:", d.reshape(N, 48, 8, 4).amax(dim=(0, 1, 3)).tolist())
        print("   g1 t0 per seq%32 (first 8):", d.reshape(N // 32, 32, -1).amax(dim=(0, 2))[:8].tolist())
# hypothesis checks on the layer-1 cell state
g1 = ws["g1"].float(); c1 = ws["c1"].float()
i, gg, f, o = g1[..., 0], g1[..., 1], g1[..., 2], g1[..., 3]
for t in range(1, Tn):
    with_prev = f[t] * c1[t - 1] + i[t] * gg[t]
    with_zero = i[t] * gg[t]
    print("c1 t", t, "err vs f*c_prev+i*g: %.4f" % (c1[t] - with_prev).abs().max().item(), " vs i*g (c_prev = 0): %.4f" % (c1[t] - with_zero).abs().max().item())
    # c_prev implied
    cimp = (c1[t] - i[t] * gg[t]) / f[t].clamp_min(1e-3)
    print("    implied c_prev vs c1[t-1]: %.4f, vs c1[t-1] of other chunk offsets:" % (cimp - c1[t - 1]).abs().max().item(),
          [round((cimp - torch.roll(c1[t - 1], 32 * k, 0)).abs().max().item(), 3) for k in range(1, 4)])
