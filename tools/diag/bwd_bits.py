"""bit-level reproducibility of the cooperative LSTM backward: two launches on the same inputs, and a checksum to compare builds
usage: python tools/diag/bwd_bits.py [lib.so]"""
import os, sys
root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(root, "generative-audio_amd"))
import torch
from nppc_audio import _hip as H
if len(sys.argv) > 1:
    H.LIB_PATH = sys.argv[1]
from nppc_audio import ops_lstm
from nppc_audio.ops_lstm import PackedLSTM, PackedLSTMBwd, lstm2_forward, lstm2_backward
dev = torch.device("cuda")
I, Hd, Tn, N, O = 34, 384, 40, 4096, 10
torch.manual_seed(0)
ws = [torch.randn(4 * Hd, I) * .05, torch.randn(4 * Hd, Hd) * .05, torch.zeros(4 * Hd), torch.zeros(4 * Hd),
      torch.randn(4 * Hd, Hd) * .05, torch.randn(4 * Hd, Hd) * .05, torch.zeros(4 * Hd), torch.zeros(4 * Hd)]
ws = [w.to(dev) for w in ws]
pk = PackedLSTM(I, Hd, 0, dev).pack(*ws)
pb = PackedLSTMBwd(I, Hd, 0, dev).pack(ws[0], ws[1], ws[4], ws[5])
x = torch.randn(Tn, N, pk.kx, device=dev).to(torch.bfloat16); x[:, :, I:] = 0
saved = lstm2_forward(x, pk, True, None)
dyt = (torch.randn(Tn, N, 16, device=dev) * .01).to(torch.bfloat16); dyt[:, :, O:] = 0
whT = torch.zeros(Hd, 32, dtype=torch.bfloat16, device=dev); whT[:, :O] = (torch.randn(Hd, O, device=dev) * .1).to(torch.bfloat16)
outs = []
for rep in range(6):
    o = [t.clone() for t in lstm2_backward(saved, None, pb, pk.kx, head=(dyt, whT))]
    torch.cuda.synchronize()
    outs.append(o)
same = all(torch.equal(a, b) for o in outs[1:] for a, b in zip(outs[0], o))
chk = [int(t.view(torch.int16).to(torch.int64).sum()) for t in outs[0]]
print("repeatable:", same, "checksums:", chk, "timeouts:", ops_lstm.coop_timeouts())
if not same:
    for r, o in enumerate(outs[1:], 1):
        for name, a, b in zip(("dx", "dg1", "dg2"), outs[0], o):
            d = (a.float() - b.float()).abs()
            print(r, name, "n_diff", int((d > 0).sum()), "max", float(d.max()))
