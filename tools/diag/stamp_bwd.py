import sys, os
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(root, "generative-audio_amd"))
import torch
from nppc_audio import _hip as H
H.LIB_PATH = os.path.join(root, "tools", "diag", "libnppc_stamp.so")
from nppc_audio import ops_lstm
from nppc_audio.ops_lstm import PackedLSTM, PackedLSTMBwd, lstm2_forward, lstm2_backward
dev = torch.device("cuda")
I, Hd, Tn, N = 34, 384, 253, 4096
torch.manual_seed(0)
ws = [torch.randn(4*Hd, I)*.05, torch.randn(4*Hd, Hd)*.05, torch.zeros(4*Hd), torch.zeros(4*Hd), torch.randn(4*Hd, Hd)*.05, torch.randn(4*Hd, Hd)*.05, torch.zeros(4*Hd), torch.zeros(4*Hd)]
ws = [w.to(dev) for w in ws]
pk = PackedLSTM(I, Hd, 0, dev).pack(*ws)
pb = PackedLSTMBwd(I, Hd, 0, dev).pack(ws[0], ws[1], ws[4], ws[5])
x = torch.randn(Tn, N, pk.kx, device=dev).to(torch.bfloat16); x[:, :, I:] = 0
dh2 = torch.randn(Tn, N, Hd, device=dev).to(torch.bfloat16)
orig = ops_lstm.workspace
def ws2(key, shape, dtype, device, zero=False):
    if key[-1] == "coop_xch":
        shape = (shape[0] + 64,)
    return orig(key, shape, dtype, device, zero)
ops_lstm.workspace = ws2
saved = lstm2_forward(x, pk, True, 1)
for _ in range(3):
    lstm2_backward(saved, dh2, pb, pk.kx, coop=True)
torch.cuda.synchronize()
xch = [t for k, t in ops_lstm._WS.items() if k[0][-1] == "coop_xch" and k[0][0] == "lstm_bwd"][-1]
dbg = xch.view(torch.uint8)[-128:].view(torch.int64).cpu().tolist()[:12]
names = ["P2 cell bwd", "barrier after P2", "publish L2 (+dgT store)", "G2 own half", "consume L2 + barrier", "G2 partner half", "G2 epilogue + barrier", "P1 + barrier", "publish L1 (+dgT)", "G1 all + consume", "final barrier", "loop top"]
tot = sum(dbg)
print(f"bwd coop: total cycles/step {tot/Tn:.0f}")
for nme, v in zip(names, dbg):
    print(f"   {nme:28s} {v/Tn:9.1f} cyc/step  {100*v/tot:5.1f}%")
