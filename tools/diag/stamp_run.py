import sys, os, shutil, time
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(root, "generative-audio_amd"))
import torch
from nppc_audio import _hip as H
H.LIB_PATH = os.path.join(root, "tools", "diag", "libnppc_stamp.so")
from nppc_audio import ops_lstm
from nppc_audio.ops_lstm import PackedLSTM, lstm2_forward
dev = torch.device("cuda")
I, Hd, Tn = 34, 384, 253
torch.manual_seed(0)
ws = [torch.randn(4*Hd, I)*.05, torch.randn(4*Hd, Hd)*.05, torch.zeros(4*Hd), torch.zeros(4*Hd), torch.randn(4*Hd, Hd)*.05, torch.randn(4*Hd, Hd)*.05, torch.zeros(4*Hd), torch.zeros(4*Hd)]
pk = PackedLSTM(I, Hd, 0, dev).pack(*[w.to(dev) for w in ws])
for N, train, force in ((4096, False, (2, 2)), (4096, True, (2, 2)), (8224, False, (2, 5))):
    x = torch.randn(Tn, N, pk.kx, device=dev).to(torch.bfloat16); x[:, :, I:] = 0
    # oversize the exchange workspace by 64 B for the stamp words: monkeypatch workspace for coop_xch
    orig = ops_lstm.workspace
    def ws2(key, shape, dtype, device, zero=False):
        if key[-1] == "coop_xch":
            shape = (shape[0] + 32,)
        return orig(key, shape, dtype, device, zero)
    ops_lstm.workspace = ws2
    for _ in range(3):
        out = lstm2_forward(x, pk, train, force)
    torch.cuda.synchronize()
    xch = [t for k, t in ops_lstm._WS.items() if k[0][-1] == "coop_xch" and k[0][2] == train][-1]
    dbg = xch.view(torch.uint8)[-64:].view(torch.int64).cpu().tolist()
    names = ["L1 gemm+cell", "consume h2+bar+LDS", "publish h1", "L2 pairA h2-half", "wait h1", "L2 rest+cell", "h2 write+publish", "loop top (x prefetch issue)"]
    tot = sum(dbg)
    print(f"N={N} train={train} force={force}: total cycles/step {tot/Tn:.0f} (100 MHz ticks? see ratio)")
    for nme, v in zip(names, dbg):
        print(f"   {nme:28s} {v/Tn:9.1f} ticks/step  {100*v/tot:5.1f}%")
    ops_lstm.workspace = orig
