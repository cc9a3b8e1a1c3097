"""Phase timers of the K-split cooperative backward (built with -DC2_STAMP into tools/diag/libnppc_stamp2.so)."""
import sys, os, subprocess, glob
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(root, "generative-audio_amd"))
STID = next((a.split("=")[1] for a in sys.argv if a.startswith("--tid=")), "0")
so = os.path.join(root, "tools", "diag", f"libnppc_stamp2_t{STID}.so")
if "--build" in sys.argv:
    srcs = sorted(glob.glob(os.path.join(root, "generative-audio_amd", "csrc", "*.hip")))
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-DC2_STAMP", "-DC2_STAMP_TID=" + STID] + (["-DC2_NO_DG"] if "--nodg" in sys.argv else []) + [
           "-I" + os.path.join(root, "include"), "-I" + os.path.join(root, "generative-audio_amd", "csrc"), "-o", so] + srcs
    subprocess.check_call(cmd)
    sys.exit(0)
import torch
from nppc_audio import _hip as H
H.LIB_PATH = so
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
dh2 = (torch.randn(Tn, N, Hd, device=dev) * .01).to(torch.bfloat16)
orig = ops_lstm.workspace
def ws2(key, shape, dtype, device, zero=False):
    if key[-1] == "coop_flags":
        shape = (shape[0] + 32,)
    return orig(key, shape, dtype, device, zero)
ops_lstm.workspace = ws2
saved = lstm2_forward(x, pk, True, None)
for _ in range(3):
    lstm2_backward(saved, dh2, pb, pk.kx, coop=True)
torch.cuda.synchronize()
fl = [t for k, t in ops_lstm._WS.items() if k[0][-1] == "coop_flags" and k[0][0] == "lstm_bwd"][-1]
ncl = (N + 31) // 32
dbg = fl[ncl * 48 + 4: ncl * 48 + 4 + 24].view(torch.int64).cpu().tolist()
names = ["P2 cell bwd", "barrier", "fetch + dg store issue", "L2 gemm pass 0 (partner tiles)", "partials: pack + store issue", "pass-1 prologue + store drain + barrier + flag",
         "L2 gemm pass 1 (own tiles; partner epoch check + partial loads inside)", "add partials + scatter own", "barrier", "P1 cell bwd", "barrier + fetch + dg store", "layer-1 gemm/exchange (all)"]
tot = sum(dbg)
print(f"bwd K-split, thread {STID} of workgroup 0: total s_memtime ticks/step {tot/Tn:.0f}")
for nme, v in zip(names, dbg):
    print(f"   {nme:44s} {v/Tn:9.1f} ticks/step  {100*v/max(tot,1):5.1f}%")
