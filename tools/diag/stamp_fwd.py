"""Phase timers of the cooperative forward kernel (built with -DCF_STAMP into tools/diag/libnppc_stampf.so)."""
import sys, os, subprocess, glob
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(root, "generative-audio_amd"))
so = os.path.join(root, "tools", "diag", "libnppc_stampf.so")
if "--build" in sys.argv:
    srcs = sorted(glob.glob(os.path.join(root, "generative-audio_amd", "csrc", "*.hip")))
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-DCF_STAMP",
                           "-I" + os.path.join(root, "include"), "-I" + os.path.join(root, "generative-audio_amd", "csrc"), "-o", so] + srcs)
    sys.exit(0)
import torch
from nppc_audio import _hip as H
H.LIB_PATH = so
from nppc_audio import ops_lstm
from nppc_audio.ops_lstm import PackedLSTM, lstm2_forward
dev = torch.device("cuda")
I, Hd, Tn = 34, 384, 253
torch.manual_seed(0)
ws = [torch.randn(4*Hd, I)*.05, torch.randn(4*Hd, Hd)*.05, torch.zeros(4*Hd), torch.zeros(4*Hd), torch.randn(4*Hd, Hd)*.05, torch.randn(4*Hd, Hd)*.05, torch.zeros(4*Hd), torch.zeros(4*Hd)]
ws = [w.to(dev) for w in ws]
pk = PackedLSTM(I, Hd, 0, dev).pack(*ws)
orig = ops_lstm.workspace
def ws2(key, shape, dtype, device, zero=False):
    if key[-1] == "coop_flags":
        shape = (shape[0] + 32,)
    return orig(key, shape, dtype, device, zero)
ops_lstm.workspace = ws2
names = ["L1 gemm (i,g)", "L1 pointwise(i,g) + gemm (f,o)", "L1 pointwise(f,o) + prime + h2 partner->LDS + barrier(1)", "h1->LDS, x->LDS, barrier (2a)",
         "publish(0) (+flush)", "L2 (i,g) h2-half gemm (+prime)", "consume(0) + barrier (2c)", "L2 (i,g) h1-half gemm", "-", "L2 pointwise(i,g) + (f,o) gemm",
         "L2 pointwise(f,o) + barrier (3)", "h2->LDS + barrier", "publish(1) + flush + loop top (poll, x prefetch)"]
for N, train, mt in ((8224, False, (2, 5)), (4096, True, (2, 2))):
    x = torch.randn(Tn, N, pk.kx, device=dev).to(torch.bfloat16); x[:, :, I:] = 0
    for _ in range(2):
        lstm2_forward(x, pk, train, mt)
    torch.cuda.synchronize()
    fl = [t for k, t in ops_lstm._WS.items() if k[0][-1] == "coop_flags"][-1]
    G, m = mt
    ncl = (N + 16 * m - 1) // (16 * m)
    dbg = fl[ncl * 2 * G + 4: ncl * 2 * G + 4 + 26].view(torch.int64).cpu().tolist()
    tot = sum(dbg)
    print(f"fwd N={N} train={train} {mt}: cycles/step {tot/Tn:.0f}")
    for nme, v in zip(names, dbg):
        print(f"   {nme:62s} {v/Tn:9.1f}  {100*v/max(tot,1):5.1f}%")
