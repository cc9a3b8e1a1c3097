"""Cycles per phase of the convolution ring kernel (conv_dma_kernel<128, 3>, built with -DCONV_STAMP into tools/diag/libconv_stamp.so):
compute wave 0 of workgroup 0 on one deep C3 layer (B=32, 16 x 62, 512 -> 512, 3 x 3).
  python tools/diag/conv_stamp.py --build       (CPU)          python tools/diag/conv_stamp.py      (GPU box)"""
import glob, os, subprocess, sys
root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
so = os.path.join(root, "tools", "diag", "libconv_stamp.so")
if "--build" in sys.argv:
    csrc = os.path.join(root, "generative-audio_amd", "csrc")
    objs = [os.path.join(root, "generative-audio_amd", "build", os.path.basename(f)[:-4] + ".o") for f in sorted(glob.glob(csrc + "/*.hip")) if not f.endswith("unet.hip")]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(root, "include"), "-I" + csrc,
                           "-Wno-unused-value", "-DCONV_STAMP", "-c", os.path.join(csrc, "unet.hip"), "-o", "/tmp/unet_stamp.o"])
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, "/tmp/unet_stamp.o"] + objs)
    sys.exit(0)
sys.path.insert(0, os.path.join(root, "generative-audio_amd"))
import torch
from nppc_audio import _hip as H
H.LIB_PATH = so
B, Hh, W, Cin, Cout, ks = 32, 16, 62, 512, 512, 3
dt = torch.bfloat16
P = B * (Hh + 2) * (W + 2)
gb = W + 4
g = torch.Generator().manual_seed(0)
X = torch.zeros((gb + P + 4096 + W + 4) * Cin, dtype=dt, device="cuda")
X[gb * Cin:(gb + P) * Cin] = (torch.randn(P * Cin, generator=g) * 0.5).to(dt).cuda()
Y = torch.zeros((gb + P + 4096 + W + 4) * Cout, dtype=dt, device="cuda")
w = torch.randn(Cout, Cin, ks, ks, generator=g) / (Cin * 9) ** 0.5
wf = torch.empty(Cout * 9 * Cin, dtype=dt, device="cuda"); wb = torch.empty(Cin * 9 * Cout, dtype=dt, device="cuda")
s = H.stream()
H.call("nppc_conv_pack", H.PREC_BF16, w.cuda(), wf, wb, Cout, Cin, ks, Cout, Cin, Cin, Cout, s)
ntiles = (P + 127) // 128
part = torch.zeros(ntiles * 2 * Cout + 64, dtype=torch.float32, device="cuda")
for _ in range(200):        # warm: the clock settles under back-to-back launches
    H.call("nppc_conv_fwd_stats", H.PREC_BF16, X[gb * Cin:], Cin, wf, Y[gb * Cout:], Cout, None, B, Hh, W, Cin, Cout, Cout, ks, part, s)
torch.cuda.synchronize()
d = part[ntiles * 2 * Cout:].view(torch.int64)[:16].cpu().tolist()
names = ["half 1 (reads set 1 between 16 MFMAs)", "LDS wait", "barrier", "-", "half 2 (reads set 0 between 16 MFMAs + wait)"]
for who, off in (("wave 0 (compute)", 0),):
    n = max(d[off + 5], 1)
    tot = sum(d[off:off + 5])
    print(f"{who}: {tot / n:.0f} cycles per stage over {n} stages")
    for nm, v in zip(names, d[off:off + 5]):
        print(f"    {nm:44s} {v / n:8.1f}  {100 * v / max(tot, 1):5.1f} %")
