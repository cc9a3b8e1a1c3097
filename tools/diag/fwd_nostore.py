"""Diagnostic: forward-train LSTM kernel with the saved-state stores compiled out (-DCF_NO_SAVE): what do they cost?"""
import sys, os, subprocess, glob, time
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(root, "generative-audio_amd"))
so = os.path.join(root, "tools", "diag", "libnppc_nosave.so")
if "--build" in sys.argv:
    srcs = sorted(glob.glob(os.path.join(root, "generative-audio_amd", "csrc", "*.hip")))
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-DCF_NO_SAVE",
                           "-I" + os.path.join(root, "include"), "-I" + os.path.join(root, "generative-audio_amd", "csrc"), "-o", so] + srcs)
    sys.exit(0)
import torch
from nppc_audio import _hip as H
if "--nosave" in sys.argv:
    H.LIB_PATH = so
from nppc_audio.ops_lstm import PackedLSTM, lstm2_forward
dev = torch.device("cuda")
I, Hd, Tn, N = 34, 384, 253, 4096
torch.manual_seed(0)
ws = [torch.randn(4*Hd, I)*.05, torch.randn(4*Hd, Hd)*.05, torch.zeros(4*Hd), torch.zeros(4*Hd), torch.randn(4*Hd, Hd)*.05, torch.randn(4*Hd, Hd)*.05, torch.zeros(4*Hd), torch.zeros(4*Hd)]
ws = [w.to(dev) for w in ws]
pk = PackedLSTM(I, Hd, 0, dev).pack(*ws)
x = torch.randn(Tn, N, pk.kx, device=dev).to(torch.bfloat16); x[:, :, I:] = 0
for _ in range(2):
    lstm2_forward(x, pk, True, None)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    lstm2_forward(x, pk, True, None)
torch.cuda.synchronize()
print(f"fwd train N={N} {'NO saved-state stores' if '--nosave' in sys.argv else 'normal'}: {(time.perf_counter()-t0)/5*1e3:.2f} ms")
