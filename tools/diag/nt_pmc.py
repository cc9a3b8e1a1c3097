"""One NT GEMM shape in a loop, for counter passes: conv1x1 forward of the TCN (Z=3, R=8192, N=512, K=576, PReLU + stats)."""
import os, sys
root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 1 and sys.argv[1] == "--summarize":
    import csv, collections
    for path in sys.argv[2:]:
        acc = collections.defaultdict(float); n = collections.Counter()
        for r in csv.DictReader(open(path)):
            if "gemm_nt_lds" not in r["Kernel_Name"]:
                continue
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
        for c, v in sorted(acc.items()):
            print(f"    {c:36s} {v / n[c]:16.0f}   (mean of {n[c]} dispatches)")
    sys.exit(0)
sys.path.insert(0, os.path.join(root, "generative-audio_amd"))
import torch
from nppc_audio import _hip as H
Z, R, Tp, Tv, N, K = 3, 8192, 256, 251, 512, 576
dt = torch.bfloat16
g = torch.Generator().manual_seed(0)
def mk(*shape): return (torch.randn(*shape, generator=g) * 0.1).to(dt).cuda()
A, W = mk(Z, R, K), mk(Z, N, K)
out = torch.empty(Z, R, N, dtype=dt, device="cuda")
slope = torch.full((Z,), 0.25, device="cuda"); stats = torch.zeros(Z, R // Tp, 2, dtype=torch.float64, device="cuda")
for _ in range(6):
    H.call("nppc_gemm_nt", 0, 1, A, K, R * K, W, K, N * K, out, N, R * N, None, 0, None, N, R * N, slope, 1, stats, (R // Tp) * 2, R, N, K, Tp, Tv, N, 0, Z, 1, H.stream())
torch.cuda.synchronize()
