"""One deep convolution layer of C3 (B=32, 16 x 62 pixels, 512 -> 512 channels, 3 x 3, bf16) in a loop, for counter passes.
  rocprofv3 --pmc <counters> ... -- python3 tools/diag/conv_pmc.py        (NPPC_CONV_DMA=0|1|2 picks the kernel)
  python tools/diag/conv_pmc.py --summarize <counter_collection.csv>"""
import os, sys
root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 1 and sys.argv[1] == "--summarize":
    import csv, collections
    for path in sys.argv[2:]:
        acc = collections.defaultdict(float); n = collections.Counter()
        for r in csv.DictReader(open(path)):
            if "conv_tiled_kernel" not in r["Kernel_Name"] and "conv_dma_kernel" not in r["Kernel_Name"]:
                continue
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
        for c, v in sorted(acc.items()):
            print(f"    {c:36s} {v / n[c]:16.0f}   (mean of {n[c]} dispatches)")
    sys.exit(0)
sys.path.insert(0, os.path.join(root, "generative-audio_amd"))
import torch
from nppc_audio import _hip as H
B, Hh, W, Cin, Cout, ks = 32, 16, 62, 512, 512, 3
dt = torch.bfloat16
P = B * (Hh + 2) * (W + 2)
gb = W + 4
g = torch.Generator().manual_seed(0)
X = torch.zeros((gb + P + 4096 + W + 4) * Cin, dtype=dt, device="cuda")
X[gb * Cin:(gb + P) * Cin] = (torch.randn(P * Cin, generator=g) * 0.5).to(dt).cuda()
Y = torch.zeros((gb + P + 4096 + W + 4) * Cout, dtype=dt, device="cuda")
w = torch.randn(Cout, Cin, ks, ks, generator=g) / (Cin * 9) ** 0.5
wf = torch.empty(Cout * 9 * Cin, dtype=dt, device="cuda")
wb = torch.empty(Cin * 9 * Cout, dtype=dt, device="cuda")
s = H.stream()
H.call("nppc_conv_pack", H.PREC_BF16, w.cuda(), wf, wb, Cout, Cin, ks, Cout, Cin, Cin, Cout, s)
xt, yt = X[gb * Cin:], Y[gb * Cout:]
n = int(os.environ.get("CONV_PMC_ITERS", "8"))
for _ in range(n):
    H.call("nppc_conv_fwd", H.PREC_BF16, xt, Cin, wf, yt, Cout, None, None, None, 0.2, B, Hh, W, Cin, Cout, Cout, ks, s)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    H.call("nppc_conv_fwd", H.PREC_BF16, xt, Cin, wf, yt, Cout, None, None, None, 0.2, B, Hh, W, Cin, Cout, Cout, ks, s)
e1.record(); torch.cuda.synchronize()
print(f"NPPC_CONV_DMA={os.environ.get('NPPC_CONV_DMA', '0')}: {e0.elapsed_time(e1) / n * 1e3:.1f} us per launch, "
      f"{2.0 * B * Hh * W * Cin * Cout * 9 / (e0.elapsed_time(e1) / n * 1e-3) / 1e12:.0f} TFLOP/s", flush=True)
