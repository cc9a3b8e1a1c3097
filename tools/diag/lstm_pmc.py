"""The three cooperative LSTM launches of the C2 step, alone, for counter passes (SQ issue / wait / LDS / MFMA counters):
restorer forward (N = 8224, fused head, 80-row pairs), training forward (N = 4096, 32-row pairs), K-split backward.
  rocprofv3 --pmc <counters> --output-format csv -d out -o r -- python3 tools/diag/lstm_pmc.py
  python tools/diag/lstm_pmc.py --summarize <counter_collection.csv> ...     (per kernel: every counter summed over the dispatch)"""
import os, sys
root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 1 and sys.argv[1] == "--summarize":
    import csv, collections
    for path in sys.argv[2:]:
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        n = collections.Counter()
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"]
            if "lstm2_coop" not in k:
                continue
            k = k.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0][:60]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[(k, r["Counter_Name"])] += 1
        for k, d in acc.items():
            print(k)
            for c, v in sorted(d.items()):
                print(f"    {c:36s} {v / n[(k, c)]:16.0f}   (mean of {n[(k, c)]} dispatches)")
    sys.exit(0)
sys.path.insert(0, os.path.join(root, "generative-audio_amd"))
import torch
from nppc_audio import ops_lstm
from nppc_audio.ops_lstm import PackedLSTM, PackedLSTMBwd, lstm2_forward, lstm2_backward
dev = torch.device("cuda")
I, Hd, Tn, O = 34, 384, 253, 10
torch.manual_seed(0)
ws = [torch.randn(4 * Hd, I) * .05, torch.randn(4 * Hd, Hd) * .05, torch.zeros(4 * Hd), torch.zeros(4 * Hd),
      torch.randn(4 * Hd, Hd) * .05, torch.randn(4 * Hd, Hd) * .05, torch.zeros(4 * Hd), torch.zeros(4 * Hd)]
ws = [w.to(dev) for w in ws]
pk = PackedLSTM(I, Hd, 0, dev).pack(*ws)
pb = PackedLSTMBwd(I, Hd, 0, dev).pack(ws[0], ws[1], ws[4], ws[5])
whp = torch.zeros(16, Hd, dtype=torch.bfloat16, device=dev); whp[:O] = (torch.randn(O, Hd, device=dev) * .1).to(torch.bfloat16)
xr = torch.randn(Tn, 8224, pk.kx, device=dev).to(torch.bfloat16); xr[:, :, I:] = 0
for _ in range(2):
    lstm2_forward(xr, pk, False, None, head=(whp, 2))
del xr
N = 4096
x = torch.randn(Tn, N, pk.kx, device=dev).to(torch.bfloat16); x[:, :, I:] = 0
for _ in range(2):
    saved = lstm2_forward(x, pk, True, None, head=(whp, O))
dyt = (torch.randn(Tn, N, 16, device=dev) * .01).to(torch.bfloat16); dyt[:, :, O:] = 0
whT = torch.zeros(Hd, 32, dtype=torch.bfloat16, device=dev); whT[:, :O] = (torch.randn(Hd, O, device=dev) * .1).to(torch.bfloat16)
for _ in range(2):
    out = lstm2_backward(saved, None, pb, pk.kx, head=(dyt, whT))
torch.cuda.synchronize()
print("timeouts", ops_lstm.coop_timeouts(), flush=True)
