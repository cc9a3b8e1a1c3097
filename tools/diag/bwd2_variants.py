"""A/B timing of build variants of the K-split cooperative LSTM backward (csrc/lstm_coop.hip, -D switches).
  python tools/diag/bwd2_variants.py --build      (CPU: cross-compiles tools/diag/libv_<name>.so for every variant)
  python tools/diag/bwd2_variants.py              (GPU box: one process per variant, C2 shape N=4096, T'=253, fused head)"""
import glob, os, subprocess, sys, time
root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
VARIANTS = {"base": [], "g4_d4": ["-DC4_DEPTH=4"], "g4_d6": ["-DC4_DEPTH=6"], "g4_d8": ["-DC4_DEPTH=8"], "g4_d10": ["-DC4_DEPTH=10"],
            "apad16": ["-DCOOP_APAD=16"], "g4_d8_nodg_nofetch": ["-DC4_DEPTH=8", "-DC2_NO_DG", "-DC2_NO_FETCH"]}       # "g4*": the four-CU kernel (NPPC_LSTM_BWD_G4=1)
R02_REV = "HEAD"        # "r02" variant = lstm_coop.hip of that commit for A/B on one box
so = lambda n: os.path.join(root, "tools", "diag", f"libv_{n}.so")
if "--build" in sys.argv:
    csrc = os.path.join(root, "generative-audio_amd", "csrc")
    objs = [os.path.join(root, "generative-audio_amd", "build", os.path.basename(f)[:-4] + ".o") for f in sorted(glob.glob(csrc + "/*.hip"))
            if not f.endswith("lstm_coop.hip")]
    for n, flags in [(n, f) for n, f in VARIANTS.items() if not [a for a in sys.argv[1:] if a in VARIANTS] or n in sys.argv]:
        o = f"/tmp/lstm_coop_{n}.o"
        src = os.path.join(csrc, "lstm_coop.hip")
        if n == "r02":
            src = "/tmp/lstm_coop_r02.hip"
            open(src, "w").write(subprocess.check_output(["git", "-C", root, "show", R02_REV + ":generative-audio_amd/csrc/lstm_coop.hip"], text=True))
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(root, "include"),
                               "-I" + csrc, "-Wno-unused-value", *flags, "-c", src, "-o", o])
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so(n), o] + objs)
        print("built", so(n), flush=True)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    name = sys.argv[2]
    if name.startswith("g4"):
        os.environ["NPPC_LSTM_BWD_G4"] = "1"
    sys.path.insert(0, os.path.join(root, "generative-audio_amd"))
    import torch
    from nppc_audio import _hip as H
    H.LIB_PATH = so(name)
    from nppc_audio import ops_lstm
    from nppc_audio.ops_lstm import PackedLSTM, PackedLSTMBwd, lstm2_forward, lstm2_backward
    dev = torch.device("cuda")
    I, Hd, Tn, N, O = 34, 384, 253, 4096, 10
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
    for _ in range(3):
        out = lstm2_backward(saved, None, pb, pk.kx, head=(dyt, whT))
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = lstm2_backward(saved, None, pb, pk.kx, head=(dyt, whT)); e1.record()
        torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    print(f"{name:22s} {min(ts):7.3f} ms (min of 5; mean {sum(ts)/5:.3f})  dx_abs_sum={float(out[0].float().abs().sum()):.6e} "
          f"dg2_abs_sum={float(out[2].float().abs().sum()):.6e} timeouts={ops_lstm.coop_timeouts()}", flush=True)
    sys.exit(0)
names = [a for a in sys.argv[1:] if a in VARIANTS] or list(VARIANTS)
for n in names:
    subprocess.call([sys.executable, os.path.abspath(__file__), "--one", n])
