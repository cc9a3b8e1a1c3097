"""A/B timing of build variants of the cooperative LSTM kernels (csrc/lstm_coop.hip, -D switches): where the time of a
recurrent step goes.  Diagnostic builds give garbage results; only their timing is used.
  python tools/diag/lstm_variants.py --build      (CPU: cross-compiles tools/diag/libv_<name>.so for every variant)
  python tools/diag/lstm_variants.py [names]      (GPU box: one process per variant; C2 shapes: restorer forward N=8224,
                                                   direction-net forward (train) and backward N=4096, T'=253, fused heads)"""
import glob, os, subprocess, sys
root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
VARIANTS = {"base": [], "w_l1": ["-DCF_W_L1"], "nopoll": ["-DCF_NO_POLL"], "w_l1_nopoll": ["-DCF_W_L1", "-DCF_NO_POLL"],
            "nosave": ["-DCF_NO_SAVE", "-DC2_NO_DG", "-DC2_NO_FETCH"],
            "w_l1_nopoll_nosave": ["-DCF_W_L1", "-DCF_NO_POLL", "-DCF_NO_SAVE", "-DC2_NO_DG", "-DC2_NO_FETCH"]}
so = lambda n: os.path.join(root, "tools", "diag", f"libv_{n}.so")
if "--build" in sys.argv:
    csrc = os.path.join(root, "generative-audio_amd", "csrc")
    objs = [os.path.join(root, "generative-audio_amd", "build", os.path.basename(f)[:-4] + ".o") for f in sorted(glob.glob(csrc + "/*.hip"))
            if not f.endswith("lstm_coop.hip")]
    for n, flags in VARIANTS.items():
        o = f"/tmp/lstm_coop_{n}.o"
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(root, "include"),
                               "-I" + csrc, "-Wno-unused-value", *flags, "-c", os.path.join(csrc, "lstm_coop.hip"), "-o", o])
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so(n), o] + objs)
        print("built", so(n), flush=True)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    name = sys.argv[2]
    sys.path.insert(0, os.path.join(root, "generative-audio_amd"))
    import torch
    from nppc_audio import _hip as H
    H.LIB_PATH = so(name)
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

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record()
            torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        return min(ts)

    xr = torch.randn(Tn, 8224, pk.kx, device=dev).to(torch.bfloat16); xr[:, :, I:] = 0
    t_rest = timed(lambda: lstm2_forward(xr, pk, False, None, head=(whp, 2)))
    del xr
    x = torch.randn(Tn, 4096, pk.kx, device=dev).to(torch.bfloat16); x[:, :, I:] = 0
    t_dir = timed(lambda: lstm2_forward(x, pk, True, None, head=(whp, O)))
    saved = lstm2_forward(x, pk, True, None, head=(whp, O))
    dyt = (torch.randn(Tn, 4096, 16, device=dev) * .01).to(torch.bfloat16); dyt[:, :, O:] = 0
    whT = torch.zeros(Hd, 32, dtype=torch.bfloat16, device=dev); whT[:, :O] = (torch.randn(Hd, O, device=dev) * .1).to(torch.bfloat16)
    t_bwd = timed(lambda: lstm2_backward(saved, None, pb, pk.kx, head=(dyt, whT)))
    print(f"{name:22s} restorer fwd {t_rest:7.3f} ms   direction fwd (train) {t_dir:7.3f} ms   direction bwd {t_bwd:7.3f} ms", flush=True)
    sys.exit(0)
names = [a for a in sys.argv[1:] if a in VARIANTS] or list(VARIANTS)
for n in names:
    subprocess.call([sys.executable, os.path.abspath(__file__), "--one", n])
