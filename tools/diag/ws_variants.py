"""A/B timing of diagnostic builds of the weight-stationary forward (csrc/lstm_ws.hip, -DWS_DIAG_* switches): where the
time of an item goes.  Diagnostic builds give garbage results (and may count time-outs); only their timing is used.
  python tools/diag/ws_variants.py --build     (CPU: cross-compiles tools/diag/libws_<name>.so)
  python tools/diag/ws_variants.py [names]     (GPU box: one process per variant)"""
import glob, os, subprocess, sys
root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
VARIANTS = {"base": [], "nocell": ["-DWS_DIAG_NOCELL"], "notilewrite": ["-DWS_DIAG_NOTILEWRITE"], "nogather": ["-DWS_DIAG_NOGATHER"],
            "gather_l2": ["-DWS_DIAG_GATHERL2"], "mfma16": ["-DWS_MFMA16=1"], "mfma16_all_but_mfma": ["-DWS_MFMA16=1", "-DWS_DIAG_NOGATHER", "-DWS_DIAG_NOTILEWRITE", "-DWS_DIAG_NOFLUSH", "-DWS_DIAG_NOCELL"],
            "noflush": ["-DWS_DIAG_NOFLUSH"], "nomfma": ["-DWS_DIAG_NOMFMA"], "nobarrier": ["-DWS_DIAG_NOBARRIER"],
            "nogather_notilewrite_noflush": ["-DWS_DIAG_NOGATHER", "-DWS_DIAG_NOTILEWRITE", "-DWS_DIAG_NOFLUSH"],
            "all_but_mfma": ["-DWS_DIAG_NOGATHER", "-DWS_DIAG_NOTILEWRITE", "-DWS_DIAG_NOFLUSH", "-DWS_DIAG_NOCELL"]}
so = lambda n: os.path.join(root, "tools", "diag", f"libws_{n}.so")
if "--build" in sys.argv:
    only = [a for a in sys.argv[1:] if not a.startswith("--")]
    if only:
        VARIANTS = {k: v for k, v in VARIANTS.items() if k in only}
    csrc = os.path.join(root, "generative-audio_amd", "csrc")
    objs = [os.path.join(root, "generative-audio_amd", "build", os.path.basename(f)[:-4] + ".o") for f in sorted(glob.glob(csrc + "/*.hip"))
            if not f.endswith("lstm_ws.hip")]
    for n, flags in VARIANTS.items():
        o = f"/tmp/lstm_ws_{n}.o"
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(root, "include"),
                               "-I" + csrc, "-Wno-unused-value", "-DWS_STAMP", *flags, "-c", os.path.join(csrc, "lstm_ws.hip"), "-o", o])
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
    from nppc_audio.ops_lstm import PackedLSTM, lstm2_forward
    dev = torch.device("cuda")
    I, Hd, Tn = 34, 384, 253
    torch.manual_seed(0)
    ws = [torch.randn(4 * Hd, I) * .05, torch.randn(4 * Hd, Hd) * .05, torch.zeros(4 * Hd), torch.zeros(4 * Hd),
          torch.randn(4 * Hd, Hd) * .05, torch.randn(4 * Hd, Hd) * .05, torch.zeros(4 * Hd), torch.zeros(4 * Hd)]
    pk = PackedLSTM(I, Hd, 0, dev).pack(*[w.to(dev) for w in ws])
    orig = ops_lstm.workspace
    def ws2(key, shape, dtype, device, zero=False):
        if key[-1] == "coop_flags":
            shape = (shape[0] + 64,)
        return orig(key, shape, dtype, device, zero)
    ops_lstm.workspace = ws2
    res = []
    for N, train, O in ((8224, False, 2), (4096, True, 10)):
        x = torch.randn(Tn, N, pk.kx, device=dev).to(torch.bfloat16); x[:, :, I:] = 0
        wh = torch.zeros(16, Hd, dtype=torch.bfloat16, device=dev); wh[:O] = (torch.randn(O, Hd) * 0.1).to(dev)
        lstm2_forward(x, pk, train, "ws", head=(wh, O))
        torch.cuda.synchronize()
        ts = []
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); lstm2_forward(x, pk, train, "ws", head=(wh, O)); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        fl = [t for k, t in ops_lstm._WS.items() if k[0][-1] == "coop_flags"][-1]
        ncl, nch = ops_lstm.ws_plan(N, pk)
        dbg = fl[ncl * nch * 16 + 4: ncl * nch * 16 + 4 + 32].view(torch.int64).cpu().tolist()
        cyc = sum(dbg[:8])
        ms = sorted(ts)[1]
        res.append(f"{'train' if train else 'infer'} N={N}: {ms:.3f} ms, {cyc / ((N // 32 + ncl - 1) // ncl * (Tn + 2)):.0f} cycles/item, {cyc / ms / 1e6:.2f} GHz")
    print(f"{name:32s} " + "   ".join(res) + f"   timeouts={ops_lstm.coop_timeouts()}", flush=True)
    sys.exit(0)
names = sys.argv[1:] or list(VARIANTS)
for n in names:
    try:
        subprocess.call([sys.executable, os.path.abspath(__file__), "--one", n], timeout=60)
    except subprocess.TimeoutExpired:
        print(f"{n:32s} timed out", flush=True)
