"""Sizing builds of the cooperative LSTM FORWARD kernel (csrc/lstm_coop.hip, -D switches): what the
pointwise phases and the in-step barriers cost a time step.  Diagnostic builds give garbage results; timing only.
  python tools/diag/fwd_variants.py --build      (CPU: cross-compiles tools/diag/libf_<name>.so for every variant)
  python tools/diag/fwd_variants.py [names]      (GPU box: one process per variant; C2 shapes, T' = 253, fused heads:
                                                  restorer forward N = 8224 (80-row pairs), training forward N = 4096)"""
import glob, os, subprocess, sys
root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
VARIANTS = {"base": [], "nocell": ["-DCF_NO_CELL"], "nobar": ["-DCF_NO_BAR"], "nocell_nobar": ["-DCF_NO_CELL", "-DCF_NO_BAR"],
            "prio_old": ["-DCF_PRIO=1"], "prio_young": ["-DCF_PRIO=2"],
            "w_l1": ["-DCF_W_L1"], "w_l1_nocell": ["-DCF_W_L1", "-DCF_NO_CELL"],
            "w_l1_nocell_nobar_nopoll": ["-DCF_W_L1", "-DCF_NO_CELL", "-DCF_NO_BAR", "-DCF_NO_POLL"],
            # A-operand (LDS) prefetch of the 80-row GEMM: agX_apfY = X row tiles per group, reads Y groups ahead of their MFMAs
            # (ag2_apf0 = round 3; VGPR spills of the restorer instantiation: ag2_apf1 6, ag1_apf1 0, ag1_apf2 6)
            "ag2_apf0": ["-DCF_AG=2", "-DCF_APF=0"], "ag2_apf1": ["-DCF_AG=2", "-DCF_APF=1"],
            "ag1_apf1": ["-DCF_AG=1", "-DCF_APF=1"], "ag1_apf2": ["-DCF_AG=1", "-DCF_APF=2"],
            # h2 hand-off polled / requested at the top of the step (round 3) or behind layer 1's first GEMM (round 4)
            "apad16": ["-DCOOP_APAD=16"], "r3": ["-DCF_AG=2", "-DCF_APF=0", "-DCF_H2_LATE=0"], "h2top_ag1_apf1": ["-DCF_AG=1", "-DCF_APF=1", "-DCF_H2_LATE=0"]}
so = lambda n: os.path.join(root, "tools", "diag", f"libf_{n}.so")
if "--build" in sys.argv:
    csrc = os.path.join(root, "generative-audio_amd", "csrc")
    objs = [os.path.join(root, "generative-audio_amd", "build", os.path.basename(f)[:-4] + ".o") for f in sorted(glob.glob(csrc + "/*.hip"))
            if not f.endswith("lstm_coop.hip")]
    import concurrent.futures as cf

    def one(item):
        n, flags = item
        o = f"/tmp/lstm_coop_f_{n}.o"
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(root, "include"),
                               "-I" + csrc, "-Wno-unused-value", *flags, "-c", os.path.join(csrc, "lstm_coop.hip"), "-o", o])
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so(n), o] + objs)
        return so(n)
    sel = [a for a in sys.argv[1:] if a in VARIANTS]
    with cf.ThreadPoolExecutor(max_workers=4) as ex:
        for r in ex.map(one, [(n, f) for n, f in VARIANTS.items() if not sel or n in sel]):
            print("built", r, flush=True)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    name = sys.argv[2]
    sys.path.insert(0, os.path.join(root, "generative-audio_amd"))
    import torch
    from nppc_audio import _hip as H
    H.LIB_PATH = so(name)
    from nppc_audio.ops_lstm import PackedLSTM, lstm2_forward
    dev = torch.device("cuda")
    I, Hd, Tn, O = 34, 384, 253, 10
    torch.manual_seed(0)
    ws = [torch.randn(4 * Hd, I) * .05, torch.randn(4 * Hd, Hd) * .05, torch.zeros(4 * Hd), torch.zeros(4 * Hd),
          torch.randn(4 * Hd, Hd) * .05, torch.randn(4 * Hd, Hd) * .05, torch.zeros(4 * Hd), torch.zeros(4 * Hd)]
    ws = [w.to(dev) for w in ws]
    pk = PackedLSTM(I, Hd, 0, dev).pack(*ws)
    whp = torch.zeros(16, Hd, dtype=torch.bfloat16, device=dev); whp[:O] = (torch.randn(O, Hd, device=dev) * .1).to(torch.bfloat16)

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record()
            torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        return min(ts), sum(ts) / len(ts)

    xr = torch.randn(Tn, 8224, pk.kx, device=dev).to(torch.bfloat16); xr[:, :, I:] = 0
    t_rest = timed(lambda: lstm2_forward(xr, pk, False, None, head=(whp, 2)))
    del xr
    x = torch.randn(Tn, 4096, pk.kx, device=dev).to(torch.bfloat16); x[:, :, I:] = 0
    t_dir = timed(lambda: lstm2_forward(x, pk, True, None, head=(whp, O)))
    print(f"{name:26s} restorer fwd {t_rest[0]:7.3f} ms (mean {t_rest[1]:7.3f})   direction fwd (train) {t_dir[0]:7.3f} ms (mean {t_dir[1]:7.3f})", flush=True)
    sys.exit(0)
names = [a for a in sys.argv[1:] if a in VARIANTS] or list(VARIANTS)
for n in names:
    subprocess.call([sys.executable, os.path.abspath(__file__), "--one", n])
