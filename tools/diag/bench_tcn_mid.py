"""Stand-alone timing of the fused TCN mid-block backward at BASELINE C2 shape (GPU box).  NPPC_TCN_RPB is read once per
process, so every setting runs in its own process:  python tools/diag/bench_tcn_mid.py [rpb ...]"""
import os, subprocess, sys
root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, os.path.join(root, "generative-audio_amd"))
    import torch
    from nppc_audio import _hip as H
    Z, B, C, Tp, Tv, dil = 3, 32, 512, 256, 253, 5
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(0)
    mk = lambda: (torch.randn(Z, B, Tp, C, generator=g) * 0.5).to(dt).cuda()
    dA, y2, y1 = mk(), mk(), mk()
    st = torch.zeros(Z, B, 2, dtype=torch.float64, device="cuda"); st[..., 1] = C * Tv * 0.3
    S = torch.empty(Z, B, C // 64, 8, dtype=torch.float64, device="cuda")
    part = torch.empty(H.mid_bwd_part_elems(B, C, Tp, Z), device="cuda")
    sP = 1 << 20
    par = lambda: torch.rand(Z * sP, device="cuda") * 0.5 + 0.25
    p = [par() for _ in range(7)]
    gr = [torch.zeros(Z * sP, device="cuda") for _ in range(9)]
    a2, dpre = torch.empty_like(dA), torch.empty_like(dA)
    def run():
        H.call("nppc_tcn_mid_bwd", 0, dA, y2, y1, st, st, S, part, p[0], p[1], p[2], p[3], p[4], p[5], p[6], a2, dpre, *gr, None, 0, 0, 0, None, B, C, Tp, Tv,
               dil, 1e-8, B * Tp * C, B * 2, sP, Z, 1, H.stream())
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    mb = 8 * Z * B * Tv * C * 2 / 1e6          # reduce: 3 reads + 1 write, apply: 3 reads + 1 write
    print(f"RPB={os.environ.get('NPPC_TCN_RPB', '64'):>4s}: {ms*1e3:8.1f} us per call  ({mb/ms/1e3:.2f} TB/s on {mb:.0f} MB algorithmic)", flush=True)
    sys.exit(0)
for rpb in (sys.argv[1:] or ["64", "32", "16", "8", "4"]):
    subprocess.call([sys.executable, os.path.abspath(__file__), "--one"], env=dict(os.environ, NPPC_TCN_RPB=rpb))
