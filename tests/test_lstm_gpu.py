"""GPU parity: fused 2-layer LSTM kernel (through the C ABI) vs the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import nppc_ref as R
from oracle import weights as W

pytestmark = pytest.mark.gpu


def _weights(I, Hd, seed):
    spec = {k: v for k, v in W.fullsubnet_spec(num_freqs=9, sb_neighbors=(I - 4) // 2, sb_hidden=Hd).items()
            if k.startswith("sb_model.sequence_model")}
    assert spec["sb_model.sequence_model.weight_ih_l0"][1] == I
    return {k: torch.from_numpy(v) for k, v in W.make_weights(spec, seed).items()}


def _run(I, Hd, N, Tn, prec, train, mtile, seed=0):
    from nppc_audio import _hip as H
    from nppc_audio.ops_lstm import PackedLSTM, lstm2_forward
    P = _weights(I, Hd, seed)
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(N, Tn, I, generator=g)
    ref = R.lstm2(x, P, "sb_model.sequence_model")                     # [N,Tn,H]
    dev = torch.device("cuda")
    pk = PackedLSTM(I, Hd, prec, dev)
    pre = "sb_model.sequence_model."
    pk.pack(*[P[pre + n].to(dev) for n in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0",
                                             "weight_ih_l1", "weight_hh_l1", "bias_ih_l1", "bias_hh_l1")])
    xt = torch.zeros(Tn, N, pk.kx, dtype=H.dtype_of(prec), device=dev)
    xt[:, :, :I] = x.permute(1, 0, 2).to(dev)
    out = lstm2_forward(xt, pk, train, mtile)
    torch.cuda.synchronize()
    got = out["h2"].float().cpu().permute(1, 0, 2)
    return got, ref, out, (x, P)


@pytest.mark.parametrize("Hd,I", [(16, 10), (384, 34)])
@pytest.mark.parametrize("train", [False, True])
def test_lstm_fwd_f32_matches_oracle(Hd, I, train):
    N, Tn = (37, 19) if Hd == 16 else (50, 23)
    got, ref, out, (x, P) = _run(I, Hd, N, Tn, 1, train, 1)
    err = (got - ref).abs().max().item()
    assert err < 2e-5, err          # fp32 tolerance: exact-f32 MFMA, rcp/exp 1-ulp activations
    if train:
        # saved state: h1 and the post-activation gates/cell of both layers against the explicit recurrence
        h1 = out["h1"].float().cpu().permute(1, 0, 2)
        ref_h1 = _layer_out(x, P, 0)
        assert (h1 - ref_h1).abs().max().item() < 2e-5
        assert float(out["h1_rows"][Tn * N:].abs().max()) == 0 and float(out["h2_rows"][Tn * N:].abs().max()) == 0
        g2 = out["g2"].float().cpu()
        assert torch.isfinite(g2).all() and g2[..., 0].min() >= 0 and g2[..., 0].max() <= 1   # i
        assert g2[..., 1].abs().max() <= 1 + 1e-6                                            # g (tanh)
        c2 = out["c2"].float().cpu().permute(1, 0, 2)
        o2 = g2[..., 3].permute(1, 0, 2)
        assert (o2 * torch.tanh(c2) - got).abs().max().item() < 2e-5                         # h = o*tanh(c)


def _layer_out(x, P, layer):
    pre = "sb_model.sequence_model."
    Hd = P[pre + "weight_hh_l0"].shape[1]
    wi, wh = P[pre + f"weight_ih_l{layer}"], P[pre + f"weight_hh_l{layer}"]
    b = P[pre + f"bias_ih_l{layer}"] + P[pre + f"bias_hh_l{layer}"]
    h = x.new_zeros(x.shape[0], Hd)
    c = x.new_zeros(x.shape[0], Hd)
    outs = []
    for t in range(x.shape[1]):
        g = x[:, t] @ wi.t() + h @ wh.t() + b
        i, f, gg, o = g.split(Hd, dim=1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
        outs.append(h)
    return torch.stack(outs, dim=1)


@pytest.mark.parametrize("Hd,I,mtile", [(16, 10, 1), (16, 10, 2), (384, 34, 1), (384, 34, 2), (384, 34, 3)])
def test_lstm_fwd_bf16_matches_oracle(Hd, I, mtile):
    N, Tn = (37, 19) if Hd == 16 else (70, 40)
    got, ref, _, _ = _run(I, Hd, N, Tn, 0, False, mtile)
    err = (got - ref).abs().max().item()
    # bf16 operands (2^-9 relative rounding of x, h, W), fp32 accumulate and cell state
    assert err < 3e-2, err


def test_lstm_fwd_bf16_train_saves_consistent_state():
    got, ref, out, _ = _run(34, 384, 40, 12, 0, True, 2)
    assert (got - ref).abs().max().item() < 3e-2
    c2 = out["c2"].float().cpu().permute(1, 0, 2)
    o2 = out["g2"].float().cpu()[..., 3].permute(1, 0, 2)
    assert (o2 * torch.tanh(c2) - got).abs().max().item() < 2e-2


def _unperm(dgT, Hd):
    """[4H (k = u*4 + g', g' in i,g,f,o)][R] -> torch gate-row order [4H (i,f,g,o blocks)][R]"""
    R = dgT.shape[1]
    d = dgT.reshape(Hd, 4, R)
    return torch.cat([d[:, 0], d[:, 2], d[:, 1], d[:, 3]], dim=0)


@pytest.mark.parametrize("Hd,I,prec,tol", [(16, 10, 1, 2e-4), (384, 34, 1, 2e-4), (384, 34, 0, 4e-2), (16, 10, 0, 4e-2)])
def test_lstm_bwd_matches_autograd(Hd, I, prec, tol):
    from nppc_audio import _hip as H
    from nppc_audio.ops_lstm import PackedLSTM, PackedLSTMBwd, lstm2_forward, lstm2_backward
    N, Tn = (37, 9) if Hd == 16 else (40, 11)
    P = _weights(I, Hd, 3)
    pre = "sb_model.sequence_model."
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, Tn, I, generator=g)
    G = torch.randn(N, Tn, Hd, generator=g)
    Pr = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    xr = x.clone().requires_grad_(True)
    (R.lstm2_steps(xr, Pr, "sb_model.sequence_model") * G).sum().backward()

    dev = torch.device("cuda")
    dt = H.dtype_of(prec)
    names = ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0", "weight_ih_l1", "weight_hh_l1", "bias_ih_l1",
             "bias_hh_l1")
    pk = PackedLSTM(I, Hd, prec, dev).pack(*[P[pre + n].to(dev) for n in names])
    pb = PackedLSTMBwd(I, Hd, prec, dev).pack(*[P[pre + n].to(dev) for n in
                                                 ("weight_ih_l0", "weight_hh_l0", "weight_ih_l1", "weight_hh_l1")])
    xt = torch.zeros(Tn, N, pk.kx, dtype=dt, device=dev)
    xt[:, :, :I] = x.permute(1, 0, 2).to(dev)
    saved = lstm2_forward(xt, pk, True, 1)
    dh2 = G.permute(1, 0, 2).contiguous().to(dev).to(dt)
    dx, dg1T, dg2T = lstm2_backward(saved, dh2, pb, pk.kx)
    torch.cuda.synchronize()

    def rel(a, b):
        return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()

    got_dx = dx.float().cpu()[:, :, :I].permute(1, 0, 2)
    assert rel(got_dx, xr.grad) < tol, rel(got_dx, xr.grad)
    R_ = Tn * N
    assert float(dg1T[R_:].abs().max()) == 0                                   # padding rows stay zero
    d1 = _unperm(dg1T[:R_].float().cpu().t().contiguous(), Hd)                 # [4H][R], R index = t*N + n
    d2 = _unperm(dg2T[:R_].float().cpu().t().contiguous(), Hd)
    xf = x.permute(1, 0, 2).reshape(Tn * N, I)
    h1 = saved["h1"].float().cpu().reshape(R_, Hd)                             # [R][H]
    h2 = saved["h2"].float().cpu().reshape(R_, Hd)
    checks = {
        "weight_ih_l0": d1 @ xf, "weight_hh_l0": d1[:, N:] @ h1[:-N], "bias_ih_l0": d1.sum(1), "bias_hh_l0": d1.sum(1),
        "weight_ih_l1": d2 @ h1, "weight_hh_l1": d2[:, N:] @ h2[:-N], "bias_ih_l1": d2.sum(1), "bias_hh_l1": d2.sum(1),
    }
    for n, got in checks.items():
        r = rel(got, Pr[pre + n].grad)
        assert r < tol, (n, r)


@pytest.mark.parametrize("N,Tn,train,force", [(50, 23, False, (2, 2)), (50, 23, True, (2, 2)), (200, 9, False, (2, 5)),
                                              (512, 7, True, (4, 4)), (100, 11, False, (4, 4)), (777, 5, True, (2, 2)),
                                              (300, 9, True, (4, 2)), (90, 13, False, (4, 2)),
                                              (100, 11, False, (8, 2)), (300, 9, True, (8, 2)), (200, 7, False, (8, 5))])
def test_cooperative_forward_matches_single_workgroup_kernel_and_oracle(N, Tn, train, force):
    """The CU-pair kernel (weights split over two workgroups, h slices exchanged through global memory) must give
    the same numbers as the single-workgroup kernel, and no bounded spin may time out."""
    from nppc_audio import _hip as H
    from nppc_audio import ops_lstm
    from nppc_audio.ops_lstm import PackedLSTM, lstm2_forward
    I, Hd = 34, 384
    P = _weights(I, Hd, 1)
    pre = "sb_model.sequence_model."
    dev = torch.device("cuda")
    pk = PackedLSTM(I, Hd, 0, dev).pack(*[P[pre + n].to(dev) for n in (
        "weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0", "weight_ih_l1", "weight_hh_l1", "bias_ih_l1", "bias_hh_l1")])
    g = torch.Generator().manual_seed(N)
    x = torch.randn(N, Tn, I, generator=g)
    xt = torch.zeros(Tn, N, pk.kx, dtype=torch.bfloat16, device=dev)
    xt[:, :, :I] = x.permute(1, 0, 2).to(dev)
    coop = {k: v.clone() for k, v in lstm2_forward(xt, pk, train, force).items()}
    assert ops_lstm.coop_timeouts() == 0
    single = {k: v.clone() for k, v in lstm2_forward(xt, pk, train, 1).items()}
    torch.cuda.synchronize()
    ref = R.lstm2(x, P, "sb_model.sequence_model")
    got = coop["h2"].float().cpu().permute(1, 0, 2)
    assert (got - ref).abs().max().item() < 3e-2
    for k in coop:
        d = (coop[k].float() - single[k].float()).abs().max().item()
        assert d < 2e-2, (k, d)      # same bf16 arithmetic, different accumulation split


@pytest.mark.parametrize("N,Tn", [(40, 11), (96, 7), (333, 5)])
def test_cooperative_backward_matches_single_workgroup_kernel(N, Tn):
    from nppc_audio import ops_lstm
    from nppc_audio.ops_lstm import PackedLSTM, PackedLSTMBwd, lstm2_forward, lstm2_backward
    I, Hd = 34, 384
    P = _weights(I, Hd, 2)
    pre = "sb_model.sequence_model."
    dev = torch.device("cuda")
    names = ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0", "weight_ih_l1", "weight_hh_l1", "bias_ih_l1", "bias_hh_l1")
    pk = PackedLSTM(I, Hd, 0, dev).pack(*[P[pre + n].to(dev) for n in names])
    pb = PackedLSTMBwd(I, Hd, 0, dev).pack(*[P[pre + n].to(dev) for n in ("weight_ih_l0", "weight_hh_l0", "weight_ih_l1", "weight_hh_l1")])
    g = torch.Generator().manual_seed(N + Tn)
    xt = torch.zeros(Tn, N, pk.kx, dtype=torch.bfloat16, device=dev)
    xt[:, :, :I] = torch.randn(Tn, N, I, generator=g).to(dev)
    dh2 = torch.randn(Tn, N, Hd, generator=g).to(dev).to(torch.bfloat16)
    saved = lstm2_forward(xt, pk, True, 1)
    a = [t.clone() for t in lstm2_backward(saved, dh2, pb, pk.kx, coop=True)]
    assert ops_lstm.coop_timeouts() == 0
    b = [t.clone() for t in lstm2_backward(saved, dh2, pb, pk.kx, coop=False)]
    torch.cuda.synchronize()
    for name, x, y in zip(("dx", "dg1T", "dg2T"), a, b):
        x, y = x.float(), y.float()
        d = (x - y).abs().max().item() / (y.abs().max().item() + 1e-30)
        assert d < 2e-2, (name, d)


@pytest.mark.parametrize("N,Tn,head", [(192, 7, False), (160, 5, True), (77, 6, True), (640, 4, False)])
def test_four_cu_cluster_backward_matches_cu_pair_backward(N, Tn, head, monkeypatch):
    """The K-split backward on four-CU clusters of 64 sequences (lstm2_coop_bwd4_kernel: three partial-sum shipments per layer and
    step in one continuous fragment stream; opt-in, NPPC_LSTM_BWD_G4=1) against the CU-pair kernel and the single-workgroup
    kernel: ragged last clusters (160 = 2.5 x 64, 77), with d h2 given and with the fused head backward; no time-outs."""
    from nppc_audio import ops_lstm
    from nppc_audio.ops_lstm import PackedLSTM, PackedLSTMBwd, lstm2_forward, lstm2_backward
    I, Hd, O = 34, 384, 10
    P = _weights(I, Hd, 5)
    pre = "sb_model.sequence_model."
    dev = torch.device("cuda")
    names = ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0", "weight_ih_l1", "weight_hh_l1", "bias_ih_l1", "bias_hh_l1")
    wnames = ("weight_ih_l0", "weight_hh_l0", "weight_ih_l1", "weight_hh_l1")
    pk = PackedLSTM(I, Hd, 0, dev).pack(*[P[pre + n].to(dev) for n in names])
    monkeypatch.setattr(ops_lstm, "BWD_G4", "1")
    pb = PackedLSTMBwd(I, Hd, 0, dev).pack(*[P[pre + n].to(dev) for n in wnames])      # packs the four-CU layout as well
    g = torch.Generator().manual_seed(N + Tn)
    xt = torch.zeros(Tn, N, pk.kx, dtype=torch.bfloat16, device=dev)
    xt[:, :, :I] = torch.randn(Tn, N, I, generator=g).to(dev)
    saved = lstm2_forward(xt, pk, True, 1)
    if head:
        dyt = torch.zeros(Tn, N, 16, dtype=torch.bfloat16, device=dev)
        dyt[:, :, :O] = (torch.randn(Tn, N, O, generator=g) * 0.1).to(dev)
        whT = torch.zeros(Hd, 32, dtype=torch.bfloat16, device=dev)
        whT[:, :O] = (torch.randn(Hd, O, generator=g) * 0.1).to(dev)
        args = dict(dh2=None, head=(dyt, whT))
    else:
        args = dict(dh2=torch.randn(Tn, N, Hd, generator=g).to(dev).to(torch.bfloat16), head=None)
    ops_lstm.clear_coop_timeouts()
    four = [t.clone() for t in lstm2_backward(saved, args["dh2"], pb, pk.kx, coop=True, head=args["head"])]
    monkeypatch.setattr(ops_lstm, "BWD_G4", "0")
    pair = [t.clone() for t in lstm2_backward(saved, args["dh2"], pb, pk.kx, coop=True, head=args["head"])]
    torch.cuda.synchronize()
    assert ops_lstm.coop_timeouts() == 0
    R_ = Tn * N
    for name, x, y in zip(("dx", "dg1", "dg2"), four, pair):
        x, y = (x[:R_].float(), y[:R_].float()) if x.dim() == 2 else (x.float(), y.float())
        assert bool(torch.isfinite(x).all()), name
        d = (x - y).abs().max().item() / (y.abs().max().item() + 1e-30)
        assert d < 2e-2, (name, d)            # both bf16 kernels: the partial sums are rounded to bf16 at different places
    if not head:
        single = [t.clone() for t in lstm2_backward(saved, args["dh2"], pb, pk.kx, coop=False)]
        for name, x, y in zip(("dx", "dg1", "dg2"), four, single):
            x, y = (x[:R_].float(), y[:R_].float()) if x.dim() == 2 else (x.float(), y.float())
            assert (x - y).abs().max().item() / (y.abs().max().item() + 1e-30) < 2e-2, name


@pytest.mark.parametrize("N,Tn,force,O,Fo,la", [(96, 9, (2, 2), 2, 48, 2), (200, 7, (2, 5), 2, 100, 0), (161, 5, (2, 5), 4, 161, 1)])
def test_fused_head_equals_separate_head_kernel(N, Tn, force, O, Fo, la):
    """inference: Linear(H -> O) fused into the CU-pair kernel (per-CU partial sums + nppc_sb_head_finalize) against the
    same kernel storing h2 followed by nppc_sb_head (ragged last row tile, look-ahead crop, several samples)"""
    from nppc_audio import _hip as H
    from nppc_audio import ops_lstm
    from nppc_audio.ops_lstm import PackedLSTM, lstm2_forward
    I, Hd = 34, 384
    P = _weights(I, Hd, 3)
    pre = "sb_model.sequence_model."
    dev = torch.device("cuda")
    pk = PackedLSTM(I, Hd, 0, dev).pack(*[P[pre + n].to(dev) for n in (
        "weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0", "weight_ih_l1", "weight_hh_l1", "bias_ih_l1", "bias_hh_l1")])
    g = torch.Generator().manual_seed(N + O)
    xt = torch.zeros(Tn, N, pk.kx, dtype=torch.bfloat16, device=dev)
    xt[:, :, :I] = torch.randn(Tn, N, I, generator=g).to(dev)
    wh = torch.zeros(16, Hd, dtype=torch.bfloat16, device=dev)
    wh[:O] = (torch.randn(O, Hd, generator=g) * 0.1).to(dev)
    bias = torch.randn(O, generator=g).to(dev)
    Bq = N // Fo
    s = H.stream()
    plain = lstm2_forward(xt, pk, False, force)
    want = torch.empty(Bq, O, Fo, Tn - la, dtype=torch.float32, device=dev)
    H.call("nppc_sb_head", 0, plain["h2"], wh, bias, want, N, Tn, la, Hd, O, Fo, s)
    fused = lstm2_forward(xt, pk, False, force, head=(wh, O))
    assert "head_partial" in fused and "h2" not in fused and ops_lstm.coop_timeouts() == 0
    got = torch.full_like(want, float("nan"))
    H.call("nppc_sb_head_finalize", fused["head_partial"], fused["head_partial"].shape[0], bias, got, N, Tn, la, O, Fo, s)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(got).all())
    assert float((got - want).abs().max()) < 2e-5 * max(1.0, float(want.abs().max()))      # same bf16 operands, fp32 sums
    # a plan without the pair kernel ignores the request and returns h2
    single = lstm2_forward(xt, pk, False, 1, head=(wh, O))
    assert "h2" in single and "head_partial" not in single
    if force == (2, 2):
        # training variant: saved state AND the fused head (K = 5 directions -> O = 10)
        O2 = 10
        wh2 = torch.zeros(16, Hd, dtype=torch.bfloat16, device=dev)
        wh2[:O2] = (torch.randn(O2, Hd, generator=g) * 0.1).to(dev)
        b2 = torch.randn(O2, generator=g).to(dev)
        ref_state = {k: v.clone() for k, v in lstm2_forward(xt, pk, True, force).items()}
        want2 = torch.empty(Bq, O2, Fo, Tn - la, dtype=torch.float32, device=dev)
        H.call("nppc_sb_head", 0, ref_state["h2"], wh2, b2, want2, N, Tn, la, Hd, O2, Fo, s)
        tr = lstm2_forward(xt, pk, True, force, head=(wh2, O2))
        got2 = torch.full_like(want2, float("nan"))
        H.call("nppc_sb_head_finalize", tr["head_partial"], tr["head_partial"].shape[0], b2, got2, N, Tn, la, O2, Fo, s)
        torch.cuda.synchronize()
        assert ops_lstm.coop_timeouts() == 0
        assert float((got2 - want2).abs().max()) < 2e-5 * max(1.0, float(want2.abs().max()))
        for k in ("h1", "h2", "c1", "c2", "g1", "g2"):
            assert torch.equal(tr[k], ref_state[k]), k


@pytest.mark.parametrize("N,Tn,O,Fo,la", [(96, 7, 10, 48, 2), (77, 5, 4, 77, 0), (256, 6, 16, 128, 1)])
def test_fused_head_backward_equals_separate_kernels(N, Tn, O, Fo, la):
    """K-split backward forming d h2 = dY . Wh itself (dY rows by LDS-DMA, two MFMAs per wave and step) against the
    head_bwd_dh kernel + the dh2-reading kernel; ragged last cluster, look-ahead rows, O up to one MFMA tile"""
    from nppc_audio import _hip as H
    from nppc_audio import ops_lstm
    from nppc_audio.ops_lstm import PackedLSTM, PackedLSTMBwd, bwd_head_fusable, lstm2_backward, lstm2_forward
    I, Hd = 34, 384
    P = _weights(I, Hd, 4)
    pre = "sb_model.sequence_model."
    dev = torch.device("cuda")
    names = ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0", "weight_ih_l1", "weight_hh_l1", "bias_ih_l1", "bias_hh_l1")
    pk = PackedLSTM(I, Hd, 0, dev).pack(*[P[pre + n].to(dev) for n in names])
    pb = PackedLSTMBwd(I, Hd, 0, dev).pack(*[P[pre + n].to(dev) for n in ("weight_ih_l0", "weight_hh_l0", "weight_ih_l1", "weight_hh_l1")])
    assert bwd_head_fusable(N, pb, coop=True)
    g = torch.Generator().manual_seed(N + O)
    xt = torch.zeros(Tn, N, pk.kx, dtype=torch.bfloat16, device=dev)
    xt[:, :, :I] = torch.randn(Tn, N, I, generator=g).to(dev)
    saved = lstm2_forward(xt, pk, True, 1)
    Bq = N // Fo
    dout = torch.randn(Bq, O, Fo, Tn - la, generator=g).to(dev)
    wh = (torch.randn(O, Hd, generator=g) * 0.2)
    whT = torch.zeros(Hd, 32, dtype=torch.bfloat16, device=dev)
    whT[:, :O] = wh.t().to(dev)
    s = H.stream()
    dh2 = torch.empty(Tn, N, Hd, dtype=torch.bfloat16, device=dev)
    dW, db = torch.zeros(O, Hd, device=dev), torch.zeros(O, device=dev)
    H.call("nppc_sb_head_bwd", 0, dout, whT, saved["h2"], dh2, dW, db, N, Tn, la, Hd, O, Fo, s)
    a = [t.clone() for t in lstm2_backward(saved, dh2, pb, pk.kx, coop=True)]
    dyt = torch.full((Tn, N, 16), float("nan"), dtype=torch.bfloat16, device=dev)
    H.call("nppc_head_dy_gather", dout, dyt, N, Tn, la, O, Fo, s)
    dW2, db2 = torch.zeros(O, Hd, device=dev), torch.zeros(O, device=dev)
    H.call("nppc_sb_head_bwd_w", 0, dout, saved["h2"], dW2, db2, N, Tn, la, Hd, O, Fo, s)
    b = [t.clone() for t in lstm2_backward(saved, None, pb, pk.kx, coop=True, head=(dyt, whT))]
    torch.cuda.synchronize()
    assert ops_lstm.coop_timeouts() == 0
    # the gathered rows: exact bf16 of dout, zero padding
    want_rows = torch.zeros(Tn, N, 16)
    want_rows[la:, :, :O] = dout.cpu().permute(3, 0, 2, 1).reshape(Tn - la, N, O)
    assert torch.equal(dyt.float().cpu(), want_rows.to(torch.bfloat16).float())
    assert torch.allclose(dW2, dW, rtol=1e-5, atol=1e-5) and torch.allclose(db2, db, rtol=1e-5, atol=1e-5)
    for name, x, y in zip(("dx", "dg1", "dg2"), a, b):
        x, y = x.float(), y.float()
        assert bool(torch.isfinite(y).all()), name
        assert float((x - y).abs().max()) < 2e-2 * float(x.abs().max()) + 1e-6, name     # dh2 rounded to bf16 on one side only


def _production_setup(N, Tn, seed):
    """same weights and inputs packed for the bf16 production kernels and for the fp32 single-workgroup kernels (the
    fp32 kernels are the ones test_lstm_fwd_f32_matches_oracle / test_lstm_bwd_matches_autograd pin to the oracle)"""
    from nppc_audio.ops_lstm import PackedLSTM, PackedLSTMBwd
    I, Hd = 34, 384
    P = _weights(I, Hd, seed)
    pre = "sb_model.sequence_model."
    dev = torch.device("cuda")
    names = ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0", "weight_ih_l1", "weight_hh_l1", "bias_ih_l1", "bias_hh_l1")
    wn = ("weight_ih_l0", "weight_hh_l0", "weight_ih_l1", "weight_hh_l1")
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(Tn, N, I, generator=g)
    out = {}
    for prec in (0, 1):
        pk = PackedLSTM(I, Hd, prec, dev).pack(*[P[pre + n].to(dev) for n in names])
        pb = PackedLSTMBwd(I, Hd, prec, dev).pack(*[P[pre + n].to(dev) for n in wn])
        xt = torch.zeros(Tn, N, pk.kx, dtype=torch.bfloat16 if prec == 0 else torch.float32, device=dev)
        xt[:, :, :I] = x.to(dev)
        out[prec] = (pk, pb, xt)
    return out, g


def test_production_restorer_forward_full_width_against_fp32_kernel():
    """BASELINE C2's restorer launch as the plan really makes it -- N = 8224 sequences, CU pairs with 80-row tiles, the
    XCD-aware pair placement with the grid rounded to a multiple of 8 and its spare workgroups, the 2-output head fused --
    against the product's own oracle-pinned fp32 single-workgroup kernel, 32 steps."""
    import ctypes
    from nppc_audio import _hip as H
    from nppc_audio import ops_lstm
    from nppc_audio.ops_lstm import lstm2_forward
    N, Tn, O, Fo, la, Hd = 8224, 32, 2, 257, 2, 384
    setup, g = _production_setup(N, Tn, 11)
    G, mt, ncl = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    H.call("nppc_lstm2_coop_plan", 0, 0, N, Hd, ops_lstm._n_cu(), ctypes.byref(G), ctypes.byref(mt), ctypes.byref(ncl))
    assert G.value == 2 and ncl.value * 2 > 200, (G.value, mt.value, ncl.value)     # the wide pair plan, most of the chip
    dev = torch.device("cuda")
    wh = (torch.randn(O, Hd, generator=g) * 0.1)
    bias = torch.randn(O, generator=g).to(dev)
    s = H.stream()
    pk32, _, x32 = setup[1]
    ref = lstm2_forward(x32, pk32, False, 1)
    wh32 = torch.zeros(16, Hd, device=dev)
    wh32[:O] = wh.to(dev)
    want = torch.empty(N // Fo, O, Fo, Tn - la, device=dev)
    H.call("nppc_sb_head", 1, ref["h2"], wh32, bias, want, N, Tn, la, Hd, O, Fo, s)
    pk16, _, x16 = setup[0]
    fused = lstm2_forward(x16, pk16, False, None, head=(wh32.to(torch.bfloat16), O))
    assert "head_partial" in fused
    got = torch.full_like(want, float("nan"))
    H.call("nppc_sb_head_finalize", fused["head_partial"], fused["head_partial"].shape[0], bias, got, N, Tn, la, O, Fo, s)
    torch.cuda.synchronize()
    assert ops_lstm.coop_timeouts() == 0
    assert bool(torch.isfinite(got).all())
    # bf16 operands vs exact fp32: the forward tolerance of test_lstm_fwd_bf16_matches_oracle, on the head outputs
    # (|Wh| ~ 0.1 x 384 units of |h| <= 1: the same order as h itself)
    assert float((got - want).abs().max()) < 3e-2 * max(1.0, float(want.abs().max()))


def test_production_direction_net_forward_backward_full_width_against_fp32_kernels():
    """BASELINE C2's direction-net launches as planned -- N = 4096, training forward with saved state and the fused
    10-output head, K-split cooperative backward with the head backward fused -- against the fp32 single-workgroup
    forward + backward kernels, 32 steps: LSTM outputs, dx and both gate-gradient tensors."""
    from nppc_audio import _hip as H
    from nppc_audio import ops_lstm
    from nppc_audio.ops_lstm import bwd_head_fusable, lstm2_backward, lstm2_forward
    N, Tn, O, Fo, la, Hd = 4096, 32, 10, 128, 2, 384
    setup, g = _production_setup(N, Tn, 12)
    dev = torch.device("cuda")
    s = H.stream()
    wh = torch.randn(O, Hd, generator=g) * 0.2
    dout = torch.randn(N // Fo, O, Fo, Tn - la, generator=g).to(dev)
    # fp32 reference path
    pk32, pb32, x32 = setup[1]
    sv32 = lstm2_forward(x32, pk32, True, 1)
    whT32 = torch.zeros(Hd, 32, device=dev)
    whT32[:, :O] = wh.t().to(dev)
    dh2 = torch.empty(Tn, N, Hd, device=dev)
    dW, db = torch.zeros(O, Hd, device=dev), torch.zeros(O, device=dev)
    H.call("nppc_sb_head_bwd", 1, dout, whT32, sv32["h2"], dh2, dW, db, N, Tn, la, Hd, O, Fo, s)
    ref = [t.float().clone() for t in lstm2_backward(sv32, dh2, pb32, pk32.kx, coop=False)]
    h2_ref = sv32["h2"].clone()
    # production path
    pk16, pb16, x16 = setup[0]
    wh16 = torch.zeros(16, Hd, dtype=torch.bfloat16, device=dev)
    wh16[:O] = wh.to(dev)
    sv16 = lstm2_forward(x16, pk16, True, None, head=(wh16, O))
    assert "head_partial" in sv16 and bwd_head_fusable(N, pb16)
    dyt = torch.empty(Tn, N, 16, dtype=torch.bfloat16, device=dev)
    H.call("nppc_head_dy_gather", dout, dyt, N, Tn, la, O, Fo, s)
    got = [t.float() for t in lstm2_backward(sv16, None, pb16, pk16.kx, head=(dyt, whT32.to(torch.bfloat16)))]
    torch.cuda.synchronize()
    assert ops_lstm.coop_timeouts() == 0
    assert float((sv16["h2"].float() - h2_ref).abs().max()) < 3e-2
    R_ = Tn * N
    for name, a, b in zip(("dx", "dg1", "dg2"), got, ref):
        a, b = (a[:R_], b[:R_]) if a.dim() == 2 else (a, b)
        assert bool(torch.isfinite(a).all()), name
        d = float((a - b).abs().max()) / float(b.abs().max())
        assert d < 4e-2, (name, d)           # the backward tolerance of test_lstm_bwd_matches_autograd (bf16)


def test_handoff_timeout_counter_is_sticky():
    """a recorded hand-off time-out must survive later launches on the same flag block (the launchers zero the epoch
    words only) until the host clears it; check_coop_timeouts raises and clears"""
    from nppc_audio import ops_lstm
    from nppc_audio.ops_lstm import lstm2_forward
    setup, _ = _production_setup(96, 5, 13)
    pk16, _, x16 = setup[0]
    lstm2_forward(x16, pk16, False, (2, 2))
    torch.cuda.synchronize()
    ops_lstm.clear_coop_timeouts()
    assert ops_lstm.coop_timeouts() == 0
    blocks = ops_lstm._flag_blocks()
    assert blocks
    blocks[-1][-4] = 7                                   # as if 7 bounded spins had given up in an earlier launch
    for _ in range(3):
        lstm2_forward(x16, pk16, False, (2, 2))          # same workspace key -> same flag block, epoch words re-zeroed
    torch.cuda.synchronize()
    assert ops_lstm.coop_timeouts() == 7
    with pytest.raises(RuntimeError, match="hand-off time-out"):
        ops_lstm.check_coop_timeouts("test")
    assert ops_lstm.coop_timeouts() == 0


@pytest.mark.parametrize("N,Tn,train,O", [(160, 9, False, 2), (320, 7, True, 10), (512, 6, False, 4), (512, 5, True, 16),
                                          (1280, 5, True, 10), (3360, 4, False, 2),
                                          # ragged last chunk (N % 32 != 0, round 4): rows >= N read zeros and are never stored
                                          (514, 6, False, 2), (1030, 5, True, 16), (2056, 4, False, 2)])
def test_weight_stationary_forward_matches_single_workgroup_kernel_and_oracle(N, Tn, train, O):
    """The 12-CU weight-stationary cluster kernel (csrc/lstm_ws.hip: weights in registers, one fused hand-off per time step,
    chunks of 32 sequences walked round robin; 1, 2, 3 (ragged: 6/5/5 chunks), 8 and 21 clusters) against the single-workgroup
    streaming kernel of the same precision and the oracle; head fused; no bounded spin may time out."""
    from nppc_audio import _hip as H
    from nppc_audio import ops_lstm
    from nppc_audio.ops_lstm import PackedLSTM, lstm2_forward, ws_plan
    I, Hd, Fo, la = 34, 384, N // 2, 1
    P = _weights(I, Hd, 7)
    pre = "sb_model.sequence_model."
    dev = torch.device("cuda")
    pk = PackedLSTM(I, Hd, 0, dev).pack(*[P[pre + n].to(dev) for n in (
        "weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0", "weight_ih_l1", "weight_hh_l1", "bias_ih_l1", "bias_hh_l1")])
    assert ws_plan(N, pk) is not None
    g = torch.Generator().manual_seed(N + Tn)
    x = torch.randn(N, Tn, I, generator=g)
    xt = torch.zeros(Tn, N, pk.kx, dtype=torch.bfloat16, device=dev)
    xt[:, :, :I] = x.permute(1, 0, 2).to(dev)
    wh = torch.zeros(16, Hd, dtype=torch.bfloat16, device=dev)
    wh[:O] = (torch.randn(O, Hd, generator=g) * 0.1).to(dev)
    bias = torch.randn(O, generator=g).to(dev)
    s = H.stream()
    ops_lstm.clear_coop_timeouts()
    single = {k: v.clone() for k, v in lstm2_forward(xt, pk, train, 1).items()}
    want = torch.empty(N // Fo, O, Fo, Tn - la, dtype=torch.float32, device=dev)
    H.call("nppc_sb_head", 0, single["h2"], wh, bias, want, N, Tn, la, Hd, O, Fo, s)
    ws = lstm2_forward(xt, pk, train, "ws", head=(wh, O))
    assert "head_partial" in ws and ws["head_partial"].shape[0] == 1
    got = torch.full_like(want, float("nan"))
    H.call("nppc_sb_head_finalize", ws["head_partial"], 1, bias, got, N, Tn, la, O, Fo, s)
    torch.cuda.synchronize()
    assert ops_lstm.coop_timeouts() == 0
    assert bool(torch.isfinite(got).all())
    assert float((got - want).abs().max()) < 2e-2 * max(1.0, float(want.abs().max()))
    if train:
        ref = R.lstm2(x, P, "sb_model.sequence_model")
        assert (ws["h2"].float().cpu().permute(1, 0, 2) - ref).abs().max().item() < 3e-2
        for k in ("h1", "h2", "c1", "c2", "g1", "g2"):
            d = (ws[k].float() - single[k].float()).abs().max().item()
            assert d < 2e-2, (k, d)      # same bf16 arithmetic, different accumulation order
        assert float(ws["h1_rows"][Tn * N:].abs().max()) == 0 and float(ws["h2_rows"][Tn * N:].abs().max()) == 0
