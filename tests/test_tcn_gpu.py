"""GPU: the fused full-band TCN kernels against fp64 torch autograd of the reference's TCNBlock arithmetic
(audio_zen/model/module/causal_conv.py:96-108: conv1x1, PReLU, GroupNorm(1, C, eps 1e-8), depthwise dilated conv k = 3,
PReLU, GroupNorm, sconv, skip) -- the sconv product with GroupNorm-2 folded in (nppc_gemm_nt_gn + nppc_tcn_pack_sconv) and
the one-reduce-one-apply backward of the block's middle (nppc_tcn_mid_bwd)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
EPS = 1e-8


def _middle(pre1, P, dil, Tv):
    """pre1 [Z,B,C,Tv] fp64 -> (y1, y2, a2): the block's middle, per branch z"""
    Z, B, C, _ = pre1.shape
    y1 = torch.where(pre1 > 0, pre1, P["a1"].view(Z, 1, 1, 1) * pre1)
    ys, a2s = [], []
    for z in range(Z):
        z1 = F.group_norm(y1[z], 1, P["g1"][z], P["b1"][z], EPS)
        u = F.conv1d(z1, P["wd"][z].view(C, 1, 3), P["bd"][z], padding=dil, dilation=dil, groups=C)
        y2 = torch.where(u > 0, u, P["a2"][z] * u)
        ys.append(y2)
        a2s.append(F.group_norm(y2, 1, P["g2"][z], P["b2"][z], EPS))
    return y1, torch.stack(ys), torch.stack(a2s)


def _tm(x, Tp, dt):
    """[Z,B,C,Tv] -> time-major zero-padded [Z,B,Tp,C] on the device"""
    Z, B, C, Tv = x.shape
    out = torch.zeros(Z, B, Tp, C, dtype=dt, device="cuda")
    out[:, :, :Tv] = x.permute(0, 1, 3, 2).to(dt).cuda()
    return out


def _stats(x_tm, Tv):
    v = x_tm[:, :, :Tv].double()
    return torch.stack([v.sum(dim=(2, 3)), (v * v).sum(dim=(2, 3))], dim=-1).contiguous()     # [Z,B,2]


@pytest.mark.parametrize("prec,dil,Tv,C", [(1, 1, 37, 64), (1, 9, 70, 128), (0, 5, 130, 192), (1, 2, 5, 64),
                                           (1, 9, 250, 64), (0, 2, 241, 128)])      # > 224 frames: two time chunks + halo rows
def test_mid_block_backward_matches_autograd(prec, dil, Tv, C):
    from nppc_audio import _hip as H
    Z, B, Tp = 3, 2, 256
    dt = H.dtype_of(prec)
    g = torch.Generator().manual_seed(100 * dil + Tv)
    P = {k: (torch.randn(Z, C, generator=g).double() * s + o) for k, s, o in
         (("g1", 0.3, 1.0), ("b1", 0.3, 0.0), ("g2", 0.3, 1.0), ("b2", 0.3, 0.0), ("bd", 0.2, 0.0))}
    P["wd"] = torch.randn(Z, C, 3, generator=g).double() * 0.5
    P["a1"] = torch.tensor([0.25, 0.1, 0.4], dtype=torch.float64)
    P["a2"] = torch.tensor([0.3, 0.2, 0.15], dtype=torch.float64)
    pre1 = torch.randn(Z, B, C, Tv, generator=g).double() * 1.5 + 0.2
    dA = torch.randn(Z, B, C, Tv, generator=g).double()
    # the kernel sees the STORED activations: round them to the storage dtype first, differentiate from there
    y1_0, _, _ = _middle(pre1, P, dil, Tv)
    y1s = y1_0.to(dt).double()
    pre1_eff = torch.where(y1s > 0, y1s, y1s / P["a1"].view(Z, 1, 1, 1)).requires_grad_(True)     # PReLU^-1 of the stored y1
    Pr = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    y1, y2, a2 = _middle(pre1_eff, Pr, dil, Tv)
    (a2 * dA).sum().backward()
    y1_tm, y2_tm, dA_tm = _tm(y1.detach(), Tp, dt), _tm(y2.detach(), Tp, dt), _tm(dA, Tp, dt)
    st1, st2 = _stats(y1_tm, Tv).cuda(), _stats(y2_tm, Tv).cuda()
    S = torch.full((Z, B, C // 64, 8), float("nan"), dtype=torch.float64, device="cuda")      # no initial state needed
    part = torch.empty(H.mid_bwd_part_elems(B, C, Tp, Z), device="cuda")
    flat = lambda t: t.float().contiguous().cuda()
    sP = 4096                                                    # parameter stride between branches (elements)

    def strided(t):                                              # [Z, n] -> flat buffer with branch stride sP
        buf = torch.zeros(Z * sP, device="cuda")
        for z in range(Z):
            buf[z * sP: z * sP + t[z].numel()] = t[z].reshape(-1).float().cuda()
        return buf

    par = {k: strided(v) for k, v in P.items()}
    grads = {k: torch.zeros(Z * sP, device="cuda") for k in ("g2", "b2", "g1", "b1", "wd", "bd", "a1", "a2", "bias1")}
    a2_out = torch.zeros(Z, B, Tp, C, dtype=dt, device="cuda")
    dpre1 = torch.full((Z, B, Tp, C), float("nan"), dtype=dt, device="cuda")
    # optional rider: tile column sums of the block's upstream gradient (what nppc_gemm_nt_colsum leaves) -> sconv bias gradient
    n_tiles, cp_ld, cp_cols = 5, 320, 257
    colpart = torch.randn(Z, n_tiles, cp_ld, generator=g).cuda()
    dbias2 = torch.full((Z * sP,), float("nan"), device="cuda")
    H.call("nppc_tcn_mid_bwd", prec, dA_tm, y2_tm, y1_tm, st1, st2, S, part, par["g1"], par["b1"], par["g2"], par["b2"], par["wd"],
           par["a1"], par["a2"], a2_out, dpre1, grads["g2"], grads["b2"], grads["g1"], grads["b1"], grads["wd"], grads["bd"],
           grads["a1"], grads["a2"], grads["bias1"], colpart, n_tiles, cp_ld, cp_cols, dbias2, B, C, Tp, Tv, dil, EPS, B * Tp * C,
           B * 2, sP, Z, 1, H.stream())
    torch.cuda.synchronize()
    tol = 2e-4 if prec == 1 else 4e-2
    for z in range(Z):
        want_b2 = colpart[z, :, :cp_cols].double().sum(0).cpu()
        assert float((dbias2[z * sP: z * sP + cp_cols].double().cpu() - want_b2).abs().max()) < 1e-5 * float(want_b2.abs().max())
        assert bool(torch.isnan(dbias2[z * sP + cp_cols: (z + 1) * sP]).all())              # nothing written past the valid columns

    def rel(got, ref):
        return float((got.double().cpu() - ref).abs().max() / (ref.abs().max() + 1e-30))

    want_dpre = pre1_eff.grad.permute(0, 1, 3, 2)                                  # [Z,B,Tv,C]
    assert bool(torch.isfinite(dpre1.float()).all())
    assert float(dpre1[:, :, Tv:].float().abs().max()) == 0.0                       # padded frames written as zero
    assert rel(dpre1[:, :, :Tv], want_dpre) < tol
    # a2 from stats of the STORED y2 (what the forward used) vs group_norm of the exact y2: storage rounding only
    assert rel(a2_out[:, :, :Tv], a2.detach().permute(0, 1, 3, 2)) < (1e-5 if prec == 1 else 2e-2)
    for k, n in (("g2", C), ("b2", C), ("g1", C), ("b1", C), ("wd", 3 * C), ("bd", C), ("a1", 1), ("a2", 1)):
        got = torch.stack([grads[k][z * sP: z * sP + n] for z in range(Z)])
        assert rel(got, Pr[k].grad.reshape(Z, n)) < tol, k
    got_b = torch.stack([grads["bias1"][z * sP: z * sP + C] for z in range(Z)])
    assert rel(got_b, pre1_eff.grad.sum(dim=(1, 3))) < tol
    # accumulate semantics: a second launch doubles the parameter gradients -- EXACTLY: no atomics anywhere (round 4), every sum
    # has a fixed order, so the second launch adds bit-identical contributions and x + x = 2 x is exact
    first = {k: v.clone() for k, v in grads.items()}
    H.call("nppc_tcn_mid_bwd", prec, dA_tm, y2_tm, y1_tm, st1, st2, S, part, par["g1"], par["b1"], par["g2"], par["b2"], par["wd"],
           par["a1"], par["a2"], None, dpre1, grads["g2"], grads["b2"], grads["g1"], grads["b1"], grads["wd"], grads["bd"],
           grads["a1"], grads["a2"], grads["bias1"], None, 0, 0, 0, None, B, C, Tp, Tv, dil, EPS, B * Tp * C, B * 2, sP, Z, 1,
           H.stream())
    torch.cuda.synchronize()
    got2 = torch.stack([grads["g1"][z * sP: z * sP + C] for z in range(Z)])
    assert rel(got2, 2 * Pr["g1"].grad) < tol
    for k in grads:
        assert torch.equal(grads[k], 2 * first[k]), k
    # deferred finishing pass (finish_now = 0) over TWO "blocks" at once: the partial rows of the call above in two part
    # buffers, gradients at a block stride sL inside one flat buffer -> both blocks receive exactly the first launch's sums
    n_part = H.mid_bwd_part_elems(B, C, Tp, Z)
    parts = torch.full((2, n_part), float("nan"), device="cuda")
    sL = 600                                                      # block stride (elements) inside the branch stride sP
    assert 3 * C <= sL and 2 * sL <= sP
    g2 = {k: torch.zeros(Z * sP, device="cuda") for k in grads}
    colparts = colpart.unsqueeze(0).repeat(2, 1, 1, 1).contiguous()
    dbias2b = torch.full((Z * sP,), float("nan"), device="cuda")
    for blk in range(2):
        H.call("nppc_tcn_mid_bwd", prec, dA_tm, y2_tm, y1_tm, st1, st2, S, parts[blk], par["g1"], par["b1"], par["g2"], par["b2"],
               par["wd"], par["a1"], par["a2"], None, dpre1, g2["g2"], g2["b2"], g2["g1"], g2["b1"], g2["wd"], g2["bd"],
               g2["a1"], g2["a2"], g2["bias1"], None, 0, 0, 0, None, B, C, Tp, Tv, dil, EPS, B * Tp * C, B * 2, sP, Z, 0, H.stream())
    torch.cuda.synchronize()
    assert all(float(v.abs().max()) == 0.0 for v in g2.values())          # nothing is added before the finishing launch
    H.call("nppc_tcn_mid_bwd_finish", parts, n_part, colparts, Z * n_tiles * cp_ld, n_tiles, cp_ld, cp_cols, g2["g2"], g2["b2"],
           g2["g1"], g2["b1"], g2["wd"], g2["bd"], g2["a1"], g2["a2"], g2["bias1"], dbias2b, B, C, sP, sL, Z, 2, H.stream())
    torch.cuda.synchronize()
    for k, n in (("g2", C), ("b2", C), ("g1", C), ("b1", C), ("wd", 3 * C), ("bd", C), ("a1", 1), ("a2", 1), ("bias1", C)):
        for z in range(Z):
            for blk in range(2):
                assert torch.equal(g2[k][z * sP + blk * sL: z * sP + blk * sL + n], first[k][z * sP: z * sP + n]), (k, z, blk)
    for z in range(Z):
        for blk in range(2):
            assert torch.equal(dbias2b[z * sP + blk * sL: z * sP + blk * sL + cp_cols], dbias2[z * sP: z * sP + cp_cols])


@pytest.mark.parametrize("prec,C,Tv", [(1, 257, 70), (0, 514, 130)])
def test_sconv_with_folded_groupnorm_matches_reference_arithmetic(prec, C, Tv):
    """out = sconv(GN2(y2)) + x  (causal_conv.py:104-108) with the normalisation folded into the weights / epilogue"""
    from nppc_audio import _hip as H
    Z, B, K, Tp = 3, 2, 512, 128 if Tv <= 128 else 256
    ldC = (C + 63) // 64 * 64
    dt = H.dtype_of(prec)
    g = torch.Generator().manual_seed(C + Tv)
    NA, lay, brs = 2, 400_000, 1_000_000                         # two "blocks" x three branches at constant strides
    n_flat = NA * lay + Z * brs
    W, gam, bet, bias = (torch.zeros(n_flat) for _ in range(4))
    ref_p = {}
    for a in range(NA):
        for z in range(Z):
            o = a * lay + z * brs
            w = torch.randn(C, K, generator=g) * 0.05
            ga, be, bi = torch.randn(K, generator=g) * 0.3 + 1, torch.randn(K, generator=g) * 0.3, torch.randn(C, generator=g) * 0.2
            W[o:o + C * K], gam[o:o + K], bet[o:o + K], bias[o:o + C] = w.reshape(-1), ga, be, bi
            ref_p[a, z] = (w.double(), ga.double(), be.double(), bi.double())
    Wg = torch.full((NA, Z, ldC, K), 3.0, dtype=dt, device="cuda")
    u = torch.full((NA, Z, ldC), 3.0, device="cuda")
    v = torch.full((NA, Z, ldC), 3.0, device="cuda")
    H.call("nppc_tcn_pack_sconv", prec, W.cuda(), gam.cuda(), bet.cuda(), bias.cuda(), Wg, u, v, C, K, ldC, K, NA, Z, lay, brs,
           Wg.stride(0), Wg.stride(1), H.stream())
    a = 1
    y2 = torch.randn(Z, B, K, Tv, generator=g).double() * 1.3 + 0.4
    x = torch.randn(Z, B, C, Tv, generator=g).double()
    y2_tm = torch.zeros(Z, B, Tp, K, dtype=dt, device="cuda")
    y2_tm[:, :, :Tv] = y2.permute(0, 1, 3, 2).to(dt).cuda()
    x_tm = torch.zeros(Z, B, Tp, ldC, dtype=dt, device="cuda")
    x_tm[:, :, :Tv, :C] = x.permute(0, 1, 3, 2).to(dt).cuda()
    st = _stats(y2_tm, Tv).cuda()
    out = torch.full((Z, B, Tp, ldC), float("nan"), dtype=dt, device="cuda")
    R = B * Tp
    H.call("nppc_gemm_nt_gn", prec, y2_tm, K, R * K, Wg[a], K, ldC * K, out, ldC, R * ldC, u[a], v[a], ldC, x_tm, ldC, R * ldC, st,
           B * 2, float(K * Tv), EPS, R, ldC, K, Tp, Tv, C, Z, H.stream())
    torch.cuda.synchronize()
    assert float(Wg[a, :, C:].float().abs().max()) == 0.0 and float(u[a, :, C:].abs().max()) == 0.0
    for z in range(Z):
        w, ga, be, bi = ref_p[a, z]
        ys = y2_tm[z, :, :Tv].double().cpu().permute(0, 2, 1)                       # stored activations [B,K,Tv]
        ref = F.conv1d(F.group_norm(ys, 1, ga, be, EPS), w.view(C, K, 1), bi) + x_tm[z, :, :Tv, :C].double().cpu().permute(0, 2, 1)
        got = out[z, :, :Tv, :C].double().cpu().permute(0, 2, 1)
        assert float((got - ref).abs().max()) < (2e-5 if prec == 1 else 3e-2) * float(ref.abs().max()), z
    assert float(out[:, :, Tv:].float().abs().max()) == 0.0 and float(out[:, :, :, C:].float().abs().max()) == 0.0


@pytest.mark.parametrize("nm,prec", [(1, 1), (2, 0)])
def test_batched_attention_front_equals_per_map_launches(nm, prec):
    """nppc_tsse_fwd_maps / nppc_tsse_bwd_maps (all 3 * nm input maps of a net in 3 / 4 launches) against the per-map entry
    points, which tests/test_forward_gpu.py and the whole-step tests pin to the reference (attention_model.py:43-98)"""
    from nppc_audio import _hip as H
    B, C, T, la = 3, 33, 40, 2
    Tp, ld, C2 = 128, 64 * nm + 64, C // 2
    dt = H.dtype_of(prec)
    g = torch.Generator().manual_seed(7 + nm)
    sW = 5000                                                     # parameter stride between the three attention layers
    ks = (3, 5, 10)
    shapes = dict(cw0=C * ks[0], cb0=C, cw1=C * ks[1], cb1=C, cw2=C * ks[2], cb2=C, fcw=3, fcb=1, w1=C2 * C, b1=C2, w2=C * C2, b2=C)
    offs, o = {}, 0
    for k, n in shapes.items():
        offs[k] = o
        o += (n + 3) // 4 * 4
    assert o < sW
    flat = (torch.randn(3 * sW, generator=g) * 0.3).cuda()
    P = lambda k, z=0: flat[z * sW + offs[k]:]
    maps = [(torch.randn(B, C, T, generator=g) * (1.0 if j % 3 == 0 else 0.5) + (0.8 if j % 3 == 0 else 0.01)).cuda().contiguous()
            for j in range(3 * nm)]
    s = H.stream()
    # ---- batched
    rs = torch.empty(3 * nm, B, C, dtype=torch.float64, device="cuda")
    scale = torch.empty(3, nm, B, C, device="cuda")
    sv = {k: torch.empty(3, nm, B, *shp, device="cuda") for k, shp in (("ns", ()), ("pre", (C, 3)), ("sq", (C,)), ("h1", (C2,)), ("sg", (C,)))}
    X0 = torch.zeros(3, B, Tp, ld, dtype=dt, device="cuda")
    H.call("nppc_tsse_fwd_maps", prec, H.ptr_array(maps), 3 * nm, rs, P("cw0"), P("cb0"), P("cw1"), P("cb1"), P("cw2"), P("cb2"), *ks,
           P("fcw"), P("fcb"), P("w1"), P("b1"), P("w2"), P("b2"), sW, scale, sv["ns"], sv["pre"], sv["sq"], sv["h1"], sv["sg"], X0,
           B * Tp * ld, B, C, T, la, Tp, ld, s)
    # ---- per map
    X0r = torch.zeros_like(X0)
    for z in range(3):
        for m in range(nm):
            x = maps[m * 3 + z]
            rs1 = torch.empty(B, C, dtype=torch.float64, device="cuda")
            H.call("nppc_rowsum", x, rs1, B * C, T, s)
            sc1 = torch.empty(B, C, device="cuda")
            one = {k: torch.empty_like(v[z, m]) for k, v in sv.items()}
            H.call("nppc_tsse_fwd", x, rs1, P("cw0", z), P("cb0", z), P("cw1", z), P("cb1", z), P("cw2", z), P("cb2", z), *ks,
                   P("fcw", z), P("fcb", z), P("w1", z), P("b1", z), P("w2", z), P("b2", z), sc1, one["ns"], one["pre"], one["sq"],
                   one["h1"], one["sg"], B, C, T, la, s)
            H.call("nppc_scale_transpose", prec, x, sc1, X0r[z], B, C, T, Tp, ld, m * C, s)
            torch.cuda.synchronize()
            assert torch.equal(rs[m * 3 + z], rs1) and torch.equal(scale[z, m], sc1), (z, m)
            for k in sv:
                assert torch.equal(sv[k][z, m], one[k]), (k, z, m)
    assert torch.equal(X0, X0r)
    # ---- backward
    dX0 = (torch.randn(3, B, Tp, ld, generator=g) * 0.1).to(dt).cuda()
    import ctypes
    nW = ctypes.c_long()
    H.call("nppc_tsse_bwd_ws_elems", 3 * nm, B, C, *ks, ctypes.byref(nW))
    ws = torch.full((nW.value,), float("nan"), device="cuda")       # nothing may rely on a cleared workspace
    G = torch.zeros_like(flat)
    Gp = lambda k, z=0: G[z * sW + offs[k]:]
    H.call("nppc_tsse_bwd_maps", prec, dX0, B * Tp * ld, H.ptr_array(maps), 3 * nm, rs, P("cw0"), P("cw1"), P("cw2"), *ks, P("fcw"),
           P("w1"), P("w2"), sW, sv["ns"], sv["pre"], sv["sq"], sv["h1"], sv["sg"], ws, Gp("cw0"), Gp("cb0"), Gp("cw1"), Gp("cb1"),
           Gp("cw2"), Gp("cb2"), Gp("fcw"), Gp("fcb"), Gp("w1"), Gp("b1"), Gp("w2"), Gp("b2"), B, C, T, la, Tp, ld, s)
    Gr = torch.zeros_like(flat)
    Gq = lambda k, z: Gr[z * sW + offs[k]:]
    H.call("nppc_tsse_bwd_ws_elems", 1, B, C, *ks, ctypes.byref(nW))
    ws1 = torch.full((nW.value,), float("nan"), device="cuda")
    for z in range(3):
        for m in range(nm):
            H.call("nppc_tsse_bwd", prec, dX0[z], maps[m * 3 + z], rs[m * 3 + z], P("cw0", z), P("cw1", z), P("cw2", z), *ks,
                   P("fcw", z), P("w1", z), P("w2", z), sv["ns"][z, m], sv["pre"][z, m], sv["sq"][z, m], sv["h1"][z, m],
                   sv["sg"][z, m], ws1, Gq("cw0", z), Gq("cb0", z), Gq("cw1", z), Gq("cb1", z), Gq("cw2", z), Gq("cb2", z),
                   Gq("fcw", z), Gq("fcb", z), Gq("w1", z), Gq("b1", z), Gq("w2", z), Gq("b2", z), B, C, T, la, Tp, ld, m * C, s)
    torch.cuda.synchronize()
    assert float(Gr.abs().max()) > 0
    # no atomics (round 4): every gradient element has one writer, samples and maps are added in index order on both paths
    assert torch.equal(G, Gr)
    # ... so a second batched launch on the same inputs reproduces the first bit for bit
    G2 = torch.zeros_like(flat)
    Gp2 = lambda k, z=0: G2[z * sW + offs[k]:]
    H.call("nppc_tsse_bwd_maps", prec, dX0, B * Tp * ld, H.ptr_array(maps), 3 * nm, rs, P("cw0"), P("cw1"), P("cw2"), *ks, P("fcw"),
           P("w1"), P("w2"), sW, sv["ns"], sv["pre"], sv["sq"], sv["h1"], sv["sg"], ws, Gp2("cw0"), Gp2("cb0"), Gp2("cw1"), Gp2("cb1"),
           Gp2("cw2"), Gp2("cb2"), Gp2("fcw"), Gp2("fcb"), Gp2("w1"), Gp2("b1"), Gp2("w2"), Gp2("b2"), B, C, T, la, Tp, ld, s)
    torch.cuda.synchronize()
    assert torch.equal(G, G2)


@pytest.mark.parametrize("prec,epi,N,K", [(0, 2, 576, 512), (0, 5, 576, 320), (1, 2, 128, 96), (0, 2, 512, 128)])
def test_gemm_epilogue_leaves_tile_column_sums_of_the_output(prec, epi, N, K):
    """nppc_gemm_nt_colsum: the same output as nppc_gemm_nt (EPI_RESIDUAL = 2 / EPI_MASK_POS = 5) plus the column sums of every
    128-row tile of the STORED output -- the bias gradient of the next 1x1 convolution down the backward chain
    (causal_conv.py:107) without a pass of its own over that tensor"""
    from nppc_audio import _hip as H
    Z, B, Tp, Tv, Nv = 3, 2, 256, 200, N - 7
    R = B * Tp
    dt = H.dtype_of(prec)
    g = torch.Generator().manual_seed(N + K + epi)
    A = torch.randn(Z, R, K, generator=g).to(dt).cuda()
    W = (torch.randn(Z, N, K, generator=g) * 0.1).to(dt).cuda()
    res = torch.randn(Z, R, N, generator=g).to(dt).cuda()
    out0 = torch.full((Z, R, N), float("nan"), dtype=dt, device="cuda")
    out1 = torch.full((Z, R, N), float("nan"), dtype=dt, device="cuda")
    cp = torch.full((Z, R // 128, N), float("nan"), device="cuda")
    H.call("nppc_gemm_nt", prec, epi, A, K, R * K, W, K, N * K, out0, N, R * N, None, 0, res, N, R * N, None, 0, None, 0, R, N, K,
           Tp, Tv, Nv, 0, Z, 1, H.stream())
    H.call("nppc_gemm_nt_colsum", prec, epi, A, K, R * K, W, K, N * K, out1, N, R * N, None, 0, res, N, R * N, R, N, K, Tp, Tv, Nv,
           Z, cp, H.stream())
    torch.cuda.synchronize()
    assert torch.equal(out0, out1)
    want = out1.double().view(Z, R // 128, 128, N).sum(2).cpu()
    assert bool(torch.isfinite(cp).all())
    assert float((cp.double().cpu() - want).abs().max()) < 1e-5 * float(want.abs().max())
    if prec == 1:
        return
    with pytest.raises(RuntimeError, match="unsupported"):               # K not a multiple of the bf16 LDS stage: no silent fallback
        H.call("nppc_gemm_nt_colsum", prec, epi, A, K, R * K, W, K, N * K, out1, N, R * N, None, 0, res, N, R * N, R, N, 32, Tp, Tv,
               Nv, Z, cp, H.stream())


@pytest.mark.parametrize("prec,epi,N,K", [(0, 0, 512, 320), (0, 1, 512, 320), (0, 3, 320, 320), (0, 1, 320, 128), (1, 1, 128, 96), (1, 0, 192, 64)])
def test_coalesced_plain_prelu_relu_epilogues_equal_the_accumulator_layout_ones(prec, epi, N, K, monkeypatch):
    """csrc/tcn.hip gemm_nt_lds_kernel, round 4: EPI_PLAIN (0) / EPI_PRELU_STATS (1) / EPI_RELU (3) leave through the LDS tile
    image as 16-byte row chunks (NPPC_NT_STAGED_PLAIN, default on) -- the same values, masks and GroupNorm statistics, bit for
    bit, as the accumulator-layout epilogue (64 two-byte stores per lane), and both equal the torch product."""
    from nppc_audio import _hip as H
    Z, B, Tp, Tv, Nv = 3, 2, 256, 200, N - 7
    R = B * Tp
    dt = H.dtype_of(prec)
    g = torch.Generator().manual_seed(N + K + epi)
    A = torch.randn(Z, R, K, generator=g).to(dt).cuda()
    W = (torch.randn(Z, N, K, generator=g) * 0.1).to(dt).cuda()
    bias = (torch.randn(Z, N, generator=g) * 0.1).cuda()
    slope = torch.tensor([0.25, 0.1, -0.3]).cuda()
    outs, stats = [], []
    for sw in ("0", "1"):
        monkeypatch.setenv("NPPC_NT_STAGED_PLAIN", sw)
        out = torch.full((Z, R, N), float("nan"), dtype=dt, device="cuda")
        st = torch.zeros(Z, B, 2, dtype=torch.float64, device="cuda")
        H.call("nppc_gemm_nt", prec, epi, A, K, R * K, W, K, N * K, out, N, R * N, bias, N, None, 0, 0, slope if epi == 1 else None,
               1, st if epi == 1 else None, B * 2, R, N, K, Tp, Tv, Nv, 0, Z, 1, H.stream())
        torch.cuda.synchronize()
        outs.append(out)
        stats.append(st)
    assert torch.equal(outs[0], outs[1])
    if prec == 0:
        assert torch.equal(stats[0], stats[1])     # (sums of bf16 values and of their squares are exact in fp64: order-free)
    else:                                          # fp32 keeps the accumulator-layout epilogue (the switch changes nothing); the
        # workgroups of a sample add their fp64 partials in arrival order
        assert float(((stats[0] - stats[1]) / stats[1].abs().clamp_min(1.0)).abs().max()) < 1e-12
    ref = torch.einsum("zrk,znk->zrn", A.double(), W.double()) + bias.double()[:, None, :]
    if epi == 1:
        ref = torch.where(ref > 0, ref, slope.double()[:, None, None] * ref)
    if epi == 3:
        ref = ref.clamp_min(0)
    ref = ref.view(Z, B, Tp, N)
    ref[:, :, Tv:] = 0
    ref[..., Nv:] = 0
    got = outs[1].double().view(Z, B, Tp, N)
    assert float((got - ref).abs().max()) < (2e-2 if prec == 0 else 2e-5) * float(ref.abs().max())
    if epi == 1:
        want = torch.stack([got.sum((2, 3)), (got * got).sum((2, 3))], -1)
        assert float(((stats[1] - want) / want.abs().clamp_min(1.0)).abs().max()) < 1e-5
