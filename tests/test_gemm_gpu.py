"""GPU: the split-K weight-gradient GEMMs (NT on transposed operands, TN on row-major operands via transposed LDS reads)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


# the last three shapes are large enough (>= 256 workgroups of 256 x 128) for the LDS-DMA ring kernel, one of them with a
# K slice shorter than the ring (2 stages) and one with a slice that is not a multiple of the ring depth
@pytest.mark.parametrize("M,N,R,S", [(128, 64, 256, 1), (256, 128, 1024, 4), (1536, 384, 2048, 2), (128, 192, 640, 2),
                                     (1536, 384, 8192, 64), (512, 256, 8192, 128), (1536, 384, 64 * 416, 16)])
def test_gemm_tn_splitk_matches_matmul(M, N, R, S):
    from nppc_audio import _hip as H
    g = torch.Generator().manual_seed(M + N)
    A = torch.randn(R, M, generator=g).cuda().to(torch.bfloat16)
    B = torch.randn(R, N, generator=g).cuda().to(torch.bfloat16)
    C = torch.empty(S, M, N, dtype=torch.float32, device="cuda")
    H.call("nppc_gemm_tn_splitk", A, M, B, N, C, N, M, N, R, S, H.stream())
    torch.cuda.synchronize()
    ref = A.float().t().cpu().double() @ B.float().cpu().double()
    got = C.sum(0).cpu().double()
    assert (got - ref).abs().max().item() < 2e-3 * ref.abs().max().item()
    # an asymmetric check of the fragment maps: A = shifted identity pattern
    A2 = torch.zeros(R, M)
    for r in range(min(R, M)):
        A2[r, (r * 7 + 3) % M] = 1.0
    A2 = A2.cuda().to(torch.bfloat16)
    H.call("nppc_gemm_tn_splitk", A2, M, B, N, C, N, M, N, R, S, H.stream())
    ref2 = A2.float().t().cpu().double() @ B.float().cpu().double()
    assert (C.sum(0).cpu().double() - ref2).abs().max().item() < 1e-6


@pytest.mark.parametrize("prec", [0, 1])
def test_gemm_nt_splitk_matches_matmul(prec):
    from nppc_audio import _hip as H
    M, N, K, S = 256, 128, 2048, 4
    dt = H.dtype_of(prec)
    g = torch.Generator().manual_seed(5)
    A = torch.randn(M, K, generator=g).cuda().to(dt)
    B = torch.randn(N, K, generator=g).cuda().to(dt)
    C = torch.empty(S, M, N, dtype=torch.float32, device="cuda")
    H.call("nppc_gemm_nt_splitk", prec, A, K, B, K, C, N, M, N, K, S, H.stream())
    torch.cuda.synchronize()
    ref = A.float().cpu().double() @ B.float().cpu().double().t()
    assert (C.sum(0).cpu().double() - ref).abs().max().item() < (2e-3 if prec == 0 else 1e-4) * ref.abs().max().item()


@pytest.mark.parametrize("rows,cols,ld,batch", [(8192, 512, 512, 3), (8192, 257, 320, 3), (200000, 64, 64, 1),
                                                (1000, 100, 100, 2), (77, 24, 24, 1)])
@pytest.mark.parametrize("prec", [0, 1])
def test_colsum_variants_match_torch(rows, cols, ld, batch, prec):
    """bias gradients: the 16-byte path (padded rows, few large workgroups for tall matrices) and the scalar fallback"""
    from nppc_audio import _hip as H
    dt = H.dtype_of(prec)
    g = torch.Generator().manual_seed(rows + cols)
    M = torch.zeros(batch, rows, ld, dtype=dt, device="cuda")
    M[:, :, :cols] = torch.randn(batch, rows, cols, generator=g).to(dt).cuda()
    out = torch.full((batch, cols + 3), 0.5, dtype=torch.float32, device="cuda")      # accumulates; neighbours untouched
    scratch = lambda n: torch.full((n,), float("nan"), dtype=torch.float32, device="cuda")
    H.colsum(prec, M, out, rows, cols, ld, rows * ld, cols + 3, batch, scratch, H.stream())
    want = M[:, :, :cols].double().sum(dim=1).cpu() + 0.5
    got = out.cpu().double()
    assert float((got[:, :cols] - want).abs().max()) < 2e-4 * (rows ** 0.5) + 1e-3
    assert float((got[:, cols:] - 0.5).abs().max()) == 0.0
    # no atomics: a second launch reproduces the first bit for bit
    out2 = torch.full_like(out, 0.5)
    H.colsum(prec, M, out2, rows, cols, ld, rows * ld, cols + 3, batch, scratch, H.stream())
    assert torch.equal(out, out2)
    with pytest.raises(RuntimeError, match="bad argument"):            # a scratch that is too small is refused, not overrun
        H.call("nppc_colsum", prec, M, out2, rows, cols, ld, rows * ld, cols + 3, batch, torch.empty(1, device="cuda"), 1, H.stream())


def test_pack_matrix_batched_equals_per_matrix_packs():
    from nppc_audio import _hip as H
    g = torch.Generator().manual_seed(1)
    na, nb, N, K, Npad, ldd = 4, 3, 37, 50, 48, 64
    lay, brs = 3000, 20000                                   # element strides of the (a, b) grid inside one flat buffer
    flat = torch.randn(na * lay + nb * brs + N * K, generator=g).cuda()
    for tr in (0, 1):
        shape = (N, K) if not tr else (K, N)                 # transpose: src is [K][N], dst[n][k] = src[k][n]
        dst = torch.full((na, nb, Npad, ldd), 7.0, dtype=torch.bfloat16, device="cuda")
        H.call("nppc_pack_matrix_batched", 0, flat, dst, N, K, Npad, ldd, tr, na, nb, lay, brs, dst.stride(0), dst.stride(1),
               H.stream())
        for a in range(na):
            for b in range(nb):
                one = torch.full((Npad, ldd), 7.0, dtype=torch.bfloat16, device="cuda")
                H.call("nppc_pack_matrix", 0, flat[a * lay + b * brs:], one, N, K, Npad, ldd, tr, H.stream())
                assert torch.equal(dst[a, b], one), (tr, a, b)
                src = flat[a * lay + b * brs: a * lay + b * brs + N * K].reshape(shape)
                ref = (src if not tr else src.t()).to(torch.bfloat16)
                assert torch.equal(one[:N, :K], ref) and float(one[N:].abs().max()) == 0.0


def test_gemm_tn_rowsum_is_the_column_sum_of_the_a_operand():
    """nppc_gemm_tn_splitk_rowsum: the product of the LDS-DMA kernel plus sum_r A[r][m] per K slice (the LSTM bias gradient
    that rides on a weight-gradient product); unsupported shapes are refused, not silently computed without the sums"""
    from nppc_audio import _hip as H
    M, N, R, S = 1536, 384, 64 * 64 * 3, 64
    g = torch.Generator().manual_seed(3)
    A = (torch.randn(R, M, generator=g) * 0.5).cuda().to(torch.bfloat16)
    B = torch.randn(R, N, generator=g).cuda().to(torch.bfloat16)
    C = torch.empty(S, M, N, dtype=torch.float32, device="cuda")
    rs = torch.full((S, M), float("nan"), device="cuda")
    H.call("nppc_gemm_tn_splitk_rowsum", A, M, B, N, C, N, M, N, R, S, rs, H.stream())
    torch.cuda.synchronize()
    ref = A.float().t().cpu().double() @ B.float().cpu().double()
    assert (C.sum(0).cpu().double() - ref).abs().max().item() < 2e-3 * ref.abs().max().item()
    want = A.float().view(S, R // S, M).double().sum(dim=1).cpu()
    assert bool(torch.isfinite(rs).all())
    assert float((rs.cpu().double() - want).abs().max()) < 1e-3 * float(want.abs().max())
    with pytest.raises(RuntimeError, match="unsupported"):
        H.call("nppc_gemm_tn_splitk_rowsum", A, M, B, N, C, N, 128, 64, 1024, 2, rs, H.stream())


@pytest.mark.parametrize("M", [1536, 1280])       # 384-row tiles (M % 384 == 0) and the 256-row fallback geometry
@pytest.mark.parametrize("N2", [64, 384])
def test_gemm_tn_two_sources_share_one_pass_over_a(N2, M):
    """nppc_gemm_tn_splitk2: [A^T . B1 | A^T . B2] in one launch (the two weight gradients of an LSTM layer behind one pass over
    its gate gradients): 192-column tiles from B1, then 192-column tiles (N2 = 384) or ONE 64-column tile (N2 = 64: the
    layer-1 input rows) from B2, slab row stride N1 + N2, optional row sums of A; B1 may start before its buffer's live rows
    (the zero guard rows that stand for h_{-1}): here a shifted view of a larger buffer"""
    from nppc_audio import _hip as H
    N1, S = 384, 64
    R, shift = 64 * S * 2, 96
    g = torch.Generator().manual_seed(5 + N2)
    A = (torch.randn(R, M, generator=g) * 0.5).cuda().to(torch.bfloat16)
    B1buf = torch.randn(R + shift, N1, generator=g).cuda().to(torch.bfloat16)
    B1buf[:shift] = 0
    B2 = torch.randn(R, N2, generator=g).cuda().to(torch.bfloat16)
    ldc = N1 + N2
    C = torch.full((S, M, ldc), float("nan"), dtype=torch.float32, device="cuda")
    rs = torch.full((S, M), float("nan"), device="cuda")
    for B1, rsum in ((B1buf[shift:], rs), (B1buf, None)):           # un-shifted with row sums; guard view without
        C.fill_(float("nan"))
        H.call("nppc_gemm_tn_splitk2", A, M, B1, N1, N1, B2, N2, N2, C, ldc, M, R, S, rsum, H.stream())
        torch.cuda.synchronize()
        got = C.sum(0).cpu().double()
        Ad = A.float().cpu().double()
        ref = torch.cat([Ad.t() @ B1[:R].float().cpu().double(), Ad.t() @ B2.float().cpu().double()], dim=1)
        assert bool(torch.isfinite(got).all())
        assert (got - ref).abs().max().item() < 2e-3 * ref.abs().max().item()
    want = A.float().view(S, R // S, M).double().sum(dim=1).cpu()
    assert float((rs.cpu().double() - want).abs().max()) < 1e-3 * float(want.abs().max())
    with pytest.raises(RuntimeError, match="unsupported"):
        H.call("nppc_gemm_tn_splitk2", A, M, B1buf, N1, 128, B2, N2, N2, C, ldc, M, R, S, None, H.stream())
