"""GPU: the split-K weight-gradient GEMMs (NT on transposed operands, TN on row-major operands via transposed LDS reads)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,N,R,S", [(128, 64, 256, 1), (256, 128, 1024, 4), (1536, 384, 2048, 2), (128, 192, 640, 2)])
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
