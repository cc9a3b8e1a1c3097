"""GPU parity of the inpainting sibling: every U-Net kernel of csrc/unet.hip through
the C ABI against torch-CPU fp32, the whole step against the reference goldens and the fp64 oracle."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import inpaint_ref as R
from oracle import weights as W

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PRECS = [("fp32", 1, torch.float32, 2e-5), ("bf16", 0, torch.bfloat16, 3e-2)]


def load(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    meta = json.load(open(os.path.join(GOLD, name + ".json")))
    return z, meta


def rel(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


class Halo:
    """test-side staging of NCHW tensors into the haloed NHWC layout of csrc/unet.hip"""

    def __init__(self, B, H, W, ld, dtype, guard_after=4096):
        self.B, self.H, self.W, self.ld = B, H, W, ld
        self.P = B * (H + 2) * (W + 2)
        gb = W + 4
        self.store = torch.zeros((gb + self.P + guard_after + W + 4) * ld, dtype=dtype, device="cuda")
        self.t = self.store[gb * ld:]

    def put(self, x, coff=0):           # x [B,C,H,W] cpu fp32
        v = self.t[: self.P * self.ld].view(self.B, self.H + 2, self.W + 2, self.ld)
        v[:, 1:-1, 1:-1, coff:coff + x.shape[1]] = x.permute(0, 2, 3, 1).to(v.dtype).cuda()
        return self

    def get(self, C, coff=0):           # -> [B,C,H,W] cpu fp32 (interior)
        v = self.t[: self.P * self.ld].view(self.B, self.H + 2, self.W + 2, self.ld)
        return v[:, 1:-1, 1:-1, coff:coff + C].float().permute(0, 3, 1, 2).cpu().contiguous()

    def halo_is_zero(self):
        v = self.t[: self.P * self.ld].view(self.B, self.H + 2, self.W + 2, self.ld).float()
        return float(v[:, 0].abs().max() + v[:, -1].abs().max() + v[:, :, 0].abs().max() + v[:, :, -1].abs().max()) == 0.0


def q(x, dtype):
    """round to the storage dtype (the kernels see bf16-rounded operands in bf16 mode)"""
    return x.to(dtype).float()


@pytest.mark.parametrize("B,H,W,Cin,Cout,ks", [(2, 9, 13, 64, 64, 3), (1, 6, 37, 128, 192, 3), (2, 9, 13, 64, 128, 3), (1, 5, 21, 256, 256, 3),
                                               (2, 4, 6, 128, 128, 1), (3, 40, 37, 64, 64, 3), (1, 30, 61, 128, 100, 3),
                                               (4, 126, 126, 64, 64, 3)])
@pytest.mark.parametrize("mode", ["1"])
def test_conv_lds_dma_ring_kernel_equals_register_staged_kernel(B, H, W, Cin, Cout, ks, mode, monkeypatch):
    """csrc/unet.hip conv_dma_kernel (bf16: 256-row tiles, LDS-DMA ring with hand-counted waits) against conv_tiled_kernel:
    the same products accumulated in the same order -> bit-identical outputs and per-tile statistics, on ragged pixel counts
    (last 256-row tile partly / wholly behind P), one to nine taps, K of 1 to 36 stages, both tile widths, folded BatchNorm."""
    from nppc_audio import _hip as Hh
    prec, dtype = Hh.PREC_BF16, torch.bfloat16
    g = torch.Generator().manual_seed(B * 100 + Cin + Cout + H)
    x = q(torch.randn(B, Cin, H, W, generator=g), dtype)
    w = torch.randn(Cout, Cin, ks, ks, generator=g) / np.sqrt(Cin * ks * ks)
    b = torch.randn(Cout, generator=g) * 0.1
    sc, sh = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.1
    Np = (Cout + 63) // 64 * 64
    nt = ks * ks
    wf = torch.empty(Np * nt * Cin, dtype=dtype, device="cuda")
    wb = torch.empty(Cin * nt * Np, dtype=dtype, device="cuda")
    s = Hh.stream()
    Hh.call("nppc_conv_pack", prec, w.cuda(), wf, wb, Cout, Cin, ks, Np, Cin, Cin, Np, s)
    X = Halo(B, H, W, Cin, dtype).put(x)
    ntiles = (X.P + 127) // 128
    out = {}
    for name, min_tiles in (("tiled", "1000000000"), ("dma", "1")):
        monkeypatch.setenv("NPPC_CONV_DMA", mode)          # the ring kernel for every tile width
        monkeypatch.setenv("NPPC_CONV_DMA_MIN_TILES", min_tiles)
        Y, Y2 = Halo(B, H, W, Np, dtype), Halo(B, H, W, Np, dtype)
        part = torch.full((ntiles * 2 * Np,), float("nan"), dtype=torch.float32, device="cuda")
        Hh.call("nppc_conv_fwd_stats", prec, X.t, Cin, wf, Y.t, Np, b.cuda(), B, H, W, Cin, Cout, Np, ks, part, s)
        Hh.call("nppc_conv_fwd", prec, X.t, Cin, wf, Y2.t, Np, b.cuda(), sc.cuda(), sh.cuda(), 0.2, B, H, W, Cin, Cout, Np, ks, s)
        torch.cuda.synchronize()
        assert Y.halo_is_zero() and Y2.halo_is_zero()
        out[name] = (Y, Y2, part)
    ref = F.conv2d(x, q(w, dtype), b, padding=ks // 2)
    assert rel(out["dma"][0].get(Cout), ref) < 8e-3
    assert torch.equal(out["dma"][0].t, out["tiled"][0].t)
    assert torch.equal(out["dma"][1].t, out["tiled"][1].t)
    assert torch.equal(out["dma"][2], out["tiled"][2])            # (no NaN left: every tile of every column was written)
    assert not torch.isnan(out["dma"][2]).any()


@pytest.mark.parametrize("pname,prec,dtype,tol", PRECS)
@pytest.mark.parametrize("B,H,W,Cin,Cout,ks", [(2, 9, 13, 64, 64, 3), (1, 6, 37, 128, 192, 3), (3, 5, 7, 64, 5, 1),
                                               # Np % 128 == 0: the 128-column tile (round 4: it had no unit test), long K
                                               (2, 9, 13, 64, 128, 3), (1, 5, 21, 256, 256, 3), (2, 4, 6, 128, 128, 1)])
def test_conv_forward_input_gradient_weight_gradient(pname, prec, dtype, tol, B, H, W, Cin, Cout, ks):
    from nppc_audio import _hip as Hh
    g = torch.Generator().manual_seed(B * 100 + Cin + Cout)
    x = q(torch.randn(B, Cin, H, W, generator=g), dtype)
    w = torch.randn(Cout, Cin, ks, ks, generator=g) / np.sqrt(Cin * ks * ks)
    b = torch.randn(Cout, generator=g) * 0.1
    Np = (Cout + 63) // 64 * 64
    nt = ks * ks
    wf = torch.empty(Np * nt * Cin, dtype=dtype, device="cuda")
    wb = torch.empty(Cin * nt * Np, dtype=dtype, device="cuda")
    s = Hh.stream()
    Hh.call("nppc_conv_pack", prec, w.cuda(), wf, wb, Cout, Cin, ks, Np, Cin, Cin, Np, s)
    X = Halo(B, H, W, Cin, dtype).put(x)
    Y = Halo(B, H, W, Np, dtype)
    Hh.call("nppc_conv_fwd", prec, X.t, Cin, wf, Y.t, Np, b.cuda(), None, None, 0.2, B, H, W, Cin, Cout, Np, ks, s)
    torch.cuda.synchronize()
    ref = F.conv2d(x, q(w, dtype), b, padding=ks // 2)
    assert rel(Y.get(Cout), ref) < tol
    assert Y.halo_is_zero()
    if ks == 3:
        # the same convolution leaving per-tile column sums of its stored output: identical output, and the BatchNorm batch
        # statistics they add up to equal the statistics pass over the tensor (nppc_bn_stats) to fp32 summation noise
        Y1 = Halo(B, H, W, Np, dtype)
        ntiles = (X.P + 127) // 128
        part = torch.full((ntiles * 2 * Np,), float("nan"), dtype=torch.float32, device="cuda")
        Hh.call("nppc_conv_fwd_stats", prec, X.t, Cin, wf, Y1.t, Np, b.cuda(), B, H, W, Cin, Cout, Np, ks, part, s)
        assert torch.equal(Y1.t, Y.t)
        Cs = Cout // 8 * 8
        st_ref = torch.zeros(2 * Cs, dtype=torch.float64, device="cuda")
        Hh.call("nppc_bn_stats", prec, Y.t, Np, X.P, Cs, st_ref, s)
        st = torch.full((2 * Cs,), float("nan"), dtype=torch.float64, device="cuda")
        scr = torch.full((2 * Cs * 128,), float("nan"), dtype=torch.float64, device="cuda")
        Hh.call("nppc_bn_stats_from_parts", part, B, H, W, Np, Cs, st, scr, s)
        st2 = torch.full_like(st, float("nan"))
        Hh.call("nppc_bn_stats_from_parts", part, B, H, W, Np, Cs, st2, scr, s)
        torch.cuda.synchronize()
        assert torch.equal(st, st2)                                   # fixed summation order
        n = B * H * W
        assert float((st[:Cs] - st_ref[:Cs]).abs().max()) < 1e-5 * n ** 0.5 * float(st_ref[Cs:].max() / n) ** 0.5 + 1e-6
        assert float(((st[Cs:] - st_ref[Cs:]) / st_ref[Cs:]).abs().max()) < 1e-5
    # folded eval-mode BatchNorm + LeakyReLU epilogue
    sc, sh = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.1
    Y2 = Halo(B, H, W, Np, dtype)
    Hh.call("nppc_conv_fwd", prec, X.t, Cin, wf, Y2.t, Np, b.cuda(), sc.cuda(), sh.cuda(), 0.2, B, H, W, Cin, Cout, Np, ks, s)
    ref2 = F.leaky_relu(ref * sc[None, :, None, None] + sh[None, :, None, None], 0.2)
    assert rel(Y2.get(Cout), ref2) < tol
    # input gradient = transposed convolution through the backward pack
    dy = q(torch.randn(B, Cout, H, W, generator=g), dtype)
    DY = Halo(B, H, W, Np, dtype).put(dy)
    DX = Halo(B, H, W, Cin, dtype)
    Hh.call("nppc_conv_fwd", prec, DY.t, Np, wb, DX.t, Cin, None, None, None, 0.2, B, H, W, Np, Cin, Cin, ks, s)
    xr = x.clone().requires_grad_(True)
    wr = q(w, dtype).requires_grad_(True)
    F.conv2d(xr, wr, None, padding=ks // 2).backward(dy)
    assert rel(DX.get(Cin), xr.grad) < tol
    # weight gradient (split-K slabs + reduce)
    M = Np
    ksplit = 3
    slabs = torch.empty(nt * ksplit * ((M + 127) // 128 * 128) * Cin, dtype=torch.float32, device="cuda")
    dW = torch.empty(Cout, Cin, ks, ks, dtype=torch.float32, device="cuda")
    Hh.call("nppc_conv_wgrad", prec, DY.t, Np, X.t, Cin, slabs, M, Cin, B, H, W, ks, ksplit, s)
    Hh.call("nppc_conv_wgrad_reduce", prec, slabs, ksplit, M, Cin, dW, Cout, Cin, ks, s)
    torch.cuda.synchronize()
    assert rel(dW.cpu(), wr.grad) < (tol if pname == "fp32" else 1e-2)


@pytest.mark.parametrize("pname,prec,dtype,tol", PRECS)
@pytest.mark.parametrize("B,H,W,Cin", [(2, 9, 13, 1), (3, 6, 37, 2)])
def test_thin_first_convolution_matches_conv2d(pname, prec, dtype, tol, B, H, W, Cin):
    """conv3x3(1 or 2 -> 64) on the direct kernels (tmp_utils.py:8-37 `inc`): forward, folded eval-BatchNorm epilogue,
    weight gradient; the activations are rounded to the storage dtype, the weights stay fp32 (unpacked parameter tensor)"""
    from nppc_audio import _hip as Hh
    Cout, ld = 64, 32
    g = torch.Generator().manual_seed(11 * B + Cin)
    x = q(torch.randn(B, Cin, H, W, generator=g), dtype)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / 3.0
    b = torch.randn(Cout, generator=g) * 0.1
    s = Hh.stream()
    X = Halo(B, H, W, ld, dtype).put(x)
    Y = Halo(B, H, W, Cout, dtype)
    Hh.call("nppc_conv3x3_thin_fwd", prec, X.t, ld, w.cuda(), b.cuda(), None, None, 0.2, Y.t, Cout, B, H, W, Cin, Cout, s)
    torch.cuda.synchronize()
    ref = F.conv2d(x, w, b, padding=1)
    assert rel(Y.get(Cout), ref) < tol
    assert Y.halo_is_zero()
    sc, sh = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.1
    Y2 = Halo(B, H, W, Cout, dtype)
    Hh.call("nppc_conv3x3_thin_fwd", prec, X.t, ld, w.cuda(), b.cuda(), sc.cuda(), sh.cuda(), 0.2, Y2.t, Cout, B, H, W, Cin, Cout, s)
    assert rel(Y2.get(Cout), F.leaky_relu(ref * sc[None, :, None, None] + sh[None, :, None, None], 0.2)) < tol
    dy = q(torch.randn(B, Cout, H, W, generator=g), dtype)
    DY = Halo(B, H, W, Cout, dtype).put(dy)
    part = torch.empty(Hh.conv_thin_part_elems(), dtype=torch.float32, device="cuda")
    dW = torch.empty(Cout, Cin, 3, 3, dtype=torch.float32, device="cuda")
    Hh.call("nppc_conv3x3_thin_wgrad", prec, DY.t, Cout, X.t, ld, part, dW, B, H, W, Cin, Cout, s)
    torch.cuda.synchronize()
    wr = w.clone().requires_grad_(True)
    F.conv2d(x, wr, None, padding=1).backward(dy)
    assert rel(dW.cpu(), wr.grad) < (tol if pname == "fp32" else 2e-3)


@pytest.mark.parametrize("pname,prec,dtype,tol", PRECS)
@pytest.mark.parametrize("B,H,W,K", [(2, 9, 13, 1), (3, 6, 37, 5), (1, 4, 9, 8)])
def test_thin_output_convolution_matches_conv2d(pname, prec, dtype, tol, B, H, W, K):
    """conv1x1(64 -> K <= 8) (`outc`, unet.py:262): forward, input gradient, weight gradient on the direct kernels"""
    from nppc_audio import _hip as Hh
    Cin, ld = 64, 64
    g = torch.Generator().manual_seed(7 * B + K)
    x = q(torch.randn(B, Cin, H, W, generator=g), dtype)
    w = torch.randn(K, Cin, 1, 1, generator=g) / 8.0
    b = torch.randn(K, generator=g) * 0.1
    s = Hh.stream()
    X = Halo(B, H, W, Cin, dtype).put(x)
    Y = Halo(B, H, W, ld, dtype)
    Hh.call("nppc_conv1x1_thin_fwd", prec, X.t, Cin, w.cuda(), b.cuda(), Y.t, ld, B, H, W, Cin, K, s)
    torch.cuda.synchronize()
    assert rel(Y.get(K), F.conv2d(x, w, b)) < tol
    assert Y.halo_is_zero() and float(Y.get(ld - K, K).abs().max()) == 0.0
    dy = q(torch.randn(B, K, H, W, generator=g), dtype)
    DY = Halo(B, H, W, ld, dtype).put(dy)
    DX = Halo(B, H, W, Cin, dtype)
    Hh.call("nppc_conv1x1_thin_bwd_data", prec, DY.t, ld, w.cuda(), DX.t, Cin, B, H, W, Cin, K, s)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    F.conv2d(xr, wr, None).backward(dy)
    assert rel(DX.get(Cin), xr.grad) < tol
    assert DX.halo_is_zero()
    part = torch.empty(Hh.conv_thin_part_elems(), dtype=torch.float32, device="cuda")
    dW = torch.empty(K, Cin, 1, 1, dtype=torch.float32, device="cuda")
    Hh.call("nppc_conv1x1_thin_wgrad", prec, DY.t, ld, X.t, Cin, part, dW, B, H, W, Cin, K, s)
    torch.cuda.synchronize()
    assert rel(dW.cpu(), wr.grad) < (tol if pname == "fp32" else 2e-3)


@pytest.mark.parametrize("pname,prec,dtype,tol", PRECS)
def test_batchnorm_leakyrelu_forward_backward(pname, prec, dtype, tol):
    from nppc_audio import _hip as Hh
    B, H, W, C = 3, 7, 11, 64
    g = torch.Generator().manual_seed(5)
    x = q(torch.randn(B, C, H, W, generator=g) * 2 + 0.5, dtype)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
    rm, rv = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
    s = Hh.stream()
    X = Halo(B, H, W, C, dtype).put(x)
    Y = Halo(B, H, W, 2 * C, dtype)                   # written as the upper channel slice of a wider buffer
    st = torch.zeros(2 * C, dtype=torch.float64, device="cuda")
    ss = torch.empty(4 * C, dtype=torch.float32, device="cuda")
    rmd, rvd = rm.cuda(), rv.cuda()
    Hh.call("nppc_bn_stats", prec, X.t, C, X.P, C, st, s)
    Hh.call("nppc_bn_finalize", st, gamma.cuda(), beta.cuda(), rmd, rvd, ss, C, float(B * H * W), 1e-5, 0.1, 1, s)
    Hh.call("nppc_bn_act", prec, X.t, C, Y.t[C:], 2 * C, ss, C, B, H, W, 0.2, s)
    xr = x.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm2, rv2 = rm.clone(), rv.clone()
    yr = F.leaky_relu(F.batch_norm(xr, rm2, rv2, gr, br, True, 0.1, 1e-5), 0.2)
    torch.cuda.synchronize()
    assert rel(Y.get(C, C), yr.detach()) < tol
    assert float(Y.get(C, 0).abs().max()) == 0.0
    assert rel(rmd.cpu(), rm2) < 1e-5 and rel(rvd.cpu(), rv2) < 1e-5
    # eval-mode scale/shift from the running buffers
    ss2 = torch.empty(4 * C, dtype=torch.float32, device="cuda")
    Hh.call("nppc_bn_finalize", None, gamma.cuda(), beta.cuda(), rmd, rvd, ss2, C, 1.0, 1e-5, 0.1, 0, s)
    sc = gamma / torch.sqrt(rv2 + 1e-5)
    assert rel(ss2[:C].cpu(), sc) < 1e-5 and rel(ss2[C:2 * C].cpu(), beta - rm2 * sc) < 1e-5
    # backward with two upstream gradients (pooling branch + skip branch)
    d1, d2 = q(torch.randn(B, C, H, W, generator=g), dtype), q(torch.randn(B, C, H, W, generator=g), dtype)
    yr.backward(d1 + d2)
    DA, DB = Halo(B, H, W, C, dtype).put(d1), Halo(B, H, W, 3 * C, dtype).put(d2, C)
    DX = Halo(B, H, W, C, dtype)
    S = torch.zeros(2 * C, dtype=torch.float64, device="cuda")
    dgam, dbet = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    Hh.call("nppc_bn_bwd", prec, DA.t, C, DB.t[C:], 3 * C, Y.t[C:], 2 * C, X.t, C, ss, S, DX.t, C, dgam, dbet, C, B, H, W,
            0.2, s)
    torch.cuda.synchronize()
    assert rel(DX.get(C), xr.grad) < tol * 3
    assert rel(dgam.cpu(), gr.grad) < tol * 3 and rel(dbet.cpu(), br.grad) < tol * 3
    assert DX.halo_is_zero()


@pytest.mark.parametrize("pname,prec,dtype,tol", PRECS)
@pytest.mark.parametrize("H,W", [(8, 13), (6, 6), (5, 9)])
def test_maxpool_and_bilinear_upsample(pname, prec, dtype, tol, H, W):
    from nppc_audio import _hip as Hh
    B, C = 2, 64
    g = torch.Generator().manual_seed(H * 31 + W)
    s = Hh.stream()
    x = q(torch.randn(B, C, H, W, generator=g), dtype)
    X = Halo(B, H, W, C, dtype).put(x)
    Ho, Wo = H // 2, W // 2
    Y = Halo(B, Ho, Wo, C, dtype)
    idx = torch.empty(Y.P * C, dtype=torch.uint8, device="cuda")
    Hh.call("nppc_maxpool2", prec, X.t, C, Y.t, C, idx, C, B, H, W, s)
    xr = x.clone().requires_grad_(True)
    yr = F.max_pool2d(xr, 2)
    torch.cuda.synchronize()
    assert rel(Y.get(C), yr.detach()) == 0.0
    dy = q(torch.randn(B, C, Ho, Wo, generator=g), dtype)
    yr.backward(dy)
    DY = Halo(B, Ho, Wo, C, dtype).put(dy)
    DX = Halo(B, H, W, C, dtype)
    Hh.call("nppc_maxpool2_bwd", prec, DY.t, C, idx, DX.t, C, C, B, H, W, s)
    torch.cuda.synchronize()
    assert rel(DX.get(C), xr.grad) == 0.0
    # bilinear x2 (align_corners) of the pooled map, padded back to (H, W) like tmp_utils.up, into a channel slice
    pr = yr.detach().clone().requires_grad_(True)
    up = F.interpolate(pr, scale_factor=2, mode="bilinear", align_corners=True)
    dyy, dxx = H - up.shape[2], W - up.shape[3]
    up = F.pad(up, (dxx // 2, dxx - dxx // 2, dyy // 2, dyy - dyy // 2))
    U = Halo(B, H, W, 2 * C, dtype)
    Hh.call("nppc_upsample2", prec, Y.t, C, U.t[C:], 2 * C, C, B, Ho, Wo, H, W, s)
    torch.cuda.synchronize()
    assert rel(U.get(C, C), up.detach()) < (1e-6 if pname == "fp32" else 1e-2)
    du = q(torch.randn(B, C, H, W, generator=g), dtype)
    up.backward(du)
    DU = Halo(B, H, W, 2 * C, dtype).put(du, C)
    DP = Halo(B, Ho, Wo, C, dtype)
    Hh.call("nppc_upsample2_bwd", prec, DU.t[C:], 2 * C, DP.t, C, C, B, Ho, Wo, H, W, s)
    torch.cuda.synchronize()
    assert rel(DP.get(C), pr.grad) < (1e-5 if pname == "fp32" else 1e-2)


def test_preprocess_matches_golden():
    from nppc_audio.inpainting.utils import preprocess_data
    for name in ("inp_tiny", "inp_c3s"):
        z, meta = load(name)
        cn, mask4, mn, mean, std = preprocess_data(torch.from_numpy(z["clean_spec"]).cuda(),
                                                   torch.from_numpy(z["masked_spec"]).cuda(),
                                                   torch.from_numpy(z["mask_frames"]).cuda(), plot_mean_std=True)
        assert abs(float(mean) - meta["mean"]) < 2e-6 * abs(meta["mean"]) + 1e-6
        assert abs(float(std) - meta["std"]) < 2e-6 * meta["std"]
        assert rel(cn.cpu(), z["clean_norm"]) < 2e-6 and rel(mn.cpu(), z["masked_norm"]) < 2e-6
        assert tuple(mask4.shape) == tuple(z["clean_norm"].shape)


def build_trainer(meta, precision, tmp_path, z, opt="Adam"):
    from nppc_audio.inpainting.trainer.nppc_trainer import NPPCAudioInpaintingTrainer, NPPCAudioInpaintingTrainerConfig
    c = meta["config"]
    wts = {k: torch.from_numpy(np.asarray(v)) for k, v in W.make_weights(W.inpainting_spec(c["K"]), c["seed"]).items()}
    pre = "pretrained_restoration_model.net."
    ck = os.path.join(str(tmp_path), "restorer.pt")
    torch.save({"model_state_dict": {k[len(pre):]: v for k, v in wts.items() if k.startswith(pre)}}, ck)
    cfg = NPPCAudioInpaintingTrainerConfig(
        nppc_model_configuration=dict(
            pretrained_restoration_model_configuration=dict(in_channels=1, out_channels=1, dropout=0.2, precision=precision),
            pretrained_restoration_model_path=ck,
            audio_pc_wrapper_configuration=dict(n_dirs=c["K"], model_configuration=dict(in_channels=2, out_channels=c["K"],
                                                                                         precision=precision)),
            device="cuda"),
        data_configuration=dict(clean_path=".", stft_configuration=dict(nfft=c["nfft"], hop_length=c["hop"],
                                                                         win_length=c["nfft"])),
        dataloader_configuration=dict(batch_size=c["B"], num_workers=0, pin_memory=False, shuffle=False),
        optimizer_configuration=dict(type=opt, args=dict(lr=1e-4, betas=[0.5, 0.999])), device="cuda")

    class Mem(torch.utils.data.Dataset):
        def __len__(self):
            return c["B"]

        def __getitem__(self, i):
            return (torch.from_numpy(z["masked_spec"][i]), torch.from_numpy(z["mask_frames"][i]),
                    torch.from_numpy(z["clean_spec"][i]))

    tr = NPPCAudioInpaintingTrainer(cfg, dataset=Mem())
    tr.nppc_model.load_state_dict(wts, strict=True)
    tr.nppc_model.to("cuda")
    assert tr.nppc_model.pc_wrapper.training and not tr.nppc_model.pretrained_restoration_model.training
    batch = tuple(torch.from_numpy(z[k]).cuda() for k in ("masked_spec", "mask_frames", "clean_spec"))
    return tr, wts, batch


@pytest.mark.parametrize("name,precision", [("inp_tiny", "fp32"), ("inp_c3s", "fp32"), ("inp_c3s", "bf16")])
def test_step_forward_and_gradients(name, precision, tmp_path):
    z, meta = load(name)
    fp32 = precision == "fp32"
    tr, wts, batch = build_trainer(meta, precision, tmp_path, z)
    model = tr.nppc_model
    with torch.no_grad():
        from nppc_audio.inpainting.utils import preprocess_data
        cn, mask4, mn = preprocess_data(batch[2], batch[0], batch[1])
        pred = model.get_pred_spec_mag_norm(mn, mask4)
    assert rel(pred.cpu(), z["pred_norm"]) < (5e-5 if fp32 else 5e-2)
    tr.step = 500
    reconst, obj, log = tr.base_step(batch)
    model.zero_grad()
    obj.backward()
    torch.cuda.synchronize()
    assert rel(log["w_mat"].cpu(), z["log.w_mat"]) < (5e-4 if fp32 else 1.5e-1)
    assert abs(float(obj) - meta["g500.objective"]) < (2e-5 if fp32 else 3e-2)
    for k in ("err_norm", "err_proj", "w_norms", "reconst_err", "second_moment_mse"):
        assert rel(log[k].cpu(), z["log." + k]) < (1e-3 if fp32 else 2e-1), k
    # BatchNorm running buffers after this one train-mode forward (the reference updates them the same way)
    sd = model.state_dict()
    for k in z.files:
        if k.startswith("bn1.") and "num_batches" not in k:
            assert rel(sd[k[4:]].cpu().numpy().reshape(-1)[:4096], z[k]) < (5e-5 if fp32 else 2e-2), k
    assert int(sd["pc_wrapper.net.up2.conv.conv.1.num_batches_tracked"]) == 8

    # every parameter gradient against the fp64 oracle (the fp32 oracle is pinned to the reference goldens on CPU)
    P = {}
    for k, v in wts.items():
        P[k] = v.double() if v.is_floating_point() else v.clone()
    names = [k for k in P if k.startswith("pc_wrapper.") and P[k].is_floating_point() and "running_" not in k]
    for k in names:
        P[k].requires_grad_(True)
    _, obj_o, _ = R.inpaint_step(*[torch.from_numpy(z[k]).double() for k in ("masked_spec", "mask_frames", "clean_spec")],
                                 P, 500)
    ref = dict(zip(names, torch.autograd.grad(obj_o, [P[k] for k in names])))
    got = dict(model.named_parameters())
    worst = {}
    dot = nn_g = nn_r = 0.0
    for n in names:
        gq = got[n].grad
        assert gq is not None, n
        gd = gq.double().cpu()
        if n.endswith((".conv.0.bias", ".conv.3.bias")):
            assert float(gd.abs().max()) < 1e-5        # exactly zero in exact arithmetic (BatchNorm absorbs the bias)
            continue
        worst[n] = (gd - ref[n]).abs().max().item() / (ref[n].abs().max().item() + 1e-300)
        dot += float((gd * ref[n]).sum())
        nn_g += float((gd * gd).sum())
        nn_r += float((ref[n] * ref[n]).sum())
    cos = dot / np.sqrt(nn_g * nn_r)
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:6]
    print(name, precision, f"cos={cos:.6f} |g|/|ref|={np.sqrt(nn_g / nn_r):.5f}", [(n[15:], f"{r:.1e}") for n, r in top])
    if fp32:
        # fp32 tolerance: 2e-2 of the tensor's max |grad| (single LeakyReLU / max-pool decisions on activations within
        # rounding of a tie move individual elements; the reference itself differs from fp64 by up to 6e-3 here),
        # plus direction and norm of the whole gradient
        bad = {n: r for n, r in worst.items() if r > 2e-2}
        assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:8]
        assert cos > 0.99999 and abs(np.sqrt(nn_g / nn_r) - 1) < 1e-3
        assert abs(np.sqrt(nn_g) - meta["g500.grad_total_l2"]) < 2e-3 * meta["g500.grad_total_l2"]
    else:
        assert cos > 0.98 and abs(np.sqrt(nn_g / nn_r) - 1) < 0.1
    assert all(p.grad is None for n, p in model.named_parameters() if n.startswith("pretrained_restoration_model"))


@pytest.mark.parametrize("name,opt", [("inp_tiny", "Adam"), ("inp_c3s", "Adam"), ("inp_tiny", "AdamW")])
def test_two_clipped_optimizer_steps(name, opt, tmp_path):
    """train_step x 2 (clip_grad_norm_ 1.0 + Adam(0.5, 0.999)) against the reference's weights after one and two steps"""
    z, meta = load(name)
    tr, wts, batch = build_trainer(meta, "fp32", tmp_path, z, opt)
    if opt == "AdamW":
        for g in tr.optimizer.param_groups:
            g["weight_decay"] = 0.0
    params = dict(tr.nppc_model.named_parameters())
    for it, step in ((1, 500), (2, 501)):
        tr.step = step
        _, obj, _ = tr.train_step(batch)
        torch.cuda.synchronize()
        if it == 2:
            assert abs(float(obj) - meta["adam1.objective_next"]) < 5e-4
        checked = 0
        for k in z.files:
            if not k.startswith(f"adam{it}."):
                continue
            n = k[len(f"adam{it}."):]
            got = params[n].detach().cpu().numpy().reshape(-1)
            want = z[k]
            w0 = wts[n].numpy().reshape(-1)[: want.size]
            d = np.abs((got[: want.size] - w0) - (want - w0))
            assert (d > 0.05 * 1e-4 * it + 1e-7).sum() <= max(1, 1e-2 * d.size) and np.median(d) < 2e-6, (it, n)
            checked += 1
        assert checked >= 8


# ---- row f3 (second half): on-device batch synthesis of the inpainting step -------------------------------------------
@pytest.mark.parametrize("name", ["c3", "tiny"])
def test_batch_synthesis_against_reference_dataset_goldens(name):
    """normalise -> gap mask -> centred STFT (nfft 255 / 63: not powers of two) -> time_to_spec_mask -> masked STFT,
    against the outputs of the reference's own dataset methods (tests/golden/inp_data.npz)."""
    from nppc_audio.inpainting.data import inpainting_batch_on_device
    z = np.load(os.path.join(GOLD, "inp_data.npz"))
    nfft, hop, L = (int(v) for v in z[f"{name}_cfg"])
    gaps = z[f"{name}_gaps"]
    clean = torch.from_numpy(z[f"{name}_clean_in"]).cuda()
    sm, mf, sc, ma = inpainting_batch_on_device(clean, gaps[:, 0], gaps[:, 0] + gaps[:, 1], nfft, hop)
    torch.cuda.synchronize()
    assert np.array_equal(mf.cpu().numpy(), z[f"{name}_mask_frames"])          # 0/1 values: bit exact
    scale = np.abs(z[f"{name}_stft_clean"]).max()
    assert np.abs(sc.cpu().numpy() - z[f"{name}_stft_clean"]).max() < 2e-6 * scale   # fp32 FFT vs fp64-accumulated DFT
    assert np.abs(sm.cpu().numpy() - z[f"{name}_stft_masked"]).max() < 2e-6 * scale
    gold_ma = z[f"{name}_masked_audio"]
    got_ma = ma[:, 0].cpu().numpy()
    assert np.array_equal(got_ma == 0, gold_ma == 0)
    assert np.abs(got_ma - gold_ma).max() < 2e-6 * np.abs(gold_ma).max()       # gain: powf / log10f vs torch's
    # masked frames are exactly zero in the masked STFT, untouched elsewhere
    keep = mf.bool()[:, None, None, :].expand_as(sm)
    assert float(sm[~keep].abs().max()) == 0.0 and torch.equal(sm[keep], sc[keep])


def test_time_to_spec_mask_arbitrary_masks_and_stft_against_oracle():
    from nppc_audio.inpainting.data import audio_to_stft, time_to_spec_mask
    g = torch.Generator().manual_seed(5)
    for (win, hop, L, center) in ((255, 128, 4000, True), (63, 32, 777, True), (64, 16, 500, False), (255, 128, 130, True)):
        T = 1 + L // hop
        m = torch.ones(5, L)
        m[0, 0] = 0
        m[1, L - 1] = 0
        m[2, torch.randint(0, L, (7,), generator=g)] = 0
        m[3, :] = 0
        got = time_to_spec_mask(m.cuda(), T + 2, L, win, hop, center).cpu()      # T + 2: frames whose window is empty
        for b in range(5):
            want = R.time_to_spec_mask(m[b:b + 1], T + 2, L, win, hop, center)
            assert torch.equal(got[b], want), (win, hop, L, center, b)
    # the general-N STFT against torch.stft on the CPU, several sizes (odd, even, power of two)
    for nfft, hop, L in ((255, 128, 3000), (63, 32, 500), (100, 25, 1000), (512, 256, 4096), (254, 127, 128)):
        x = torch.randn(3, L, generator=g)
        want = R.stft_pair(x.double(), nfft, hop, nfft)
        got = audio_to_stft(x.cuda(), nfft, hop).cpu().double()
        assert got.shape == want.shape
        assert float((got - want).abs().max() / want.abs().max()) < 1e-6, (nfft, hop, L)


def test_synthesised_batch_feeds_the_train_step():
    """the synthesised tuple drives NPPCAudioInpaintingTrainer.base_step unchanged (shapes, dtypes, value ranges)"""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    import bench_inpainting as bi
    from nppc_audio.inpainting.data import inpainting_batch_on_device
    B, L = 4, 8000
    _, clean = W.synth_batch(B, L)
    sm, mf, sc, ma = inpainting_batch_on_device(torch.from_numpy(clean).cuda(), [3000, 0, 7000, 4000],
                                                [4600, 800, 8000, 4400])
    Fq, T = sc.shape[2], sc.shape[3]
    assert (Fq, T) == (128, 63) and ma.shape == (B, 1, L)
    tr = bi.build("fp32", B, Fq, T)
    tr.step = 500
    rec, obj, log = tr.base_step((sm, mf, sc))
    obj.backward()
    torch.cuda.synchronize()
    assert torch.isfinite(obj) and float(rec.min()) > -1e-4 and float(rec.max()) < 1 + 1e-4
    w = log["w_mat"]
    assert float(w[mf.bool()[:, None, None, :].expand_as(w)].abs().max()) == 0.0


# ---- row f4: MC-dropout + PCA baseline ----------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["mc50", "small", "odd"])
def test_pca_batch_against_reference_pca_goldens(name):
    """batched Gram + Jacobi + components (csrc/mc_pca.hip) against the outputs of the reference's
    compute_pca_sklearn_batch (tests/golden/pca.npz) and the fp64 oracle"""
    from nppc_audio.inpainting.mc_baseline import compute_pca_batch
    z = np.load(os.path.join(GOLD, "pca.npz"))
    x = torch.from_numpy(z[f"{name}_x"])
    pcs, scaled, w, mean, sv = (t.cpu().numpy() for t in compute_pca_batch(x.cuda(), 5))
    assert np.abs(pcs - z[f"{name}_pcs"]).max() < 2e-5
    assert np.abs(sv - z[f"{name}_svals"]).max() < 2e-6 * z[f"{name}_svals"].max()
    assert np.abs(scaled - z[f"{name}_scaled"]).max() < 2e-5 * np.abs(z[f"{name}_scaled"]).max()
    assert np.abs(w - z[f"{name}_weights"]).max() < 1e-6 and np.abs(mean - z[f"{name}_mean"]).max() < 5e-6
    o = [t.numpy() for t in R.pca_batch(x, 5)]
    assert np.abs(pcs - o[0]).max() < 5e-6 and np.abs(sv - o[4]).max() < 1e-6 * o[4].max()
    # properties: orthonormal components, scaled = component * singular value
    g = np.einsum("bid,bjd->bij", pcs.astype(np.float64), pcs.astype(np.float64))
    assert np.abs(g - np.eye(g.shape[1])).max() < 1e-5
    assert np.abs(scaled - pcs * sv[:, :, None]).max() < 1e-6 * sv.max()


def test_pca_batch_edge_cases():
    from nppc_audio.inpainting.mc_baseline import compute_pca_batch
    g = torch.Generator().manual_seed(3)
    # more components than samples -> min(n, K); K = 2; a constant stack (all singular values 0) stays finite
    x = torch.randn(2, 3, 40, generator=g)
    pcs, scaled, w, mean, sv = compute_pca_batch(x.cuda(), 5)
    assert pcs.shape == (3, 2, 40) and sv.shape == (3, 2)
    o = R.pca_batch(x, 5)
    assert float((pcs[:, 0].cpu() - o[0][:, 0]).abs().max()) < 1e-5 and float((sv.cpu() - o[4]).abs().max()) < 1e-5
    c = torch.ones(6, 1, 33)
    pcs, scaled, w, mean, sv = compute_pca_batch(c.cuda(), 3)
    assert float(sv.abs().max()) == 0.0 and bool(torch.isfinite(pcs).all()) and float(mean.min()) == 1.0
    with pytest.raises(RuntimeError, match="unsupported"):
        compute_pca_batch(torch.zeros(61, 1, 8).cuda(), 3)


def _dropout_unet(precision, seed=0):
    from nppc_audio.inpainting.networks.unet import RestorationWrapper, UNet, UNetConfig
    spec = W.unet_spec(1, 1)
    wts = {k: torch.from_numpy(v) for k, v in W.make_weights(spec, 21).items()}
    net = UNet(UNetConfig(in_channels=1, out_channels=1, dropout=0.2, precision=precision))
    net.load_state_dict(wts, strict=True)
    net.dropout_seed = 1234 + seed
    return RestorationWrapper(net).cuda().eval(), wts


def _tap_to_nchw(tap, B, H, Wd):
    keep, level, C = tap
    return keep.reshape(B, H + 2, Wd + 2, C)[:, 1:-1, 1:-1].permute(0, 3, 1, 2).float().cpu()


@pytest.mark.parametrize("precision,tol", [("fp32", 2e-4), ("bf16", 6e-2)])
def test_mc_dropout_pass_against_oracle_with_the_same_keep_masks(precision, tol):
    """one stochastic pass: the keep bits the kernel drew (tapped) fed to the oracle's U-Net -> same output; the bits
    are Bernoulli(0.8), differ between passes, repeat for the same (seed, pass)"""
    from nppc_audio.inpainting.mc_baseline import enable_dropout
    B, Fq, T = 2, 32, 37
    model, wts = _dropout_unet(precision)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(B, 1, Fq, T, generator=g)
    mask = torch.ones(B, 1, Fq, T)
    mask[:, :, :, 10:17] = 0
    enable_dropout(model)
    assert not model.net.training and all(m.training for m in model.net.modules() if isinstance(m, torch.nn.Dropout))
    model.net.dropout_tap = {}
    with torch.no_grad():
        y0 = model(x.cuda(), mask.cuda()).cpu()
    taps0 = dict(model.net.dropout_tap)
    dims = {"down3": (Fq // 8, T // 8), "down4": (Fq // 16, T // 16), "up1": (Fq // 8, T // 8), "up2": (Fq // 4, T // 4)}
    keep = {blk: _tap_to_nchw(taps0[blk], B, *dims[blk]) for blk in dims}
    frac = np.mean([float(k.mean()) for k in keep.values()])
    assert 0.77 < frac < 0.83, frac
    P = {k: v.double() for k, v in wts.items()}
    want = R.unet_forward(x.double(), P, "", train=False, keep={k: v.double() for k, v in keep.items()}, p_drop=0.2)
    want = x.double() * mask + want * (1 - mask)
    assert rel(y0.numpy(), want.numpy()) < tol
    # second pass: new bits, different output in the gap, identical known frames
    model.net.dropout_tap = {}
    with torch.no_grad():
        y1 = model(x.cuda(), mask.cuda()).cpu()
    assert not torch.equal(model.net.dropout_tap["down4"][0], taps0["down4"][0])
    assert float((y1 - y0).abs().max()) > 1e-3 and torch.equal((y1 * mask), (y0 * mask))
    # replaying pass 0 reproduces it bit for bit
    model.net.dropout_pass = 0
    model.net.dropout_tap = {}
    with torch.no_grad():
        y2 = model(x.cuda(), mask.cuda()).cpu()
    assert torch.equal(y2, y0) and torch.equal(model.net.dropout_tap["up2"][0], taps0["up2"][0])
    # eval mode again (dropout modules back to eval): deterministic identity dropout == the plain restorer
    model.eval()
    with torch.no_grad():
        y3 = model(x.cuda(), mask.cuda()).cpu()
    plain = R.restorer_forward(x.double(), mask.double(), {"n." + k: v for k, v in P.items()}, pre="n.")
    assert rel(y3.numpy(), plain.numpy()) < tol


def test_calculate_unet_baseline_end_to_end():
    """calculate_unet_baseline's dict: shapes, support on the gap only, PCA of the very samples the passes produced
    (oracle PCA on the sampled stack), importance weights sum to 1"""
    from nppc_audio.inpainting import mc_baseline as MB
    B, Fq, T, K = 3, 32, 40, 12
    model, _ = _dropout_unet("fp32", seed=5)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(B, 1, Fq, T, generator=g).cuda()
    mask = torch.ones(B, 1, Fq, T)
    for b, s in enumerate((5, 17, 30)):
        mask[b, :, :, s:s + 6] = 0
    mask = mask.cuda()
    out = MB.calculate_unet_baseline(model, x, mask, n_mc_samples=K, n_components=5)
    assert out["mean_prediction"].shape == (B, 1, Fq, T) and out["principal_components"].shape == (B, 5, Fq, T)
    assert out["scaled_principal_components"].shape == (B, 5, Fq, T) and out["singular_vals"].shape == (B, 5)
    known = mask.bool().expand(-1, 5, -1, -1)
    assert float(out["principal_components"][known].abs().max()) == 0.0
    assert float(out["mean_prediction"][mask.bool()].abs().max()) == 0.0
    assert torch.allclose(out["importance_weights"].sum(1), torch.ones(B, device="cuda"), atol=1e-5)
    # replay the same K passes and run the oracle PCA on that stack
    model.net.dropout_pass = 0
    preds, hole = MB.mc_dropout_samples(model, x, mask, K)
    o = R.pca_batch(preds.cpu(), 5)
    full = R.scatter_gap(o[0].float(), mask.cpu())
    assert float((out["principal_components"].cpu() - full).abs().max()) < 2e-4
    assert float((out["singular_vals"].cpu() - o[4].float()).abs().max()) < 1e-4 * float(o[4].max())
    assert float((out["mean_prediction"][:, 0].cpu() - R.scatter_gap(o[3].float(), mask.cpu())).abs().max()) < 1e-5
    with pytest.raises(ValueError, match="same number of masked"):
        bad = mask.clone()
        bad[0, :, :, 0] = 0
        MB.mc_dropout_samples(model, x, bad, 2)


def test_base_step2_loss_and_gradient_against_oracle():
    """PairProjectionLoss (nppc_pair_loss / _bwd) against the fp64 torch restatement of nppc_trainer.py:285-323"""
    from nppc_audio.inpainting.mc_baseline import PairProjectionLoss
    g = torch.Generator().manual_seed(8)
    B, K, Fq, T = 3, 5, 16, 23
    w = torch.randn(B, K, Fq, T, generator=g) * torch.rand(B, K, 1, 1, generator=g) * 3
    m = torch.randn(B, K, Fq, T, generator=g) + 0.5 * w
    sv = torch.rand(B, K, generator=g) * 20
    gr = torch.randn(B, generator=g)
    for lam in (1e-6, 0.7):
        wd = w.double().requires_grad_(True)
        rec_o, sm_o, obj_o, proj_o, wn_o = R.base_step2_loss(wd, m.double(), sv.double(), lam)
        (obj_o * 1.3 + (rec_o * gr.double()).sum()).backward()
        wg = w.cuda().requires_grad_(True)
        rec, obj, proj, wn, sm = PairProjectionLoss.apply(wg, m.cuda(), sv.cuda(), lam)
        (obj * 1.3 + (rec * gr.cuda()).sum()).backward()
        assert rel(rec.detach().cpu(), rec_o.detach()) < 1e-5 and rel(sm.cpu(), sm_o.detach()) < 1e-5
        assert abs(float(obj) - float(obj_o)) < 1e-5 * abs(float(obj_o))
        assert rel(proj.cpu(), proj_o.detach()) < 1e-5 and rel(wn.cpu(), wn_o.detach()) < 1e-6
        assert rel(wg.grad.cpu(), wd.grad) < 1e-5


def test_base_step2_runs_on_the_trainer():
    """the alternative-target step end to end: MC passes with the restorer in train mode, PCA, projection loss, backward"""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    import bench_inpainting as bi
    B, Fq, T = 4, 128, 80
    tr = bi.build("fp32", B, Fq, T)
    masked, mask, clean = bi.synth(B, Fq, T, "cuda")
    tr.step = 500
    before = tr.nppc_model.pretrained_restoration_model.net.state_dict()["down3.mpconv.1.conv.1.running_mean"].clone()
    rec, obj, log = tr.base_step2((masked, mask, clean), n_mc_samples=8)
    tr.nppc_model.zero_grad()
    obj.backward()
    torch.cuda.synchronize()
    assert set(log) == {"w_mat", "w_mc", "proj_W_mc_on_W_nppc", "w_norms", "reconst_err", "second_moment_mse", "objective"}
    assert log["w_mc"].shape == log["w_mat"].shape and log["proj_W_mc_on_W_nppc"].shape == (B, bi.K_DIRS)
    assert torch.isfinite(obj) and float(rec.min()) > -1e-5 and float(rec.max()) < 1 + 1e-5
    grad = tr.nppc_model.pc_wrapper.net.engine().fp.grad
    assert bool(torch.isfinite(grad).all()) and float(grad.abs().max()) > 0
    net = tr.nppc_model.pretrained_restoration_model.net
    assert not net.training and not any(m.training for m in net.modules())
    after = net.state_dict()["down3.mpconv.1.conv.1.running_mean"]
    assert not torch.equal(before, after)           # restoration_model.train(): the reference moves these buffers too
    # the w_mc rows are the scaled components: their norms are the singular values
    n = log["w_mc"].flatten(2).norm(dim=2)
    assert float(n.min()) > 0


@pytest.mark.parametrize("case", ["a", "b"])
def test_compute_metrics_against_reference_goldens(case):
    """device Gram + host 5 x 5 algebra against the reference's own compute_metrics output (tests/golden/metrics.npz)"""
    from nppc_audio.inpainting.mc_baseline import compute_metrics
    z = np.load(os.path.join(GOLD, "metrics.npz"))
    t = {k: torch.from_numpy(z[f"{case}_{k}"]).cuda() for k in ("nppc", "mc", "pred", "mean", "clean", "mask")}
    m = compute_metrics(t["nppc"], t["mc"], t["pred"], t["mean"], t["clean"], t["mask"])
    got = np.array([m["nppc"]["rmse"], m["nppc"]["residual_error"], m["mc_dropout"]["rmse"], m["mc_dropout"]["residual_error"]])
    assert np.abs(got - z[f"{case}_scalars"]).max() < 2e-6 * z[f"{case}_scalars"].max()
    assert np.abs(np.array(m["principal_angles"]) - z[f"{case}_angles"]).max() < 1e-3      # degrees; reference is fp32
