"""Launch sequence of the inpainting U-Net on the HIP kernels of csrc/unet.hip (no torch compute ops).

Mirrors UNet.forward (nppc_audio/inpainting/networks/unet.py:277-290) block by block on haloed NHWC
activations: one `UNetEngine` owns the packed weights, the activation / gradient buffers of every
resolution level and the launch order of forward and backward.  The frozen restorer runs with its
BatchNorm folded into the convolution epilogue (eval mode); the direction U-Net runs in train mode
(batch statistics, running-buffer update) and keeps what its backward needs.
"""
import os

import numpy as np
import torch

from . import _hip as H
from .engine import FlatParams

LEAK = 0.2
BN_EPS = 1e-5
BN_MOMENTUM = 0.1
# (block, path of the double_conv inside it, in channels (None = net input), out channels, level)
PLAN = (("inc", "conv.conv", None, 64, 0), ("down1", "mpconv.1.conv", 64, 128, 1), ("down2", "mpconv.1.conv", 128, 256, 2),
        ("down3", "mpconv.1.conv", 256, 512, 3), ("down4", "mpconv.1.conv", 512, 512, 4), ("up1", "conv.conv", 1024, 256, 3),
        ("up2", "conv.conv", 512, 128, 2), ("up3", "conv.conv", 256, 64, 1), ("up4", "conv.conv", 128, 64, 0))
IN_LD = 64          # the 1- or 2-channel net input is staged into a 64-wide zero-padded pixel row
OUT_LD = 64         # so is the K-channel output of the 1x1 convolution
DROP_BLOCKS = ("down3", "down4", "up1", "up2")    # double_convs that end in nn.Dropout (unet.py:254-257)


BN_STATS_IN_CONV = os.environ.get("NPPC_BN_STATS_IN_CONV", "1") != "0"   # train-mode BatchNorm statistics from the convolution's epilogue (A/B switch)
THIN = os.environ.get("NPPC_UNET_THIN", "1") != "0"      # direct kernels for the first (1-2 channel) and last (1x1, K <= 8) convolutions
PROFILE = None      # bench hook: list of (kind, algorithmic flops, start event, end event)


def rup(a, b):
    return (a + b - 1) // b * b


class _timed:
    """records a HIP-event pair on the current stream around the launches inside the block (bench only)"""

    def __init__(self, kind, flops):
        self.kind, self.flops = kind, flops

    def __enter__(self):
        if PROFILE is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *exc):
        if PROFILE is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            PROFILE.append((self.kind, self.flops, self.e0, e1))


class _Buf:
    """haloed NHWC matrix [P rows][ld] inside a zeroed allocation with guard rows on both sides"""

    def __init__(self, rows, ld, guard_before, guard_after, dtype, device):
        self.ld = ld
        self.store = torch.zeros((guard_before + rows + guard_after) * ld, dtype=dtype, device=device)
        self.t = self.store[guard_before * ld:]

    def at(self, coff=0):
        return self.t[coff:] if coff else self.t


class UNetEngine:
    def __init__(self, module, in_ch, out_ch, prec, trainable):
        H.require_gpu()
        self.mod = module
        self.prec = prec
        self.dtype = H.dtype_of(prec)
        self.in_ch, self.out_ch = in_ch, out_ch
        self.trainable = trainable
        self.dev = next(module.parameters()).device
        self.fp = FlatParams(module, self.dev)
        self.bufs = {}
        self.geo = None
        self._packed_version = None
        self._gsel = 0
        self._gbufs = [None, None]
        self.convs = []                      # (name prefix, cin, cout) of every convolution, forward order
        for blk, path, cin, cout, _ in PLAN:
            cin = in_ch if cin is None else cin
            self.convs.append((f"{blk}.{path}.0", cin, cout, 3))
            self.convs.append((f"{blk}.{path}.3", cout, cout, 3))
        self.convs.append(("outc.conv", 64, out_ch, 1))
        self.wf, self.wb = {}, {}
        self._buffers = dict(module.named_buffers())

    # ------------------------------------------------------------------------------------------ parameters
    def p(self, name):
        return self.fp.view(name)

    def g(self, name):
        return self._grad.narrow(0, self.fp.off[name][0], int(np.prod(self.fp.off[name][1])))

    def bnbuf(self, name):
        return self._buffers[name]

    def _dims(self, name, cin, cout):
        cinp = IN_LD if cin < 32 else cin            # K of the forward GEMM (multiple of 32)
        np_ = rup(cout, 64)                          # N of the forward GEMM
        return cinp, np_

    def pack_weights(self, train):
        """fp32 parameters -> packed bf16/f32 GEMM operands (forward: [Np][taps*Cinp], input-gradient:
        [Cinp][taps*Np]).  Re-done whenever a parameter or BatchNorm buffer was written since the last pack."""
        ver = (self.fp.version(), sum(b._version for b in self._buffers.values()), bool(train))
        if self._packed_version == ver:
            return
        s = H.stream()
        for name, cin, cout, ks in self.convs:
            cinp, np_ = self._dims(name, cin, cout)
            nt = ks * ks
            if name not in self.wf:
                self.wf[name] = torch.empty(np_ * nt * cinp, dtype=self.dtype, device=self.dev)
                need_b = self.trainable and not name.startswith("inc.conv.conv.0")
                self.wb[name] = torch.empty(cinp * nt * np_, dtype=self.dtype, device=self.dev) if need_b else None
            H.call("nppc_conv_pack", self.prec, self.p(name + ".weight"), self.wf[name], self.wb[name], cout, cin, ks, np_,
                   cinp, cinp, np_, s)
        if not train:
            # eval-mode BatchNorm folds into a per-channel scale / shift applied in the convolution epilogue
            self.ss = {}
            for name, cin, cout, ks in self.convs[:-1]:
                bn = name[:-1] + str(int(name[-1]) + 1)
                ss = torch.empty(4 * cout, dtype=torch.float32, device=self.dev)
                H.call("nppc_bn_finalize", None, self.p(bn + ".weight"), self.p(bn + ".bias"), self.bnbuf(bn + ".running_mean"),
                       self.bnbuf(bn + ".running_var"), ss, cout, 1.0, BN_EPS, BN_MOMENTUM, 0, s)
                self.ss[name] = ss
        self._packed_version = ver

    # ------------------------------------------------------------------------------------------ buffers
    def _setup(self, B, F, T):
        if self.geo == (B, F, T):
            return
        self.geo = (B, F, T)
        self.bufs = {}
        self.lv = []
        h, w = F, T
        for _ in range(5):
            if h < 1 or w < 1:
                raise ValueError(f"spectrogram {F}x{T} is too small for four 2x2 poolings")
            self.lv.append((h, w))
            h, w = h // 2, w // 2
        self.P = [B * (h + 2) * (w + 2) for h, w in self.lv]
        self.ksplit = [int(min(64, max(1, p // 1024))) for p in self.P]   # 64 slices x 9 taps already fill the chip; more only costs reduction traffic
        self.ksplit = [k // 8 * 8 if k >= 8 else k for k in self.ksplit]     # multiples of 8: XCD-aware tile order
        self.slab = None
        self._thin_part = None
        self.st = torch.zeros(2 * 1024, dtype=torch.float64, device=self.dev)
        self._statpart = None        # per-tile column sums of a convolution's output (nppc_conv_fwd_stats), sized at first use
        self._statscr = None

    def buf(self, tag, level, ld):
        key = (tag, level, ld)
        b = self.bufs.get(key)
        if b is None:
            h, w = self.lv[level]
            gb = w + 3 + 1
            ga = 64 * self.ksplit[level] + 128 + w + 4
            b = self.bufs[key] = _Buf(self.P[level], ld, gb, ga, self.dtype, self.dev)
        return b

    # ------------------------------------------------------------------------------------------ forward
    def _conv(self, name, cin, cout, ks, x, ldx, y, ldy, level, fold, stats=False):
        """stats=True (train-mode BatchNorm follows): returns the per-tile column sums of the stored output when the tiled kernel
        runs (they replace the statistics pass over the tensor), else None"""
        B = self.geo[0]
        h, w = self.lv[level]
        cinp, np_ = self._dims(name, cin, cout)
        if (stats and BN_STATS_IN_CONV and not fold and ks == 3 and not (THIN and cin <= 2 and cout <= 64 and cout % 8 == 0)
                and cinp % (64 if self.prec == H.PREC_BF16 else 32) == 0):
            if cinp != ldx:
                raise RuntimeError(f"{name}: input row width {ldx} != packed K {cinp}")
            ntiles = (self.P[level] + 127) // 128
            if self._statpart is None or self._statpart.numel() < ntiles * 2 * np_:
                self._statpart = torch.empty(max(ntiles * 2 * np_, (self.P[0] + 127) // 128 * 2 * 128), dtype=torch.float32,
                                             device=self.dev)
            with _timed("conv_fwd", 2.0 * B * h * w * cin * cout * ks * ks):
                H.call("nppc_conv_fwd_stats", self.prec, x, ldx, self.wf[name], y, ldy, self.p(name + ".bias"), B, h, w, cinp, cout,
                       np_, ks, self._statpart, H.stream())
            return (self._statpart, np_)
        if ks == 3 and cinp != ldx:
            raise RuntimeError(f"{name}: input row width {ldx} != packed K {cinp}")
        ss = self.ss[name] if fold else None
        if THIN and ks == 3 and cin <= 2 and cout <= 64 and cout % 8 == 0:
            # first layer: memory-bound, 18 MACs per output -- direct kernel on the fp32 parameter tensor (csrc/unet.hip)
            with _timed("conv_fwd", 2.0 * B * h * w * cin * cout * ks * ks):
                H.call("nppc_conv3x3_thin_fwd", self.prec, x, ldx, self.p(name + ".weight"), self.p(name + ".bias"), ss,
                       ss[cout:] if fold else None, LEAK, y, ldy, B, h, w, cin, cout, H.stream())
            return
        if THIN and ks == 1 and cin == 64 and cout <= 8 and not fold:
            with _timed("conv_fwd", 2.0 * B * h * w * cin * cout * ks * ks):
                H.call("nppc_conv1x1_thin_fwd", self.prec, x, ldx, self.p(name + ".weight"), self.p(name + ".bias"), y, ldy, B, h, w,
                       cin, cout, H.stream())
            return
        with _timed("conv_fwd", 2.0 * B * h * w * cin * cout * ks * ks):
            H.call("nppc_conv_fwd", self.prec, x, ldx, self.wf[name], y, ldy, self.p(name + ".bias"),
                   ss, ss[cout:] if fold else None, LEAK, B, h, w, cinp, cout, np_, ks, H.stream())

    def _bn_train(self, conv_name, raw, cout, level, y, ldy, parts=None):
        """batch statistics -> scale/shift (+ running update) -> LeakyReLU, output possibly a channel slice.
        parts: (per-tile column sums left by the convolution's epilogue, their row width) or None (a pass over `raw`)"""
        B = self.geo[0]
        h, w = self.lv[level]
        bn = conv_name[:-1] + str(int(conv_name[-1]) + 1)
        s = H.stream()
        st = self.st[:2 * cout]
        if parts is not None:
            if self._statscr is None:
                self._statscr = torch.empty(2 * 1024 * 128, dtype=torch.float64, device=self.dev)
            H.call("nppc_bn_stats_from_parts", parts[0], B, h, w, parts[1], cout, st, self._statscr, s)
        else:
            st.zero_()
            H.call("nppc_bn_stats", self.prec, raw.t, raw.ld, self.P[level], cout, st, s)
        ss = torch.empty(4 * cout, dtype=torch.float32, device=self.dev)
        H.call("nppc_bn_finalize", st, self.p(bn + ".weight"), self.p(bn + ".bias"), self.bnbuf(bn + ".running_mean"),
               self.bnbuf(bn + ".running_var"), ss, cout, float(B * h * w), BN_EPS, BN_MOMENTUM, 1, s)
        self.bnbuf(bn + ".num_batches_tracked").add_(1)
        H.call("nppc_bn_act", self.prec, raw.t, raw.ld, y, ldy, ss, cout, B, h, w, LEAK, s)
        return ss

    def _double_conv(self, blk, path, cin, cout, level, x, ldx, out, ldo, train, saved):
        """(conv3x3 -> BatchNorm -> LeakyReLU) x 2 (tmp_utils.py:8-37); `out` may be a channel slice of a concat buffer"""
        na, nb = f"{blk}.{path}.0", f"{blk}.{path}.3"
        if not train:
            mid = self.buf("act_a", level, cout)
            self._conv(na, cin, cout, 3, x, ldx, mid.t, cout, level, True)
            self._conv(nb, cout, cout, 3, mid.t, cout, out, ldo, level, True)
            return
        raw_a, act_a, raw_b = self.buf(blk + ".raw_a", level, cout), self.buf(blk + ".act_a", level, cout), self.buf(
            blk + ".raw_b", level, cout)
        pa = self._conv(na, cin, cout, 3, x, ldx, raw_a.t, cout, level, False, stats=True)
        ss_a = self._bn_train(na, raw_a, cout, level, act_a.t, cout, pa)
        pb = self._conv(nb, cout, cout, 3, act_a.t, cout, raw_b.t, cout, level, False, stats=True)
        ss_b = self._bn_train(nb, raw_b, cout, level, out, ldo, pb)
        saved[blk] = dict(x=x, ldx=ldx, raw_a=raw_a, act_a=act_a, raw_b=raw_b, out=out, ldo=ldo, ss_a=ss_a, ss_b=ss_b,
                          cin=cin, cout=cout, level=level, path=path)

    def _dropout(self, blk, x, ldx, cout, level, dropout):
        """nn.Dropout at the end of the block's double_conv (tmp_utils.py:28-29), in place; MC-dropout inference only"""
        if not dropout or blk not in DROP_BLOCKS:
            return
        keep = None
        if dropout.get("tap") is not None:
            keep = torch.empty(self.P[level] * cout, dtype=torch.uint8, device=self.dev)
            dropout["tap"][blk] = (keep, level, cout)
        H.call("nppc_dropout", self.prec, x, ldx, self.P[level], cout, float(dropout["p"]), int(dropout["seed"]),
               int(dropout["pass_id"]) * len(DROP_BLOCKS) + DROP_BLOCKS.index(blk), keep, H.stream())

    def forward(self, shape, maps, map_bstride, mask, out, out_pstride, xin=None, xin_bstride=0, train=False, dropout=None):
        """dropout: None or dict(p, seed, pass_id[, tap]) -- MC-dropout passes (utils.py:334-338, 561-577).
        maps: list of fp32 tensors holding [B][F*T] planes with batch stride `map_bstride` (the net's input
        channels); mask [B,T] fp32.  Writes out[(b*K+k)*out_pstride + f*T + t]:
          xin is None : U-Net(maps) * (1 - mask)                      (pc_wrapper.py:77-83)
          xin given   : xin * mask + U-Net(maps) * (1 - mask)         (RestorationWrapper, unet.py:299-312)"""
        B, F, T = shape
        self._setup(B, F, T)
        self.pack_weights(train)
        s = H.stream()
        x0 = self.buf("in", 0, IN_LD)
        for c, m in enumerate(maps):
            H.call("nppc_unet_stage_map", self.prec, m, map_bstride, x0.t, IN_LD, c, B, F, T, s)
        saved = {} if train else None
        cat = {0: self.buf("cat", 0, 128), 1: self.buf("cat", 1, 256), 2: self.buf("cat", 2, 512), 3: self.buf("cat", 3, 1024)}
        x, ldx = x0.t, IN_LD
        pools = {}
        for blk, path, cin, cout, level in PLAN[:5]:
            cin = self.in_ch if cin is None else cin
            if level > 0:
                ph, pw = self.lv[level - 1]
                pooled = self.buf("pool", level, cin)
                idx = None
                if train:
                    idx = pools[level] = self._pool_idx(level, self.P[level] * cin)
                else:
                    idx = self._scratch_idx(self.P[level] * cin)
                H.call("nppc_maxpool2", self.prec, x, ldx, pooled.t, cin, idx, cin, B, ph, pw, s)
                x, ldx = pooled.t, cin
            if level < 4:
                out_t, ldo = cat[level].t, cat[level].ld      # skip goes to channels [0, cout) of the concat buffer
            else:
                x5 = self.buf("x5", 4, cout)
                out_t, ldo = x5.t, cout
            self._double_conv(blk, path, cin, cout, level, x, ldx, out_t, ldo, train, saved)
            self._dropout(blk, out_t, ldo, cout, level, dropout)
            x, ldx = out_t, ldo
        prev, prev_c, prev_level = x, 512, 4
        for blk, path, cin, cout, level in PLAN[5:]:
            hi, wi = self.lv[prev_level]
            ht, wt = self.lv[level]
            cskip = cin - prev_c
            H.call("nppc_upsample2", self.prec, prev, prev_c, cat[level].at(cskip), cat[level].ld, prev_c, B, hi, wi, ht, wt, s)
            u = self.buf(blk + ".out", level, cout)
            self._double_conv(blk, path, cin, cout, level, cat[level].t, cat[level].ld, u.t, cout, train, saved)
            self._dropout(blk, u.t, cout, cout, level, dropout)
            prev, prev_c, prev_level = u.t, cout, level
        raw_o = self.buf("raw_o", 0, OUT_LD)
        self._conv("outc.conv", 64, self.out_ch, 1, prev, 64, raw_o.t, OUT_LD, 0, False)
        H.call("nppc_unet_out", self.prec, raw_o.t, OUT_LD, mask, xin, xin_bstride, out, out_pstride, self.out_ch, B, F, T,
               0 if xin is None else 1, s)
        if train:
            self.saved = None if dropout else dict(blocks=saved, pools=pools, u4=prev, mask=mask)

    def _pool_idx(self, level, n):
        key = ("pool_idx", level)
        t = self.bufs.get(key)
        if t is None or t.numel() < n:
            t = self.bufs[key] = torch.empty(n, dtype=torch.uint8, device=self.dev)
        return t

    def _scratch_idx(self, n):
        t = getattr(self, "_idx_scratch", None)
        if t is None or t.numel() < n:
            t = self._idx_scratch = torch.empty(n, dtype=torch.uint8, device=self.dev)
        return t

    # ------------------------------------------------------------------------------------------ backward
    def _grad_buffer(self):
        """two flat gradient buffers used alternately, so a gradient still referenced by the caller (or by a
        parameter's .grad) is never overwritten by the next backward"""
        self._gsel ^= 1
        if self._gbufs[self._gsel] is None:
            self._gbufs[self._gsel] = torch.zeros_like(self.fp.flat)
        self._grad = self._gbufs[self._gsel]
        self.fp.grad = self._grad
        return self._grad

    def _slabs(self, n):
        if self.slab is None or self.slab.numel() < n:
            self.slab = torch.empty(n, dtype=torch.float32, device=self.dev)
        return self.slab

    def _wgrad(self, name, cin, cout, ks, dy, lddy, x, ldx, level):
        """dW[co][ci][tap] = sum_p dY[p][co] * X[p + off(tap)][ci]"""
        B = self.geo[0]
        h, w = self.lv[level]
        M, N = lddy, ldx
        S = self.ksplit[level]
        s = H.stream()
        thin3, thin1 = THIN and ks == 3 and cin <= 2 and cout <= 64 and cout % 8 == 0, THIN and ks == 1 and cin == 64 and cout <= 8
        if thin3 or thin1:
            if self._thin_part is None:
                self._thin_part = torch.empty(H.conv_thin_part_elems(), dtype=torch.float32, device=self.dev)
            with _timed("conv_wgrad", 2.0 * B * h * w * cin * cout * ks * ks):
                H.call("nppc_conv3x3_thin_wgrad" if thin3 else "nppc_conv1x1_thin_wgrad", self.prec, dy, lddy, x, ldx, self._thin_part,
                       self.g(name + ".weight"), B, h, w, cin, cout, s)
            return
        slabs = self._slabs(ks * ks * S * rup(M, 128) * N)
        with _timed("conv_wgrad", 2.0 * B * h * w * cin * cout * ks * ks):
            H.call("nppc_conv_wgrad", self.prec, dy, lddy, x, ldx, slabs, M, N, B, h, w, ks, S, s)
            H.call("nppc_conv_wgrad_reduce", self.prec, slabs, S, M, N, self.g(name + ".weight"), cout, cin, ks, s)

    def _conv_bwd_data(self, name, cin, cout, ks, dy, lddy, dx, lddx, level):
        """dX = transposed convolution of dY: the forward kernel on the flipped / transposed pack"""
        B = self.geo[0]
        h, w = self.lv[level]
        cinp, np_ = self._dims(name, cin, cout)
        if THIN and ks == 1 and cin == 64 and cout <= 8:
            with _timed("conv_bwd_data", 2.0 * B * h * w * cin * cout * ks * ks):
                H.call("nppc_conv1x1_thin_bwd_data", self.prec, dy, lddy, self.p(name + ".weight"), dx, lddx, B, h, w, cin, cout,
                       H.stream())
            return
        with _timed("conv_bwd_data", 2.0 * B * h * w * cin * cout * ks * ks):
            H.call("nppc_conv_fwd", self.prec, dy, lddy, self.wb[name], dx, lddx, None, None, None, LEAK, B, h, w, np_, cin,
                   cinp, ks, H.stream())

    def _double_conv_bwd(self, blk, dyA, ldA, dyB, ldB, need_dx):
        sv = self.saved["blocks"][blk]
        B = self.geo[0]
        level, cin, cout, path = sv["level"], sv["cin"], sv["cout"], sv["path"]
        h, w = self.lv[level]
        na, nb = f"{blk}.{path}.0", f"{blk}.{path}.3"
        bna, bnb = f"{blk}.{path}.1", f"{blk}.{path}.4"
        s = H.stream()
        S = self.st[:2 * cout]
        draw = self.buf("draw", level, cout)
        H.call("nppc_bn_bwd", self.prec, dyA, ldA, dyB, ldB, sv["out"], sv["ldo"], sv["raw_b"].t, cout, sv["ss_b"], S, draw.t,
               cout, self.g(bnb + ".weight"), self.g(bnb + ".bias"), cout, B, h, w, LEAK, s)
        self._wgrad(nb, cout, cout, 3, draw.t, cout, sv["act_a"].t, cout, level)
        dact = self.buf("dact", level, cout)
        self._conv_bwd_data(nb, cout, cout, 3, draw.t, cout, dact.t, cout, level)
        H.call("nppc_bn_bwd", self.prec, dact.t, cout, None, 0, sv["act_a"].t, cout, sv["raw_a"].t, cout, sv["ss_a"], S, draw.t,
               cout, self.g(bna + ".weight"), self.g(bna + ".bias"), cout, B, h, w, LEAK, s)
        self._wgrad(na, cin, cout, 3, draw.t, cout, sv["x"], sv["ldx"], level)
        # the bias of a convolution that feeds a BatchNorm has an exactly zero gradient (the batch mean absorbs it);
        # the flat gradient buffer is zeroed at the start of backward, so nothing to write
        if not need_dx:
            return None
        dx = self.buf("dx", level, sv["ldx"])
        self._conv_bwd_data(na, cin, cout, 3, draw.t, cout, dx.t, sv["ldx"], level)
        return dx

    def backward(self, dout, dout_pstride):
        """dout: gradient wrt the masked output, planes [(b*K+k)*dout_pstride + f*T + t] -> flat parameter gradient"""
        B, F, T = self.geo
        s = H.stream()
        grad = self._grad_buffer()
        grad.zero_()
        K = self.out_ch
        d_rawo = self.buf("d_rawo", 0, OUT_LD)
        H.call("nppc_unet_out_bwd", self.prec, dout, dout_pstride, self.saved["mask"], d_rawo.t, OUT_LD, K, B, F, T, s)
        bias_acc = torch.zeros(OUT_LD, dtype=torch.float32, device=self.dev)
        H.colsum(self.prec, d_rawo.t, bias_acc, self.P[0], OUT_LD, OUT_LD, 0, 0, 1,
                 lambda n: torch.empty(n, dtype=torch.float32, device=self.dev), s)
        self.g("outc.conv.bias").copy_(bias_acc[:K])
        self._wgrad("outc.conv", 64, K, 1, d_rawo.t, OUT_LD, self.saved["u4"], 64, 0)
        d_u = self.buf("d_u", 0, 64)
        self._conv_bwd_data("outc.conv", 64, K, 1, d_rawo.t, OUT_LD, d_u.t, 64, 0)
        dy, ldy = d_u.t, 64
        skips = {}
        ups = list(PLAN[5:])
        for i in range(len(ups) - 1, -1, -1):
            blk, path, cin, cout, level = ups[i]
            dcat = self._double_conv_bwd(blk, dy, ldy, None, 0, True)       # [P_level][cin]
            prev_c = 512 if blk == "up1" else ups[i - 1][3]
            prev_level = level + 1
            cskip = cin - prev_c
            skips[level] = (dcat.t, dcat.ld)                                # channels [0, cskip): gradient of the skip
            hi, wi = self.lv[prev_level]
            ht, wt = self.lv[level]
            dprev = self.buf("d_up", prev_level, prev_c)
            H.call("nppc_upsample2_bwd", self.prec, dcat.at(cskip), dcat.ld, dprev.t, prev_c, prev_c, B, hi, wi, ht, wt, s)
            dy, ldy = dprev.t, prev_c
        # encoder, deepest first: gradient = (upsample / pooling branch) + (skip branch)
        dyA, ldA = dy, ldy                                                 # d x5
        dyB, ldB = None, 0
        for blk, path, cin, cout, level in reversed(PLAN[:5]):
            cin = self.in_ch if cin is None else cin
            dx = self._double_conv_bwd(blk, dyA, ldA, dyB, ldB, level > 0)
            if level == 0:
                break
            ph, pw = self.lv[level - 1]
            dpool = self.buf("d_pool", level - 1, cin)
            H.call("nppc_maxpool2_bwd", self.prec, dx.t, dx.ld, self.saved["pools"][level], dpool.t, cin, cin, B, ph, pw, s)
            dyA, ldA = dpool.t, cin
            dyB, ldB = skips[level - 1]
        return grad
