"""Launch sequences of the FullSubNet+-shaped nets on the HIP kernels (no torch compute ops).

`FSNEngine` owns the packed weights, the time-major activation buffers and the launch order for
one net (the frozen restorer: 3 input maps; the direction net: 6 maps, 2K outputs).  It mirrors
FullSubNet_Plus.forward (FullSubNet_plus/.../fullsubnet_plus.py:143-230) and
MultiDirectionFullSubNet_Plus.forward (nppc_audio/networks.py:63-163) stage by stage.
"""
import ctypes
import os

import numpy as np
import torch

from . import _hip as H
from .ops_lstm import (PackedLSTM, PackedLSTMBwd, ROW_PAD, WGRAD_SPLITS, bwd_head_fusable, forward_x_ld, lstm2_backward, lstm2_forward,
                       own_workspaces, padded_rows,
                       rows_view, workspace)

TCN_HIDDEN = 512
TCN_DILATIONS = (1, 2, 5, 9, 1, 2, 5, 9)
BRANCHES = ("", "_real", "_imag")   # channel_attention{,_real,_imag} / fb_model{,_real,_imag}
EPI_PLAIN, EPI_PRELU_STATS, EPI_RESIDUAL, EPI_RELU, EPI_PLAIN_F32, EPI_MASK_POS = 0, 1, 2, 3, 4, 5


def rup(a, b):
    return (a + b - 1) // b * b


TCN_WGRAD_ON_SIDE = os.environ.get("NPPC_TCN_WGRAD_SIDE", "1") != "0"    # A/B switch (tools/diag)
TCN_WGRAD_SPLITS = int(os.environ.get("NPPC_TCN_S2", "8"))            # K-slices of the TCN weight-gradient GEMMs
FC_WGRAD_AT_END = os.environ.get("NPPC_FC_AT_END", "1") != "0"       # fc_output_layer weight gradient at the end of the main chain (A/B switch)
TCN_WGRAD_MAIN_BLOCKS = int(os.environ.get("NPPC_TCN_WGRAD_MAIN_BLOCKS", "1"))   # TCN blocks (0 .. n-1) whose weight gradients stay on the main queue
MID_FINISH_DEFER = os.environ.get("NPPC_MID_FINISH_DEFER", "1") != "0"   # ONE finishing launch for the eight fused middle backwards (A/B switch)
COLSUM_IN_GEMM = os.environ.get("NPPC_COLSUM_IN_GEMM", "1") != "0"   # sconv bias gradients from the producing GEMM's epilogue (A/B switch)
TN_PAIRED = os.environ.get("NPPC_TN_PAIRED", "1") != "0"              # one pass over the gate gradients per LSTM layer (A/B switch)
FUSED_HEAD = int(os.environ.get("NPPC_FUSED_HEAD", "3"))    # 0: head kernels; 1: fused in the inference forward; 2: + training forward; 3: + backward


class FlatParams:
    """All parameters of a module re-homed into ONE fp32 device buffer (named_parameters order).

    The three full-band branches are identical sub-trees laid out back to back, so any parameter of
    branch z sits at a constant element stride from branch 0: kernels batch the branches over
    blockIdx.z with that single stride.  Parameters stay individual nn.Parameters (views), so
    state_dict()/optimizers see the reference names and shapes.
    """

    def __init__(self, module, device):
        named = list(module.named_parameters())
        total = sum(p.numel() for _, p in named)
        self.flat = torch.empty(total, dtype=torch.float32, device=device)
        self.grad = None
        self.off = {}
        o = 0
        for n, p in named:
            k = p.numel()
            self.flat[o:o + k].copy_(p.detach().reshape(-1))
            p.data = self.flat[o:o + k].view(p.shape)
            self.off[n] = (o, tuple(p.shape))
            o += k
        self.named = named

    def view(self, name):
        o, shp = self.off[name]
        return self.flat[o:o + int(np.prod(shp))].view(shp)

    def gview(self, name):
        o, shp = self.off[name]
        return self.grad[o:o + int(np.prod(shp))].view(shp)

    def version(self):
        """changes whenever any parameter was written: in-place torch ops (load_state_dict, torch optimizers) bump the
        parameter's own counter, the HIP optimizers bump the flat buffer's (trainer.HipAdam / FlatAdamStepper)"""
        return (self.flat._version, sum(p._version for _, p in self.named))

    def ensure_grad(self):
        if self.grad is None:
            self.grad = torch.zeros_like(self.flat)
        return self.grad

    def attention_stride(self):
        """element stride between the three (equally shaped) attention layers channel_attention{,_real,_imag}"""
        names = [n for n, _ in self.named if n.startswith("channel_attention.")]
        a = self.off["channel_attention." + names[0].split(".", 1)[1]][0]
        strides = set()
        for n in names:
            leaf = n.split(".", 1)[1]
            o0, o1, o2 = (self.off[f"channel_attention{br}.{leaf}"][0] for br in BRANCHES)
            strides.add(o1 - o0)
            strides.add(o2 - o1)
        assert len(strides) == 1 and a >= 0
        return strides.pop()

    def branch_stride(self):
        a = self.off["fb_model.fc_output_layer.weight"][0]
        b = self.off["fb_model_real.fc_output_layer.weight"][0]
        c = self.off["fb_model_imag.fc_output_layer.weight"][0]
        assert b - a == c - b > 0
        a2 = self.off["fb_model.sequence_model.0.conv1x1.weight"][0]
        b2 = self.off["fb_model_real.sequence_model.0.conv1x1.weight"][0]
        assert b2 - a2 == b - a
        return b - a


def unfold_multiplicity(F, nb):
    """mult[f] = number of (bin, tap) pairs whose reflected source bin is f (sum of the unfolded block)."""
    m = np.zeros(F, np.float32)
    for f in range(F):
        for j in range(-nb, nb + 1):
            i = f + j
            if i < 0:
                i = -i
            if i >= F:
                i = 2 * (F - 1) - i
            m[i] += 1
    return m


class FSNEngine:
    # engines with side-stream work (weight gradients of a backward in defer_join mode, the re-pack of the updated weights)
    # that the main stream has not been made to wait for yet.  A cooperative LSTM kernel needs every CU for its own
    # workgroups, so NOTHING of ours may be in flight on another stream when one is launched: every engine joins ALL of them
    # right before its LSTM launches (the trainer's pre_lstm_hook normally has done so already), and its OWN pending work
    # at the top of forward / backward, before any kernel reads the packed weights the side stream may still be rewriting.
    import weakref as _weakref
    _unjoined = _weakref.WeakValueDictionary()

    def __init__(self, flat, *, num_freqs, n_maps, out_size, sb_neighbors, look_ahead, sb_hidden, groups, kersize,
                 prec, trainable):
        self.fp = flat
        self.F, self.nm, self.O = num_freqs, n_maps, out_size
        self.nb, self.la, self.Hd, self.G = sb_neighbors, look_ahead, sb_hidden, groups
        self.ks = tuple(kersize)
        self.prec = prec
        self.dt = H.dtype_of(prec)
        self.trainable = trainable
        self.C = n_maps * num_freqs
        self.ldC = rup(self.C, 64)
        self.ldF = rup(num_freqs, 64)
        self.KC = self.ldC                     # K of the input GEMMs: the zero-padded row width (multiple of 64: LDS-staged GEMM path)
        self.dev = flat.flat.device
        self.I = 2 * sb_neighbors + 1 + 3
        self.sP = flat.branch_stride()
        self.sAtt = flat.attention_stride()
        self.packed_version = None
        self._side = None                     # side stream for the LSTM weight-gradient GEMMs (backward)
        self._gen = 0                         # train-forward generation: backward must consume the LATEST train forward
        self.last_train = None
        self.grad_range_hook = None           # fn(flat_grad, lo, hi): range final for this step (dp.FlatGradientReducer)
        self.early_buckets_ok = self._check_bucket_layout() if trainable else False
        self.pre_lstm_hook = None             # one-shot callback run right before this engine's next LSTM launch (trainer)
        self.defer_join = False               # backward leaves the side-stream join to the caller (join_side)
        self.join_pending = False
        self.bufs = {}
        own_workspaces(self)                  # step-persistent workspaces keyed by id(self) die with the engine
        self.lstm = PackedLSTM(self.I, self.Hd, prec, self.dev)
        self.KX = self.lstm.kx
        self.mult = torch.from_numpy(unfold_multiplicity(self.F, self.nb)).to(self.dev)
        C, ldC, ldF = self.C, self.ldC, self.ldF
        self.W1p = torch.zeros(8, 3, TCN_HIDDEN, ldC, dtype=self.dt, device=self.dev)
        self.W2p = torch.zeros(8, 3, ldC, TCN_HIDDEN, dtype=self.dt, device=self.dev)      # sconv weights x norm2.weight
        self.u2 = torch.zeros(8, 3, ldC, dtype=torch.float32, device=self.dev)            # norm2.bias through sconv + sconv.bias
        self.v2 = torch.zeros(8, 3, ldC, dtype=torch.float32, device=self.dev)            # row sums of W2p
        self.Wfcp = torch.zeros(3, ldF, ldC, dtype=self.dt, device=self.dev)
        self.Opad = rup(self.O, 16)
        self.Whp = torch.zeros(self.Opad, self.Hd, dtype=self.dt, device=self.dev)
        if trainable:
            if self.O > 32:
                raise NotImplementedError("n_directions <= 16 is what the head-backward kernel is built for")
            self.lstm_bwd = PackedLSTMBwd(self.I, self.Hd, prec, self.dev)
            self.W1T = torch.zeros(8, 3, ldC, TCN_HIDDEN, dtype=self.dt, device=self.dev)   # [c][k] = W1[k][c]
            self.W2T = torch.zeros(8, 3, TCN_HIDDEN, ldC, dtype=self.dt, device=self.dev)   # [k][c] = W2[c][k]
            self.WfcT = torch.zeros(3, ldC, ldF, dtype=self.dt, device=self.dev)            # [c][f] = Wfc[f][c]
            self.WhT = torch.zeros(self.Hd, 32, dtype=self.dt, device=self.dev)             # [u][o] = Wh[o][u]
            self.gbuf = [None, None]

    def _check_bucket_layout(self):
        """the early gradient buckets (backward: the sub-band tail of the flat buffer; TCN blocks 4..7 of a branch) are
        handed to the exchange as RANGES: true only when the flat buffer really has that layout -- everything from
        sb_model.sequence_model.weight_ih_l0 to the end belongs to sb_model, and the range conv1x1.weight of block 4 ..
        sconv.bias of block 7 of a branch holds nothing but that branch's blocks 4..7.  A subclass that registers parameters
        elsewhere falls back to the single exchange in `finish` (no early bucket is announced)."""
        off = self.fp.off
        try:
            t0 = off["sb_model.sequence_model.weight_ih_l0"][0]
            if any(o >= t0 and not n.startswith("sb_model.") for n, (o, _) in off.items()):
                return False
            for br in BRANCHES:
                a = off[f"fb_model{br}.sequence_model.4.conv1x1.weight"][0]
                b, shp = off[f"fb_model{br}.sequence_model.7.sconv.bias"]
                b += int(np.prod(shp))
                pre = tuple(f"fb_model{br}.sequence_model.{i}." for i in (4, 5, 6, 7))
                if any(a <= o < b and not n.startswith(pre) for n, (o, _) in off.items()):
                    return False
        except KeyError:
            return False
        return True

    # ------------------------------------------------------------------ weights
    def p(self, name):
        return self.fp.view(name)

    def pack_weights(self, force=False):
        ver = self.fp.version()
        if not force and self.packed_version == ver:
            return
        s = H.stream()
        C, ldC, ldF = self.C, self.ldC, self.ldF
        # the 8 TCN blocks x 3 branches are equally shaped and sit at constant strides in the flat buffer: one launch each
        lay = self.fp.off["fb_model.sequence_model.1.conv1x1.weight"][0] - self.fp.off["fb_model.sequence_model.0.conv1x1.weight"][0]
        brs = self.fp.branch_stride()
        if self.packed_version is None:            # first pack: check the constant-stride assumption once
            for i in range(8):
                for z, br in enumerate(BRANCHES):
                    for leaf in ("conv1x1.weight", "sconv.weight", "conv1x1.bias", "sconv.bias", "norm1.weight", "norm1.bias",
                                 "norm2.weight", "norm2.bias", "depthwise_conv.weight", "depthwise_conv.bias", "prelu1.weight",
                                 "prelu2.weight"):
                        assert (self.fp.off[f"fb_model{br}.sequence_model.{i}.{leaf}"][0]
                                == self.fp.off[f"fb_model.sequence_model.0.{leaf}"][0] + i * lay + z * brs)
        w1, w2 = self.p("fb_model.sequence_model.0.conv1x1.weight"), self.p("fb_model.sequence_model.0.sconv.weight")

        def packed(src, dst, N, K, Npad, ldd, tr):
            H.call("nppc_pack_matrix_batched", self.prec, src, dst, N, K, Npad, ldd, tr, 8, 3, lay, brs, dst.stride(0),
                   dst.stride(1), s)

        packed(w1, self.W1p, TCN_HIDDEN, C, TCN_HIDDEN, ldC, 0)
        # sconv with norm2 folded in (csrc/tcn.hip: EPI_RESIDUAL_GN): the normalised activation is never materialised
        H.call("nppc_tcn_pack_sconv", self.prec, w2, self.p("fb_model.sequence_model.0.norm2.weight"),
               self.p("fb_model.sequence_model.0.norm2.bias"), self.p("fb_model.sequence_model.0.sconv.bias"), self.W2p, self.u2,
               self.v2, C, TCN_HIDDEN, ldC, TCN_HIDDEN, 8, 3, lay, brs, self.W2p.stride(0), self.W2p.stride(1), s)
        for z, br in enumerate(BRANCHES):
            H.call("nppc_pack_matrix", self.prec, self.p(f"fb_model{br}.fc_output_layer.weight"), self.Wfcp[z], self.F, C,
                   ldF, ldC, 0, s)
        H.call("nppc_pack_matrix", self.prec, self.p("sb_model.fc_output_layer.weight"), self.Whp, self.O, self.Hd,
               self.Opad, self.Hd, 0, s)
        if self.trainable:
            packed(w1, self.W1T, C, TCN_HIDDEN, ldC, TCN_HIDDEN, 1)
            packed(w2, self.W2T, TCN_HIDDEN, C, TCN_HIDDEN, ldC, 1)
            for z, br in enumerate(BRANCHES):
                H.call("nppc_pack_matrix", self.prec, self.p(f"fb_model{br}.fc_output_layer.weight"), self.WfcT[z], C,
                       self.F, ldC, ldF, 1, s)
            H.call("nppc_pack_matrix", self.prec, self.p("sb_model.fc_output_layer.weight"), self.WhT, self.Hd, self.O,
                   self.Hd, 32, 1, s)
            q = "sb_model.sequence_model."
            self.lstm_bwd.pack(*[self.p(q + n) for n in ("weight_ih_l0", "weight_hh_l0", "weight_ih_l1", "weight_hh_l1")])
        q = "sb_model.sequence_model."
        self.lstm.pack(*[self.p(q + n) for n in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0",
                                                  "weight_ih_l1", "weight_hh_l1", "bias_ih_l1", "bias_hh_l1")])
        self.packed_version = ver

    # ------------------------------------------------------------------ buffers
    def _buffers(self, B, T, train):
        key = (B, T, train)
        if key in self.bufs:
            return self.bufs[key]
        dev, dt = self.dev, self.dt
        Tv = T + self.la
        Tp = rup(Tv, 128)
        nblk = 8 if train else 1
        d = dict(B=B, T=T, Tv=Tv, Tp=Tp)
        d["rs"] = torch.empty(3 * self.nm, B, self.F, dtype=torch.float64, device=dev)
        d["scale"] = torch.empty(3, self.nm, B, self.F, dtype=torch.float32, device=dev)
        d["X"] = torch.zeros(9 if train else 3, 3, B, Tp, self.ldC, dtype=dt, device=dev)   # X[0] = TCN input (kept)
        d["y1"] = torch.zeros(nblk, 3, B, Tp, TCN_HIDDEN, dtype=dt, device=dev)
        d["y2"] = torch.zeros(nblk, 3, B, Tp, TCN_HIDDEN, dtype=dt, device=dev)
        d["a2"] = torch.zeros(3, B, Tp, TCN_HIDDEN, dtype=dt, device=dev)
        d["stats"] = torch.zeros(8, 2, 3, B, 2, dtype=torch.float64, device=dev)
        d["fb"] = torch.zeros(3, B, Tp, self.ldF, dtype=dt, device=dev)
        if self.nm == 2:
            d["rawmag"] = torch.zeros(B, Tp, self.ldF, dtype=dt, device=dev)
        d["sbscale"] = torch.empty(B, dtype=torch.float32, device=dev)
        d["sbwork"] = torch.zeros(2 * B, dtype=torch.float64, device=dev)     # nppc_subband_mean: sums + arrival counters
        G = self.G if B > 1 else 1
        Fo = self.F if G <= 1 else (self.F - self.F % G) // G
        d["G"], d["Fo"], d["Nseq"] = G, Fo, B * Fo
        d["x_rows"] = torch.zeros(padded_rows(Tv * B * Fo, B * Fo), self.KX, dtype=dt, device=dev)   # GEMM operand
        d["x_tm"] = rows_view(d["x_rows"], Tv, B * Fo)
        if train:
            d["tsse_saved"] = {k: torch.empty(3, self.nm, B, *shp, dtype=torch.float32, device=dev)
                               for k, shp in (("ns", ()), ("pre", (self.F, 3)), ("sq", (self.F,)),
                                              ("h1", (self.F // 2,)), ("sg", (self.F,)))}
        self.bufs[key] = d
        return d

    # ------------------------------------------------------------------ forward
    def forward(self, maps, train=False, mtile=None):
        """maps: 3*n_maps tensors [B,1,F,T] fp32 ordered (mag, real, imag)[, (enh mag, real, imag)].
        Returns the net output [B', O, F', T] fp32 (B' in drop-band order) and keeps what backward needs."""
        H.require_gpu()
        assert len(maps) == 3 * self.nm
        for m in maps:
            assert m.dim() == 4, "inputs are [B, 1, F, T]"      # fullsubnet_plus.py:157
            assert m.shape[1] == 1 and m.shape[2] == self.F
        B, _, F, T = maps[0].shape
        if B > 1:
            assert B > self.G, f"Batch size = {B}, num_groups = {self.G}."   # feature.py:263
        self.join_side()                      # own side-stream work (re-pack of the updated weights) before anything reads it
        self.pack_weights()
        d = self._buffers(B, T, train)
        s = H.stream()
        Tv, Tp, prec = d["Tv"], d["Tp"], self.prec
        C, ldC, ldF, sP = self.C, self.ldC, self.ldF, self.sP
        R = B * Tp
        maps = [m.contiguous().float() for m in maps]
        d["maps"] = maps if train else None
        sv = d.get("tsse_saved")
        # 1-3: laplace norm + TSSE attention scale, transposed into the TCN input: all 3 * n_maps maps in three launches
        att = "channel_attention."
        sv_ = [sv[k] if sv else None for k in ("ns", "pre", "sq", "h1", "sg")]
        H.call("nppc_tsse_fwd_maps", prec, H.ptr_array(maps), 3 * self.nm, d["rs"],
               self.p(att + "smallConv1d.0.weight"), self.p(att + "smallConv1d.0.bias"),
               self.p(att + "middleConv1d.0.weight"), self.p(att + "middleConv1d.0.bias"),
               self.p(att + "largeConv1d.0.weight"), self.p(att + "largeConv1d.0.bias"), self.ks[0], self.ks[1], self.ks[2],
               self.p(att + "feature_concate_fc.weight"), self.p(att + "feature_concate_fc.bias"),
               self.p(att + "fc1.weight"), self.p(att + "fc1.bias"), self.p(att + "fc2.weight"), self.p(att + "fc2.bias"),
               self.sAtt, d["scale"], *sv_, d["X"][0], R * ldC, B, F, T, self.la, Tp, ldC, s)
        if self.nm == 2:
            H.call("nppc_scale_transpose", prec, maps[0], None, d["rawmag"], B, F, T, Tp, ldF, 0, s)
        # 4: eight TCN blocks, the three branches batched over blockIdx.z
        d["stats"].zero_()
        sAct = B * Tp * TCN_HIDDEN
        for i, dil in enumerate(TCN_DILATIONS):
            pre = f"fb_model.sequence_model.{i}."
            xi, xo = (i, i + 1) if train else ((0 if i == 0 else 1 + (i - 1) % 2), 1 + i % 2)
            bi = i if train else 0
            Xin, Xout = d["X"][xi], d["X"][xo]
            y1, y2 = d["y1"][bi], d["y2"][bi]
            st1, st2 = d["stats"][i, 0], d["stats"][i, 1]
            H.call("nppc_gemm_nt", prec, EPI_PRELU_STATS, Xin, ldC, R * ldC, self.W1p[i], ldC, TCN_HIDDEN * ldC,
                   y1, TCN_HIDDEN, sAct, self.p(pre + "conv1x1.bias"), sP, None, 0, 0, self.p(pre + "prelu1.weight"), sP,
                   st1, B * 2, R, TCN_HIDDEN, self.KC, Tp, Tv, TCN_HIDDEN, 0, 3, 1, s)
            H.call("nppc_tcn_dwconv", prec, y1, y2, st1, st2, self.p(pre + "norm1.weight"), self.p(pre + "norm1.bias"),
                   self.p(pre + "depthwise_conv.weight"), self.p(pre + "depthwise_conv.bias"),
                   self.p(pre + "prelu2.weight"), B, TCN_HIDDEN, TCN_HIDDEN, Tp, Tv, dil, 1e-8, sAct, B * 2, sP, 3, s)
            H.call("nppc_gemm_nt_gn", prec, y2, TCN_HIDDEN, sAct, self.W2p[i], TCN_HIDDEN, ldC * TCN_HIDDEN, Xout, ldC, R * ldC,
                   self.u2[i], self.v2[i], ldC, Xin, ldC, R * ldC, st2, B * 2, float(TCN_HIDDEN * Tv), 1e-8, R, ldC,
                   TCN_HIDDEN, Tp, Tv, C, 3, s)
        Xlast = d["X"][8 if train else 2]
        # 5: trailing ReLU + Linear(C -> F) + ReLU
        H.call("nppc_gemm_nt", prec, EPI_RELU, Xlast, ldC, R * ldC, self.Wfcp, ldC, ldF * ldC, d["fb"], ldF, R * ldF,
               self.p("fb_model.fc_output_layer.bias"), sP, None, 0, 0, None, 0, None, 0, R, ldF, self.KC, Tp, Tv, F, 1,
               3, 1, s)
        # 6: sub-band unfold + concat + norm + drop-band, staged time-major for the LSTM
        if self.nm == 1:
            src, ldS = d["X"][0, 0], ldC          # attention-scaled, normalised magnitude (fullsubnet_plus.py:203)
        else:
            src, ldS = d["rawmag"], ldF           # RAW padded magnitude (networks.py:133)
        H.call("nppc_subband_mean", prec, src, ldS, d["fb"], ldF, R * ldF, self.mult, d["sbscale"], d["sbwork"], B, F, Tp,
               Tv, self.I, s)
        # mtile None: cooperative kernel when the shape allows; it also takes the output head to fuse (bf16 pair kernel)
        head = (self.Whp, self.O) if (FUSED_HEAD > int(train) and prec == H.PREC_BF16 and self.Opad == 16) else None
        # the frozen net's fused-head inference launch reads rows of 40 columns (the same buffer, viewed narrower): the
        # staging kernel writes 37 % less, the recurrent kernel fetches 80 instead of 128 bytes per sequence and step
        x_tm = d["x_tm"]
        xld = forward_x_ld(d["Nseq"], self.lstm, train, head) if mtile is None else self.KX
        if xld != self.KX:
            x_tm = d["x_rows"].view(-1)[:Tv * d["Nseq"] * xld].view(Tv, d["Nseq"], xld)
        H.call("nppc_subband_stage", prec, src, ldS, d["fb"], ldF, R * ldF, d["sbscale"], x_tm, B, F, Tp, Tv,
               self.nb, self.G, xld, int(train), s)
        # 7: two-layer LSTM over T' steps for the B*F' sequences
        if self.pre_lstm_hook is not None:
            # everything another stream still has in flight must be joined BEFORE a cooperative (CU-pair) kernel goes out:
            # its workgroups need the whole chip to themselves.  The trainer parks the previous step's deferred tail
            # (side-stream weight gradients, gradient exchange, Adam) here: it overlaps this net's small front kernels.
            hook, self.pre_lstm_hook = self.pre_lstm_hook, None
            hook()
        FSNEngine.join_all()                  # (a no-op after the trainer's hook; any other caller gets the join for free)
        lo = lstm2_forward(x_tm, self.lstm, train, mtile, head=head)
        d["lstm"] = lo
        # 8: Linear(H -> O) + re-layout + look-ahead crop
        out = torch.empty(B, self.O, d["Fo"], T, dtype=torch.float32, device=self.dev)
        if "head_partial" in lo:
            H.call("nppc_sb_head_finalize", lo["head_partial"], lo["head_partial"].shape[0], self.p("sb_model.fc_output_layer.bias"), out, d["Nseq"], Tv,
                   self.la, self.O, d["Fo"], s)
        else:
            H.call("nppc_sb_head", prec, lo["h2"], self.Whp, self.p("sb_model.fc_output_layer.bias"), out, d["Nseq"], Tv,
                   self.la, self.Hd, self.O, d["Fo"], s)
        self.last = d
        if train:
            self._gen += 1
            d["gen"] = self._gen
            d["consumed"] = False            # the buffer dict is reused by every step of this shape
            self.last_train = d
        return out

    # ------------------------------------------------------------------ backward (direction net)
    def _grad_buffer(self):
        """Flat gradient buffer that does NOT alias the parameters' current .grad (so autograd's accumulate
        semantics stay right whether or not the caller cleared the grads)."""
        first = self.fp.named[0][1]
        for k in (0, 1):
            if self.gbuf[k] is None:
                self.gbuf[k] = torch.empty_like(self.fp.flat)
            if first.grad is None or first.grad.data_ptr() != self.gbuf[k].data_ptr():
                self.fp.grad = self.gbuf[k]
                return self.gbuf[k]
        raise RuntimeError("unreachable")

    def g(self, name):
        return self.fp.gview(name)

    def _wgrad(self, AT, lda, sA, BT, ldb, sB, rows, ncolsN, K, S, dst_name_or_tensor, dst_ld, out_rows, ncols, slab,
               batch=1, sDst=0, permH=0, col0=0, accumulate=0):
        """dst[r][c] = sum_k AT[r][k] * BT[c][k]  via split-K fp32 slabs + reduction (one launch pair)."""
        s = H.stream()
        dst = self.g(dst_name_or_tensor) if isinstance(dst_name_or_tensor, str) else dst_name_or_tensor
        bk = 64 if self.prec == H.PREC_BF16 else 32
        if batch == 1 and rows % 128 == 0 and ncolsN % 128 == 0 and K % (bk * S) == 0:
            H.call("nppc_gemm_nt_splitk", self.prec, AT, lda, BT, ldb, slab, ncolsN, rows, ncolsN, K, S, s)
        else:
            H.call("nppc_gemm_nt", self.prec, EPI_PLAIN_F32, AT, lda, sA, BT, ldb, sB, slab, ncolsN, rows * ncolsN, None,
                   0, None, 0, 0, None, 0, None, 0, rows, ncolsN, K, rows, rows, ncolsN, 0, batch, S, s)
        H.call("nppc_reduce_slabs", slab, S, rows * ncolsN, ncolsN, dst, dst_ld, out_rows, col0, ncols, permH, accumulate,
               S * rows * ncolsN, sDst, batch, s)

    def _lstm_wgrad(self, dg1, dg2, x_rows, h1_rows, h2_rows, Tv, Nseq, head=None, guards=None):
        """LSTM weight / bias gradients from the row-major gate gradients (runs on the side stream).
        head = (dyt_rows [Rpad][16] bf16, O): also the output head's weight / bias gradients, as one more TN product
        h2^T . dY (the 16-column dY rows are read as a 64-column operand: the extra columns are ignored)."""
        s = H.stream()
        dt, dev, Hd, I, KX = self.dt, self.dev, self.Hd, self.I, self.KX
        K4 = 4 * Hd
        q = "sb_model.sequence_model."
        Rr = Tv * Nseq
        ws = lambda name, shape, dtype=dt, zero=False: workspace(("eng", id(self), name), shape, dtype, dev, zero)
        jobs = [   # (dgates, row offset of dgates, input rows, input width, destination(s))
            (dg1, 0, x_rows, KX, ("ih", q + "weight_ih_l0", q + "bias_ih_l0", q + "bias_hh_l0")),
            (dg2, 0, x_rows, KX, ("bias", None, q + "bias_ih_l1", q + "bias_hh_l1")),
            (dg2, 0, h1_rows, Hd, ("w", q + "weight_ih_l1")),
        ]
        if Tv > 1:
            jobs += [(dg1, Nseq, h1_rows, Hd, ("w", q + "weight_hh_l0")), (dg2, Nseq, h2_rows, Hd, ("w", q + "weight_hh_l1"))]

        def scatter(slab, S, rows, ncolsN, dest):
            kind = dest[0]
            if kind in ("ih", "w"):
                ncol = I if kind == "ih" else Hd
                H.call("nppc_reduce_slabs", slab, S, rows * ncolsN, ncolsN, self.g(dest[1]), ncol, K4, 0, ncol, Hd, 0, 0, 0, 1, s)
            if kind in ("ih", "bias"):
                H.call("nppc_reduce_slabs", slab, S, rows * ncolsN, ncolsN, self.g(dest[2]), 1, K4, I, 1, Hd, 0, 0, 0, 1, s)
                self.g(dest[3]).copy_(self.g(dest[2]))

        if self.prec == H.PREC_BF16 and K4 % 128 == 0 and Hd % 64 == 0 and KX % 64 == 0:
            # 64 K-slices: (4H/128) * (H/128) * 64 = 2304 workgroups for 512 slots (16 slices = 576 left the second
            # round of workgroups 1/8 full: 481 -> 674 TFLOP/s on the H x 4H products, tools/bench_tn.py)
            S = WGRAD_SPLITS
            assert 64 * S <= ROW_PAD                 # operand buffers are padded by ROW_PAD rows (ops_lstm.padded_rows)
            slab = ws("slab", (S * K4 * max(Hd, KX),), torch.float32)
            # the layer-2 bias gradient (column sums of dg2) comes out of the W_ih_l1 product as row sums of its A operand
            # (LDS-DMA kernel, csrc/tcn.hip) instead of out of a 64-column product of its own that re-read all of dg2
            fold_bias = K4 % 256 == 0 and Hd % 128 == 0 and (K4 // 256) * (Hd // 128) * S >= 256
            # ONE pass over each layer's gate gradients for both of its weight gradients (`nppc_gemm_tn_splitk2`):
            #   layer 1: dg1^T . [h1_{t-1} | x_t]   (x's spare column carries the bias gradient)
            #   layer 2: dg2^T . [h1_t | h2_{t-1}]  (+ row sums of dg2 = the bias gradient)
            # h_{t-1} is the guard view of the h rows (N zero rows in front), so the gate gradients are not shifted
            if (guards is not None and TN_PAIRED and fold_bias and Hd % 192 == 0 and KX == 64 and Tv > 1
                    and (K4 // 256) * (Hd // 192 + 1) * S >= 256):
                h1_guard, h2_guard = guards
                rows = (Rr + 64 * S - 1) // (64 * S) * (64 * S)
                slab = ws("slab2", (S * K4 * 2 * Hd,), torch.float32)
                rsum = ws("rowsum", (S, K4), torch.float32)
                H.call("nppc_gemm_tn_splitk2", dg1, K4, h1_guard, Hd, Hd, x_rows, KX, KX, slab, Hd + KX, K4, rows, S, None, s)
                n1 = K4 * (Hd + KX)
                H.call("nppc_reduce_slabs", slab, S, n1, Hd + KX, self.g(q + "weight_hh_l0"), Hd, K4, 0, Hd, Hd, 0, 0, 0, 1, s)
                H.call("nppc_reduce_slabs", slab, S, n1, Hd + KX, self.g(q + "weight_ih_l0"), I, K4, Hd, I, Hd, 0, 0, 0, 1, s)
                H.call("nppc_reduce_slabs", slab, S, n1, Hd + KX, self.g(q + "bias_ih_l0"), 1, K4, Hd + I, 1, Hd, 0, 0, 0, 1, s)
                self.g(q + "bias_hh_l0").copy_(self.g(q + "bias_ih_l0"))
                H.call("nppc_gemm_tn_splitk2", dg2, K4, h1_rows, Hd, Hd, h2_guard, Hd, Hd, slab, 2 * Hd, K4, rows, S, rsum, s)
                n2 = K4 * 2 * Hd
                H.call("nppc_reduce_slabs", slab, S, n2, 2 * Hd, self.g(q + "weight_ih_l1"), Hd, K4, 0, Hd, Hd, 0, 0, 0, 1, s)
                H.call("nppc_reduce_slabs", slab, S, n2, 2 * Hd, self.g(q + "weight_hh_l1"), Hd, K4, Hd, Hd, Hd, 0, 0, 0, 1, s)
                H.call("nppc_reduce_slabs", rsum, S, K4, 1, self.g(q + "bias_ih_l1"), 1, K4, 0, 1, Hd, 0, 0, 0, 1, s)
                self.g(q + "bias_hh_l1").copy_(self.g(q + "bias_ih_l1"))
                jobs = []
            for dg, off, inp, width, dest in jobs:
                rows = (Rr - off + 64 * S - 1) // (64 * S) * (64 * S)
                if fold_bias and dest[0] == "bias":
                    continue
                if fold_bias and dest == ("w", q + "weight_ih_l1"):
                    rsum = ws("rowsum", (S, K4), torch.float32)
                    H.call("nppc_gemm_tn_splitk_rowsum", dg.view(-1)[off * K4:], K4, inp, width, slab, width, K4, width, rows, S,
                           rsum, s)
                    H.call("nppc_reduce_slabs", rsum, S, K4, 1, self.g(q + "bias_ih_l1"), 1, K4, 0, 1, Hd, 0, 0, 0, 1, s)
                    self.g(q + "bias_hh_l1").copy_(self.g(q + "bias_ih_l1"))
                else:
                    H.call("nppc_gemm_tn_splitk", dg.view(-1)[off * K4:], K4, inp, width, slab, width, K4, width, rows, S, s)
                scatter(slab, S, K4, width, dest)
            if head is not None:
                dyt_rows, O = head
                rows = (Rr + 64 * S - 1) // (64 * S) * (64 * S)
                H.call("nppc_gemm_tn_splitk", h2_rows, Hd, dyt_rows, 16, slab, 64, Hd, 64, rows, S, s)
                H.call("nppc_reduce_slabs_t", slab, S, Hd * 64, 64, self.g("sb_model.fc_output_layer.weight"), Hd, O, Hd, 0, 0, 1, s)
                H.colsum(self.prec, dyt_rows, self.g("sb_model.fc_output_layer.bias"), Rr, O, 16, 0, 0, 1,
                         lambda n: ws("cs_head", (n,), torch.float32), s)
            return
        # generic path (fp32 parity mode, small hidden sizes): transposed copies + the NT split-K GEMM
        bk = 64 if self.prec == H.PREC_BF16 else 32
        # split-K: the products are (4H x H) outputs over T'*N rows -- 72 workgroups at C2 without it, each walking a million
        # rows (the fp32 step spent 372 of its 556 ms in these five launches, profiles/r03_bench_c2_fp32_kernel_stats.csv)
        Sg = max(1, min(64, Rr // (16 * bk)))
        Rp = rup(Rr, bk * Sg)
        K4p, Hp, KXp = rup(K4, 128), rup(Hd, 128), rup(KX, 128)
        dgT = [ws("dg1T", (K4p, Rp), zero=True), ws("dg2T", (K4p, Rp), zero=True)]
        for src, dst in ((dg1, dgT[0]), (dg2, dgT[1])):
            H.call("nppc_transpose", self.prec, src, dst, Rr, K4, K4, Rp, 0, 0, 0, 1, s)
        slab = ws("slab", (Sg * K4p * max(Hp, KXp),), torch.float32)
        for dg, off, inp, width, dest in jobs:
            wp = rup(width, 128)
            name = f"inT_{id(inp)}_{off}"
            inT = ws(name, (wp, Rp), zero=True)
            # shifted product: input row r pairs with dgates row r + off -> place it at column r + off (scalar stores)
            H.call("nppc_transpose", self.prec, inp, inT.view(-1)[off:], Rr - off, width, width, Rp, 0, 0, 0, 1, s)
            A = dgT[0] if dg is dg1 else dgT[1]
            H.call("nppc_gemm_nt_splitk", self.prec, A, Rp, inT, Rp, slab, wp, K4p, wp, Rp, Sg, s)
            scatter(slab, Sg, K4p, wp, dest)

    def backward(self, dout, gen=None):
        """dout [B', O, F', T] fp32 -> flat parameter gradient (same layout as the flat parameter buffer).

        Contract: ONE backward per train-mode forward, and it must be the latest one -- the saved activations live in
        step-persistent workspaces that the next train-mode forward of this engine overwrites.  `gen` (the generation
        number the forward returned in `last_train["gen"]`) makes a violation an error instead of a wrong gradient."""
        d = self.last_train
        if d is None or "lstm" not in d or "g1" not in d["lstm"]:
            raise RuntimeError("backward needs a forward(train=True) first")
        FSNEngine.join_all()                  # the cooperative LSTM backward wants the chip to itself
        if gen is not None and gen != d["gen"]:
            raise RuntimeError(
                f"direction-net backward for forward #{gen}, but forward #{d['gen']} has since overwritten the saved "
                "activations: this engine keeps ONE set of saved state (one backward per train-mode forward, no "
                "gradient accumulation over micro-batches); run backward before the next forward")
        if d.get("consumed"):
            raise RuntimeError("direction-net backward called twice for one forward (retain_graph / double backward are "
                               "not supported: the flat gradient buffer is overwritten, not accumulated)")
        d["consumed"] = True
        s = H.stream()
        prec, dt, dev = self.prec, self.dt, self.dev
        B, T, Tv, Tp = d["B"], d["T"], d["Tv"], d["Tp"]
        F, C, ldC, ldF, sP, Hd, O = self.F, self.C, self.ldC, self.ldF, self.sP, self.Hd, self.O
        R = B * Tp
        Nseq, Fo = d["Nseq"], d["Fo"]
        G = self._grad_buffer()
        G.zero_()
        lo = d["lstm"]
        # ---- 1. head: dh2 = dY Wh, dWh, dbh
        ws = lambda name, shape, dtype=dt, zero=False: workspace(("eng", id(self), name), shape, dtype, dev, zero)
        if FUSED_HEAD >= 3 and prec == H.PREC_BF16 and O <= 16 and bwd_head_fusable(Nseq, self.lstm_bwd):
            # the K-split cooperative kernel forms d h2 = dY . Wh itself from the gathered dY rows: no dh2 tensor
            # dY rows [T'*N][16] (zero padded like the other GEMM operands: the head's weight gradient h2^T . dY runs with
            # the LSTM weight gradients on the side stream, behind the recurrence, instead of in front of it)
            dyt_rows = ws("dyt_rows", (padded_rows(Tv * Nseq) + 8, 16), zero=True)
            dyt = rows_view(dyt_rows, Tv, Nseq)
            H.call("nppc_head_dy_gather", dout, dyt, Nseq, Tv, self.la, O, Fo, s)
            head_wgrad = (dyt_rows, O) if (Hd % 128 == 0 and self.KX % 64 == 0) else None
            if head_wgrad is None:
                H.call("nppc_sb_head_bwd_w", prec, dout, lo["h2"], self.g("sb_model.fc_output_layer.weight"),
                       self.g("sb_model.fc_output_layer.bias"), Nseq, Tv, self.la, Hd, O, Fo, s)
            dx, dg1, dg2 = lstm2_backward(lo, None, self.lstm_bwd, self.KX, head=(dyt, self.WhT))
        else:
            head_wgrad = None
            dh2 = ws("dh2", (Tv, Nseq, Hd))
            H.call("nppc_sb_head_bwd", prec, dout, self.WhT, lo["h2"], dh2, self.g("sb_model.fc_output_layer.weight"),
                   self.g("sb_model.fc_output_layer.bias"), Nseq, Tv, self.la, Hd, O, Fo, s)
            # ---- 2. LSTM recurrence backward
            dx, dg1, dg2 = lstm2_backward(lo, dh2, self.lstm_bwd, self.KX)
        # ---- 3. LSTM weight gradients: dW[k][c] = sum_rows dgates[row][k] * input[row][c], rows = (t, sequence).
        # Row-major operands straight from the recurrent kernels; h_{t-1} is the same buffer one time block (Nseq rows)
        # earlier; the staged input carries a ones column (index I), so its product column is the bias gradient.
        # They only feed the flat gradient, so they run on a side stream beside the full-band backward chain below
        # (dozens of small kernels that leave most CUs idle); joined at the end of backward.
        main = torch.cuda.current_stream()
        if self._side is None:
            # lowest priority: the dependent chain on the main stream is what the step waits for; the weight-gradient
            # GEMMs are throughput work (NPPC_SIDE_PRIO=0: default priority, A/B switch)
            prio = 0
            if os.environ.get("NPPC_SIDE_PRIO", "1") != "0" and hasattr(torch.cuda.Stream, "priority_range"):
                prio = max(torch.cuda.Stream.priority_range())
            self._side = torch.cuda.Stream(device=dev, priority=prio)
        self._side.wait_stream(main)
        with torch.cuda.stream(self._side):
            self._lstm_wgrad(dg1, dg2, d["x_rows"], lo["h1_rows"], lo["h2_rows"], Tv, Nseq, head=head_wgrad,
                             guards=(lo["h1_guard"], lo["h2_guard"]) if "h1_guard" in lo else None)
            if self.grad_range_hook is not None and self.early_buckets_ok:
                # the sub-band segment (LSTM + head: the tail of the flat buffer) is final once these GEMMs are done
                self.grad_range_hook(G, self.fp.off["sb_model.sequence_model.weight_ih_l0"][0], G.numel())
        # ---- 4. staging backward -> gradient of the pre-ReLU full-band outputs
        dpre_fb = ws("dpre_fb", (3, B, Tp, ldF), zero=True)
        Dsb = ws("Dsb", (B,), torch.float64)
        H.call("nppc_subband_stage_bwd", prec, dx, d["x_tm"], d["fb"], d["sbscale"], Dsb, dpre_fb, B, F, Tp, Tv, ldF, R * ldF,
               self.nb, self.G, self.KX, s)
        # ---- 5. fc_output_layer backward
        S2 = TCN_WGRAD_SPLITS if R % (64 * TCN_WGRAD_SPLITS) == 0 else (8 if R % (32 * 8) == 0 else 1)
        Fr = rup(F, 128)
        Cr = rup(ldC, 128)
        tA = ws("tA", (3, max(Cr, TCN_HIDDEN, Fr), R), zero=True)       # transposed dY operand
        tB = ws("tB", (3, max(ldC, TCN_HIDDEN), R), zero=True)          # transposed activation operand
        slab2 = ws("slab2", (3 * S2 * max(Cr * TCN_HIDDEN, TCN_HIDDEN * ldC, Fr * ldC),), torch.float32)
        X8 = d["X"][8]
        sTA, sTB = tA.shape[1] * R, tB.shape[1] * R

        def on_side(fn):
            """run fn (which launches on the current stream) on the side stream once the main stream got here"""
            if not TCN_WGRAD_ON_SIDE:
                return fn()
            ev = torch.cuda.Event()
            ev.record(main)
            with torch.cuda.stream(self._side):
                self._side.wait_event(ev)
                fn()

        def fc_wgrad(slab):
            # parameter gradients only (bias: column sums; weight: two transposes + the NT split-K product -- F and C are not
            # multiples of the TN kernel's tiles)
            q = H.stream()
            H.colsum(prec, dpre_fb, self.g("fb_model.fc_output_layer.bias"), R, F, ldF, R * ldF, sP, 3,
                     lambda n: ws("cs_fc", (n,), torch.float32), q)
            H.call("nppc_transpose", prec, dpre_fb, tA, R, F, ldF, R, R * ldF, sTA, 0, 3, q)
            H.call("nppc_transpose", prec, X8, tB, R, ldC, ldC, R, R * ldC, sTB, 1, 3, q)
            self._wgrad(tA, R, sTA, tB, R, sTB, Fr, ldC, R, S2, "fb_model.fc_output_layer.weight", C, F, C, slab, batch=3, sDst=sP)
        # bf16: the product waits until the END of the main chain -- the side queue (LSTM + TCN weight gradients) is the longer
        # one, the main queue would idle ~0.4 ms in front of the join -- with a slab of its own (slab2 belongs to the side queue)
        # (the generic path below re-uses tA / tB / slab2 on the main stream: there the product stays in line)
        fc_ok = prec == H.PREC_BF16 and TCN_HIDDEN % 128 == 0 and ldC % 64 == 0 and R % (64 * S2) == 0
        fc_at_end = fc_ok and FC_WGRAD_AT_END
        if fc_ok and not fc_at_end:
            on_side(lambda: fc_wgrad(slab2))
        elif not fc_ok:
            fc_wgrad(slab2)
        # bf16: the 1x1-conv weight gradients run on the TN GEMM straight from the row-major activations (no transposes)
        tn_ok = prec == H.PREC_BF16 and TCN_HIDDEN % 128 == 0 and ldC % 64 == 0 and R % (64 * S2) == 0
        if tn_ok:
            # ... and on the side stream, behind the LSTM weight gradients: nothing downstream reads them before the
            # optimizer, and the end of the main chain (TSSE backward, small kernels) leaves most CUs idle.  Their operands
            # get one buffer per TCN block, so the main chain never waits for the side stream inside a step.
            dXL = [ws(f"dX_{i}", (3, B, Tp, ldC)) for i in range(9)]
            a2L = [ws(f"a2_{i}", (3, B, Tp, TCN_HIDDEN), zero=True) for i in range(8)]
            h2L = [ws(f"h2b_{i}", (3, B, Tp, TCN_HIDDEN)) for i in range(8)]
            dXa, dXb = dXL[8], dXL[7]
        else:
            dXa = ws("dXa", (3, B, Tp, ldC))
            dXb = ws("dXb", (3, B, Tp, ldC))

        # the GEMMs that PRODUCE a block's upstream gradient also leave its column sums per 128-row tile (epilogue): the sconv
        # bias gradient is their sum over the tiles, added up by the block's fused middle backward -- no pass over dXo for it
        bk = 64 if prec == H.PREC_BF16 else 32
        cs_ok = tn_ok and COLSUM_IN_GEMM and ldF % bk == 0 and TCN_HIDDEN % bk == 0
        cp_all = ws("colpart", (8, 3, R // 128, ldC), torch.float32) if cs_ok else None      # one block per TCN block
        cpL = [cp_all[i] for i in range(8)] if cs_ok else None
        if cs_ok:
            H.call("nppc_gemm_nt_colsum", prec, EPI_MASK_POS, dpre_fb, ldF, R * ldF, self.WfcT, ldF, ldC * ldF, dXa, ldC, R * ldC,
                   None, 0, X8, ldC, R * ldC, R, ldC, ldF, Tp, Tv, C, 3, cpL[7], s)
        else:
            H.call("nppc_gemm_nt", prec, EPI_MASK_POS, dpre_fb, ldF, R * ldF, self.WfcT, ldF, ldC * ldF, dXa, ldC, R * ldC, None, 0,
                   X8, ldC, R * ldC, None, 0, None, 0, R, ldC, ldF, Tp, Tv, C, 0, 3, 1, s)
        # ---- 6. TCN blocks in reverse
        sAct = B * Tp * TCN_HIDDEN
        h1b = ws("h1b", (3, B, Tp, TCN_HIDDEN))
        h2b = ws("h2b", (3, B, Tp, TCN_HIDDEN))
        Smid = ws("Smid", (3, B, TCN_HIDDEN // 64, 8), torch.float64)      # channel-group shares of the per-sample sums
        # partial rows of the fused middle backward, one set per TCN block: their finishing pass (sum over the samples into the
        # parameter gradients) runs as ONE launch for all eight blocks behind the loop -- or as two (blocks 7..4, then 3..0) when the
        # data-parallel exchange sends blocks 7..4 early.  Eight 5-us launches each waited up to 0.5 ms for CUs beside the
        # weight-gradient GEMMs of the side queue (profiles/r03_bench_c2_bf16_windows.txt)
        n_part = H.mid_bwd_part_elems(B, TCN_HIDDEN, Tp, 3)
        Pmid = ws("Pmid", (8, n_part), torch.float32)
        lay = self.fp.off["fb_model.sequence_model.1.conv1x1.weight"][0] - self.fp.off["fb_model.sequence_model.0.conv1x1.weight"][0]

        def finish_mid(lo, hi):
            """finishing pass of blocks lo .. hi-1"""
            pre_ = f"fb_model.sequence_model.{lo}."
            H.call("nppc_tcn_mid_bwd_finish", Pmid[lo], n_part, cpL[lo] if cs_ok else None, 3 * (R // 128) * ldC, R // 128, ldC, C,
                   self.g(pre_ + "norm2.weight"), self.g(pre_ + "norm2.bias"), self.g(pre_ + "norm1.weight"), self.g(pre_ + "norm1.bias"),
                   self.g(pre_ + "depthwise_conv.weight"), self.g(pre_ + "depthwise_conv.bias"), self.g(pre_ + "prelu1.weight"),
                   self.g(pre_ + "prelu2.weight"), self.g(pre_ + "conv1x1.bias"), self.g(pre_ + "sconv.bias") if cs_ok else None,
                   B, TCN_HIDDEN, sP, lay, 3, hi - lo, s)
        early = self.grad_range_hook is not None and self.early_buckets_ok
        defer_finish = MID_FINISH_DEFER        # (A/B switch NPPC_MID_FINISH_DEFER=0: one finishing launch per block, as in round 3)
        dXo, dXi = dXa, dXb
        for i in range(7, -1, -1):
            pre = f"fb_model.sequence_model.{i}."
            dil = TCN_DILATIONS[i]
            st1, st2 = d["stats"][i, 0], d["stats"][i, 1]
            y1, y2, Xin = d["y1"][i], d["y2"][i], d["X"][i]
            if tn_ok:
                dXo, dXi, a2, h2b = dXL[i + 1], dXL[i], a2L[i], h2L[i]
            else:
                a2 = d["a2"]
            if not cs_ok:
                H.colsum(prec, dXo, self.g(pre + "sconv.bias"), R, C, ldC, R * ldC, sP, 3,
                         lambda n: ws("cs_sconv", (n,), torch.float32), s)
            # dA2 = dXo W2  (gradient of the normalised depthwise output)
            H.call("nppc_gemm_nt", prec, EPI_PLAIN, dXo, ldC, R * ldC, self.W2T[i], ldC, TCN_HIDDEN * ldC, h1b, TCN_HIDDEN,
                   sAct, None, 0, None, 0, 0, None, 0, None, 0, R, TCN_HIDDEN, ldC, Tp, Tv, TCN_HIDDEN, 0, 3, 1, s)
            # GroupNorm-2, PReLU-2, depthwise conv, GroupNorm-1, PReLU-1 backward in one reduce + one apply pass: h1b -> h2b
            # (= gradient of the conv1x1 output) with every parameter gradient of those stages and the conv1x1 bias gradient;
            # the reduce pass also leaves a2 = GN2(y2), the operand of the sconv weight gradient
            H.call("nppc_tcn_mid_bwd", prec, h1b, y2, y1, st1, st2, Smid, Pmid[i], self.p(pre + "norm1.weight"), self.p(pre + "norm1.bias"),
                   self.p(pre + "norm2.weight"), self.p(pre + "norm2.bias"), self.p(pre + "depthwise_conv.weight"),
                   self.p(pre + "prelu1.weight"), self.p(pre + "prelu2.weight"), a2, h2b, self.g(pre + "norm2.weight"),
                   self.g(pre + "norm2.bias"), self.g(pre + "norm1.weight"), self.g(pre + "norm1.bias"),
                   self.g(pre + "depthwise_conv.weight"), self.g(pre + "depthwise_conv.bias"), self.g(pre + "prelu1.weight"),
                   self.g(pre + "prelu2.weight"), self.g(pre + "conv1x1.bias"),
                   cpL[i] if cs_ok else None, R // 128, ldC, C, self.g(pre + "sconv.bias") if cs_ok else None,
                   B, TCN_HIDDEN, Tp, Tv, dil, 1e-8, sAct, B * 2, sP, 3, 0 if defer_finish else 1, s)
            if i == 4 and early and defer_finish:
                finish_mid(4, 8)              # blocks 7..4 are handed to the exchange below: their gradients must be final
            # sconv weight gradient: dW2[c][k] = sum_r dXo[r][c] * a2[r][k]
            if tn_ok:
                # row-major operands as they are: slab[k][c] = sum_r a2[r][k] * dXo[r][c] = dW2^T, transposed in the reduction
                # (blocks i < TCN_WGRAD_MAIN_BLOCKS -- the last ones of the chain -- keep their two products on the MAIN queue,
                # with a slab of their own: since the eight finishing launches became one the side queue is the longer one,
                # profiles/r04_bench_c2_bf16_windows.txt)
                inline = i < TCN_WGRAD_MAIN_BLOCKS
                wslab = ws("slab_main", (3 * S2 * TCN_HIDDEN * ldC,), torch.float32) if inline else slab2

                def sconv_wgrad(a2=a2, dXo=dXo, dest=self.g(pre + "sconv.weight"), wslab=wslab):
                    q = H.stream()
                    H.call("nppc_gemm_tn_splitk_batched", a2, TCN_HIDDEN, sAct, dXo, ldC, R * ldC, wslab, ldC,
                           S2 * TCN_HIDDEN * ldC, TCN_HIDDEN, ldC, R, S2, 3, q)
                    H.call("nppc_reduce_slabs_t", wslab, S2, TCN_HIDDEN * ldC, ldC, dest, TCN_HIDDEN, C, TCN_HIDDEN,
                           S2 * TCN_HIDDEN * ldC, sP, 3, q)
                if inline:
                    sconv_wgrad()
                else:
                    on_side(sconv_wgrad)
            else:
                H.call("nppc_transpose", prec, dXo, tA, R, ldC, ldC, R, R * ldC, sTA, 0, 3, s)
                H.call("nppc_transpose", prec, a2, tB, R, TCN_HIDDEN, TCN_HIDDEN, R, sAct, sTB, 0, 3, s)
                self._wgrad(tA, R, sTA, tB, R, sTB, Cr, TCN_HIDDEN, R, S2, pre + "sconv.weight", TCN_HIDDEN, C, TCN_HIDDEN,
                            slab2, batch=3, sDst=sP)
            # conv1x1: weight, input gradients (the bias gradient came out of the fused kernel)
            if tn_ok:
                # slab[k][c] = sum_r dpre1[r][k] * Xin[r][c] = dW1
                def c1_wgrad(h2b=h2b, Xin=Xin, dest=self.g(pre + "conv1x1.weight"), wslab=wslab):
                    q = H.stream()
                    H.call("nppc_gemm_tn_splitk_batched", h2b, TCN_HIDDEN, sAct, Xin, ldC, R * ldC, wslab, ldC,
                           S2 * TCN_HIDDEN * ldC, TCN_HIDDEN, ldC, R, S2, 3, q)
                    H.call("nppc_reduce_slabs", wslab, S2, TCN_HIDDEN * ldC, ldC, dest, C, TCN_HIDDEN, 0, C, 0, 0,
                           S2 * TCN_HIDDEN * ldC, sP, 3, q)
                if inline:
                    c1_wgrad()
                else:
                    on_side(c1_wgrad)
            else:
                H.call("nppc_transpose", prec, h2b, tA, R, TCN_HIDDEN, TCN_HIDDEN, R, sAct, sTA, 0, 3, s)
                H.call("nppc_transpose", prec, Xin, tB, R, ldC, ldC, R, R * ldC, sTB, 0, 3, s)
                self._wgrad(tA, R, sTA, tB, R, sTB, TCN_HIDDEN, ldC, R, S2, pre + "conv1x1.weight", C, TCN_HIDDEN, C, slab2,
                            batch=3, sDst=sP)
            if cs_ok and i > 0:
                H.call("nppc_gemm_nt_colsum", prec, EPI_RESIDUAL, h2b, TCN_HIDDEN, sAct, self.W1T[i], TCN_HIDDEN, ldC * TCN_HIDDEN,
                       dXi, ldC, R * ldC, None, 0, dXo, ldC, R * ldC, R, ldC, TCN_HIDDEN, Tp, Tv, C, 3, cpL[i - 1], s)
            else:
                H.call("nppc_gemm_nt", prec, EPI_RESIDUAL, h2b, TCN_HIDDEN, sAct, self.W1T[i], TCN_HIDDEN, ldC * TCN_HIDDEN, dXi, ldC,
                       R * ldC, None, 0, dXo, ldC, R * ldC, None, 0, None, 0, R, ldC, TCN_HIDDEN, Tp, Tv, C, 0, 3, 1, s)
            dXo, dXi = dXi, dXo
            if i == 4 and self.grad_range_hook is not None and self.early_buckets_ok:
                # TCN blocks 7..4 of every branch are final once both queues have passed this point
                def blocks_done():
                    for br in BRANCHES:
                        a = self.fp.off[f"fb_model{br}.sequence_model.4.conv1x1.weight"][0]
                        b, shp = self.fp.off[f"fb_model{br}.sequence_model.7.sconv.bias"]
                        self.grad_range_hook(G, a, b + int(np.prod(shp)))
                on_side(blocks_done)
        if defer_finish:
            finish_mid(0, 4 if early else 8)
        if tn_ok:
            dXo = dXL[0]
        # ---- 7. TSSE attention backward (parameter gradients only: the maps are data)
        sv = d["tsse_saved"]
        n_ws = ctypes.c_long()
        H.call("nppc_tsse_bwd_ws_elems", 3 * self.nm, B, F, self.ks[0], self.ks[1], self.ks[2], ctypes.byref(n_ws))
        dsg = ws("dsg", (n_ws.value,), torch.float32)      # per map: dsg | da2 | da1 | per-sample contributions (csrc/spec.hip)
        att = "channel_attention."
        H.call("nppc_tsse_bwd_maps", prec, dXo, R * ldC, H.ptr_array(d["maps"]), 3 * self.nm, d["rs"],
               self.p(att + "smallConv1d.0.weight"), self.p(att + "middleConv1d.0.weight"), self.p(att + "largeConv1d.0.weight"),
               self.ks[0], self.ks[1], self.ks[2], self.p(att + "feature_concate_fc.weight"), self.p(att + "fc1.weight"),
               self.p(att + "fc2.weight"), self.sAtt, sv["ns"], sv["pre"], sv["sq"], sv["h1"], sv["sg"], dsg,
               self.g(att + "smallConv1d.0.weight"), self.g(att + "smallConv1d.0.bias"),
               self.g(att + "middleConv1d.0.weight"), self.g(att + "middleConv1d.0.bias"),
               self.g(att + "largeConv1d.0.weight"), self.g(att + "largeConv1d.0.bias"),
               self.g(att + "feature_concate_fc.weight"), self.g(att + "feature_concate_fc.bias"),
               self.g(att + "fc1.weight"), self.g(att + "fc1.bias"), self.g(att + "fc2.weight"), self.g(att + "fc2.bias"),
               B, F, T, self.la, Tp, ldC, s)
        if fc_at_end:
            fc_wgrad(ws("slab_fc", (3 * S2 * Fr * ldC,), torch.float32))
        if self.defer_join:
            self.join_pending = True          # joined (join_side) before anything reads the gradient
            FSNEngine._unjoined[id(self)] = self
        else:
            torch.cuda.current_stream().wait_stream(self._side)       # join the weight-gradient stream
        return G

    def side_stream(self):
        return self._side

    def prepack_on_side(self):
        """re-pack the (just updated) weights on the side stream behind everything the current stream has been given so far;
        the caller joins (join_side) before the packed copies are used and before the next cooperative LSTM launch"""
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        with torch.cuda.stream(self._side):
            self._side.wait_event(ev)
            self.pack_weights()
        self.join_pending = True
        FSNEngine._unjoined[id(self)] = self

    def join_side(self):
        """make the current stream wait for this engine's side-stream work (weight gradients of the last backward in
        defer_join mode, re-packed weights); idempotent"""
        if self.join_pending:
            torch.cuda.current_stream().wait_stream(self._side)
            self.join_pending = False
            FSNEngine._unjoined.pop(id(self), None)

    @staticmethod
    def join_all():
        """make the current stream wait for the side-stream work of EVERY engine (before a cooperative LSTM launch)"""
        for e in list(FSNEngine._unjoined.values()):
            e.join_side()
