"""Host side of the fused 2-layer LSTM kernels (csrc/lstm.hip)."""
import ctypes
import os

import torch

from . import _hip as H


# bench.py sets this to a list to collect (label, start_event, end_event) around every LSTM launch; the events
# are recorded on the current stream, i.e. the stream the kernels are enqueued on.
PROFILE = None


def _timed(label, fn):
    if PROFILE is None:
        return fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = fn()
    e1.record()
    PROFILE.append((label, e0, e1))
    return r


class PackedLSTM:
    """Per-wave MFMA B-fragment packing of nn.LSTM(I,Hd,2) weights (+ summed biases)."""

    def __init__(self, I, Hd, prec, device):
        n1, n2, kx = ctypes.c_long(), ctypes.c_long(), ctypes.c_int()
        H.call("nppc_lstm2_packed_elems", I, Hd, ctypes.byref(n1), ctypes.byref(n2), ctypes.byref(kx))
        self.I, self.Hd, self.prec, self.kx = I, Hd, prec, kx.value
        dt = H.dtype_of(prec)
        self.wp1 = torch.empty(n1.value, dtype=dt, device=device)
        self.wp2 = torch.empty(n2.value, dtype=dt, device=device)
        self.bias1 = torch.empty(4 * Hd, dtype=torch.float32, device=device)
        self.bias2 = torch.empty(4 * Hd, dtype=torch.float32, device=device)
        # weight-stationary cluster kernel (csrc/lstm_ws.hip): its own A-fragment packing of the same weights
        self.ws = prec == H.PREC_BF16 and Hd == 384 and I <= 64
        if self.ws:
            m1, m2 = ctypes.c_long(), ctypes.c_long()
            H.call("nppc_lstm2_ws_packed_elems", ctypes.byref(m1), ctypes.byref(m2))
            self.wsp1 = torch.empty(m1.value, dtype=dt, device=device)
            self.wsp2 = torch.empty(m2.value, dtype=dt, device=device)
        own_workspaces(self)

    def pack(self, w_ih0, w_hh0, b_ih0, b_hh0, w_ih1, w_hh1, b_ih1, b_hh1):
        ws = [t.detach().contiguous() for t in (w_ih0, w_hh0, b_ih0, b_hh0, w_ih1, w_hh1, b_ih1, b_hh1)]
        H.call("nppc_lstm2_pack_weights", self.prec, *ws, self.I, self.Hd, self.wp1, self.wp2, self.bias1,
               self.bias2, H.stream())
        if self.ws:
            H.call("nppc_lstm2_ws_pack", ws[0], ws[1], ws[4], ws[5], self.I, self.wsp1, self.wsp2, H.stream())
        return self


def pick_mtile(n_seq, prec, train, n_cu=256):
    """Rows per workgroup = 16*mtile: fill the CUs first, then grow the tile (weight reuse)."""
    if prec == H.PREC_F32:
        return 1
    for mt in ((2,) if train else (3, 2)):
        if (n_seq + 16 * mt - 1) // (16 * mt) >= n_cu * 2 // 3:
            return mt
    return 1


_WS = {}
COOP = True          # use the cooperative (weights split over CU pairs) kernels when the shape allows it
# weight-stationary 12-CU clusters (csrc/lstm_ws.hip) for the forward when the plan allows: "0" never (default), "1" inference
# launches, "2" also the training forward.  Measured at C2 (profiles/r03_lstm_ws_variants.txt, r03_bench_ws_modes.txt): ALONE
# the restorer launch takes 8.9 ms against 9.8 ms for the CU-pair kernel and the no-grad direction launch 4.4 against 5.6,
# but INSIDE the train step the same launches are slower (9.3 vs 9.1 ms) and so are the launches behind them: these kernels
# are bound by what the chip may draw, and the cluster kernel's hidden-state gather (4.3 TB/s from L2) costs 20 % of its clock
# Round 4 default "auto": the cluster kernel takes the INFERENCE launches of small problems -- those for which the streaming
# plan falls back to four-CU clusters because CU pairs would leave most of the chip idle (BASELINE config 5: 2056 sequences
# x 1878 steps: 23.4 ms against 48.3 ms for the four-CU streaming plan, step 145.3 -> 120.1 ms; the training forward at
# N = 1024 stays on the streaming kernel: 35.4 vs 32.9 ms; profiles/r04_bench_c5_ws_modes.txt)
WS_MODE = os.environ.get("NPPC_LSTM_WS", "auto")
WS = WS_MODE != "0"
N_CU = None


def _n_cu():
    global N_CU
    if N_CU is None:
        N_CU = torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count
    return N_CU


def _flag_blocks():
    return [t for k, t in _WS.items() if k[0][-1] == "coop_flags"]


_RETIRED_TIMEOUTS = 0      # time-outs counted in flag blocks that have been released with their owner


def coop_timeouts():
    """Bounded hand-off spins that gave up since the counters were last cleared (0 on a healthy run).  The counter word
    of every flag block is STICKY: kernels only add to it and no launcher clears it (csrc/lstm_coop.hip), so a time-out
    in ANY earlier launch is still visible here.  Reads the device (one small copy per flag block: call it where the
    host synchronises anyway)."""
    return _RETIRED_TIMEOUTS + sum(int(t[-4].item()) for t in _flag_blocks())


_GUARDS = [None, ()]


def timeout_guards():
    """(device int64 tensor of the addresses of every live sticky time-out counter, count) for nppc_adam_step_guarded;
    rebuilt only when the set of flag blocks changes (a host -> device copy of a few words)"""
    blocks = _flag_blocks()
    key = tuple(t.data_ptr() for t in blocks)
    if _GUARDS[1] != key or _GUARDS[0] is None:
        ptrs = [t[-4:].data_ptr() for t in blocks]
        _GUARDS[0] = torch.tensor(ptrs or [0], dtype=torch.int64, device=blocks[0].device if blocks else "cuda")
        _GUARDS[1] = key
    return _GUARDS[0], len(blocks)


def clear_coop_timeouts():
    global _RETIRED_TIMEOUTS
    _RETIRED_TIMEOUTS = 0
    for t in _flag_blocks():
        t[-4:].zero_()


def release_workspaces(owner_id):
    """drop every step-persistent workspace of one owner (a packed-weight object or an engine; key[1] of its workspace
    keys is id(owner)).  Registered as a weakref finalizer by the owners, so the tens of GB of saved LSTM state of a
    deleted model return to the allocator instead of living as long as the process."""
    global _RETIRED_TIMEOUTS
    for k in [k for k in _WS if len(k[0]) > 1 and k[0][1] == owner_id]:
        t = _WS.pop(k)
        if k[0][-1] == "coop_flags":
            try:
                _RETIRED_TIMEOUTS += int(t[-4].item())
            except Exception:          # interpreter / device shutting down
                pass


def own_workspaces(owner):
    import weakref
    weakref.finalize(owner, release_workspaces, id(owner))


def check_coop_timeouts(where):
    """raise if any cooperative LSTM hand-off timed out: the kernels carry on with wrong numbers after a time-out, so
    everything computed since (gradients, Adam state) is suspect"""
    n = coop_timeouts()
    if n:
        clear_coop_timeouts()
        raise RuntimeError(f"{n} cooperative LSTM hand-off time-out(s) detected at {where}: a partner workgroup was not "
                           "co-resident or stalled; results since the last check are invalid")


def workspace(key, shape, dtype, device, zero=False):
    """Step-persistent device workspace (allocated, and if asked zeroed, ONCE: kernels never touch the padding)."""
    k = (key, tuple(shape), dtype, str(device))
    t = _WS.get(k)
    if t is None:
        t = (torch.zeros if zero else torch.empty)(*shape, dtype=dtype, device=device)
        _WS[k] = t
    return t


COOP_BWD_KSPLIT = True    # cooperative backward variant: True = K-split (bf16 partial-sum exchange), False = output-split
# K-split backward on four-CU clusters of 64 sequences instead of CU pairs of 32 (half the weight-fragment bytes per CU and step):
# "1" whenever ceil(N / 64) * 4 CUs are free, "0" never
BWD_G4 = os.environ.get("NPPC_LSTM_BWD_G4", "0")
WGRAD_SPLITS = int(os.environ.get("NPPC_WGRAD_SPLITS", "64"))    # K-slices of the weight-gradient GEMMs (engine._lstm_wgrad)
ROW_PAD = 64 * WGRAD_SPLITS   # row granularity of their operands: 64-row stages x K-slices; buffers carry this much slack


def padded_rows(R, shift=0):
    """rows to allocate for a [R][cols] GEMM operand: whole ROW_PAD blocks, plus slack for the one-step shift"""
    return (R + ROW_PAD - 1) // ROW_PAD * ROW_PAD + ROW_PAD + shift


def rows_view(flat, Tn, N):
    """[Rpad][cols] zero-padded operand buffer -> its live [Tn][N][cols] prefix"""
    return flat[:Tn * N].view(Tn, N, flat.shape[1])


FUSED_HEAD_MAX_O = 16    # one MFMA column tile


def lstm2_forward(x_tm, packed, train, mtile=None, head=None):
    """x_tm [Tn][N][kx] (time-major, zero padded to kx) -> dict(h2[, h1, g1, g2, c1, c2]) time-major.
    In train mode h1/h2 are prefixes of zero-padded row buffers (`h1_rows`, `h2_rows`: [Rpad][H]) that the
    weight-gradient GEMMs read directly.
    head = (whp [16][H] packed head weights, O): when the cooperative pair kernel runs, the output head is fused into it
    and the result carries head_partial [2][Tn][N][O] fp32 (the caller finishes with nppc_sb_head_finalize) -- in
    inference INSTEAD of h2, in training beside the saved state; any other plan ignores `head`."""
    Tn, N, kx = x_tm.shape
    # kx < packed.kx: rows staged at a packed width (forward_x_ld) -- only the fused-head CU-pair inference launch takes them
    assert (kx == packed.kx or (kx == forward_x_ld(N, packed, train, head) and mtile is None)) and x_tm.dtype == H.dtype_of(packed.prec)
    Hd = packed.Hd
    dt, dev = x_tm.dtype, x_tm.device
    tag = ("lstm", id(packed), train)
    out = {}
    if train:
        # N zero guard rows in front (h_{-1} = 0): `h1_guard` row r is h1 row r - N, the B operand of the weight-gradient
        # product that pairs the gate gradients of step t with h_{t-1} without shifting the gate gradients
        Rp = padded_rows(Tn * N, N)
        # (N, Tn) in the key: the guard rows in front and the padding rows behind Tn * N must be zero and are zeroed once per
        # buffer; two (N, Tn) pairs with equal N + Rp must not share one (the larger N's guard rows would hold stale h)
        out["h1_guard"] = workspace(tag + ("h1", N, Tn), (N + Rp, Hd), dt, dev, zero=True)
        out["h2_guard"] = workspace(tag + ("h2", N, Tn), (N + Rp, Hd), dt, dev, zero=True)
        out["h1_rows"] = out["h1_guard"][N:]
        out["h2_rows"] = out["h2_guard"][N:]
        out["h1"] = rows_view(out["h1_rows"], Tn, N)
        out["h2"] = rows_view(out["h2_rows"], Tn, N)
        out["c1"] = workspace(tag + ("c1",), (Tn, N, Hd), dt, dev)
        out["c2"] = workspace(tag + ("c2",), (Tn, N, Hd), dt, dev)
        out["g1"] = workspace(tag + ("g1",), (Tn, N, Hd, 4), dt, dev)
        out["g2"] = workspace(tag + ("g2",), (Tn, N, Hd, 4), dt, dev)
    if mtile == "ws" or (WS and COOP and mtile is None and (not train or WS_MODE == "2") and _ws_wanted(N, packed, train)):
        ws_out = _lstm2_forward_ws(x_tm, packed, train, head, out, tag)
        if ws_out is not None:
            return ws_out
        assert mtile != "ws", "the weight-stationary plan does not apply to this shape"
        mtile = None
    if (COOP and mtile is None) or isinstance(mtile, tuple):
        G, cmt, ncl = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        if isinstance(mtile, tuple):               # (G, mtile) forced by a test / benchmark
            G.value, cmt.value = mtile
            ncl.value = (N + 16 * cmt.value - 1) // (16 * cmt.value)
            assert ncl.value * G.value <= _n_cu(), "cooperative launch needs one CU per workgroup"
        else:
            H.call("nppc_lstm2_coop_plan", packed.prec, int(train), N, Hd, _n_cu(), ctypes.byref(G), ctypes.byref(cmt),
                   ctypes.byref(ncl))
        if ncl.value > 0:
            G, cmt, ncl = G.value, cmt.value, ncl.value
            xch = workspace(tag + ("coop_xch",), (ncl * 2 * 2 * 16 * cmt * Hd,), dt, dev)
            flags = workspace(tag + ("coop_flags",), (ncl * 2 * G + 4,), torch.int32, dev, zero=True)
            if head is not None and G == 2 and head[1] <= FUSED_HEAD_MAX_O and (not train or cmt == 2):
                whp, O = head
                out["head_partial"] = workspace(tag + ("hpart", O), (2, Tn, N, O), torch.float32, dev)
                _timed((f"lstm2_fwd_coop_g{G}", int(train), N, Tn, cmt), lambda: H.call(
                    "nppc_lstm2_fwd_coop_head", packed.prec, int(train), cmt, x_tm, packed.wp1, packed.wp2, packed.bias1,
                    packed.bias2, out.get("h2"), out.get("h1"), out.get("g1"), out.get("g2"), out.get("c1"), out.get("c2"),
                    xch, xch.numel() * xch.element_size(), flags, N, Tn, packed.I, Hd, whp, out["head_partial"], O, kx,
                    H.stream()))
                return out
            assert kx == packed.kx
            if not train:
                out["h2"] = workspace(tag + ("h2",), (Tn, N, Hd), dt, dev)
            _timed((f"lstm2_fwd_coop_g{G}", int(train), N, Tn, cmt), lambda: H.call(
                "nppc_lstm2_fwd_coop", packed.prec, int(train), G, cmt, x_tm, packed.wp1, packed.wp2, packed.bias1,
                packed.bias2, out["h2"], out.get("h1"), out.get("g1"), out.get("g2"), out.get("c1"), out.get("c2"), xch,
                xch.numel() * xch.element_size(), flags, N, Tn, packed.I, Hd, H.stream()))
            return out
    if mtile is None:
        mtile = pick_mtile(N, packed.prec, train)
    if not train:
        out["h2"] = workspace(tag + ("h2",), (Tn, N, Hd), dt, dev)
    _timed(("lstm2_fwd", int(train), N, Tn, mtile), lambda: H.call(
        "nppc_lstm2_fwd", packed.prec, int(train), mtile, x_tm, packed.wp1, packed.wp2, packed.bias1, packed.bias2,
        out["h2"], out.get("h1"), out.get("g1"), out.get("g2"), out.get("c1"), out.get("c2"), N, Tn, packed.I, Hd,
        H.stream()))
    return out


# packed input rows for the frozen restorer's inference launch (CU-pair kernel with the fused head): 40 columns instead of 64
# (34 features + the spare column + pad to 16 bytes), -37 % of the staged bytes and of the kernel's x reads.  "0" = 64 columns.
X_PACKED = os.environ.get("NPPC_X_PACKED", "1") != "0"


def forward_x_ld(N, packed, train, head):
    """row width (elements) the staging kernel should write for lstm2_forward(x, packed, train, None, head): packed.kx unless the
    launch will be the fused-head CU-pair inference kernel, which fetches rows of any 16-byte multiple width"""
    if (not X_PACKED or train or head is None or not COOP or packed.prec != H.PREC_BF16 or packed.Hd != 384
            or head[1] > FUSED_HEAD_MAX_O or packed.I + 1 > 40):
        return packed.kx
    if WS and _ws_wanted(N, packed, train) and ws_plan(N, packed) is not None:
        return packed.kx
    G, cmt, ncl = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    H.call("nppc_lstm2_coop_plan", packed.prec, 0, N, packed.Hd, _n_cu(), ctypes.byref(G), ctypes.byref(cmt), ctypes.byref(ncl))
    return 40 if (ncl.value > 0 and G.value == 2) else packed.kx


def _ws_wanted(N, packed, train):
    """NPPC_LSTM_WS=1|2: whenever the plan applies; "auto": only where the streaming plan would not be CU pairs"""
    if WS_MODE != "auto":
        return True
    G, cmt, ncl = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    H.call("nppc_lstm2_coop_plan", packed.prec, int(train), N, packed.Hd, _n_cu(), ctypes.byref(G), ctypes.byref(cmt), ctypes.byref(ncl))
    return ncl.value > 0 and G.value >= 4


def ws_plan(N, packed):
    """(clusters, nch_max) of the weight-stationary forward for N sequences, or None"""
    if not getattr(packed, "ws", False):
        return None
    cl, nch = ctypes.c_int(), ctypes.c_int()
    H.call("nppc_lstm2_ws_plan", packed.prec, N, packed.Hd, packed.I, _n_cu(), ctypes.byref(cl), ctypes.byref(nch))
    return (cl.value, nch.value) if cl.value > 0 else None


def _lstm2_forward_ws(x_tm, packed, train, head, out, tag):
    """weight-stationary cluster forward (csrc/lstm_ws.hip).  Inference needs the fused head (h2 only lives in a two-slot
    exchange ring); training keeps the whole saved state and takes the head along when asked.  Returns None when the
    plan does not apply."""
    Tn, N, _ = x_tm.shape
    plan = ws_plan(N, packed)
    if plan is None or (not train and head is None) or (head is not None and head[1] > FUSED_HEAD_MAX_O):
        return None
    ncl, nch = plan
    dt, dev, Hd = x_tm.dtype, x_tm.device, packed.Hd
    cst = workspace(tag + ("ws_cst",), (ncl * 12 * nch * 2048,), torch.float32, dev)
    flags = workspace(tag + ("ws", "coop_flags"), (ncl * nch * 16 + 4,), torch.int32, dev, zero=True)
    if train:
        h1, h2 = out["h1"], out["h2"]
    else:
        h1 = workspace(tag + ("ws_h1ring",), (2, N, Hd), dt, dev)
        h2 = workspace(tag + ("ws_h2ring",), (2, N, Hd), dt, dev)
    whp, O, hpart = None, 0, None
    if head is not None:
        whp, O = head
        hpart = out["head_partial"] = workspace(tag + ("ws_hpart", O), (1, Tn, N, O), torch.float32, dev)
    _timed(("lstm2_fwd_ws", int(train), N, Tn, nch), lambda: H.call(
        "nppc_lstm2_fwd_ws", int(train), x_tm, packed.wsp1, packed.wsp2, packed.bias1, packed.bias2, h1, h2, out.get("g1"),
        out.get("g2"), out.get("c1"), out.get("c2"), cst, flags, whp, hpart, O, N, Tn, ncl, nch, H.stream()))
    return out


class PackedLSTMBwd:
    """Backward (transposed) packing: B fragment column = input feature, k = unit*4 + gate(i,g,f,o)."""

    def __init__(self, I, Hd, prec, device):
        n1, n2 = ctypes.c_long(), ctypes.c_long()
        H.call("nppc_lstm2_bwd_packed_elems", I, Hd, ctypes.byref(n1), ctypes.byref(n2))
        self.I, self.Hd, self.prec = I, Hd, prec
        dt = H.dtype_of(prec)
        self.wb1 = torch.empty(n1.value, dtype=dt, device=device)
        self.wb2 = torch.empty(n2.value, dtype=dt, device=device)
        own_workspaces(self)
        self.coop = prec == H.PREC_BF16 and Hd == 384 and I <= 64
        if self.coop:
            nc = ctypes.c_long()
            H.call("nppc_lstm2_coop_bwd_packed_elems", ctypes.byref(nc))
            self.cwb1 = torch.empty(nc.value, dtype=dt, device=device)
            self.cwb2 = torch.empty(nc.value, dtype=dt, device=device)
            H.call("nppc_lstm2_coop_bwd2_packed_elems", ctypes.byref(nc))
            self.kwb1 = torch.empty(nc.value, dtype=dt, device=device)     # K-split variant
            self.kwb2 = torch.empty(nc.value, dtype=dt, device=device)
            n4, nx, nf = ctypes.c_long(), ctypes.c_long(), ctypes.c_long()
            H.call("nppc_lstm2_coop_bwd4_sizes", 64, ctypes.byref(n4), ctypes.byref(nx), ctypes.byref(nf))
            self.k4wb1 = torch.empty(n4.value, dtype=dt, device=device)    # K-split on four-CU clusters
            self.k4wb2 = torch.empty(n4.value, dtype=dt, device=device)

    def pack(self, w_ih0, w_hh0, w_ih1, w_hh1):
        ws = [t.detach().contiguous() for t in (w_ih0, w_hh0, w_ih1, w_hh1)]
        H.call("nppc_lstm2_pack_weights_bwd", self.prec, *ws, self.I, self.Hd, self.wb1, self.wb2, H.stream())
        if self.coop:
            if COOP_BWD_KSPLIT:
                H.call("nppc_lstm2_coop_bwd2_pack", *ws, self.I, self.kwb1, self.kwb2, H.stream())
                if BWD_G4 != "0":
                    H.call("nppc_lstm2_coop_bwd4_pack", *ws, self.I, self.k4wb1, self.k4wb2, H.stream())
            else:
                H.call("nppc_lstm2_coop_bwd_pack", *ws, self.I, self.cwb1, self.cwb2, H.stream())
        return self


def bwd_head_fusable(N, packed_bwd, coop=None):
    """True when lstm2_backward will run the K-split cooperative kernel, which can take the head backward (dyt, whT)"""
    return bool((COOP if coop is None else coop) and packed_bwd.coop and ((N + 31) // 32) * 2 <= _n_cu() and COOP_BWD_KSPLIT)


def lstm2_backward(saved, dh2, packed_bwd, kx, coop=None, head=None):
    """saved = lstm2_forward(train=True) dict; dh2 [Tn][N][H] -> (dx [Tn][N][kx], dg1_rows, dg2_rows): the gate gradients
    as zero-padded row buffers [Rpad][4H] (row t*N + n, column unit*4 + gate in (i,g,f,o) order).
    head = (dyt [Tn][N][16] bf16, whT [H][32] bf16) with dh2 None: the K-split cooperative kernel forms
    d h2 = dY . Wh itself (only valid when bwd_head_fusable(N, packed_bwd))."""
    Tn, N, Hd = saved["h2"].shape
    dt, dev = saved["h2"].dtype, saved["h2"].device
    tag = ("lstm_bwd", id(packed_bwd))
    dx = workspace(tag + ("dx",), (Tn, N, kx), dt, dev)
    Rp = padded_rows(Tn * N, N)
    dg1 = workspace(tag + ("dg1",), (Rp, 4 * Hd), dt, dev, zero=True)
    dg2 = workspace(tag + ("dg2",), (Rp, 4 * Hd), dt, dev, zero=True)
    use_coop = (COOP if coop is None else coop) and packed_bwd.coop and ((N + 31) // 32) * 2 <= _n_cu()
    if use_coop and COOP_BWD_KSPLIT and BWD_G4 != "0" and ((N + 63) // 64) * 4 <= _n_cu():
        npk, nx, nf = ctypes.c_long(), ctypes.c_long(), ctypes.c_long()
        H.call("nppc_lstm2_coop_bwd4_sizes", N, ctypes.byref(npk), ctypes.byref(nx), ctypes.byref(nf))
        xch = workspace(tag + ("coop_xch4",), (nx.value,), torch.uint8, dev)
        flags = workspace(tag + ("g4", "coop_flags"), (nf.value,), torch.int32, dev, zero=True)
        dyt, whT = head if head is not None else (None, None)
        _timed(("lstm2_bwd_coop_ksplit_g4", 1, N, Tn, 4), lambda: H.call(
            "nppc_lstm2_bwd_coop4", saved["g1"], saved["g2"], saved["c1"], saved["c2"], dh2 if head is None else None, dyt, whT,
            packed_bwd.k4wb1, packed_bwd.k4wb2, dx, dg1, dg2, xch, nx.value, flags, N, Tn, _n_cu(), H.stream()))
        return dx, dg1, dg2
    if use_coop and COOP_BWD_KSPLIT:
        ncl = (N + 31) // 32
        xch = workspace(tag + ("coop_xch",), (ncl * 2 * 2 * 2 * 32 * 384,), dt, dev)
        flags = workspace(tag + ("coop_flags",), (ncl * 48 + 4,), torch.int32, dev, zero=True)   # [cluster][layer][cu][wave] epochs + time-out words
        if head is not None:
            dyt, whT = head
            _timed(("lstm2_bwd_coop_ksplit", 1, N, Tn, 2), lambda: H.call(
                "nppc_lstm2_bwd_coop2_head", saved["g1"], saved["g2"], saved["c1"], saved["c2"], dyt, whT, packed_bwd.kwb1,
                packed_bwd.kwb2, dx, dg1, dg2, xch, xch.numel() * xch.element_size(), flags, N, Tn, _n_cu(), H.stream()))
            return dx, dg1, dg2
        _timed(("lstm2_bwd_coop_ksplit", 1, N, Tn, 2), lambda: H.call(
            "nppc_lstm2_bwd_coop2", saved["g1"], saved["g2"], saved["c1"], saved["c2"], dh2, packed_bwd.kwb1, packed_bwd.kwb2,
            dx, dg1, dg2, xch, xch.numel() * xch.element_size(), flags, N, Tn, _n_cu(), H.stream()))
        return dx, dg1, dg2
    assert head is None, "the fused head backward needs the K-split cooperative kernel"
    if use_coop:
        ncl = (N + 31) // 32
        xch = workspace(tag + ("coop_xch",), (ncl * 2 * 2 * 2 * 32 * 768,), dt, dev)
        flags = workspace(tag + ("coop_flags",), (ncl * 4 + 4,), torch.int32, dev, zero=True)
        _timed(("lstm2_bwd_coop_g2", 1, N, Tn, 2), lambda: H.call(
            "nppc_lstm2_bwd_coop", saved["g1"], saved["g2"], saved["c1"], saved["c2"], dh2, packed_bwd.cwb1, packed_bwd.cwb2,
            dx, dg1, dg2, xch, xch.numel() * xch.element_size(), flags, N, Tn, _n_cu(), H.stream()))
        return dx, dg1, dg2
    _timed(("lstm2_bwd", 1, N, Tn, 1), lambda: H.call(
        "nppc_lstm2_bwd", packed_bwd.prec, saved["g1"], saved["g2"], saved["c1"], saved["c2"], dh2, packed_bwd.wb1,
        packed_bwd.wb2, dx, dg1, dg2, N, Tn, packed_bwd.I, Hd, H.stream()))
    return dx, dg1, dg2
