"""TEST INFRASTRUCTURE -- CPU oracle for the NPPC-audio train step (never shipped, never measured
as the product).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.

A functional fp32 restatement, in plain torch-CPU ops, of the reference hot path (citations are
file:line under /root/reference).  It takes a flat {name: tensor} weight dict in the reference
state-dict naming (oracle/weights.py) and shares no code with generative-audio_amd/.

PINNED: tests/test_oracle_golden.py checks every function here against tests/golden/*.npz, which
were produced by running the reference itself in the build container
(tests/golden/make_goldens.py).  The reference has no tests of its own for this path.
"""
import math

import torch
import torch.nn.functional as Fn

EPS32 = float(torch.finfo(torch.float32).eps)   # audio_zen/constant.py:8


# ------------------------------------------------------------------ signal front end
def stft_parts(wave, nfft, hop, win):
    """utils.py:107-147 -> (mag, real, imag), each [B,1,F,T]; centred, periodic hann, onesided."""
    if wave.dim() == 1:
        wave = wave[None]
    w = torch.hann_window(win, dtype=wave.dtype)
    z = torch.stft(wave, n_fft=nfft, hop_length=hop, win_length=win, window=w, center=True,
                   return_complex=True)
    re, im = z.real, z.imag
    return torch.sqrt(re * re + im * im)[:, None], re[:, None], im[:, None]


def compress_mask(m, K=10.0, C=0.1):
    """mask.py:44-54."""
    m = torch.where(m <= -100, torch.full_like(m, -100.0), m)
    e = torch.exp(-C * m)
    return K * (1 - e) / (1 + e)


def decompress_mask(m, K=10.0, limit=9.9):
    """mask.py:57-60."""
    m = torch.clamp(m, -limit, limit)
    return -K * torch.log((K - m) / (K + m))


def ideal_mask(n_re, n_im, c_re, c_im):
    """mask.py:24-41: compressed cIRM, [B,F,T] maps -> [B,2,F,T] (the layout trainer.py:360 permutes to)."""
    den = n_re * n_re + n_im * n_im + EPS32
    mr = (n_re * c_re + n_im * c_im) / den
    mi = (n_re * c_im - n_im * c_re) / den
    return compress_mask(torch.stack((mr, mi), dim=1))


def apply_mask_swapped(mask_re, mask_im, n_re, n_im):
    """utils.py:241-249 calling utils.py:75-79 with real/imag swapped at the call AND the unpack:
    net effect conj(mask) * noisy.  All [B,F,T]."""
    e_re = mask_re * n_re + mask_im * n_im
    e_im = mask_re * n_im - mask_im * n_re
    return torch.sqrt(e_re * e_re + e_im * e_im), e_re, e_im


def istft_wave(re, im, nfft, hop, length):
    """utils.py:60-70 / validator.py:136-143: torch.istft with the periodic hann window, centred, length-cropped."""
    w = torch.hann_window(nfft, dtype=re.dtype)
    return torch.istft(torch.complex(re, im), n_fft=nfft, hop_length=hop, win_length=nfft, window=w, center=True,
                       return_complex=False, length=length)


def outputs_to_waveforms(mask, n_re, n_im, length, nfft=512, hop=256):
    """utils.py:37-72 (model_outputs_to_waveforms): decompress, TRUE product mask*noisy, iSTFT.  mask [B,2,F,T]."""
    d = decompress_mask(mask)
    e_re = d[:, 0] * n_re - d[:, 1] * n_im
    e_im = d[:, 1] * n_re + d[:, 0] * n_im
    return istft_wave(e_re, e_im, nfft, hop, length)


def band_drop(x, groups):
    """feature.py:254-285 on [B,C,F,T]."""
    B, _, F, _ = x.shape
    assert B > groups, f"Batch size = {B}, num_groups = {groups}."
    if groups <= 1:
        return x
    F = F - F % groups
    parts = [x[g::groups, :, g:F:groups, :] for g in range(groups)]
    return torch.cat(parts, dim=0)


# ------------------------------------------------------------------ FullSubNet+ building blocks
def laplace_norm(x):
    """base_model.py:210-224: divide by the per-sample mean over (C,F,T) + 1e-5."""
    mu = x.mean(dim=tuple(range(1, x.dim())), keepdim=True)
    return x / (mu + 1e-5)


def tsse(x, P, pre):
    """attention_model.py:78-98 on [B,C,T']."""
    C = x.shape[1]
    feats = []
    for nm in ("smallConv1d", "middleConv1d", "largeConv1d"):
        y = Fn.conv1d(x, P[f"{pre}.{nm}.0.weight"], P[f"{pre}.{nm}.0.bias"], groups=C)
        feats.append(torch.relu(y.mean(dim=2)))
    f = torch.stack(feats, dim=2)                                     # [B,C,3]
    sq = Fn.linear(f, P[f"{pre}.feature_concate_fc.weight"], P[f"{pre}.feature_concate_fc.bias"])[..., 0]
    h = torch.relu(Fn.linear(sq, P[f"{pre}.fc1.weight"], P[f"{pre}.fc1.bias"]))
    s = torch.sigmoid(Fn.linear(h, P[f"{pre}.fc2.weight"], P[f"{pre}.fc2.bias"]))
    return x * s[:, :, None]


TCN_DILATIONS = (1, 2, 5, 9, 1, 2, 5, 9)   # sequence_model.py:48-57


def tcn_block(x, P, pre, dil):
    """causal_conv.py:96-108 (non-causal, skip connection)."""
    y = Fn.conv1d(x, P[f"{pre}.conv1x1.weight"], P[f"{pre}.conv1x1.bias"])
    y = Fn.prelu(y, P[f"{pre}.prelu1.weight"])
    y = Fn.group_norm(y, 1, P[f"{pre}.norm1.weight"], P[f"{pre}.norm1.bias"], eps=1e-8)
    y = Fn.conv1d(y, P[f"{pre}.depthwise_conv.weight"], P[f"{pre}.depthwise_conv.bias"],
                  padding=dil, dilation=dil, groups=y.shape[1])
    y = Fn.prelu(y, P[f"{pre}.prelu2.weight"])
    y = Fn.group_norm(y, 1, P[f"{pre}.norm2.weight"], P[f"{pre}.norm2.bias"], eps=1e-8)
    return x + Fn.conv1d(y, P[f"{pre}.sconv.weight"], P[f"{pre}.sconv.bias"])


def fullband_model(x, P, pre):
    """sequence_model.py:106-112 with the 8-block TCN + trailing ReLU, Linear, ReLU. [B,C,T']->[B,F,T']."""
    for i, d in enumerate(TCN_DILATIONS):
        x = tcn_block(x, P, f"{pre}.sequence_model.{i}", d)
    x = torch.relu(x)
    o = Fn.linear(x.transpose(1, 2), P[f"{pre}.fc_output_layer.weight"], P[f"{pre}.fc_output_layer.bias"])
    return torch.relu(o).transpose(1, 2)


def lstm2(x, P, pre):
    """nn.LSTM(I,H,2,batch_first) restated step by step (gate order i,f,g,o; zero initial state;
    sequence_model.py:113-123).  x [N,T,I] -> [N,T,H].  Uses the fused torch op when no hook needs
    the steps; `lstm2_steps` below is the explicit recurrence used to cross-check it."""
    w = [P[f"{pre}.weight_ih_l0"], P[f"{pre}.weight_hh_l0"], P[f"{pre}.bias_ih_l0"], P[f"{pre}.bias_hh_l0"],
         P[f"{pre}.weight_ih_l1"], P[f"{pre}.weight_hh_l1"], P[f"{pre}.bias_ih_l1"], P[f"{pre}.bias_hh_l1"]]
    H = w[1].shape[1]
    z = x.new_zeros(2, x.shape[0], H)
    out, _, _ = torch.lstm(x, (z, z), w, True, 2, 0.0, False, False, True)
    return out


def lstm2_steps(x, P, pre):
    N, T, _ = x.shape
    H = P[f"{pre}.weight_hh_l0"].shape[1]
    inp = x
    for layer in (0, 1):
        wi, wh = P[f"{pre}.weight_ih_l{layer}"], P[f"{pre}.weight_hh_l{layer}"]
        b = P[f"{pre}.bias_ih_l{layer}"] + P[f"{pre}.bias_hh_l{layer}"]
        h = x.new_zeros(N, H)
        c = x.new_zeros(N, H)
        outs = []
        for t in range(T):
            g = inp[:, t] @ wi.t() + h @ wh.t() + b
            i, f, gg, o = g.split(H, dim=1)
            c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
            h = torch.sigmoid(o) * torch.tanh(c)
            outs.append(h)
        inp = torch.stack(outs, dim=1)
    return inp


def subband_unfold(x, n):
    """base_model.py:15-46 on [B,1,F,T'] -> [B,F,2n+1,T'] (reflect pad in F, sliding window)."""
    if n < 1:
        return x.permute(0, 2, 1, 3)
    B, _, F, T = x.shape
    xp = Fn.pad(x, [0, 0, n, n], mode="reflect")[:, 0]               # [B,F+2n,T]
    win = xp.unfold(1, 2 * n + 1, 1)                                   # [B,F,T,2n+1]
    return win.permute(0, 1, 3, 2)


def subband_model(sb_in, P, pre):
    """[N,34,T'] -> [N,O,T'] (LSTM + Linear, no output activation; sequence_model.py:113-123)."""
    h = lstm2(sb_in.transpose(1, 2).contiguous(), P, f"{pre}.sequence_model")
    o = Fn.linear(h, P[f"{pre}.fc_output_layer.weight"], P[f"{pre}.fc_output_layer.bias"])
    return o.transpose(1, 2)


def _sub(P, prefix):
    n = len(prefix)
    return {k[n:]: v for k, v in P.items() if k.startswith(prefix)}


def restorer_forward(mag, re, im, P, groups=1, look_ahead=2, sb_neighbors=15, taps=None):
    """FullSubNet_Plus.forward, fullsubnet_plus.py:143-230.  [B,1,F,T] x3 -> compressed cIRM [B,2,F',T]."""
    mag, re, im = (Fn.pad(t, [0, look_ahead]) for t in (mag, re, im))
    B, _, F, T = mag.shape
    fb_in = tsse(laplace_norm(mag).reshape(B, F, T), P, "channel_attention")
    fbr_in = tsse(laplace_norm(re).reshape(B, F, T), P, "channel_attention_real")
    fbi_in = tsse(laplace_norm(im).reshape(B, F, T), P, "channel_attention_imag")
    fb = fullband_model(fb_in, P, "fb_model")
    fbr = fullband_model(fbr_in, P, "fb_model_real")
    fbi = fullband_model(fbi_in, P, "fb_model_imag")
    if taps is not None:
        taps.update(att_mag=fb_in, att_real=fbr_in, att_imag=fbi_in, fb_mag=fb, fb_real=fbr, fb_imag=fbi)
    unf = subband_unfold(fb_in.reshape(B, 1, F, T), sb_neighbors)     # attention-scaled input (:203)
    sb = torch.cat([unf, fb[:, :, None], fbr[:, :, None], fbi[:, :, None]], dim=2)
    sb = laplace_norm(sb)
    Fp = F
    if B > 1:
        sb = band_drop(sb.permute(0, 2, 1, 3), groups).permute(0, 2, 1, 3)
        Fp = sb.shape[1]
    sb = sb.reshape(B * Fp, sb.shape[2], T)
    if taps is not None:
        taps["sb_in"] = sb
    o = subband_model(sb, P, "sb_model")                              # [B*F',2,T']
    if taps is not None:
        taps["sb_out"] = o
    o = o.reshape(B, Fp, -1, T).permute(0, 2, 1, 3)
    return o[..., look_ahead:]


def direction_forward(nmag, nre, nim, emag, ere, eim, P, n_dirs, groups=1, look_ahead=2, sb_neighbors=15,
                      taps=None):
    """MultiDirectionFullSubNet_Plus.forward, networks.py:63-163 -> [B,2K,F',T]."""
    nmag, nre, nim, emag, ere, eim = (Fn.pad(t, [0, look_ahead]) for t in (nmag, nre, nim, emag, ere, eim))
    B, _, F, T = nmag.shape

    def att(x, nm):
        return tsse(laplace_norm(x).reshape(B, F, T), P, nm)

    fb_in = torch.cat([att(nmag, "channel_attention"), att(emag, "channel_attention")], dim=1)
    fbr_in = torch.cat([att(nre, "channel_attention_real"), att(ere, "channel_attention_real")], dim=1)
    fbi_in = torch.cat([att(nim, "channel_attention_imag"), att(eim, "channel_attention_imag")], dim=1)
    fb = fullband_model(fb_in, P, "fb_model")
    fbr = fullband_model(fbr_in, P, "fb_model_real")
    fbi = fullband_model(fbi_in, P, "fb_model_imag")
    if taps is not None:
        taps.update(att_mag=fb_in, att_real=fbr_in, att_imag=fbi_in, fb_mag=fb, fb_real=fbr, fb_imag=fbi)
    unf = subband_unfold(nmag, sb_neighbors)                           # RAW padded magnitude (:133)
    sb = torch.cat([unf, fb[:, :, None], fbr[:, :, None], fbi[:, :, None]], dim=2)
    sb = laplace_norm(sb)
    Fp = F
    if B > 1:
        sb = band_drop(sb.permute(0, 2, 1, 3), groups).permute(0, 2, 1, 3)
        Fp = sb.shape[1]
    sb = sb.reshape(B * Fp, sb.shape[2], T)
    if taps is not None:
        taps["sb_in"] = sb
    o = subband_model(sb, P, "sb_model")                              # [B*F',2K,T']
    if taps is not None:
        taps["sb_out"] = o
    o = o.reshape(B, Fp, n_dirs, 2, T).permute(0, 2, 3, 1, 4)[..., look_ahead:]
    return o.reshape(B, 2 * n_dirs, Fp, -1)


# ------------------------------------------------------------------ NPPC pieces
def gram_schmidt_crm(x):
    """pc_wrapper.py:8-44 on [B,K,2,F,T]; coefficient is sum(conj(w) * w_hat),
    w_hat detached, no epsilon."""
    B, K, _, F, T = x.shape
    z = torch.complex(x[:, :, 0], x[:, :, 1]).reshape(B, K, -1)
    outs, hats = [], []
    for i in range(K):
        w = z[:, i]
        for wh in hats:
            w = w - wh * (w.conj() * wh).sum(dim=1, keepdim=True)
        wd = w.detach()
        hats.append(wd / wd.norm(dim=1, keepdim=True))
        outs.append(w)
    o = torch.stack(outs, dim=1).reshape(B, K, F, T)
    return torch.stack([o.real, o.imag], dim=2)


def second_moment_weight(step, grace=500, lam=1.0):
    """trainer.py:337-340."""
    return max(min(-1 + 2 * step / grace, 1), 1e-6) * lam


def nppc_loss(w_mat, gt, pred, step, grace=500, lam=1.0):
    """trainer.py:259-317 -> (reconst_err[B], objective, log dict)."""
    B, K = w_mat.shape[:2]
    w = w_mat.reshape(B, K, 2, -1)
    g = gt.reshape(B, 2, -1)
    p = pred.reshape(B, 2, -1)
    w_norms = torch.sqrt((w * w).sum(dim=(2, 3)))
    w_hat = w / (w_norms[..., None, None] + 1e-8)
    err = g - p
    err_norm = torch.sqrt((err * err).sum(dim=(1, 2)))
    err = err / (err_norm[:, None, None] + 1e-8)
    w_norms = w_norms / (err_norm[:, None] + 1e-8)
    ec = torch.complex(err[:, 0], err[:, 1])
    wc = torch.complex(w_hat[:, :, 0], w_hat[:, :, 1])
    proj = (wc.conj() * ec[:, None]).sum(dim=-1)
    pm = proj.abs()
    reconst = 1 - (pm ** 2).sum(dim=1)
    sm = (w_norms ** 2 - pm.detach() ** 2) ** 2
    obj = reconst.mean() + second_moment_weight(step, grace, lam) * sm.mean()
    log = dict(err_norm=err_norm.detach(), err_proj=proj.detach(), err_proj_mag=pm.detach(),
               w_norms=w_norms.detach(), reconst_err=reconst.detach(), second_moment_mse=sm.detach(),
               objective=obj.detach(), w_mat=w_mat.detach(), pred_crm=pred.detach())
    return reconst, obj, log


def nppc_forward(noisy, P, n_dirs, stft=(512, 256, 512), g_rest=1, g_pc=1, sb_neighbors=15, taps=None):
    """NPPCModel.forward (nppc_model.py:58-115) with the restorer run once.
    Returns (w_mat [B,K,2,F',T], pred_crm_full [B,2,F,T] compressed, parts dict)."""
    Pr = _sub(P, "pretrained_restoration_model.")
    Pd = _sub(P, "audio_pc_wrapper.net.")
    mag, re, im = stft_parts(noisy, *stft)
    with torch.no_grad():
        tr = {} if taps is not None else None
        pred = restorer_forward(mag, re, im, Pr, groups=g_rest, sb_neighbors=sb_neighbors, taps=tr)
        dec = decompress_mask(pred)
        emag, ere, eim = apply_mask_swapped(dec[:, 0], dec[:, 1], re[:, 0], im[:, 0])
    tp = {} if taps is not None else None
    raw = direction_forward(mag, re, im, emag[:, None], ere[:, None], eim[:, None], Pd, n_dirs,
                            groups=g_pc, sb_neighbors=sb_neighbors, taps=tp)
    B, _, Fp, T = raw.shape
    w_mat = gram_schmidt_crm(raw.reshape(B, n_dirs, 2, Fp, T))
    parts = dict(mag=mag, re=re, im=im, pred=pred, dec=dec, emag=emag, ere=ere, eim=eim, raw=raw)
    if taps is not None:
        taps["rest"], taps["pc"] = tr, tp
    return w_mat, pred, parts


def nppc_step(noisy, clean, P, n_dirs, step, stft=(512, 256, 512), g_rest=1, g_pc=1, sb_neighbors=15,
              grace=500, lam=1.0, reference_shaped=False, taps=None):
    """One base_step (trainer.py:234-317).  reference_shaped=True re-executes the frozen restorer and
    the noisy STFT the way the reference does (2x restorer, 3x STFT) -- same numbers, used only for
    the 'reference-shaped CPU' timing row."""
    w_mat, pred_full, parts = nppc_forward(noisy, P, n_dirs, stft, g_rest, g_pc, sb_neighbors, taps)
    if reference_shaped:
        Pr = _sub(P, "pretrained_restoration_model.")
        with torch.no_grad():
            m2, r2, i2 = stft_parts(noisy, *stft)
            pred_full = restorer_forward(m2, r2, i2, Pr, groups=g_rest, sb_neighbors=sb_neighbors)
            stft_parts(noisy, *stft)
    _, c_re, c_im = stft_parts(clean, *stft)
    gt = ideal_mask(parts["re"][:, 0], parts["im"][:, 0], c_re[:, 0], c_im[:, 0])
    gt = band_drop(gt, g_pc)
    pred = band_drop(pred_full, g_pc)
    reconst, obj, log = nppc_loss(w_mat, gt, pred, step, grace, lam)
    log["gt_crm"] = gt
    log["pred_crm_full"] = pred_full
    log["parts"] = parts
    return reconst, obj, log


def adam_step(params, grads, state, t, lr=1e-4, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam (no amsgrad, no weight decay) restated; in-place on `params`; t is 1-based."""
    for k, p in params.items():
        g = grads.get(k)
        if g is None:
            continue
        m, v = state.setdefault(k, (torch.zeros_like(p), torch.zeros_like(p)))
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1 = 1 - b1 ** t
        bc2 = 1 - b2 ** t
        denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(m, denom, value=-lr / bc1)


def mix_with_snr(clean, noise, snr_db, target_dbfs=-25.0):
    """AudioDataset._normalize_audio + _mix_with_snr (dataset/audio_dataset.py:92-152) for one clip [L] each:
    clean -> target dBFS, noise scaled to the SNR, clip guard at 0.99.  Returns (noisy, clean)."""
    rms = clean.pow(2).mean().sqrt()
    gain = 10 ** ((target_dbfs - 20 * torch.log10(rms + 1e-8)) / 20)
    clean = clean * gain
    clean_power, noise_power = clean.pow(2).mean(), noise.pow(2).mean()
    scale = torch.sqrt(clean_power / (10 ** (snr_db / 10) * noise_power + 1e-8))
    noisy = clean + noise * scale
    max_amp = torch.max(torch.abs(noisy))
    if max_amp > 0.99:
        f = 0.99 / max_amp
        noisy, clean = noisy * f, clean * f
    return noisy, clean
