"""TEST INFRASTRUCTURE -- CPU restatement of the inpainting NPPC train step.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product (generative-audio_amd/) never does.  Pinned against outputs of the reference itself
(tests/golden/inp_*.npz, written by tests/golden/make_goldens_inpainting.py in the build container).

Functional torch-CPU code over a flat {state_dict name: tensor} mapping; every function cites the
reference lines it follows.  dtype follows the inputs (fp32 to mirror the reference, fp64 for the
numerical floor).
"""
import math

import torch
import torch.nn.functional as F

from .weights import UNET_PLAN

LEAK = 0.2          # nn.LeakyReLU(0.2), tmp_utils.py:19,26
BN_EPS = 1e-5       # nn.BatchNorm2d defaults
BN_MOMENTUM = 0.1


def preprocess(clean_spec, masked_spec, mask_frames):
    """utils.preprocess_data (utils.py:294-306) + preprocess_log_magnitude (utils.py:273-291):
    |.| of the [B,2,F,T] STFT pairs, log(mag + 1e-6), BATCH-GLOBAL mean / unbiased std of the clean
    log-magnitude applied to both; frame mask [B,T] -> [B,1,F,T]."""
    B, _, Fq, T = clean_spec.shape
    mask = mask_frames[:, None, None, :].expand(-1, 1, Fq, -1)
    cmag = torch.sqrt(clean_spec[:, 0] ** 2 + clean_spec[:, 1] ** 2)[:, None]
    mmag = torch.sqrt(masked_spec[:, 0] ** 2 + masked_spec[:, 1] ** 2)[:, None]
    clog = torch.log(cmag + 1e-6)
    mean, std = clog.mean(), clog.std()
    return (clog - mean) / std, mask, (torch.log(mmag + 1e-6) - mean) / std, mean, std


def _bn(x, P, pre, train, new_stats):
    """nn.BatchNorm2d: batch statistics (biased variance) in train mode, running statistics in eval;
    train mode also produces the momentum-0.1 running update with the UNBIASED variance."""
    if not train:
        return F.batch_norm(x, P[pre + "running_mean"].to(x.dtype), P[pre + "running_var"].to(x.dtype),
                            P[pre + "weight"], P[pre + "bias"], False, 0.0, BN_EPS)
    mean = x.mean(dim=(0, 2, 3))
    var = x.var(dim=(0, 2, 3), unbiased=False)
    n = x.numel() // x.shape[1]
    if new_stats is not None:
        with torch.no_grad():
            new_stats[pre + "running_mean"] = (1 - BN_MOMENTUM) * P[pre + "running_mean"].to(x.dtype) + BN_MOMENTUM * mean
            new_stats[pre + "running_var"] = ((1 - BN_MOMENTUM) * P[pre + "running_var"].to(x.dtype)
                                              + BN_MOMENTUM * var * (n / max(n - 1, 1)))
            new_stats[pre + "num_batches_tracked"] = P[pre + "num_batches_tracked"] + 1
    xh = (x - mean[None, :, None, None]) * torch.rsqrt(var + BN_EPS)[None, :, None, None]
    return xh * P[pre + "weight"][None, :, None, None] + P[pre + "bias"][None, :, None, None]


def double_conv(x, P, pre, train, new_stats, taps=None, keep=None, p_drop=0.0):
    """tmp_utils.double_conv (tmp_utils.py:8-37): (conv3x3 pad 1 -> BatchNorm -> LeakyReLU(0.2)) x 2.
    Dropout (down3/down4/up1/up2 of the restorer) is the identity on the train path: the restorer runs in eval mode
    (nppc_model.py:95,113) and the direction U-Net is configured with dropout 0 (config_nppc.yaml:33-36).
    MC-dropout (utils.enable_dropout, utils.py:334-338): `keep` is the 0/1 keep mask of nn.Dropout(p_drop)
    (tmp_utils.py:28-29), applied as x * keep / (1 - p)."""
    for ci, bi in ((0, 1), (3, 4)):
        x = F.conv2d(x, P[f"{pre}{ci}.weight"], P[f"{pre}{ci}.bias"], padding=1)
        if taps is not None:
            taps[f"{pre}{ci}.raw"] = x
        x = F.leaky_relu(_bn(x, P, f"{pre}{bi}.", train, new_stats), LEAK)
    if keep is not None:
        x = x * keep.to(x.dtype) / (1.0 - p_drop)
    return x


def unet_forward(x, P, pre="", train=False, new_stats=None, taps=None, keep=None, p_drop=0.0):
    """UNet.forward (inpainting/networks/unet.py:277-290) with down = MaxPool2d(2) + double_conv
    (tmp_utils.py:50-60) and up = bilinear x2 (align_corners=True) + pad-to-skip + cat([skip, up]) +
    double_conv (tmp_utils.py:63-91), outc = 1x1 conv (tmp_utils.py:94-99)."""
    feats = []
    h = x
    for blk, path, _, _ in UNET_PLAN[:5]:
        if blk != "inc":
            h = F.max_pool2d(h, 2)
        h = double_conv(h, P, f"{pre}{blk}.{path}.", train, new_stats, taps, None if keep is None else keep.get(blk), p_drop)
        feats.append(h)
        if taps is not None:
            taps[f"{pre}{blk}.out"] = h
    h = feats[4]
    for (blk, path, _, _), skip in zip(UNET_PLAN[5:], (feats[3], feats[2], feats[1], feats[0])):
        h = F.interpolate(h, scale_factor=2, mode="bilinear", align_corners=True)
        dy, dx = skip.shape[2] - h.shape[2], skip.shape[3] - h.shape[3]
        h = F.pad(h, (dx // 2, dx - dx // 2, dy // 2, dy - dy // 2))
        h = double_conv(torch.cat([skip, h], dim=1), P, f"{pre}{blk}.{path}.", train, new_stats, taps,
                        None if keep is None else keep.get(blk), p_drop)
        if taps is not None:
            taps[f"{pre}{blk}.out"] = h
    return F.conv2d(h, P[pre + "outc.conv.weight"], P[pre + "outc.conv.bias"])


def restorer_forward(masked_norm, mask, P, pre="pretrained_restoration_model.net."):
    """RestorationWrapper.forward (unet.py:293-313), single-channel input: keep the known frames,
    fill the gap with the (eval-mode) U-Net's prediction."""
    with torch.no_grad():
        x = unet_forward(masked_norm, P, pre, train=False)
        return masked_norm * mask + x * (1 - mask)


def gram_schmidt_real(x):
    """gram_schmidt_to_spec_mag (inpainting/nppc/pc_wrapper.py:43-59): sequential projections onto the
    DETACHED unit vectors, no epsilon; the returned rows are orthogonal but not normalised."""
    shp = x.shape
    x = x.flatten(2)
    rows, units = [], []
    for i in range(x.shape[1]):
        w = x[:, i, :]
        for u in units:
            w = w - u * torch.sum(w * u, dim=-1, keepdim=True)
        wd = w.detach()
        units.append(wd / wd.norm(dim=-1, keepdim=True))
        rows.append(w)
    return torch.stack(rows, dim=1).view(*shp)


def direction_forward(masked_norm, pred_norm, mask, P, pre="pc_wrapper.net.", train=True, new_stats=None, taps=None):
    """NPPCModel.forward tail + AudioInpaintingPCWrapper.forward (nppc_model.py:136-145, pc_wrapper.py:75-87):
    U-Net(2 -> K) on cat(masked, restored), zero outside the gap, Gram-Schmidt."""
    raw = unet_forward(torch.cat((masked_norm, pred_norm), dim=1), P, pre, train, new_stats, taps)
    raw = raw * (1 - mask)
    if taps is not None:
        taps["pc_raw_masked"] = raw
    return gram_schmidt_real(raw)


def second_moment_weight(step, grace=500, lam=1.0):
    """_calculate_final_objective's ramp (nppc_trainer.py:680-685)."""
    return max(min(-1 + 2 * step / grace, 1), 1e-6) * lam


def inpaint_loss(w_mat, clean_norm, pred_norm, step, grace=500, lam=1.0):
    """base_step (nppc_trainer.py:352-385): real-valued residual projection with +1e-6 on both norms."""
    w = w_mat.flatten(2)
    w_norms = w.norm(dim=2) + 1e-6
    w_hat = w / w_norms[:, :, None]
    err = (clean_norm - pred_norm).flatten(1)
    err_norm = err.norm(dim=1) + 1e-6
    err = err / err_norm[:, None]
    w_norms = w_norms / err_norm[:, None]
    err_proj = torch.einsum("bki,bi->bk", w_hat, err)
    reconst_err = 1 - err_proj.pow(2).sum(dim=1)
    second_moment_mse = (w_norms.pow(2) - err_proj.detach().pow(2)).pow(2)
    objective = reconst_err.mean() + second_moment_weight(step, grace, lam) * second_moment_mse.mean()
    log = dict(w_mat=w_mat.detach(), err_norm=err_norm.detach(), err_proj=err_proj.detach(), w_norms=w_norms.detach(),
               reconst_err=reconst_err.detach(), second_moment_mse=second_moment_mse.detach(),
               objective=objective.detach())
    return reconst_err, objective, log


def inpaint_step(masked_spec, mask_frames, clean_spec, P, step, grace=500, lam=1.0, reference_shaped=False,
                 new_stats=None, taps=None):
    """One NPPCAudioInpaintingTrainer.base_step (nppc_trainer.py:338-385).  reference_shaped=True runs the
    frozen restorer twice like the reference (nppc_model.py:137, nppc_trainer.py:358) -- same numbers."""
    clean_norm, mask, masked_norm, mean, std = preprocess(clean_spec, masked_spec, mask_frames)
    pred = restorer_forward(masked_norm, mask, P)
    w_mat = direction_forward(masked_norm, pred, mask, P, train=True, new_stats=new_stats, taps=taps)
    if reference_shaped:
        pred = restorer_forward(masked_norm, mask, P)
    reconst, obj, log = inpaint_loss(w_mat, clean_norm, pred, step, grace, lam)
    log.update(clean_norm=clean_norm, masked_norm=masked_norm, pred_norm=pred, mean=mean, std=std)
    return reconst, obj, log


def clip_coef(grads, max_norm=1.0):
    """torch.nn.utils.clip_grad_norm_ (nppc_trainer.py:152): L2 norm over all gradients that exist,
    coefficient max_norm / (total + 1e-6) clamped to 1."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads if g is not None))
    return min(float(max_norm / (total + 1e-6)), 1.0), float(total)


def adam_step(params, grads, state, t, lr=1e-4, b1=0.5, b2=0.999, eps=1e-8):
    """torch.optim.Adam with the inpainting yaml's betas (config_nppc.yaml:63-67); t is 1-based."""
    for k, p in params.items():
        g = grads.get(k)
        if g is None:
            continue
        m, v = state.setdefault(k, (torch.zeros_like(p), torch.zeros_like(p)))
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (v.sqrt() / math.sqrt(1 - b2 ** t)).add_(eps)
        p.addcdiv_(m, denom, value=-lr / (1 - b1 ** t))


# ---- batch synthesis for the inpainting step (second half) ----------------------------
def normalize_dbfs(wave, target_db=-25.0):
    """AudioInpaintingDataset._normalize_audio (dataset/audio_dataset_inpainting.py:154-168), fixed target:
    rms over the whole clip, gain = 10 ** ((target - 20 log10(rms + 1e-8)) / 20)."""
    rms = wave.pow(2).mean(dim=-1, keepdim=True).sqrt()
    return wave * 10 ** ((target_db - 20 * torch.log10(rms + 1e-8)) / 20)


def gap_mask(length, start, end):
    """_create_random_mask with a given start (audio_dataset_inpainting.py:170-181): ones, zeros on [start, end)."""
    m = torch.ones(1, length)
    m[:, start:end] = 0
    return m


def time_to_spec_mask(mask_time, n_frames, length, win, hop, center=True):
    """time_to_spec_mask (audio_dataset_inpainting.py:223-251): frame t is 1 iff every sample of its (clamped)
    window [t*hop - win//2, +win) is 1; an empty window gives 0."""
    out = torch.zeros(n_frames)
    for t in range(n_frames):
        s = t * hop - (win // 2 if center else 0)
        e = min(s + win, length)
        s = max(s, 0)
        out[t] = 0.0 if e <= s else float(bool(mask_time[0, s:e].min() == 1))
    return out


def stft_pair(wave, nfft, hop, win):
    """utils.audio_to_stft (utils.py:150-175): centred hann STFT -> [B,2,F,T]."""
    s = torch.stft(wave, nfft, hop, win, window=torch.hann_window(win, dtype=wave.dtype), center=True, return_complex=True)
    return torch.stack((s.real, s.imag), dim=1)


def inpaint_sample(clean, start, end, nfft=255, hop=128, win=255):
    """__getitem__ after the crop (audio_dataset_inpainting.py:291-313) for one clip [1,L]:
    -> (stft_masked[2,F,T], mask_frames[T], stft_clean[2,F,T], masked_audio[1,L]); the masking is applied to the
    SPECTROGRAM frames (stft_clean * mask), not by transforming the masked audio."""
    L = clean.shape[1]
    mask = gap_mask(L, start, end).to(clean.dtype)
    spec = stft_pair(clean, nfft, hop, win)
    mf = time_to_spec_mask(mask, spec.shape[3], L, win, hop).to(clean.dtype)
    return (spec * mf[None, None, None, :])[0], mf, spec[0], clean * mask


# ---- MC-dropout + PCA baseline ---------------------------------------------------------
def pca_batch(outputs, n_components=5):
    """compute_pca_sklearn_batch (utils.py:393-496) restated with an exact SVD: outputs [K,B,D] ->
    (components [B,n,D], components * singular value, singular / sum(kept singular) [B,n], mean [B,D], singular [B,n]).
    Sign: sklearn.utils.extmath.svd_flip(u_based_decision=False) -- the largest-magnitude entry of each component is
    positive.  (scikit-learn picks its RANDOMIZED solver for 50 x ~1800 inputs, so the reference's own numbers are an
    approximation of these; compute_pca_and_importance_weights, utils.py:498-545, is the exact torch.linalg.svd variant
    without the sign rule.)"""
    K, B, D = outputs.shape
    n = min(n_components, K)
    pcs, scaled, wts, means, svs = [], [], [], [], []
    for b in range(B):
        x = outputs[:, b, :]
        mean = x.mean(dim=0)
        U, S, Vh = torch.linalg.svd((x - mean).double(), full_matrices=False)
        V, S = Vh[:n], S[:n]
        idx = V.abs().argmax(dim=1)
        V = V * torch.sign(V[torch.arange(n), idx])[:, None]
        pcs.append(V)
        scaled.append(V * S[:, None])
        wts.append(S / S.sum())
        means.append(mean)
        svs.append(S)
    return torch.stack(pcs), torch.stack(scaled), torch.stack(wts), torch.stack(means), torch.stack(svs)


def scatter_gap(values, mask4):
    """reconstruct_full_spec_batch (utils.py:600-625): values [B,N] or [B,n,N] -> zeros with the gap (mask == 0) filled"""
    B, _, Fq, T = mask4.shape
    hole = (mask4.reshape(B, Fq * T) == 0)
    if values.dim() == 3:
        out = torch.zeros(B, values.shape[1], Fq * T, dtype=values.dtype)
        for b in range(B):
            out[b][:, hole[b]] = values[b]
        return out.reshape(B, values.shape[1], Fq, T)
    out = torch.zeros(B, Fq * T, dtype=values.dtype)
    for b in range(B):
        out[b][hole[b]] = values[b]
    return out.reshape(B, Fq, T)


def base_step2_loss(w_mat, w_mc, singular_values, lam):
    """base_step2's loss terms (inpainting/trainer/nppc_trainer.py:255-323): row-normalised (norm + 1e-6) NPPC
    directions against row-normalised scaled MC components -> (reconst_err [B], second_moment_mse [B], objective,
    proj [B,K], w_norms [B,K])."""
    w_ = w_mat.flatten(2)
    w_norms = w_.norm(dim=2) + 1e-6
    w_hat = w_ / w_norms[:, :, None]
    m_ = w_mc.flatten(2)
    m_hat = m_ / (m_.norm(dim=2) + 1e-6)[:, :, None]
    proj = (w_hat * m_hat).sum(dim=2)
    reconst_err = (1 - proj.pow(2)).mean(dim=1)
    second = (w_norms.pow(2) - singular_values.pow(2)).pow(2).mean(dim=1)
    return reconst_err, second, reconst_err.mean() + lam * second.mean(), proj, w_norms


def compute_metrics(nppc_directions, mc_dropout_directions, pred_spec_mag, mean_prediction, clean_spec_mag, mask):
    """validator_nppc_model.compute_metrics (validator_nppc_model.py:742-828), one item: gap-only RMSE of the two
    predictions, ||e - W W^T e|| with row-normalised (norm + 1e-6) directions and the UNMASKED error, principal angles
    (QR bases, svdvals, degrees)."""
    import numpy as np

    def rmse(p):
        return torch.norm((p - clean_spec_mag)[mask == 0]).item()

    def residual(err, dirs):
        ef = err.reshape(err.shape[1], -1)
        d = dirs.reshape(dirs.shape[1], -1)
        d = d / (d.norm(dim=1) + 1e-6)[:, None]
        return torch.norm(ef.T - d.T @ (d @ ef.T)).item()

    def angles(a, b):
        qa, _ = torch.linalg.qr(a.reshape(a.shape[1], -1).T)
        qb, _ = torch.linalg.qr(b.reshape(b.shape[1], -1).T)
        s = torch.linalg.svdvals(qa.T @ qb)
        return (torch.arccos(torch.clamp(s, -1, 1)) * 180 / np.pi).tolist()

    err = pred_spec_mag - clean_spec_mag
    return {"nppc": {"rmse": rmse(pred_spec_mag), "residual_error": residual(err, nppc_directions)},
            "mc_dropout": {"rmse": rmse(mean_prediction), "residual_error": residual(err, mc_dropout_directions)},
            "principal_angles": angles(nppc_directions, mc_dropout_directions)}
