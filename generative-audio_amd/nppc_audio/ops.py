"""Thin host wrappers over the element-wise / front-end entry points of libnppc_hip.so.

Each function validates shapes the way the reference call site would fail, allocates the outputs
with torch (device memory plumbing only) and enqueues the HIP kernel on the current stream.
"""
import torch

from . import _hip as H

EPS32 = float(torch.finfo(torch.float32).eps)   # audio_zen/constant.py:8


def _f32c(x):
    H.require_gpu()
    if not x.is_cuda:
        raise RuntimeError("NPPC-audio HIP ops take device tensors")
    return x.contiguous().float()


def stft(wave, nfft, hop, want_mag=True):
    """utils.py:107-147 / trainer.py:349-355: centred, periodic-hann, onesided STFT.
    wave [B,L] -> (mag|None, real, imag) each [B,F,T]."""
    wave = _f32c(wave)
    if wave.dim() == 1:
        wave = wave[None]
    B, L = wave.shape
    F, T = nfft // 2 + 1, 1 + L // hop
    re = torch.empty(B, F, T, dtype=torch.float32, device=wave.device)
    im = torch.empty_like(re)
    mag = torch.empty_like(re) if want_mag else None
    H.call("nppc_stft", wave, re, im, mag, B, L, nfft, hop, H.stream())
    return mag, re, im


def drop_band(x, num_groups=2):
    """audio_zen/acoustics/feature.py:254-285 on [B,C,F,T]."""
    B, C, F, T = x.shape
    assert B > num_groups, f"Batch size = {B}, num_groups = {num_groups}. The batch size should larger than the num_groups."
    if num_groups <= 1:
        return x
    x = _f32c(x)
    Fo = (F - F % num_groups) // num_groups
    out = torch.empty(B, C, Fo, T, dtype=torch.float32, device=x.device)
    H.call("nppc_dropband", x, out, B, C, F, T, num_groups, H.stream())
    return out


def cirm_build_compress(n_re, n_im, c_re, c_im, num_groups=1):
    """mask.py:24-54 + trainer.py:359-362: compressed cIRM in [B,2,F',T], drop-band applied."""
    n_re, n_im, c_re, c_im = (_f32c(t) for t in (n_re, n_im, c_re, c_im))
    B, F, T = n_re.shape
    assert B > num_groups, f"Batch size = {B}, num_groups = {num_groups}. The batch size should larger than the num_groups."
    Fo = F if num_groups <= 1 else (F - F % num_groups) // num_groups
    out = torch.empty(B, 2, Fo, T, dtype=torch.float32, device=n_re.device)
    H.call("nppc_cirm_build_compress", n_re, n_im, c_re, c_im, out, B, F, T, num_groups, EPS32, H.stream())
    return out


def cirm_decompress_apply_conj(crm, n_re, n_im, want_dec=False):
    """mask.py:57-60 then utils.py:241-249 (-> :75-79 with real/imag swapped = conj(mask)*noisy).
    crm [B,2,F,T] compressed; n_re/n_im [B,F,T] -> (dec [B,F,T,2]|None, enh_mag, enh_real, enh_imag)."""
    crm, n_re, n_im = _f32c(crm), _f32c(n_re), _f32c(n_im)
    B, _, F, T = crm.shape
    emag = torch.empty(B, F, T, dtype=torch.float32, device=crm.device)
    ere, eim = torch.empty_like(emag), torch.empty_like(emag)
    dec = torch.empty(B, F, T, 2, dtype=torch.float32, device=crm.device) if want_dec else None
    H.call("nppc_cirm_decompress_apply_conj", crm, n_re, n_im, dec, emag, ere, eim, B, F, T, H.stream())
    return dec, emag, ere, eim


def cirm_decompress_apply(crm, n_re, n_im):
    """decompress_cIRM + TRUE complex product mask*noisy (utils.py:37-58: model_outputs_to_waveforms' mask application).
    crm [B,2,F,T] compressed -> (enh_mag, enh_real, enh_imag) [B,F,T]."""
    crm, n_re, n_im = _f32c(crm), _f32c(n_re), _f32c(n_im)
    B, _, F, T = crm.shape
    emag = torch.empty(B, F, T, dtype=torch.float32, device=crm.device)
    ere, eim = torch.empty_like(emag), torch.empty_like(emag)
    H.call("nppc_cirm_decompress_apply", crm, n_re, n_im, None, emag, ere, eim, B, F, T, H.stream())
    return emag, ere, eim


def istft(re, im, nfft, hop, length):
    """torch.istft(n_fft, hop, win_length=n_fft, hann window, center=True, length=length): [B,F,T] x2 -> [B,length]
    (utils.py:60-70, nppc_audio/validator.py:136-143)."""
    re, im = _f32c(re), _f32c(im)
    B, F, T = re.shape
    assert F == nfft // 2 + 1
    out = torch.empty(B, length, dtype=torch.float32, device=re.device)
    H.call("nppc_istft", re, im, out, B, T, nfft, hop, length, H.stream())
    return out


def model_outputs_to_waveforms(enhanced_masks, noisy_reals, noisy_imags, orig_length, nfft=512, hop=256):
    """utils.py:37-72: compressed cIRM [B,2,F,T] + noisy STFT [B,1,F,T] -> enhanced waveforms [B, orig_length]."""
    _, ere, eim = cirm_decompress_apply(enhanced_masks, noisy_reals.squeeze(1), noisy_imags.squeeze(1))
    return istft(ere, eim, nfft, hop, orig_length)


def crm_directions_to_spectrograms(w_mat, noisy_re, noisy_im):
    """NPPCAudioValidator._crm_directions_to_spectograms (nppc_audio/validator.py:55-102): every PC direction
    w_mat[:, k] (compressed cIRM [B,2,F,T]) is decompressed and applied to the noisy STFT with the TRUE complex product
    (utils.crm_to_spectogram, utils.py:252-256).  w_mat [B,K,2,F,T], noisy_re/im [B,F,T] -> (real, imag) [B,K,F,T]."""
    B, K, _, F, T = w_mat.shape
    re = torch.empty(B, K, F, T, dtype=torch.float32, device=w_mat.device)
    im = torch.empty_like(re)
    for k in range(K):
        _, r, i = cirm_decompress_apply(w_mat[:, k], noisy_re, noisy_im)
        re[:, k], im[:, k] = r, i
    return re, im


def pc_direction_waveforms(pred_crm, w_mat, alphas, noisy_re, noisy_im, length, nfft=512, hop=256):
    """PC synthesis of the validator (nppc_audio/validator.py:148-302): waveforms of `enhanced + alpha * PC_k` for every
    direction k and every alpha, all (K * len(alphas) + 1) inverse STFTs as ONE batched overlap-add launch.
    pred_crm [B,2,F,T] (compressed), w_mat [B,K,2,F,T], alphas: sequence of floats -> (enhanced [B,length],
    variants [B,K,len(alphas),length])."""
    B, K = w_mat.shape[:2]
    _, e_re, e_im = cirm_decompress_apply(pred_crm, noisy_re, noisy_im)
    d_re, d_im = crm_directions_to_spectrograms(w_mat, noisy_re, noisy_im)
    a = torch.as_tensor(list(alphas), dtype=torch.float32, device=w_mat.device)
    A = a.numel()
    v_re = (e_re[:, None, None] + a[None, None, :, None, None] * d_re[:, :, None]).reshape(B * K * A, *e_re.shape[1:])
    v_im = (e_im[:, None, None] + a[None, None, :, None, None] * d_im[:, :, None]).reshape(B * K * A, *e_im.shape[1:])
    waves = istft(torch.cat((e_re, v_re)), torch.cat((e_im, v_im)), nfft, hop, length)
    return waves[:B], waves[B:].view(B, K, A, length)
