"""TEST INFRASTRUCTURE (oracle side) -- deterministic, reference-free weights.

No pretrained checkpoint exists offline, so goldens, parity tests,
smoke() and bench.py all use weights regenerated from a seed.  The generator is numpy
PCG64 keyed by (seed, crc32(parameter name)), so any process -- the golden script that
feeds the reference modules in this container, the CPU oracle, and the HIP product on the
GPU box -- rebuilds bit-identical tensors without shipping 95 MB of floats.

`state_spec()` restates the reference wire format (parameter names
and shapes created by FullSubNet_plus/.../fullsubnet_plus.py:46-139,
audio_zen/model/module/{attention_model.py:49-74, causal_conv.py:68-81,
sequence_model.py:30-82} and nppc_audio/networks.py:18-61).  The golden script asserts
it against the reference modules' own state_dict().
"""
from collections import OrderedDict
import zlib

import numpy as np


TCN_HIDDEN = 512   # TCNBlock default hidden_channel; SequenceModel("TCN") never overrides it
                   # (sequence_model.py:47-58, causal_conv.py:68) -> fb_model_hidden_size is dead config


def fullsubnet_spec(num_freqs=257, fb_in=None, sb_neighbors=15, fb_neighbors=0,
                    sb_hidden=384, out_size=2, kersize=(3, 5, 10)):
    """Ordered {name: shape} of one FullSubNet+-shaped net.

    fb_in = channel count of the full-band TCN input: num_freqs for the restorer,
    2*num_freqs for the direction net (noisy ++ enhanced).
    """
    F = num_freqs
    C = F if fb_in is None else fb_in
    fb_hidden = TCN_HIDDEN
    spec = OrderedDict()
    for att in ("channel_attention", "channel_attention_real", "channel_attention_imag"):
        for conv, k in zip(("smallConv1d", "middleConv1d", "largeConv1d"), kersize):
            spec[f"{att}.{conv}.0.weight"] = (F, 1, k)
            spec[f"{att}.{conv}.0.bias"] = (F,)
        spec[f"{att}.feature_concate_fc.weight"] = (1, 3)
        spec[f"{att}.feature_concate_fc.bias"] = (1,)
        spec[f"{att}.fc1.weight"] = (F // 2, F)
        spec[f"{att}.fc1.bias"] = (F // 2,)
        spec[f"{att}.fc2.weight"] = (F, F // 2)
        spec[f"{att}.fc2.bias"] = (F,)
    for fb in ("fb_model", "fb_model_real", "fb_model_imag"):
        for i in range(8):
            p = f"{fb}.sequence_model.{i}"
            spec[f"{p}.conv1x1.weight"] = (fb_hidden, C, 1)
            spec[f"{p}.conv1x1.bias"] = (fb_hidden,)
            spec[f"{p}.prelu1.weight"] = (1,)
            spec[f"{p}.norm1.weight"] = (fb_hidden,)
            spec[f"{p}.norm1.bias"] = (fb_hidden,)
            spec[f"{p}.depthwise_conv.weight"] = (fb_hidden, 1, 3)
            spec[f"{p}.depthwise_conv.bias"] = (fb_hidden,)
            spec[f"{p}.prelu2.weight"] = (1,)
            spec[f"{p}.norm2.weight"] = (fb_hidden,)
            spec[f"{p}.norm2.bias"] = (fb_hidden,)
            spec[f"{p}.sconv.weight"] = (C, fb_hidden, 1)
            spec[f"{p}.sconv.bias"] = (C,)
        spec[f"{fb}.fc_output_layer.weight"] = (F, C)
        spec[f"{fb}.fc_output_layer.bias"] = (F,)
    sb_in = (2 * sb_neighbors + 1) + 3 * (2 * fb_neighbors + 1)
    H = sb_hidden
    spec["sb_model.sequence_model.weight_ih_l0"] = (4 * H, sb_in)
    spec["sb_model.sequence_model.weight_hh_l0"] = (4 * H, H)
    spec["sb_model.sequence_model.bias_ih_l0"] = (4 * H,)
    spec["sb_model.sequence_model.bias_hh_l0"] = (4 * H,)
    spec["sb_model.sequence_model.weight_ih_l1"] = (4 * H, H)
    spec["sb_model.sequence_model.weight_hh_l1"] = (4 * H, H)
    spec["sb_model.sequence_model.bias_ih_l1"] = (4 * H,)
    spec["sb_model.sequence_model.bias_hh_l1"] = (4 * H,)
    spec["sb_model.fc_output_layer.weight"] = (out_size, H)
    spec["sb_model.fc_output_layer.bias"] = (out_size,)
    return spec


def restorer_spec(**kw):
    return fullsubnet_spec(**kw)


def direction_spec(n_directions, num_freqs=257, **kw):
    return fullsubnet_spec(num_freqs=num_freqs, fb_in=2 * num_freqs,
                           out_size=2 * n_directions, **kw)


def nppc_spec(n_directions, **kw):
    """Spec of NPPCModel.state_dict(): both nets under their reference prefixes."""
    spec = OrderedDict()
    for k, v in restorer_spec(**kw).items():
        spec["pretrained_restoration_model." + k] = v
    for k, v in direction_spec(n_directions, **kw).items():
        spec["audio_pc_wrapper.net." + k] = v
    return spec


UNET_PLAN = (   # (block, conv-path prefix inside the block, in_ch, out_ch) -- tmp_utils.py:8-99, unet.py:247-260
    ("inc", "conv.conv", None, 64), ("down1", "mpconv.1.conv", 64, 128), ("down2", "mpconv.1.conv", 128, 256),
    ("down3", "mpconv.1.conv", 256, 512), ("down4", "mpconv.1.conv", 512, 512), ("up1", "conv.conv", 1024, 256),
    ("up2", "conv.conv", 512, 128), ("up3", "conv.conv", 256, 64), ("up4", "conv.conv", 128, 64))


def unet_spec(in_channels, out_channels):
    """Ordered {name: shape} of the inpainting U-Net's state_dict (inpainting/networks/unet.py:247-290;
    double_conv = Sequential[conv3x3, BatchNorm2d, LeakyReLU, conv3x3, BatchNorm2d, LeakyReLU(, Dropout)],
    tmp_utils.py:8-37).  BatchNorm buffers are part of the wire format."""
    spec = OrderedDict()
    for blk, path, cin, cout in UNET_PLAN:
        cin = in_channels if cin is None else cin
        for conv_i, bn_i, ci in ((0, 1, cin), (3, 4, cout)):
            spec[f"{blk}.{path}.{conv_i}.weight"] = (cout, ci, 3, 3)
            spec[f"{blk}.{path}.{conv_i}.bias"] = (cout,)
            spec[f"{blk}.{path}.{bn_i}.weight"] = (cout,)
            spec[f"{blk}.{path}.{bn_i}.bias"] = (cout,)
            spec[f"{blk}.{path}.{bn_i}.running_mean"] = (cout,)
            spec[f"{blk}.{path}.{bn_i}.running_var"] = (cout,)
            spec[f"{blk}.{path}.{bn_i}.num_batches_tracked"] = ()
    spec["outc.conv.weight"] = (out_channels, 64, 1, 1)
    spec["outc.conv.bias"] = (out_channels,)
    return spec


def inpainting_spec(n_dirs):
    """Spec of the inpainting NPPCModel.state_dict() (inpainting/nppc/nppc_model.py:33-117):
    RestorationWrapper(UNet(1,1)) under `pretrained_restoration_model.net.`, UNet(2,K) under `pc_wrapper.net.`."""
    spec = OrderedDict()
    for k, v in unet_spec(1, 1).items():
        spec["pretrained_restoration_model.net." + k] = v
    for k, v in unet_spec(2, n_dirs).items():
        spec["pc_wrapper.net." + k] = v
    return spec


def _is_unet_bn(name):
    parts = name.split(".")
    return len(parts) >= 3 and parts[-3] == "conv" and parts[-2] in ("1", "4")


def _draw(name, shape, seed):
    rng = np.random.Generator(np.random.PCG64([seed, zlib.crc32(name.encode())]))
    if _is_unet_bn(name):
        last = name.rsplit(".", 1)[-1]
        if last == "num_batches_tracked":
            return np.asarray(7, np.int64)
        if last == "weight":
            return (1.0 + 0.1 * rng.standard_normal(shape)).astype(np.float32)
        if last == "running_var":
            return rng.uniform(0.5, 1.5, size=shape).astype(np.float32)
        return (0.1 * rng.standard_normal(shape)).astype(np.float32)      # bias, running_mean
    leaf = name.rsplit(".", 2)
    last = leaf[-1]
    parent = leaf[-2] if len(leaf) > 1 else ""
    if parent.startswith("norm"):
        if last == "weight":
            return (1.0 + 0.1 * rng.standard_normal(shape)).astype(np.float32)
        return (0.1 * rng.standard_normal(shape)).astype(np.float32)
    if parent.startswith("prelu"):
        return (0.25 + 0.05 * rng.standard_normal(shape)).astype(np.float32)
    if last.startswith("weight"):
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else int(shape[0])
        if "sequence_model.weight_" in name:      # nn.LSTM convention: 1/sqrt(hidden)
            fan_in = shape[0] // 4
        b = 1.0 / np.sqrt(max(fan_in, 1))
        return rng.uniform(-b, b, size=shape).astype(np.float32)
    # biases
    return rng.uniform(-0.05, 0.05, size=shape).astype(np.float32)


def make_weights(spec, seed):
    """{name: float32 ndarray} for every entry of `spec` (order-independent draws)."""
    return OrderedDict((k, _draw(k, tuple(v), seed)) for k, v in spec.items())


def synth_batch(batch, length, first_clip=0):
    """SURVEY.md section 8d synthetic 16 kHz noisy-speech batch (float32 numpy).

    clip i (seed 1234+i): clean = 0.05-scale low-passed white noise with a 4 Hz raised-cosine
    envelope; noise = white, SNR ~ U(-5, 20) dB; noisy = clean + noise, peak-limited to 0.99.
    Returns (noisy[B,L], clean[B,L]).
    """
    noisy = np.empty((batch, length), np.float32)
    clean = np.empty((batch, length), np.float32)
    t = np.arange(length, dtype=np.float64) / 16000.0
    for i in range(batch):
        rng = np.random.Generator(np.random.PCG64(1234 + first_clip + i))
        white = rng.standard_normal(length)
        a = 0.95
        # one-pole low-pass, vectorised through the closed form of the recursion
        # y[n] = a*y[n-1] + (1-a)*x[n]
        from scipy.signal import lfilter
        col = lfilter([1.0 - a], [1.0, -a], white)
        col /= (np.std(col) + 1e-12)
        env = 0.5 * (1.0 - np.cos(2.0 * np.pi * 4.0 * t + rng.uniform(0, 2 * np.pi)))
        c = 0.05 * col * (0.2 + env)
        snr_db = rng.uniform(-5.0, 20.0)
        n = rng.standard_normal(length)
        n *= np.sqrt(np.mean(c ** 2) / (10.0 ** (snr_db / 10.0)) / (np.mean(n ** 2) + 1e-20))
        y = c + n
        peak = np.max(np.abs(y))
        if peak > 0.99:
            s = 0.99 / peak
            y *= s
            c = c * s
        noisy[i] = y.astype(np.float32)
        clean[i] = c.astype(np.float32)
    return noisy, clean


def synth_inpaint_batch(batch, n_frames, nfft=255, hop=128, gap_frames=(3, 6), first_clip=0):
    """Synthetic inpainting batch at the API boundary of NPPCAudioInpaintingTrainer.base_step
    (inpainting/trainer/nppc_trainer.py:338-350): (masked_spec[B,2,F,T], mask_frames[B,T], clean_spec[B,2,F,T]).
    Restates the dataset's preparation (dataset/audio_dataset_inpainting.py:154-168, 223-251, 296-313):
    clean clip scaled to -25 dBFS, STFT (hann, centred), a run of zero frames, masked = clean * mask."""
    import torch
    length = (n_frames - 1) * hop + nfft - 2 * (nfft // 2)
    _, clean = synth_batch(batch, length, first_clip)
    c = torch.from_numpy(clean)
    rms = c.pow(2).mean(dim=1, keepdim=True).sqrt()
    c = c * 10 ** ((-25.0 - 20 * torch.log10(rms + 1e-8)) / 20)
    spec = torch.stft(c, nfft, hop, nfft, window=torch.hann_window(nfft), return_complex=True)
    clean_spec = torch.stack((spec.real, spec.imag), dim=1).float().numpy()
    mask = np.ones((batch, n_frames), np.float32)
    for i in range(batch):
        rng = np.random.Generator(np.random.PCG64(4321 + first_clip + i))
        g = int(rng.integers(gap_frames[0], gap_frames[1] + 1))
        s = int(rng.integers(2, max(3, n_frames - g - 2)))
        mask[i, s:s + g] = 0.0
    masked_spec = clean_spec * mask[:, None, None, :]
    return masked_spec.astype(np.float32), mask, clean_spec
