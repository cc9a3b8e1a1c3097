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


def _draw(name, shape, seed):
    rng = np.random.Generator(np.random.PCG64([seed, zlib.crc32(name.encode())]))
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
