#!/usr/bin/env python3
"""Golden vectors for the batch-synthesis step by running the reference's own
AudioDataset._mix_with_snr / _normalize_audio (dataset/audio_dataset.py:92-152) on synthetic clips.
Runs only in the build container; writes tests/golden/mix_snr.npz (data only)."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
from make_goldens import install_placeholders  # noqa: E402
from oracle import weights as W  # noqa: E402


def main():
    install_placeholders()
    sys.path.insert(0, "/root/reference")
    from dataset.audio_dataset import AudioDataset
    fake = types.SimpleNamespace(config=types.SimpleNamespace(target_dB_FS=-25.0, target_dB_FS_floating_value=0.0))
    fake._normalize_audio = types.MethodType(AudioDataset._normalize_audio, fake)
    B, L = 6, 16000
    noisy_src, clean_src = W.synth_batch(B, L)
    rng = np.random.Generator(np.random.PCG64(99))
    noise = (noisy_src - clean_src).astype(np.float32)
    clean_src = clean_src.copy()
    clean_src[4] *= 30.0            # loud clip + low SNR -> the 0.99 clip guard triggers
    snr = np.array([0.0, 5.0, 10.0, 20.0, -30.0, 15.0], np.float32)   # clip 4: the 0.99 clip guard triggers
    noisy_out, clean_out = [], []
    for i in range(B):
        n, c = AudioDataset._mix_with_snr(fake, torch.from_numpy(clean_src[i:i + 1]), torch.from_numpy(noise[i:i + 1]),
                                          float(snr[i]))
        noisy_out.append(n.numpy())
        clean_out.append(c.numpy())
    out = dict(clean=clean_src, noise=noise, snr=snr, noisy_out=np.stack(noisy_out), clean_out=np.stack(clean_out))
    np.savez_compressed(os.path.join(HERE, "mix_snr.npz"), **out)
    print("peak per clip:", np.abs(out["noisy_out"]).max(axis=1))


if __name__ == "__main__":
    main()
