#!/usr/bin/env python3
"""Golden vectors for the wav-folder dataset's per-item arithmetic by running the reference's own
AudioDataset.__getitem__ (dataset/audio_dataset.py:160-188: crop / pad, _get_noise_segment :113-132, SNR draw,
_mix_with_snr :134-158) on pre-decoded clips.  torchaudio is absent offline, so the one method that touches it
(_load_and_process_audio, file decoding) is replaced by a lookup into in-memory clips; everything after it is the
reference's code, driven by Python's global `random` seeded per item.  Runs only in the build container; writes
tests/golden/dataset_items.npz (data only)."""
import os
import random
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
from make_goldens import install_placeholders  # noqa: E402

SR = 16000
# SNR draws down to -25 dB: the noise then peaks above 0.99 and the clip guard (audio_dataset.py:151-156) triggers
CFG = dict(sample_rate=SR, snr_range=(-25, 20), silence_length=0.01, sub_sample_length_seconds=0.25, target_dB_FS=-25.0)
CLEAN_LEN = [9000, 4000, 2500, 12000, 4001]        # longer than / equal to / shorter than the 4000-sample crop
NOISE_LEN = [1500, 2600, 900]                      # several segments (+160 samples of silence each) per item


def clips():
    rng = np.random.Generator(np.random.PCG64(2024))
    clean = [(0.1 * rng.standard_normal(n) * (0.3 + np.abs(np.sin(np.arange(n) / 700.0)))).astype(np.float32) for n in CLEAN_LEN]
    noise = [(0.03 * (k + 1) * rng.standard_normal(n)).astype(np.float32) for k, n in enumerate(NOISE_LEN)]
    return clean, noise


def main():
    install_placeholders()
    sys.path.insert(0, "/root/reference")
    from dataset.audio_dataset import AudioDataSetConfig, AudioDataset
    clean, noise = clips()
    out = {f"clean{i}": c for i, c in enumerate(clean)}
    out.update({f"noise{i}": c for i, c in enumerate(noise)})
    for fl in (0.0, 3.0):                           # fixed and floating normalisation level
        cfg = AudioDataSetConfig(clean_path=".", noisy_path=".", target_dB_FS_floating_value=fl, **CFG)
        ds = AudioDataset.__new__(AudioDataset)     # the constructor only globs directories for *.wav
        ds.config = cfg
        ds.clean_files = [f"clean{i}" for i in range(len(clean))]
        ds.noise_files = [f"noise{i}" for i in range(len(noise))]
        table = {**{f"clean{i}": c for i, c in enumerate(clean)}, **{f"noise{i}": c for i, c in enumerate(noise)}}
        ds._load_and_process_audio = types.MethodType(lambda self, f: torch.from_numpy(table[f].copy())[None], ds)
        for idx in range(len(clean)):
            for rep in range(2):
                seed = 1000 * idx + rep + (500 if fl else 0)
                random.seed(seed)
                n, c = ds[idx]
                out[f"fl{fl:g}.i{idx}.s{seed}.noisy"] = n.numpy()
                out[f"fl{fl:g}.i{idx}.s{seed}.clean"] = c.numpy()
    np.savez_compressed(os.path.join(HERE, "dataset_items.npz"), **out)
    peaks = [float(np.abs(v).max()) for k, v in out.items() if k.endswith(".noisy")]
    print(f"wrote {len(out)} arrays; max |noisy| per item: {np.round(peaks, 3)}")


if __name__ == "__main__":
    main()
