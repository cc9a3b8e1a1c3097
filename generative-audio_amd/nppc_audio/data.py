"""Config classes the reference trainer imports for its data side, plus the synthetic on-device batch source
used by bench.py / smoke().

AudioDataSetConfig mirrors dataset/audio_dataset.py:9-29, DataConfig / DataLoaderConfig mirror
use_pre_trained_model/model_validator/config/schema.py:19-27.  The wav-folder `AudioDataset` itself (random crop,
dBFS normalisation, SNR mixing, dataset/audio_dataset.py:45-188) is CPU I/O in front of the hot path and stays out
of scope this round: constructing it raises with that explanation.
"""
from pathlib import Path
from typing import Tuple, Union

import numpy as np
import pydantic
import torch


class AudioDataSetConfig(pydantic.BaseModel):
    clean_path: Union[str, Path]
    noisy_path: Union[str, Path]
    sample_rate: int = 16000
    snr_range: Tuple[int, int] = pydantic.Field(default_factory=lambda: (0, 20))
    silence_length: float = 0.2
    sub_sample_length_seconds: float = 3.0
    target_dB_FS: float = -25.0
    target_dB_FS_floating_value: float = 0.0
    sub_sample_length: int = pydantic.Field(None, description="Computed from sub_sample_length_seconds * sample_rate")
    silence_sample_length: int = pydantic.Field(None, description="Computed from silence_length * sample_rate")

    @pydantic.model_validator(mode='after')
    def compute_lengths(self) -> 'AudioDataSetConfig':
        self.sub_sample_length = int(self.sub_sample_length_seconds * self.sample_rate)
        self.silence_sample_length = int(self.silence_length * self.sample_rate)
        return self


class DataLoaderConfig(pydantic.BaseModel):
    batch_size: int = 8
    num_workers: int = 4
    pin_memory: bool = True
    shuffle: bool = False


class DataConfig(pydantic.BaseModel):
    dataset: AudioDataSetConfig
    data_path: str


class AudioDataset(torch.utils.data.Dataset):
    def __init__(self, config: AudioDataSetConfig):
        raise NotImplementedError(
            "the wav-folder AudioDataset (dataset/audio_dataset.py) is outside this build's hot path; pass "
            "`dataset=` to NPPCAudioTrainer (any torch Dataset yielding (noisy[L], clean[L]) float tensors)")


def synth_clip(i, length):
    """SURVEY section 8d synthetic clip i: (noisy, clean) float32 numpy, same recipe as oracle/weights.synth_batch
    (restated here so the product does not import test infrastructure)."""
    from scipy.signal import lfilter
    rng = np.random.Generator(np.random.PCG64(1234 + i))
    t = np.arange(length, dtype=np.float64) / 16000.0
    white = rng.standard_normal(length)
    a = 0.95
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
    return y.astype(np.float32), c.astype(np.float32)


class SyntheticNoisySpeech(torch.utils.data.Dataset):
    """Deterministic synthetic 16 kHz noisy/clean pairs (clip index -> seed 1234+index)."""

    def __init__(self, n_clips, length, first_clip=0):
        self.n, self.length, self.first = n_clips, length, first_clip

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        y, c = synth_clip(self.first + i, self.length)
        return torch.from_numpy(y), torch.from_numpy(c)


def mix_with_snr_on_device(clean, noise, snr_db, target_dB_FS=-25.0):
    """Batched, HBM-resident version of AudioDataset._mix_with_snr (dataset/audio_dataset.py:92-152):
    clean [B,L], noise [B,L], snr_db [B] (all on the HIP device) -> (noisy [B,L], clean [B,L]).  One kernel launch;
    lets the training loop synthesise minibatches from resident clip pools instead of a CPU DataLoader."""
    from . import _hip as H
    H.require_gpu()
    clean, noise = clean.contiguous().float(), noise.contiguous().float()
    snr_db = snr_db.contiguous().float()
    B, L = clean.shape
    assert noise.shape == (B, L) and snr_db.shape == (B,)
    noisy_out, clean_out = torch.empty_like(clean), torch.empty_like(clean)
    H.call("nppc_mix_snr", clean, noise, snr_db, float(target_dB_FS), noisy_out, clean_out, B, L, H.stream())
    return noisy_out, clean_out
