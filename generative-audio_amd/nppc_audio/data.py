"""Data side of the NPPC-audio trainer: the config classes the reference trainer imports, the wav-folder dataset, the
on-device batch mixer and the synthetic batch source used by bench.py / smoke().

AudioDataSetConfig mirrors dataset/audio_dataset.py:9-29, DataConfig / DataLoaderConfig mirror
use_pre_trained_model/model_validator/config/schema.py:19-27.  `AudioDataset` mirrors dataset/audio_dataset.py:45-188
(random crop, noise assembly with silence gaps, dBFS normalisation, SNR mixing, clip guard) on PRE-DECODED clips: files are
decoded once at construction, items are cut from host memory.  `dataset[i]` returns (noisy[L], clean[L]) like the
reference; `DeviceMixLoader` (what NPPCAudioTrainer uses on a HIP device) draws the same random decisions, uploads the
un-mixed crops and mixes a whole minibatch with ONE nppc_mix_snr launch.
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


def _decode_wav(path, sample_rate):
    """one wav file -> mono float32 [n] at `sample_rate` (dataset/audio_dataset.py:69-90 uses torchaudio.load + mean over
    channels + torchaudio Resample; torchaudio is not part of this build: scipy decodes, and a rate mismatch is resampled
    with scipy's polyphase filter, which is NOT bit-identical to torchaudio's windowed-sinc kernel)"""
    from scipy.io import wavfile
    sr, a = wavfile.read(str(path))
    if a.size == 0:
        return None
    if a.dtype.kind == "i":
        a = a.astype(np.float32) / float(1 << (8 * a.dtype.itemsize - 1))
    elif a.dtype.kind == "u":                           # 8-bit PCM is unsigned
        a = (a.astype(np.float32) - 128.0) / 128.0
    else:
        a = a.astype(np.float32)
    if a.ndim > 1:
        a = a.mean(axis=1)
    if sr != sample_rate:
        from math import gcd
        from scipy.signal import resample_poly
        g = gcd(int(sr), int(sample_rate))
        a = resample_poly(a.astype(np.float64), sample_rate // g, sr // g).astype(np.float32)
    return torch.from_numpy(np.ascontiguousarray(a))


class AudioDataset(torch.utils.data.Dataset):
    """dataset/audio_dataset.py:45-188 on pre-decoded clips.

    AudioDataset(config)                       scans config.clean_path / config.noisy_path for *.wav (ValueError when a
                                               directory holds none, like the reference) and decodes every file ONCE
    AudioDataset(config, clean_clips=[...], noise_clips=[...])   tensor-backed: 1-D float tensors already at sample_rate
    seed: None = entropy-seeded like the reference's global `random`; an int makes the item stream reproducible.
    The random decisions of one item are drawn in the reference's order (crop start, noise files [+ their levels], SNR,
    [clean level]), so a generator in the same state as the reference's `random` yields the same item."""

    def __init__(self, config: AudioDataSetConfig, clean_clips=None, noise_clips=None, seed=None):
        import random
        self.config = config
        if clean_clips is None:
            self.clean_path = Path(config.clean_path).resolve()
            self.noisy_path = Path(config.noisy_path).resolve()
            self.clean_files = sorted(self.clean_path.rglob("*.wav"))
            self.noise_files = sorted(self.noisy_path.rglob("*.wav"))
            if not self.clean_files:
                raise ValueError(f"No WAV files found in clean directory: {self.clean_path}")
            if not self.noise_files:
                raise ValueError(f"No WAV files found in noise directory: {self.noisy_path}")
            clean_clips = [_decode_wav(f, config.sample_rate) for f in self.clean_files]
            noise_clips = [_decode_wav(f, config.sample_rate) for f in self.noise_files]
            clean_clips = [c for c in clean_clips if c is not None]
            noise_clips = [c for c in noise_clips if c is not None]
            if not clean_clips or not noise_clips:
                raise ValueError("every WAV file in the clean or noise directory is empty")
        self.clean = [torch.as_tensor(c, dtype=torch.float32).reshape(-1) for c in clean_clips]
        self.noise = [torch.as_tensor(c, dtype=torch.float32).reshape(-1) for c in noise_clips]
        # level of every noise file, once: _normalize_audio (:103-104) measures the WHOLE file each time it is drawn
        self.noise_rms_db = [20 * torch.log10(c.pow(2).mean().sqrt() + 1e-8) for c in self.noise]
        self.rng = random.Random(seed)

    def __len__(self) -> int:
        return len(self.clean)

    def _level(self):
        c = self.config
        if c.target_dB_FS_floating_value > 0.0:                        # :94-99
            return self.rng.uniform(c.target_dB_FS - c.target_dB_FS_floating_value,
                                    c.target_dB_FS + c.target_dB_FS_floating_value)
        return c.target_dB_FS

    def draw(self, idx):
        """the un-mixed ingredients of item idx: (clean crop [L], noise [L], snr dB, clean level dBFS) -- host copies and
        one scalar gain per noise file, no per-sample arithmetic besides that gain"""
        L = self.config.sub_sample_length
        clean = self.clean[idx]
        if clean.numel() > L:                                          # :170-172
            start = self.rng.randint(0, clean.numel() - L)
            clean = clean[start:start + L]
        else:                                                          # :173-176
            clean = torch.nn.functional.pad(clean, (0, L - clean.numel()))
        noise = torch.zeros(L)
        pos = 0
        while pos < L:                                                 # _get_noise_segment :113-132
            k = self.rng.randrange(len(self.noise))                    # == random.choice(self.noise_files)
            gain = 10 ** ((self._level() - self.noise_rms_db[k]) / 20)
            seg = self.noise[k]
            n = min(seg.numel(), L - pos)
            noise[pos:pos + n] = seg[:n] * gain
            pos += seg.numel() + self.config.silence_sample_length     # the silence gap stays zero
        snr = self.rng.uniform(self.config.snr_range[0], self.config.snr_range[1])
        return clean, noise, snr, self._level()

    def __getitem__(self, idx: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """(noisy [L], clean [L]) on the host, like the reference's item (whose dataset is CPU I/O in front of the step);
        the training loop does not come through here: DeviceMixLoader mixes whole minibatches on the device"""
        clean, noise, snr, level = self.draw(idx)
        rms = clean.pow(2).mean().sqrt()                               # _normalize_audio :103-108
        clean = clean * 10 ** ((level - 20 * torch.log10(rms + 1e-8)) / 20)
        clean_power, noise_power = clean.pow(2).mean(), noise.pow(2).mean()
        scale = torch.sqrt(clean_power / (10 ** (snr / 10) * noise_power + 1e-8))     # :142-145
        noisy = clean + noise * scale
        max_amp = torch.max(torch.abs(noisy))
        if max_amp > 0.99:                                             # clip guard :151-156
            noisy, clean = noisy * (0.99 / max_amp), clean * (0.99 / max_amp)
        return noisy, clean


class DeviceMixLoader:
    """Minibatches of an `AudioDataset` mixed ON the HIP device: per batch the host draws each item's random decisions
    and gathers the un-mixed crops (memcpy-sized work), ONE upload per tensor, ONE nppc_mix_snr launch (dBFS
    normalisation, SNR scaling, clip guard for the whole batch).  Iterates like the DataLoader it replaces:
    yields (noisy [B,L], clean [B,L]) device tensors; `batch_sampler` yields index lists (any torch sampler)."""

    def __init__(self, dataset: AudioDataset, batch_sampler, device="cuda", pin_memory=True):
        self.dataset, self.batch_sampler, self.device, self.pin = dataset, batch_sampler, device, pin_memory

    def __len__(self):
        return len(self.batch_sampler)

    def __iter__(self):
        ds = self.dataset
        for idxs in self.batch_sampler:
            items = [ds.draw(i) for i in idxs]
            clean = torch.stack([it[0] for it in items])
            noise = torch.stack([it[1] for it in items])
            meta = torch.tensor([[it[2], it[3]] for it in items], dtype=torch.float32)
            if self.pin:
                clean, noise, meta = clean.pin_memory(), noise.pin_memory(), meta.pin_memory()
            clean, noise, meta = (t.to(self.device, non_blocking=True) for t in (clean, noise, meta))
            yield mix_with_snr_on_device(clean, noise, meta[:, 0].contiguous(), target_item=meta[:, 1].contiguous())


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


def mix_with_snr_on_device(clean, noise, snr_db, target_dB_FS=-25.0, target_item=None):
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
    if target_item is not None:
        target_item = target_item.contiguous().float()
        assert target_item.shape == (B,)
    H.call("nppc_mix_snr", clean, noise, snr_db, float(target_dB_FS), target_item, noisy_out, clean_out, B, L, H.stream())
    return noisy_out, clean_out
