"""The wav-folder dataset's host side (nppc_audio/data.py: AudioDataset) against items produced by the REFERENCE's own
AudioDataset.__getitem__ (tests/golden/make_goldens_dataset.py), wav decoding, and the constructor's error behaviour."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dataset_items.npz")
CFG = dict(sample_rate=16000, snr_range=(-25, 20), silence_length=0.01, sub_sample_length_seconds=0.25, target_dB_FS=-25.0)


def _clips(z):
    return [z[f"clean{i}"] for i in range(5)], [z[f"noise{i}"] for i in range(3)]


@pytest.mark.parametrize("floating", [0.0, 3.0])
def test_items_equal_the_reference_items(floating):
    """same generator state -> same crop, noise files, levels and SNR -> the reference's (noisy, clean), bit for bit
    (host fp32 arithmetic in the reference's order), incl. the pad path (clip 2), the exact-length clip and items whose
    0.99 clip guard triggers"""
    from nppc_audio.data import AudioDataSetConfig, AudioDataset
    z = np.load(GOLD)
    clean, noise = _clips(z)
    ds = AudioDataset(AudioDataSetConfig(clean_path=".", noisy_path=".", target_dB_FS_floating_value=floating, **CFG),
                      clean_clips=clean, noise_clips=noise)
    assert len(ds) == 5
    guarded = 0
    for idx in range(5):
        for rep in range(2):
            seed = 1000 * idx + rep + (500 if floating else 0)
            ds.rng.seed(seed)
            noisy, cl = ds[idx]
            want_n, want_c = z[f"fl{floating:g}.i{idx}.s{seed}.noisy"], z[f"fl{floating:g}.i{idx}.s{seed}.clean"]
            assert noisy.shape == (4000,) and cl.shape == (4000,)
            assert np.abs(noisy.numpy() - want_n).max() < 1e-7 and np.abs(cl.numpy() - want_c).max() < 1e-7
            guarded += abs(float(np.abs(want_n).max()) - 0.99) < 1e-6
    assert guarded >= 2


def test_wav_folder_constructor(tmp_path):
    from scipy.io import wavfile
    from nppc_audio.data import AudioDataSetConfig, AudioDataset
    z = np.load(GOLD)
    clean, noise = _clips(z)
    cdir, ndir = tmp_path / "clean", tmp_path / "noise" / "sub"
    cdir.mkdir(), ndir.mkdir(parents=True)
    for i, c in enumerate(clean):
        wavfile.write(str(cdir / f"c{i}.wav"), 16000, c)                                  # float32 wav
    wavfile.write(str(ndir / "n0.wav"), 16000, (noise[0] * 32767).astype(np.int16))       # 16-bit PCM
    wavfile.write(str(ndir / "n1.wav"), 8000, np.stack([noise[1], noise[1]], axis=1))      # stereo, other rate
    cfg = AudioDataSetConfig(clean_path=str(cdir), noisy_path=str(tmp_path / "noise"), **CFG)
    ds = AudioDataset(cfg, seed=3)
    assert len(ds) == 5 and len(ds.noise) == 2
    assert torch.equal(ds.clean[0], torch.from_numpy(clean[0]))
    assert float((ds.noise[0] - torch.from_numpy(noise[0])).abs().max()) < 2.0 / 32768     # PCM quantisation (truncation) + the 32767/32768 scale
    assert ds.noise[1].numel() == 2 * noise[1].size                                        # 8 kHz -> 16 kHz, mono
    noisy, cl = ds[1]
    assert noisy.shape == (4000,) and bool(torch.isfinite(noisy).all()) and float(noisy.abs().max()) <= 0.99 + 1e-6
    # reproducible item stream under a seed
    a = AudioDataset(cfg, seed=5)[0][0]
    b = AudioDataset(cfg, seed=5)[0][0]
    assert torch.equal(a, b)
    empty = tmp_path / "empty"
    empty.mkdir()
    with pytest.raises(ValueError, match="No WAV files found in clean directory"):
        AudioDataset(AudioDataSetConfig(clean_path=str(empty), noisy_path=str(tmp_path / "noise"), **CFG))
    with pytest.raises(ValueError, match="No WAV files found in noise directory"):
        AudioDataset(AudioDataSetConfig(clean_path=str(cdir), noisy_path=str(empty), **CFG))
