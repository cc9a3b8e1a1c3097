"""GPU: the on-device minibatch mixer behind the wav-folder dataset (DeviceMixLoader -> nppc_mix_snr) against items of the
REFERENCE's AudioDataset.__getitem__, and NPPCAudioTrainer(config) constructing from a wav folder like
nppc_audio/scripts/run_denoising_nppc.py:24-26."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dataset_items.npz")
CFG = dict(sample_rate=16000, snr_range=(-25, 20), silence_length=0.01, sub_sample_length_seconds=0.25, target_dB_FS=-25.0)


class _Replay:
    """batch sampler that re-seeds the dataset's generator before every item, as the fixture generator did"""

    def __init__(self, ds, items):
        self.ds, self.items = ds, items

    def __len__(self):
        return 1

    def __iter__(self):
        yield [i for i, _ in self.items]


@pytest.mark.parametrize("floating", [0.0, 3.0])
def test_device_mixed_minibatch_equals_reference_items(floating):
    from nppc_audio.data import AudioDataSetConfig, AudioDataset, DeviceMixLoader
    z = np.load(GOLD)
    clean, noise = [z[f"clean{i}"] for i in range(5)], [z[f"noise{i}"] for i in range(3)]
    ds = AudioDataset(AudioDataSetConfig(clean_path=".", noisy_path=".", target_dB_FS_floating_value=floating, **CFG),
                      clean_clips=clean, noise_clips=noise)
    items = [(idx, 1000 * idx + rep + (500 if floating else 0)) for idx in range(5) for rep in range(2)]
    seeds = iter(items)
    draw = ds.draw

    def seeded_draw(idx):                      # the fixture seeded `random` per item; replay that
        i, seed = next(seeds)
        assert i == idx
        ds.rng.seed(seed)
        return draw(idx)

    ds.draw = seeded_draw
    (noisy, cl), = list(DeviceMixLoader(ds, _Replay(ds, items), device="cuda", pin_memory=False))
    torch.cuda.synchronize()
    assert noisy.is_cuda and tuple(noisy.shape) == (10, 4000)
    for row, (idx, seed) in enumerate(items):
        want_n, want_c = z[f"fl{floating:g}.i{idx}.s{seed}.noisy"], z[f"fl{floating:g}.i{idx}.s{seed}.clean"]
        # the kernel sums the powers in fp64 (the reference: fp32 pairwise): 1e-6 of the 0.99 full scale
        assert np.abs(noisy[row].cpu().numpy() - want_n).max() < 2e-6, (idx, seed)
        assert np.abs(cl[row].cpu().numpy() - want_c).max() < 2e-6, (idx, seed)


def test_trainer_constructs_from_a_wav_folder_and_trains(tmp_path):
    """NPPCAudioTrainer(config) with NO dataset argument (the reference's constructor call): wav folders -> AudioDataset
    -> DeviceMixLoader -> two optimisation steps on the HIP path"""
    from scipy.io import wavfile
    from nppc_audio.data import AudioDataset, DeviceMixLoader
    from nppc_audio.fullsubnet import FullSubNet_Plus, FullSubNetPlusConfig
    from nppc_audio.trainer import NPPCAudioTrainer, NPPCAudioTrainerConfig
    rng = np.random.Generator(np.random.PCG64(5))
    cdir, ndir = tmp_path / "clean", tmp_path / "noise"
    cdir.mkdir(), ndir.mkdir()
    for i in range(6):
        wavfile.write(str(cdir / f"c{i}.wav"), 16000, (0.05 * rng.standard_normal(9000 + 700 * i)).astype(np.float32))
    for i in range(2):
        wavfile.write(str(ndir / f"n{i}.wav"), 16000, (0.02 * rng.standard_normal(5000)).astype(np.float32))
    torch.manual_seed(0)
    ck = str(tmp_path / "restorer.tar")
    torch.save({"model": FullSubNet_Plus(FullSubNetPlusConfig()).state_dict()}, ck)
    cfg = NPPCAudioTrainerConfig(
        nppc_model_configuration=dict(
            pretrained_restoration_model_configuration=dict(num_groups_in_drop_band=1), pretrained_restoration_model_path=ck,
            audio_pc_wrapper_configuration=dict(multi_direction_configuration=dict(num_groups_in_drop_band=2, n_directions=2)),
            stft_configuration=dict(nfft=512, hop_length=256, win_length=512), device="cuda"),
        data_configuration=dict(data_path=str(tmp_path), dataset=dict(clean_path=str(cdir), noisy_path=str(ndir),
                                                                      sub_sample_length_seconds=0.5, snr_range=(0, 20))),
        data_loader_configuration=dict(batch_size=3, num_workers=0, pin_memory=False, shuffle=True),
        optimizer_configuration=dict(type="Adam", args=dict(lr=1e-4, betas=[0.9, 0.999], eps=1e-8, weight_decay=0)),
        device="cuda", log_interval=1)
    tr = NPPCAudioTrainer(cfg)
    assert isinstance(tr.dataloader, DeviceMixLoader) and isinstance(tr.dataloader.dataset, AudioDataset)
    assert len(tr.dataloader) == 2
    tr.train(n_steps=2, checkpoint_dir=str(tmp_path / "ck"))
    assert tr.step == 2
    assert any(f.startswith("checkpoint_final_") for f in os.listdir(tmp_path / "ck"))
    assert all(bool(torch.isfinite(p).all()) for p in tr.nppc_model.audio_pc_wrapper.parameters())
