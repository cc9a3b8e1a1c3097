#!/usr/bin/env python3
"""Golden vectors for the inpainting batch synthesis (second half) by running the
reference's own AudioInpaintingDataset methods (_normalize_audio, _create_random_mask, time_to_spec_mask:
dataset/audio_dataset_inpainting.py:154-181, 223-251) and utils.audio_to_stft (utils.py:150-175) on synthetic clips,
composed as __getitem__ composes them (:291-313).  Runs only in the build container; writes
tests/golden/inp_data.npz (data only)."""
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

CASES = {
    # name: (nfft = win, hop, L, [(gap start, gap length) per clip])
    "c3": (255, 128, 8000, [(3200, 1600), (0, 300), (7700, 300), (127, 1), (4000, 129), (6400, 1600)]),
    "tiny": (63, 32, 1000, [(500, 100), (31, 1), (32, 1), (990, 10)]),
}


def main():
    install_placeholders()
    sys.path.insert(0, "/root/reference")
    from dataset.audio_dataset_inpainting import AudioInpaintingDataset as DS
    from utils import StftConfig, audio_to_stft
    out = {}
    for name, (nfft, hop, L, gaps) in CASES.items():
        stft = StftConfig(nfft=nfft, hop_length=hop, win_length=nfft)
        _, clean = W.synth_batch(len(gaps), L)
        out[f"{name}_clean_in"] = clean
        res = dict(norm=[], mask_frames=[], stft_clean=[], stft_masked=[], masked_audio=[])
        for i, (g0, glen) in enumerate(gaps):
            fake = types.SimpleNamespace(config=types.SimpleNamespace(
                target_dB_FS=-25.0, target_dB_FS_floating_value=0.0, missing_start_seconds=g0 / 16000.0,
                sample_rate=16000, missing_length=glen, stft_configuration=stft))
            audio = DS._normalize_audio(fake, torch.from_numpy(clean[i:i + 1]))
            mask, s, e = DS._create_random_mask(fake, L)
            assert (s, e) == (g0, g0 + glen), (s, e, g0, glen)
            spec = audio_to_stft(audio, stft, torch.device("cpu"))
            mf = DS.time_to_spec_mask(fake, mask, spec.shape[3], L)
            res["norm"].append(audio[0].numpy())
            res["mask_frames"].append(mf.numpy())
            res["stft_clean"].append(spec[0].numpy())
            res["stft_masked"].append((spec * mf[None, None, None, :])[0].numpy())
            res["masked_audio"].append((audio * mask)[0].numpy())
        for k, v in res.items():
            out[f"{name}_{k}"] = np.stack(v).astype(np.float32)
        out[f"{name}_gaps"] = np.array(gaps, np.int32)
        out[f"{name}_cfg"] = np.array([nfft, hop, L], np.int32)
        print(name, "frames", out[f"{name}_mask_frames"].shape, "zero frames per clip", (out[f"{name}_mask_frames"] == 0).sum(1))
    np.savez_compressed(os.path.join(HERE, "inp_data.npz"), **out)


if __name__ == "__main__":
    main()
