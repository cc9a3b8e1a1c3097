#!/usr/bin/env python3
"""Golden fixture for SURVEY row f1 (validator PC synthesis + iSTFT), made by RUNNING THE REFERENCE's own functions:
utils.model_outputs_to_waveforms (utils.py:37-72), utils.crm_to_spectogram (utils.py:252-256) on decompress_cIRM'd
directions exactly as NPPCAudioValidator._crm_directions_to_spectograms does (validator.py:81-101), and the alpha
variations + torch.istft of visualize_pc_spectrograms (validator.py:245-283).  Inputs come from the committed g1_c1
fixture (BASELINE C1: 2 x 1 s, nfft 512 -- the reference function hard-codes 512 / 256), so no model is run here.

Build container only (imports /root/reference); writes tests/golden/validator_f1.npz."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_goldens as MG  # noqa: E402

MG.install_placeholders()
sys.path.insert(0, MG.REF)
import utils as ref_utils  # noqa: E402
from FullSubNet_plus.speech_enhance.audio_zen.acoustics.mask import decompress_cIRM  # noqa: E402

z = np.load(os.path.join(HERE, "g1_c1.npz"))
noisy = torch.from_numpy(z["noisy"])
L = noisy.shape[1]
n_re, n_im = torch.from_numpy(z["noisy_real"]), torch.from_numpy(z["noisy_imag"])          # [B,1,F,T]
pred = torch.from_numpy(z["pred_crm_full"])                                                # compressed cIRM [B,2,F,T]
w_mat = torch.from_numpy(z["log.w_mat"])                                                   # [B,K,2,F,T]
g = torch.Generator().manual_seed(11)
w_big = torch.randn(w_mat.shape[0], 1, 2, w_mat.shape[3], w_mat.shape[4], generator=g) * 4.0   # exercises the +-9.9 clip

out = {"w_big": w_big.numpy()}
out["enhanced_wave"] = ref_utils.model_outputs_to_waveforms(pred, n_re, n_im, L).numpy()       # utils.py:37-72
window = torch.hann_window(512)
noisy_complex = torch.stft(noisy, 512, hop_length=256, win_length=512, window=window, return_complex=True)   # validator.py:72-79
pred_crm = decompress_cIRM(pred).permute(0, 2, 3, 1)                                           # validator.py:189-190
enhanced_complex = ref_utils.crm_to_spectogram(pred_crm, noisy_complex)                       # validator.py:194
out["enhanced_re"], out["enhanced_im"] = enhanced_complex.real.numpy(), enhanced_complex.imag.numpy()
alphas = torch.linspace(-3, 3, 6)                                                             # validator.py:244
KEEP = [0, 2, 5]                                                                              # fixture size: 3 of the 6 variations
out["alphas"] = alphas[KEEP].numpy()
for tag, W in (("w", w_mat), ("wbig", w_big)):
    specs, waves = [], []
    for k in range(W.shape[1]):
        crm = decompress_cIRM(W[:, k]).permute(0, 2, 3, 1)                                    # validator.py:90-93
        pc = ref_utils.crm_to_spectogram(crm, noisy_complex)                                  # validator.py:96
        specs.append(pc)
        waves.append(torch.stack([torch.istft(enhanced_complex + a * pc, 512, hop_length=256, win_length=512, window=window,
                                              length=L) for a in alphas[KEEP]], dim=1))      # validator.py:264-283
    out[f"{tag}_spec_re"] = torch.stack([s.real for s in specs], dim=1).numpy()
    out[f"{tag}_spec_im"] = torch.stack([s.imag for s in specs], dim=1).numpy()
    out[f"{tag}_waves"] = torch.stack(waves, dim=1).numpy()                                   # [B,K,3,L]
np.savez_compressed(os.path.join(HERE, "validator_f1.npz"), **{k: v.astype(np.float32) for k, v in out.items()})
print({k: v.shape for k, v in out.items()})
