#!/usr/bin/env python3
"""Golden vectors for the NPPC-vs-MC-dropout comparison metrics by running the reference's
own compute_metrics (nppc_audio/inpainting/validator/validator_nppc_model.py:742-828) on synthetic directions.
Runs only in the build container; writes tests/golden/metrics.npz (data only)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
# the validator module imports these lazily-resolved names; resolve them before the placeholder modules exist
from transformers import Wav2Vec2CTCTokenizer, Wav2Vec2FeatureExtractor, Wav2Vec2ForCTC, Wav2Vec2Processor  # noqa: E402,F401
from make_goldens import install_placeholders  # noqa: E402


def main():
    install_placeholders()
    sys.path.insert(0, "/root/reference")
    from nppc_audio.inpainting.validator.validator_nppc_model import compute_metrics
    out = {}
    for case, (n, Fq, T, seed) in {"a": (5, 16, 25, 1), "b": (3, 32, 40, 2)}.items():
        g = torch.Generator().manual_seed(seed)
        mask = torch.ones(1, 1, Fq, T)
        mask[..., 7:13] = 0
        hole = 1 - mask
        nppc = torch.randn(1, n, Fq, T, generator=g) * hole
        mc = (0.6 * nppc + 0.8 * torch.randn(1, n, Fq, T, generator=g)) * hole
        clean = torch.randn(1, 1, Fq, T, generator=g)
        pred = clean + 0.3 * torch.randn(1, 1, Fq, T, generator=g) * hole + 0.05 * nppc[:, :1]
        mean = clean + 0.4 * torch.randn(1, 1, Fq, T, generator=g) * hole
        m = compute_metrics(nppc, mc, pred, mean, clean, mask)
        for k, v in dict(nppc=nppc, mc=mc, clean=clean, pred=pred, mean=mean, mask=mask).items():
            out[f"{case}_{k}"] = v.numpy()
        out[f"{case}_scalars"] = np.array([m["nppc"]["rmse"], m["nppc"]["residual_error"], m["mc_dropout"]["rmse"],
                                           m["mc_dropout"]["residual_error"]], np.float64)
        out[f"{case}_angles"] = np.array(m["principal_angles"], np.float64)
        print(case, m)
    np.savez_compressed(os.path.join(HERE, "metrics.npz"), **out)


if __name__ == "__main__":
    main()
