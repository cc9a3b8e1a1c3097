#!/usr/bin/env python3
"""Golden vectors for the MC-dropout PCA baseline by running the reference's own
utils.compute_pca_sklearn_batch (utils.py:393-496; scikit-learn PCA, whose auto solver is the RANDOMIZED one for these
shapes -- numpy's global seed is fixed first) and utils.compute_pca_and_importance_weights (utils.py:498-545; exact
torch.linalg.svd, no sign rule) on synthetic sample stacks.  Runs only in the build container; writes
tests/golden/pca.npz (data only)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
from make_goldens import install_placeholders  # noqa: E402


def stack(K, B, D, seed):
    """K samples per item around an item mean, with a decaying spectrum (10 strong directions + white floor)"""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = np.empty((K, B, D), np.float32)
    for b in range(B):
        mean = rng.standard_normal(D)
        basis, _ = np.linalg.qr(rng.standard_normal((D, 10)))
        sig = 3.0 * 0.6 ** np.arange(10) * (1.0 + 0.3 * b)
        coef = rng.standard_normal((K, 10)) * sig
        out[:, b] = (mean + coef @ basis.T + 0.02 * rng.standard_normal((K, D))).astype(np.float32)
    return out


CASES = {"mc50": (50, 3, 1792, 11), "small": (7, 2, 65, 12), "odd": (33, 2, 500, 13)}


def main():
    install_placeholders()
    sys.path.insert(0, "/root/reference")
    import utils as RU
    out = {}
    for name, (K, B, D, seed) in CASES.items():
        x = stack(K, B, D, seed)
        np.random.seed(0)
        pcs, scaled, w, mean, sv = RU.compute_pca_sklearn_batch(torch.from_numpy(x), n_components=5)
        pcs2, w2, mean2 = RU.compute_pca_and_importance_weights(torch.from_numpy(x))
        out[f"{name}_x"] = x
        for k, v in dict(pcs=pcs, scaled=scaled, weights=w, mean=mean, svals=sv, svd_pcs=pcs2, svd_weights=w2,
                         svd_mean=mean2).items():
            out[f"{name}_{k}"] = v.numpy().astype(np.float32)
        print(name, "singular values item 0:", sv[0].numpy())
    np.savez_compressed(os.path.join(HERE, "pca.npz"), **out)


if __name__ == "__main__":
    main()
