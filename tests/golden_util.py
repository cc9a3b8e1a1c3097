"""Shared helpers for the reference-generated fixtures under tests/golden/ (see make_goldens.py)."""
import json
import os
import zlib

import numpy as np

from oracle import weights as W

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    meta = json.load(open(os.path.join(GOLD, name + ".json")))
    return z, meta


def rel(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def lat(a, meta):
    """'lite' fixtures (long clips) keep maps > 64 K elements on a strided (bin, frame) lattice (make_goldens.py: keep)"""
    a = np.asarray(a)
    if meta["config"].get("lite") and a.ndim >= 2 and a.size > 65536:
        return a[..., ::meta["lattice"][0], ::meta["lattice"][1]]
    return a


def waves(z, meta):
    """fixture waveforms; 'lite' fixtures regenerate them (seeded recipe) and check the CRC the generator recorded"""
    if "noisy" in z.files:
        return z["noisy"], z["clean"]
    c = meta["config"]
    noisy, clean = W.synth_batch(c["B"], c["L"])
    assert int(zlib.crc32(np.ascontiguousarray(noisy).tobytes())) == meta["noisy_crc"]
    assert int(zlib.crc32(np.ascontiguousarray(clean).tobytes())) == meta["clean_crc"]
    return noisy, clean
