"""Input preparation of the inpainting step (reference: root utils.py:273-306) as HIP kernels."""
import torch

from .. import _hip as H


def preprocess_data(clean_spec, masked_spec, mask, plot_mean_std=False):
    """(clean_spec [B,2,F,T], masked_spec [B,2,F,T], mask [B,T]) ->
    (clean log-magnitude normalised by its batch-global mean / unbiased std [B,1,F,T], mask [B,1,F,T],
     masked log-magnitude normalised with the same statistics [B,1,F,T])          utils.py:294-306"""
    H.require_gpu()
    clean_spec, masked_spec = clean_spec.contiguous().float(), masked_spec.contiguous().float()
    B, two, F, T = clean_spec.shape
    assert two == 2 and masked_spec.shape == clean_spec.shape and mask.shape == (B, T)
    s = H.stream()
    dev = clean_spec.device
    st = torch.zeros(2, dtype=torch.float64, device=dev)
    cn = torch.empty(B, 1, F, T, dtype=torch.float32, device=dev)
    mn = torch.empty_like(cn)
    ms = torch.empty(2, dtype=torch.float32, device=dev)
    H.call("nppc_logmag", clean_spec, cn, F * T, B, F * T, st, s)
    H.call("nppc_logmag", masked_spec, mn, F * T, B, F * T, None, s)
    H.call("nppc_standardize", cn, mn, F * T, B, F * T, st, ms, s)
    mask4 = mask.float()[:, None, None, :].expand(-1, 1, F, -1)
    if plot_mean_std:
        return cn, mask4, mn, ms[0], ms[1]
    return cn, mask4, mn
