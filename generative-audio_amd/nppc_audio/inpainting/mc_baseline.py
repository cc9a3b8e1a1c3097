"""MC-dropout + PCA baseline of the inpainting path on the device.

Mirrors the reference's root utils.py: enable_dropout (:334-338), compute_pca_sklearn_batch (:393-496) and
calculate_unet_baseline (:548-648).  The stochastic U-Net passes run on the HIP engine (Philox dropout after
down3/down4/up1/up2), the per-item PCA of the K samples is one batched launch sequence (Gram + Jacobi + components)
instead of B scikit-learn fits on the host; gather / scatter of the gap elements is torch indexing (plumbing).
"""
import ctypes

import torch
import torch.nn as nn

from .. import _hip as H


def enable_dropout(model):
    """Enable Dropout layers during inference for MC-Dropout (utils.py:334-338)"""
    for module in model.modules():
        if isinstance(module, nn.Dropout):
            module.train()


def compute_pca_batch(outputs, n_components=5):
    """outputs [K, B, D] (K <= 60 samples per item) ->
    (principal_components [B,n,D], scaled_principal_components [B,n,D], importance_weights [B,n],
     mean_prediction [B,D], singular_vals [B,n])   -- compute_pca_sklearn_batch's return tuple (utils.py:393-496).
    Exact (Jacobi) singular pairs where scikit-learn's auto solver is randomized for these shapes; same sign rule
    (largest-magnitude entry of every component positive)."""
    H.require_gpu()
    outputs = outputs.contiguous().float()
    K, B, D = outputs.shape
    n = min(int(n_components), K)
    dev = outputs.device
    elems = ctypes.c_long()
    H.call("nppc_pca_work_elems", K, B, n, ctypes.byref(elems))
    work = torch.empty(elems.value, dtype=torch.float64, device=dev)
    mean = torch.empty(B, D, dtype=torch.float32, device=dev)
    comps = torch.empty(B, n, D, dtype=torch.float32, device=dev)
    scaled = torch.empty_like(comps)
    svals = torch.empty(B, n, dtype=torch.float32, device=dev)
    weights = torch.empty_like(svals)
    H.call("nppc_pca_batch", outputs, K, B, D, n, mean, comps, scaled, svals, weights, work, H.stream())
    return comps, scaled, weights, mean, svals


compute_pca_sklearn_batch = compute_pca_batch        # the reference's name for the same contract


def _scatter_gap(values, hole, F, T):
    """reconstruct_full_spec_batch (utils.py:600-625): zeros with the gap elements (mask == 0) filled row-major"""
    B = values.shape[0]
    if values.dim() == 3:
        full = torch.zeros(B, values.shape[1], F * T, dtype=values.dtype, device=values.device)
        full.masked_scatter_(hole[:, None, :].expand_as(full), values)
        return full.reshape(B, values.shape[1], F, T)
    full = torch.zeros(B, F * T, dtype=values.dtype, device=values.device)
    full.masked_scatter_(hole, values)
    return full.reshape(B, F, T)


def mc_dropout_samples(model, masked_spec, mask, n_mc_samples=50):
    """the K stochastic passes of calculate_unet_baseline (utils.py:561-582): [K, B, N_masked] gap values"""
    enable_dropout(model)
    B, _, F, T = masked_spec.shape
    hole = (mask.reshape(B, F * T) == 0)
    counts = hole.sum(dim=1)
    if not bool((counts == counts[0]).all()):
        raise ValueError("every item of the batch must have the same number of masked elements (utils.py:571-582)")
    n_masked = int(counts[0])
    preds = torch.empty(n_mc_samples, B, n_masked, dtype=torch.float32, device=masked_spec.device)
    with torch.no_grad():
        for k in range(n_mc_samples):
            pred = model(masked_spec, mask)                                   # [B, 1, F, T]
            preds[k] = pred.reshape(B, F * T)[hole].reshape(B, n_masked)
    return preds, hole


def calculate_unet_baseline(model, masked_spec, mask, n_mc_samples=50, n_components=5):
    """utils.calculate_unet_baseline (utils.py:548-648): model = RestorationWrapper (dropout U-Net), masked_spec
    [B,1,F,T], mask [B,1,F,T] (1 = known) -> dict(mean_prediction [B,1,F,T], principal_components [B,n,F,T],
    scaled_principal_components [B,n,F,T], importance_weights [B,n], singular_vals [B,n])."""
    B, _, F, T = masked_spec.shape
    preds, hole = mc_dropout_samples(model, masked_spec, mask, n_mc_samples)
    pcs, scaled, weights, mean, svals = compute_pca_batch(preds, n_components)
    return {
        "mean_prediction": _scatter_gap(mean, hole, F, T).unsqueeze(1),
        "principal_components": _scatter_gap(pcs, hole, F, T),
        "scaled_principal_components": _scatter_gap(scaled, hole, F, T),
        "importance_weights": weights,
        "singular_vals": svals,
    }


class PairProjectionLoss(torch.autograd.Function):
    """base_step2's loss (inpainting/trainer/nppc_trainer.py:285-323): (w_mat [B,K,F,T], W_mc [B,K,F,T], singular_vals
    [B,K], lambda) -> (reconst_err [B], objective [], proj [B,K], w_norms [B,K], second_moment_mse [B]);
    differentiable in w_mat only (W_mc comes from no_grad passes)."""

    @staticmethod
    def forward(ctx, w_mat, w_mc, singular_vals, lam):
        H.require_gpu()
        w, m = w_mat.contiguous().float(), w_mc.contiguous().float()
        sv = singular_vals.contiguous().float()
        B, K = w.shape[:2]
        N = w[0, 0].numel()
        assert m.shape == w.shape and sv.shape == (B, K)
        dev = w.device
        sums = torch.empty(B * K * 3, dtype=torch.float64, device=dev)
        coef = torch.empty(B * K * 3, dtype=torch.float64, device=dev)
        proj = torch.empty(B, K, dtype=torch.float32, device=dev)
        w_norms = torch.empty_like(proj)
        reconst = torch.empty(B, dtype=torch.float32, device=dev)
        second = torch.empty_like(reconst)
        H.call("nppc_pair_loss", w, m, sv, sums, proj, w_norms, reconst, second, coef, B, K, N, 1e-6, H.stream())
        objective = reconst.mean() + lam * second.mean()
        ctx.save_for_backward(w, m, coef)
        ctx.lam = float(lam)
        ctx.mark_non_differentiable(proj, w_norms, second)
        return reconst, objective, proj, w_norms, second

    @staticmethod
    def backward(ctx, g_rec, g_obj, *_):
        w, m, coef = ctx.saved_tensors
        B, K = w.shape[:2]
        N = w[0, 0].numel()
        go = float(g_obj) if g_obj is not None else 0.0
        grec = g_rec.contiguous().float() if g_rec is not None else None
        dw = torch.empty_like(w)
        H.call("nppc_pair_loss_bwd", w, m, coef, grec, go / B, go * ctx.lam / B, dw, B, K, N, H.stream())
        return dw, None, None, None


def compute_metrics(nppc_directions, mc_dropout_directions, pred_spec_mag, mean_prediction, clean_spec_mag, mask):
    """validator_nppc_model.compute_metrics (inpainting/validator/validator_nppc_model.py:742-828) for one item:
    directions [1,n,F,T] x2, pred / mean / clean / mask [1,1,F,T] -> {'nppc': {rmse, residual_error},
    'mc_dropout': {rmse, residual_error}, 'principal_angles': [degrees]}.  All inner products come from ONE device
    Gram of the 2n + 3 rows; the n x n algebra (whitening instead of QR: same singular values) is host numpy fp64."""
    import numpy as np
    H.require_gpu()
    n = nppc_directions.shape[1]
    N = nppc_directions[0, 0].numel()
    assert nppc_directions.shape[0] == 1 and mc_dropout_directions.shape == nppc_directions.shape
    dev = nppc_directions.device
    rows = torch.empty(2 * n + 3, N, dtype=torch.float32, device=dev)
    rows[:n].copy_(nppc_directions.reshape(n, N))
    rows[n:2 * n].copy_(mc_dropout_directions.reshape(n, N))
    s = H.stream()
    f32 = lambda t: t.contiguous().float()
    H.call("nppc_metric_rows", f32(pred_spec_mag), f32(clean_spec_mag), f32(mean_prediction), f32(mask), rows[2 * n:], N, s)
    G = torch.empty(2 * n + 3, 2 * n + 3, dtype=torch.float64, device=dev)
    H.call("nppc_rows_gram", rows, 2 * n + 3, rows, 2 * n + 3, N, G, s)
    G = G.cpu().numpy()
    e, e_np, e_mc = 2 * n, 2 * n + 1, 2 * n + 2

    def residual(idx):
        norms = np.sqrt(np.diag(G)[idx]) + 1e-6
        c = G[idx, e] / norms                                     # W^T e
        gh = G[np.ix_(idx, idx)] / np.outer(norms, norms)
        return float(np.sqrt(max(G[e, e] - 2.0 * (c * c).sum() + c @ gh @ c, 0.0)))

    def whiten(idx):
        lam, U = np.linalg.eigh(G[np.ix_(idx, idx)])
        keep = lam > 1e-12 * lam.max()
        return U[:, keep] / np.sqrt(lam[keep])                    # columns w: (w^T A) rows are orthonormal

    a, b = np.arange(n), np.arange(n, 2 * n)
    sv = np.linalg.svd(whiten(a).T @ G[np.ix_(a, b)] @ whiten(b), compute_uv=False)
    return {
        'nppc': {'rmse': float(np.sqrt(G[e_np, e_np])), 'residual_error': residual(a)},
        'mc_dropout': {'rmse': float(np.sqrt(G[e_mc, e_mc])), 'residual_error': residual(b)},
        'principal_angles': (np.arccos(np.clip(sv, -1, 1)) * 180 / np.pi).tolist(),
    }
