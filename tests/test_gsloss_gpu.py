"""GPU parity: Gram-Schmidt + NPPC loss kernels (forward and backward) vs the CPU oracle's autograd."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import nppc_ref as R

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rel(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def test_gram_schmidt_golden_incl_collinear():
    from nppc_audio.pc_ops import gram_schmidt_to_crm
    z = np.load(os.path.join(GOLD, "g0_tiny.npz"))
    got = gram_schmidt_to_crm(torch.from_numpy(z["gs.in"]).cuda()).cpu().numpy()
    assert rel(got, z["gs.out"]) < 2e-5


@pytest.mark.parametrize("B,K,F,T", [(3, 2, 9, 11), (4, 5, 128, 33), (2, 8, 33, 40), (2, 1, 7, 5)])
def test_gram_schmidt_fwd_bwd_vs_oracle(B, K, F, T):
    from nppc_audio.pc_ops import gram_schmidt_to_crm
    g = torch.Generator().manual_seed(B * 100 + K)
    x = torch.randn(B, K, 2, F, T, generator=g)
    x[:, -1] += 0.5 * x[:, 0]
    gy = torch.randn(B, K, 2, F, T, generator=g)
    xr = x.clone().requires_grad_(True)
    wr = R.gram_schmidt_crm(xr)
    (wr * gy).sum().backward()
    xd = x.cuda().requires_grad_(True)
    wd = gram_schmidt_to_crm(xd)
    (wd * gy.cuda()).sum().backward()
    assert rel(wd.detach().cpu().numpy(), wr.detach().numpy()) < 2e-5
    assert rel(xd.grad.cpu().numpy(), xr.grad.numpy()) < 5e-5


@pytest.mark.parametrize("B,K,F,T,step", [(4, 3, 16, 9, 0), (4, 5, 128, 33, 500), (2, 2, 257, 63, 375)])
def test_loss_fwd_bwd_vs_oracle(B, K, F, T, step):
    from nppc_audio.pc_ops import NPPCLoss, second_moment_weight
    g = torch.Generator().manual_seed(step + K)
    w = torch.randn(B, K, 2, F, T, generator=g) * 0.3
    gt = torch.randn(B, 2, F, T, generator=g)
    pred = torch.randn(B, 2, F, T, generator=g)
    wr = w.clone().requires_grad_(True)
    rec_r, obj_r, log = R.nppc_loss(wr, gt, pred, step)
    extra = torch.linspace(0.5, 1.5, B)
    (obj_r + (rec_r * extra).sum()).backward()
    wd = w.cuda().requires_grad_(True)
    lam = second_moment_weight(step, 500, 1.0)
    assert lam == R.second_moment_weight(step)
    rec, obj, en, pr, pi, pm, wn, sm = NPPCLoss.apply(wd, gt.cuda(), pred.cuda(), lam)
    (obj + (rec * extra.cuda()).sum()).backward()
    assert abs(float(obj) - float(obj_r)) < 2e-6 * max(1.0, abs(float(obj_r)))
    assert rel(rec.detach().cpu().numpy(), rec_r.detach().numpy()) < 1e-5
    assert rel(en.cpu().numpy(), log["err_norm"].numpy()) < 1e-5
    assert rel(pr.cpu().numpy(), log["err_proj"].real.numpy()) < 2e-5
    assert rel(pi.cpu().numpy(), log["err_proj"].imag.numpy()) < 2e-5
    assert rel(pm.cpu().numpy(), log["err_proj_mag"].numpy()) < 1e-5
    assert rel(wn.cpu().numpy(), log["w_norms"].numpy()) < 1e-5
    assert rel(sm.cpu().numpy(), log["second_moment_mse"].numpy()) < 2e-5
    assert rel(wd.grad.cpu().numpy(), wr.grad.numpy()) < 5e-5


def test_loss_terms_golden_g2():
    from nppc_audio.pc_ops import NPPCLoss
    z = np.load(os.path.join(GOLD, "g2_k5.npz"))
    meta = json.load(open(os.path.join(GOLD, "g2_k5.json")))
    w, gt, pred = (torch.from_numpy(z[k]).cuda() for k in ("log.w_mat", "gt_crm", "pred_crm"))
    rec, obj, en, pr, pi, pm, wn, sm = NPPCLoss.apply(w, gt, pred, 1e-6)
    assert abs(float(obj) - meta["objective_at_step"]["0"]) < 2e-6
    assert rel(rec.cpu().numpy(), z["log.reconst_err"]) < 1e-5
    assert rel(sm.cpu().numpy(), z["log.second_moment_mse"]) < 1e-4
    assert rel(pr.cpu().numpy(), z["log.err_proj_re"]) < 1e-4
