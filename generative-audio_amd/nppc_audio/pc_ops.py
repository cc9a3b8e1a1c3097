"""autograd.Function wrappers of the Gram-Schmidt and NPPC-loss kernels (csrc/gsloss.hip)."""
import torch

from . import _hip as H


def _z(B, KV, dev):
    return torch.zeros(B, KV, KV, 2, dtype=torch.float64, device=dev)


_EYES = {}


def _eye(B, K, dev):
    """constant complex identity [B,K,K,2] (fp64), built once per shape: index_put in every backward cost three launches"""
    key = (B, K, str(dev))
    e = _EYES.get(key)
    if e is None:
        e = _z(B, K, dev)
        idx = torch.arange(K, device=dev)
        e[:, idx, idx, 0] = 1.0
        _EYES[key] = e
    return e


class GramSchmidtCRM(torch.autograd.Function):
    """pc_wrapper.py:8-44 on [B,K,2,F,T] (w_hat detached, conj-coefficient quirk, no epsilon)."""

    @staticmethod
    def forward(ctx, x):
        H.require_gpu()
        x = x.contiguous().float()
        B, K, _, F, T = x.shape
        N = F * T
        s = H.stream()
        G = _z(B, K, x.device)
        H.call("nppc_gram", x, None, None, None, G, B, K, N, s)
        C, Ch = torch.empty_like(G), torch.empty_like(G)
        H.call("nppc_gs_solve", G, C, Ch, B, K, K, s)
        w = torch.empty_like(x)
        H.call("nppc_combine", x, C, None, None, None, None, w, B, K, N, s)
        ctx.save_for_backward(x, G, Ch)
        return w

    @staticmethod
    def backward(ctx, g):
        x, G, Ch = ctx.saved_tensors
        B, K, _, F, T = x.shape
        N = F * T
        s = H.stream()
        g = g.contiguous().float()
        P = _z(B, K, x.device)
        H.call("nppc_gram", g, x, None, None, P, B, K, N, s)
        D = torch.empty_like(G)
        H.call("nppc_gs_bwd_solve", G, P, Ch, D, B, K, K, s)
        eye = _eye(B, K, x.device)
        dx = torch.empty_like(x)
        H.call("nppc_combine", g, eye, x, D, None, None, dx, B, K, N, s)
        return dx


def gram_schmidt_to_crm(x: torch.Tensor) -> torch.Tensor:
    return GramSchmidtCRM.apply(x)


class NPPCLoss(torch.autograd.Function):
    """trainer.py:259-298.  (w_mat [B,K,2,F',T], gt/pred [B,2,F',T], lambda) ->
    (reconst_err [B], objective [], err_norm, err_proj_re, err_proj_im, err_proj_mag, w_norms, second_moment_mse)."""

    @staticmethod
    def forward(ctx, w_mat, gt, pred, lam, eps=1e-8, eps_in_norms=0):
        H.require_gpu()
        w = w_mat.contiguous().float()
        gt, pred = gt.contiguous().float(), pred.contiguous().float()
        B, K = w.shape[:2]
        N = w[0, 0, 0].numel()
        dev = w.device
        s = H.stream()
        G = _z(B, K + 1, dev)
        H.call("nppc_gram", w, None, gt, pred, G, B, K, N, s)
        f = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)
        err_norm, reconst = f(B), f(B)
        pr, pi, pm, wn, sm = f(B, K), f(B, K), f(B, K), f(B, K), f(B, K)
        coefA = torch.empty(B, K, 4, dtype=torch.float64, device=dev)
        coefE = torch.empty(B, K, 2, dtype=torch.float64, device=dev)
        if B <= 1024:        # the objective from the same launch (two means, a product and a sum less on the step's critical chain)
            objective = torch.empty((), dtype=torch.float32, device=dev)
            H.call("nppc_loss_solve_obj", G, err_norm, pr, pi, pm, wn, reconst, sm, coefA, coefE, B, K, float(eps),
                   int(eps_in_norms), float(lam), objective, s)
        else:
            H.call("nppc_loss_solve_eps", G, err_norm, pr, pi, pm, wn, reconst, sm, coefA, coefE, B, K, float(eps),
                   int(eps_in_norms), s)
            objective = reconst.mean() + lam * sm.mean()
        ctx.save_for_backward(w, gt, pred, coefA, coefE)
        ctx.lam = float(lam)
        ctx.mark_non_differentiable(err_norm, pr, pi, pm, wn, sm)
        ctx.set_materialize_grads(False)       # unused outputs arrive as None in backward, not as seven freshly filled zero tensors
        return reconst, objective, err_norm, pr, pi, pm, wn, sm

    @staticmethod
    def backward(ctx, g_rec, g_obj, *_):
        w, gt, pred, coefA, coefE = ctx.saved_tensors
        B, K = w.shape[:2]
        N = w[0, 0, 0].numel()
        s = H.stream()
        if g_rec is None and g_obj is None:
            return None, None, None, None, None, None
        grec = g_rec.contiguous().float() if g_rec is not None else None
        M1 = torch.empty(B, K + 1, K + 1, 2, dtype=torch.float64, device=w.device)
        if g_obj is None:
            H.call("nppc_loss_bwd_coef", coefA, coefE, grec, 0.0, 0.0, M1, B, K, s)
        else:       # the upstream scalar stays on the device: float(g_obj) here stalled the host ~1 ms every step
            H.call("nppc_loss_bwd_coef_dev", coefA, coefE, grec, g_obj.contiguous().float(), 1.0 / B, ctx.lam / (B * K), M1,
                   B, K, s)
        dw = torch.empty_like(w)
        H.call("nppc_combine", w, M1, None, None, gt, pred, dw, B, K, N, s)
        return dw, None, None, None, None, None


def second_moment_weight(step, grace, lam):
    """trainer.py:337-340 (python floats, as in the reference)."""
    v = -1 + 2 * step / grace
    v = max(min(v, 1), 1e-6)
    return v * lam


def planes(x):
    """real [B, K, ...] -> complex-layout [B, K, 2, ...] with a zero imaginary plane (memory plumbing only):
    the Gram-Schmidt / loss kernels are written for complex vectors; a real vector is the special case im = 0."""
    z = torch.zeros(x.shape[0], x.shape[1], 2, *x.shape[2:], dtype=torch.float32, device=x.device)
    z[:, :, 0].copy_(x)
    return z


def gram_schmidt_to_spec_mag(x: torch.Tensor) -> torch.Tensor:
    """inpainting/nppc/pc_wrapper.py:43-59 on [B, K, F, T]: sequential projections onto the detached unit vectors,
    no epsilon.  With zero imaginary parts the complex kernel's conjugate coefficient is the real dot product."""
    return GramSchmidtCRM.apply(planes(x))[:, :, 0]
