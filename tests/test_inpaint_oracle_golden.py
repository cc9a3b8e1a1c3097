"""CPU: pin oracle/inpaint_ref.py to the reference-generated inpainting goldens."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import inpaint_ref as R
from oracle import weights as W

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    with open(os.path.join(GOLD, name + ".json")) as f:
        meta = json.load(f)
    return z, meta


def params_for(meta, dtype=torch.float32):
    c = meta["config"]
    wts = W.make_weights(W.inpainting_spec(c["K"]), c["seed"])
    P = {}
    for k, v in wts.items():
        t = torch.from_numpy(np.asarray(v).copy())
        P[k] = t.to(dtype) if t.is_floating_point() else t
    return P


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


@pytest.fixture(scope="module", params=["inp_tiny", "inp_c3s"])
def case(request):
    z, meta = load(request.param)
    return request.param, z, meta


def test_synthetic_inputs_are_reproducible(case):
    _, z, meta = case
    c = meta["config"]
    m, k, cl = W.synth_inpaint_batch(c["B"], c["T"], c["nfft"], c["hop"])
    np.testing.assert_array_equal(k, z["mask_frames"])
    np.testing.assert_allclose(cl, z["clean_spec"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(m, z["masked_spec"], rtol=0, atol=1e-6)


def test_state_dict_wire_format(case):
    _, _, meta = case
    spec = W.inpainting_spec(meta["config"]["K"])
    assert len(spec) == meta["n_tensors"]
    assert sum(int(np.prod(s)) for s in spec.values()) == meta["n_elems"]


def test_preprocess_and_restorer(case):
    _, z, meta = case
    P = params_for(meta)
    cn, mask, mn, mean, std = R.preprocess(torch.from_numpy(z["clean_spec"]), torch.from_numpy(z["masked_spec"]),
                                           torch.from_numpy(z["mask_frames"]))
    assert abs(float(mean) - meta["mean"]) < 1e-5 and abs(float(std) - meta["std"]) < 1e-5
    assert rel(cn, z["clean_norm"]) < 1e-6 and rel(mn, z["masked_norm"]) < 1e-6
    pred = R.restorer_forward(mn, mask, P)
    assert rel(pred, z["pred_norm"]) < 2e-5


def test_forward_loss_and_taps(case):
    _, z, meta = case
    P = params_for(meta)
    taps, stats = {}, {}
    _, obj, log = R.inpaint_step(torch.from_numpy(z["masked_spec"]), torch.from_numpy(z["mask_frames"]),
                                 torch.from_numpy(z["clean_spec"]), P, 0, new_stats=stats, taps=taps)
    raw0 = taps["pc_wrapper.net.inc.conv.conv.0.raw"].detach().numpy()
    g = z["tap.inc.raw0"]
    assert rel(raw0[:, : g.shape[1]] if g.shape != raw0.shape else raw0, g) < 1e-5
    assert rel(log["w_mat"], z["log.w_mat"]) < 2e-4
    for k in ("err_norm", "err_proj", "w_norms", "reconst_err", "second_moment_mse"):
        assert rel(log[k], z["log." + k]) < 5e-4, k
    assert abs(float(obj) - float(z["log.objective"])) < 1e-5
    for st, want in meta["objective_at_step"].items():
        got = float(log["reconst_err"].mean() + R.second_moment_weight(int(st)) * log["second_moment_mse"].mean())
        assert abs(got - want) < 1e-5 * max(1.0, abs(want)), st
    for n in ("pc_wrapper.net.inc.conv.conv.1.running_mean", "pc_wrapper.net.inc.conv.conv.1.running_var",
              "pc_wrapper.net.down4.mpconv.1.conv.4.running_mean", "pc_wrapper.net.down4.mpconv.1.conv.4.running_var",
              "pc_wrapper.net.up4.conv.conv.4.running_var"):
        assert rel(stats[n].numpy().reshape(-1)[:4096], z["bn1." + n]) < 2e-5, n
    assert int(stats["pc_wrapper.net.up2.conv.conv.1.num_batches_tracked"]) == int(z[
        "bn1.pc_wrapper.net.up2.conv.conv.1.num_batches_tracked"][0])


def test_gradients_clip_and_adam(case):
    _, z, meta = case
    P = params_for(meta)
    train = {k: v.requires_grad_(True) for k, v in P.items()
             if k.startswith("pc_wrapper.") and v.is_floating_point() and "running_" not in k}
    args = (torch.from_numpy(z["masked_spec"]), torch.from_numpy(z["mask_frames"]), torch.from_numpy(z["clean_spec"]))
    state = {}
    for t, step in ((1, 500), (2, 501)):
        stats = {}
        _, obj, _ = R.inpaint_step(*args, P, step, new_stats=stats)
        grads = dict(zip(train, torch.autograd.grad(obj, list(train.values()))))
        if t == 1:
            assert abs(float(obj) - meta["g500.objective"]) < 1e-5
            for n, (amax, l2) in meta["g500.grad_absmax_l2"].items():
                got = grads[n].numpy().reshape(-1)
                want = z["g500.grad." + n]
                if n.endswith((".conv.0.bias", ".conv.3.bias")):
                    # bias of a conv that feeds BatchNorm: the exact gradient is 0, both sides hold rounding noise
                    assert np.abs(got).max() < 1e-5 and np.abs(want).max() < 1e-5, n
                    continue
                d = np.abs(got[: want.size] - want)
                # a LeakyReLU / max-pool decision on an activation within 1e-7 of a tie flips between two correct
                # fp32 evaluations and moves single gradient elements by a discrete amount: allow a few
                assert (d > 2e-3 * amax + 1e-7).sum() <= max(1, 5e-3 * d.size) and d.max() <= 2e-2 * amax + 1e-7, n
                assert abs(np.sqrt((got.astype(np.float64) ** 2).sum()) - l2) <= 2e-3 * l2 + 1e-7, n
            coef, total = R.clip_coef(grads.values())
            assert abs(total - meta["clip.total_norm"]) < 1e-3 * meta["clip.total_norm"]
            assert abs(total - meta["g500.grad_total_l2"]) < 1e-3 * total
        else:
            assert abs(float(obj) - meta["adam1.objective_next"]) < 5e-4
            coef, _ = R.clip_coef(grads.values())
        with torch.no_grad():
            R.adam_step(train, {k: g * coef for k, g in grads.items()}, state, t)
            for k, v in stats.items():
                P[k] = v
        for n in [k[6:] for k in z.files if k.startswith(f"adam{t}.")]:
            got = train[n].detach().numpy().reshape(-1)
            want = z[f"adam{t}.{n}"]
            # Adam's first steps move every weight by ~lr; compare the UPDATE, not the weight
            w0 = W.make_weights({n: tuple(train[n].shape)}, meta["config"]["seed"])[n].reshape(-1)[: want.size]
            d = np.abs((got[: want.size] - w0) - (want - w0))
            # elements whose gradient is below fp32 noise (|g| ~ 1e-9 vs Adam's eps 1e-8) take an arbitrary step
            assert (d > 0.05 * 1e-4 * t + 1e-7).sum() <= max(1, 1e-2 * d.size) and np.median(d) < 2e-6, n


def test_gram_schmidt_real_golden():
    z, _ = load("inp_tiny")
    got = R.gram_schmidt_real(torch.from_numpy(z["gs.in"])).numpy()
    assert np.abs(got - z["gs.out"]).max() < 1e-5
    flat = got.reshape(got.shape[0], got.shape[1], -1).astype(np.float64)
    for b in range(flat.shape[0]):
        g = flat[b] @ flat[b].T
        off = g - np.diag(np.diag(g))
        assert np.abs(off).max() < 1e-3 * np.abs(np.diag(g)).max()


@pytest.mark.parametrize("name", ["c3", "tiny"])
def test_batch_synthesis_oracle_matches_reference_dataset(name):
    """row f3 (second half): the oracle's restatement of AudioInpaintingDataset's preparation against the outputs of the
    reference's own methods (tests/golden/inp_data.npz, written by make_goldens_inpaint_data.py)."""
    import os
    import numpy as np
    import torch
    from oracle import inpaint_ref as IR
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "inp_data.npz"))
    nfft, hop, L = (int(v) for v in z[f"{name}_cfg"])
    clean = torch.from_numpy(z[f"{name}_clean_in"])
    for i, (g0, glen) in enumerate(z[f"{name}_gaps"]):
        audio = IR.normalize_dbfs(clean[i:i + 1])
        assert np.array_equal(audio[0].numpy(), z[f"{name}_norm"][i])
        sm, mf, sc, ma = IR.inpaint_sample(audio, int(g0), int(g0 + glen), nfft, hop, nfft)
        assert np.array_equal(mf.numpy(), z[f"{name}_mask_frames"][i])
        assert np.array_equal(sc.numpy(), z[f"{name}_stft_clean"][i])
        assert np.array_equal(sm.numpy(), z[f"{name}_stft_masked"][i])
        assert np.array_equal(ma[0].numpy(), z[f"{name}_masked_audio"][i])
        assert 0 < int((mf == 0).sum()) < mf.numel()


@pytest.mark.parametrize("name", ["mc50", "small", "odd"])
def test_pca_oracle_matches_reference_pca(name):
    """row f4: the oracle's exact-SVD restatement against the reference's compute_pca_sklearn_batch (scikit-learn,
    randomized solver, seeded) and compute_pca_and_importance_weights (torch.linalg.svd; sign free) outputs."""
    import os
    import numpy as np
    import torch
    from oracle import inpaint_ref as IR
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pca.npz"))
    pcs, scaled, w, mean, sv = (t.numpy() for t in IR.pca_batch(torch.from_numpy(z[f"{name}_x"]), 5))
    assert np.abs(pcs - z[f"{name}_pcs"]).max() < 2e-5                       # unit vectors, sign rule included
    assert np.abs(sv - z[f"{name}_svals"]).max() < 2e-6 * z[f"{name}_svals"].max()
    assert np.abs(scaled - z[f"{name}_scaled"]).max() < 2e-5 * np.abs(z[f"{name}_scaled"]).max()
    assert np.abs(w - z[f"{name}_weights"]).max() < 1e-6 and np.abs(mean - z[f"{name}_mean"]).max() < 5e-6
    p2 = z[f"{name}_svd_pcs"]
    sgn = np.sign((pcs * p2).sum(-1, keepdims=True))
    assert np.abs(pcs - sgn * p2).max() < 2e-5 and np.abs(w - z[f"{name}_svd_weights"]).max() < 1e-6


@pytest.mark.parametrize("case", ["a", "b"])
def test_metrics_oracle_matches_reference_compute_metrics(case):
    import os
    import numpy as np
    import torch
    from oracle import inpaint_ref as IR
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "metrics.npz"))
    t = {k: torch.from_numpy(z[f"{case}_{k}"]) for k in ("nppc", "mc", "pred", "mean", "clean", "mask")}
    m = IR.compute_metrics(t["nppc"], t["mc"], t["pred"], t["mean"], t["clean"], t["mask"])
    got = np.array([m["nppc"]["rmse"], m["nppc"]["residual_error"], m["mc_dropout"]["rmse"], m["mc_dropout"]["residual_error"]])
    assert np.abs(got - z[f"{case}_scalars"]).max() < 1e-6 * z[f"{case}_scalars"].max()
    assert np.abs(np.array(m["principal_angles"]) - z[f"{case}_angles"]).max() < 1e-3
