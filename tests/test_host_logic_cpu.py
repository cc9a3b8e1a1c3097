"""CPU: host-side logic of the drop-in surface (configs, module tree / wire format, samplers, schedules)."""
import numpy as np
import pytest
import torch

from oracle import weights as W


def test_state_dict_wire_format_matches_reference_spec():
    """Names, order and shapes of state_dict() == the spec that make_goldens.py asserted against the reference."""
    from nppc_audio.fullsubnet import FullSubNet_Plus, FullSubNetPlusConfig
    from nppc_audio.networks import MultiDirectionConfig, MultiDirectionFullSubNet_Plus
    r = FullSubNet_Plus(FullSubNetPlusConfig())
    spec = W.restorer_spec()
    sd = r.state_dict()
    assert list(sd) == list(spec) and all(tuple(sd[k].shape) == tuple(spec[k]) for k in spec)
    d = MultiDirectionFullSubNet_Plus(MultiDirectionConfig(n_directions=5))
    spec = W.direction_spec(5)
    sd = d.state_dict()
    assert list(sd) == list(spec) and all(tuple(sd[k].shape) == tuple(spec[k]) for k in spec)
    assert sum(v.numel() for v in sd.values()) + sum(v.numel() for v in r.state_dict().values()) == 23873631
    assert d.output_size == 10 and d.n_directions == 5


def test_same_seed_same_init_as_torch_modules():
    """parameter containers are the torch modules the reference builds, created in the same order"""
    from nppc_audio.fullsubnet import FullSubNet_Plus, FullSubNetPlusConfig
    torch.manual_seed(3)
    a = FullSubNet_Plus(FullSubNetPlusConfig(num_freqs=33, sb_num_neighbors=3, sb_model_hidden_size=16))
    torch.manual_seed(3)
    b = FullSubNet_Plus(FullSubNetPlusConfig(num_freqs=33, sb_num_neighbors=3, sb_model_hidden_size=16))
    for (n, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        assert torch.equal(p, q), n


def test_config_validation_and_unsupported_variants():
    import pydantic
    from nppc_audio.fullsubnet import FullSubNet_Plus, FullSubNetPlusConfig
    from nppc_audio.nppc_model import NPPCModelConfig, StftConfig
    from nppc_audio.trainer import NPPCAudioTrainerConfig
    assert StftConfig().nfft == 512 and StftConfig().hop_length == 256
    with pytest.raises(pydantic.ValidationError):
        FullSubNetPlusConfig(kersize=3)
    with pytest.raises(NotImplementedError):
        FullSubNet_Plus(FullSubNetPlusConfig(channel_attention_model="SE"))
    with pytest.raises(NotImplementedError):
        FullSubNet_Plus(FullSubNetPlusConfig(norm_type="cumulative_laplace_norm"))
    with pytest.raises(NotImplementedError):
        FullSubNet_Plus(FullSubNetPlusConfig(sequence_model="GRU"))
    with pytest.raises(pydantic.ValidationError):
        NPPCModelConfig(pretrained_restoration_model_configuration={}, pretrained_restoration_model_path="x",
                        audio_pc_wrapper_configuration={}, stft_configuration={})
    f = NPPCAudioTrainerConfig.model_fields
    assert f["second_moment_loss_grace"].default == 500 and f["second_moment_loss_lambda"].default == 1.0
    assert f["learning_rate"].default == 1e-4 and f["device"].default == "cuda"


def test_missing_restorer_checkpoint_is_an_assertion(tmp_path):
    from nppc_audio.nppc_model import NPPCModel, NPPCModelConfig
    cfg = NPPCModelConfig(pretrained_restoration_model_configuration={}, pretrained_restoration_model_path=str(tmp_path / "no.tar"),
                          audio_pc_wrapper_configuration=dict(multi_direction_configuration=dict(n_directions=2)),
                          stft_configuration={}, device="cpu")
    with pytest.raises(AssertionError, match="is not exist"):
        NPPCModel(cfg)


def test_flat_params_views_and_branch_stride():
    from nppc_audio.engine import FlatParams
    from nppc_audio.networks import MultiDirectionConfig, MultiDirectionFullSubNet_Plus
    net = MultiDirectionFullSubNet_Plus(MultiDirectionConfig(num_freqs=33, sb_num_neighbors=3, sb_model_hidden_size=16,
                                                             n_directions=3))
    before = {k: v.clone() for k, v in net.state_dict().items()}
    fp = FlatParams(net, torch.device("cpu"))
    assert fp.flat.numel() == sum(v.numel() for v in before.values())
    for n, p in net.named_parameters():
        assert torch.equal(p, before[n])
        o, shp = fp.off[n]
        assert p.data_ptr() == fp.flat.data_ptr() + 4 * o and tuple(p.shape) == shp
    s = fp.branch_stride()
    for leaf in ("sequence_model.3.norm1.weight", "sequence_model.7.sconv.bias", "fc_output_layer.bias"):
        a, b, c = (fp.off[f"fb_model{br}.{leaf}"][0] for br in ("", "_real", "_imag"))
        assert b - a == s and c - b == s
    # load_state_dict writes through the views (checkpoints keep working after flattening)
    new = {k: torch.full_like(v, 0.5) for k, v in before.items()}
    net.load_state_dict(new, strict=True)
    assert float(fp.flat.min()) == 0.5 and float(fp.flat.max()) == 0.5


def test_unfold_multiplicity_counts_reflected_windows():
    from nppc_audio.engine import unfold_multiplicity
    from oracle import nppc_ref as R
    F, nb = 33, 3
    m = unfold_multiplicity(F, nb)
    assert m.sum() == F * (2 * nb + 1)
    x = torch.arange(F, dtype=torch.float32).reshape(1, 1, F, 1) + 1
    unf = R.subband_unfold(x, nb)                               # oracle unfold (reflect)
    assert abs(float(unf.sum()) - float((torch.from_numpy(m) * x[0, 0, :, 0]).sum())) < 1e-3


def test_lambda_ramp_and_loop_loader():
    from nppc_audio.pc_ops import second_moment_weight
    from nppc_audio.trainer import LoopLoader
    assert second_moment_weight(0, 500, 1.0) == 1e-6
    assert second_moment_weight(250, 500, 1.0) == 1e-6
    assert second_moment_weight(375, 500, 2.0) == 1.0
    assert second_moment_weight(10 ** 6, 500, 1.0) == 1.0
    ll = LoopLoader([1, 2, 3], n_steps=7)
    assert list(ll) == [1, 2, 3, 1, 2, 3, 1] and len(ll) == 7
    assert len(LoopLoader([1, 2, 3], n_epochs=2)) == 6 and len(LoopLoader([1, 2, 3], n_steps=10, n_epochs=2)) == 6
    with pytest.raises(Exception):
        LoopLoader([1])


def test_sharded_batch_sampler_contiguous_equal_shards():
    from nppc_audio.trainer import ShardedBatchSampler
    n, gb, W_ = 37, 8, 4
    got = [list(ShardedBatchSampler(n, gb, W_, r, shuffle=False)) for r in range(W_)]
    assert all(len(g) == n // gb for g in got)
    for step in range(n // gb):
        glob = sum((got[r][step] for r in range(W_)), [])
        assert glob == list(range(step * gb, (step + 1) * gb))       # concatenation of the shards = the global batch
        for r in range(W_):
            # even local batch + contiguous shards -> local index parity == global index parity (drop-band groups)
            assert all((i_loc % 2) == (i_glob % 2) for i_loc, i_glob in enumerate(got[r][step]))
    a = [list(ShardedBatchSampler(n, gb, W_, r, shuffle=True, seed=5)) for r in range(W_)]
    flat = sorted(sum((sum(x, []) for x in a), []))
    assert len(flat) == len(set(flat)) == (n // gb) * gb


def test_synthetic_dataset_matches_oracle_recipe():
    from nppc_audio.data import SyntheticNoisySpeech
    ds = SyntheticNoisySpeech(3, 2048)
    noisy, clean = W.synth_batch(3, 2048)
    for i in range(3):
        y, c = ds[i]
        assert np.array_equal(y.numpy(), noisy[i]) and np.array_equal(c.numpy(), clean[i])


def test_checkpoint_wire_format_round_trip(tmp_path):
    """row f2: save_checkpoint writes {model_state_dict, optimizer_state_dict, step} with the reference's tensor names
    (trainer.py:319-335) and it loads back strictly; the restorer file format {"model": sd} is read by preload_model."""
    import numpy as np
    import torch
    from oracle import weights as W
    from nppc_audio.data import SyntheticNoisySpeech
    from nppc_audio.fullsubnet import FullSubNet_Plus, FullSubNetPlusConfig
    from nppc_audio.trainer import NPPCAudioTrainer, NPPCAudioTrainerConfig
    tiny = dict(num_freqs=33, sb_num_neighbors=3, sb_model_hidden_size=16)
    ck = str(tmp_path / "restorer.tar")
    torch.save({"model": FullSubNet_Plus(FullSubNetPlusConfig(**tiny)).state_dict()}, ck)
    cfg = NPPCAudioTrainerConfig(
        nppc_model_configuration=dict(
            pretrained_restoration_model_configuration=tiny, pretrained_restoration_model_path=ck,
            audio_pc_wrapper_configuration=dict(multi_direction_configuration=dict(tiny, n_directions=3)),
            stft_configuration=dict(nfft=64, hop_length=32, win_length=64), device="cpu"),
        data_configuration=dict(data_path=".", dataset=dict(clean_path=".", noisy_path=".")),
        data_loader_configuration=dict(batch_size=2, num_workers=0, pin_memory=False, shuffle=False),
        optimizer_configuration=dict(type="Adam", args=dict(lr=1e-4)), device="cpu")
    tr = NPPCAudioTrainer(cfg, dataset=SyntheticNoisySpeech(2, 1024))
    tr.step = 7
    path = str(tmp_path / "ck" / "checkpoint.pt")
    tr.save_checkpoint(path)
    ckpt = torch.load(path, map_location="cpu")
    assert set(ckpt) == {"model_state_dict", "optimizer_state_dict", "step"} and ckpt["step"] == 7
    spec = W.nppc_spec(3, num_freqs=33, sb_neighbors=3, sb_hidden=16)
    assert list(ckpt["model_state_dict"]) == list(spec)
    assert all(tuple(ckpt["model_state_dict"][k].shape) == tuple(spec[k]) for k in spec)
    ref_opt = torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=1e-4).state_dict()
    got_opt = ckpt["optimizer_state_dict"]
    assert set(got_opt) == set(ref_opt) and len(got_opt["param_groups"]) == 1
    for k in ("lr", "betas", "eps", "weight_decay"):
        assert got_opt["param_groups"][0][k] == ref_opt["param_groups"][0][k], k
    tr2 = NPPCAudioTrainer(cfg, dataset=SyntheticNoisySpeech(2, 1024))
    tr2.nppc_model.load_state_dict(ckpt["model_state_dict"], strict=True)
    tr2.optimizer.load_state_dict(ckpt["optimizer_state_dict"])
    for (n1, p1), (n2, p2) in zip(tr.nppc_model.state_dict().items(), tr2.nppc_model.state_dict().items()):
        assert n1 == n2 and np.array_equal(p1.numpy(), p2.numpy())


def test_workspaces_die_with_their_owner():
    """step-persistent workspaces are keyed by id(owner) and released by a finalizer of the owner (a deleted model must
    give its saved LSTM state back); the sticky hand-off time-out count of a released flag block is kept"""
    import gc
    from nppc_audio import ops_lstm

    class Owner:
        pass

    o = Owner()
    ops_lstm.own_workspaces(o)
    n0 = len(ops_lstm._WS)
    flags = ops_lstm.workspace(("lstm", id(o), True, "coop_flags"), (12,), torch.int32, "cpu", zero=True)
    flags[-4] = 3
    ops_lstm.workspace(("lstm", id(o), True, "h1"), (12,), torch.float32, "cpu")
    assert ops_lstm.workspace(("lstm", id(o), True, "h1"), (12,), torch.float32, "cpu") is ops_lstm._WS[
        (("lstm", id(o), True, "h1"), (12,), torch.float32, "cpu")]
    assert len(ops_lstm._WS) == n0 + 2 and ops_lstm.coop_timeouts() == 3
    del o, flags
    gc.collect()
    assert len(ops_lstm._WS) == n0 and ops_lstm.coop_timeouts() == 3
    ops_lstm.clear_coop_timeouts()
    assert ops_lstm.coop_timeouts() == 0
