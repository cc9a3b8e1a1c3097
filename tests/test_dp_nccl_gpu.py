"""GPU, >= 2 devices (skipped on the one-GPU boxes): the PRODUCTION data-parallel path -- one process per GPU, backend
`nccl` (= RCCL over xGMI), the bf16 train step at BASELINE-C2 sequence counts, so that the cooperative CU-pair LSTM kernels
(csrc/lstm_coop.hip) run beside RCCL's own kernels -- against one process on the global batch.

What it pins: (i) the recipe (contiguous shards, one summed flat-gradient exchange, 1/W in
the Adam scale) reproduces the global-batch update at the bf16 tolerance; (ii) no bounded hand-off spin of a cooperative
kernel times out on any rank while RCCL is active (sticky counters, all-reduced with MAX like trainer.check_coop_timeouts);
(iii) the same with the exchange bucketed INSIDE backward (NPPC_DP_OVERLAP=1) and the parked update (NPPC_DP_PIPELINE=1).
Precedent in the reference: nppc/auxil.py:212-252, nppc/nppc.py:370-386 (per-rank batch = batch_size // world, averaged
gradients)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import weights as W

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL over xGMI)")]

WORLD = 2
B_LOCAL, SECONDS, K = 32, 2, 5            # C2 sequence counts per rank (8224 / 4096 sequences: CU-pair plans), 2 s clips
L = SECONDS * 16000
STEPS = 2
SEED = 11


def _trainer(tmp, noisy, clean, batch_size, device):
    from nppc_audio.nppc_model import NPPCModelConfig
    from nppc_audio.trainer import NPPCAudioTrainer, NPPCAudioTrainerConfig
    spec = W.nppc_spec(K)
    wts = {k: torch.from_numpy(v) for k, v in W.make_weights(spec, SEED).items()}
    pre = "pretrained_restoration_model."
    ck = os.path.join(tmp, f"restorer_{os.getpid()}.tar")
    torch.save({"model": {k[len(pre):]: v for k, v in wts.items() if k.startswith(pre)}}, ck)
    mc = NPPCModelConfig(
        pretrained_restoration_model_configuration=dict(num_groups_in_drop_band=1, precision="bf16"),
        pretrained_restoration_model_path=ck,
        audio_pc_wrapper_configuration=dict(multi_direction_configuration=dict(num_groups_in_drop_band=2, n_directions=K,
                                                                               precision="bf16")),
        stft_configuration=dict(nfft=512, hop_length=256, win_length=512), device="cuda")
    cfg = NPPCAudioTrainerConfig(
        nppc_model_configuration=mc, data_configuration=dict(data_path=".", dataset=dict(clean_path=".", noisy_path=".")),
        data_loader_configuration=dict(batch_size=batch_size, num_workers=0, pin_memory=False, shuffle=False),
        optimizer_configuration=dict(type="Adam", args=dict(lr=1e-4, betas=[0.9, 0.999], eps=1e-8, weight_decay=0)),
        device="cuda")

    class Mem(torch.utils.data.Dataset):
        def __len__(self):
            return noisy.shape[0]

        def __getitem__(self, i):
            return noisy[i], clean[i]

    tr = NPPCAudioTrainer(cfg, dataset=Mem())
    tr.nppc_model.load_state_dict(wts, strict=True)
    tr.nppc_model.to(device)
    return tr, {k[len("audio_pc_wrapper."):]: v for k, v in wts.items() if k.startswith("audio_pc_wrapper.")}


def _steps(tr, batches, device):
    from nppc_audio import ops_lstm
    tr.step = 500
    ops_lstm.clear_coop_timeouts()
    objs = []
    for b in batches:
        _, obj, _ = tr.train_step(tuple(x.to(device) for x in b))
        objs.append(float(obj))
    tr.flush()
    torch.cuda.synchronize()
    return objs, {k: v.detach().float().cpu().clone() for k, v in tr.nppc_model.audio_pc_wrapper.state_dict().items()}, ops_lstm.coop_timeouts()


def _data():
    noisy, clean = W.synth_batch(WORLD * B_LOCAL, L)
    return torch.from_numpy(noisy), torch.from_numpy(clean)


def _worker(rank, world, port, tmp, out, overlap):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
                      NPPC_DP_OVERLAP="1" if overlap else "0", NPPC_DP_PIPELINE="1" if overlap else "0")
    torch.cuda.set_device(rank)
    device = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    noisy, clean = _data()
    tr, _ = _trainer(tmp, noisy, clean, WORLD * B_LOCAL, device)      # GLOBAL batch size in the config, like the reference yaml
    assert tr.world == world and tr.rank == rank
    tr.pipeline_update = bool(overlap)
    batches = list(tr.dataloader) * STEPS                              # this rank's contiguous shard of the one global minibatch
    assert batches[0][0].shape[0] == B_LOCAL
    objs, sd, nto = _steps(tr, batches, device)
    o = torch.tensor(objs, dtype=torch.float64, device=device)
    dist.all_reduce(o)
    t = torch.tensor([nto], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)                           # a time-out on ANY rank fails every rank
    # the ranks hold the same weights after the exchange: replicated optimizer, summed gradient
    flat = torch.cat([v.reshape(-1) for v in sd.values()]).to(device)
    lo, hi = flat.clone(), flat.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    if rank == 0:
        torch.save({"objs": (o / world).tolist(), "sd": sd, "timeouts": int(t), "rank_spread": float((hi - lo).abs().max()),
                    "backend": dist.get_backend(), "overlap": bool(tr._reducer.overlap)}, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [False, True])
def test_two_gpus_over_rccl_equal_one_process_on_the_global_batch(tmp_path, overlap):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "dp.pt")
    mp.spawn(_worker, args=(WORLD, port, str(tmp_path), out, overlap), nprocs=WORLD, join=True)
    got = torch.load(out)
    assert got["backend"] == "nccl" and got["overlap"] == overlap
    assert got["timeouts"] == 0, "a cooperative LSTM hand-off timed out beside RCCL"
    assert got["rank_spread"] == 0.0, "the ranks' weights diverged"
    # one process, the global batch (its 16448 / 8192 sequences exceed the CU-pair plans: other kernels, same arithmetic)
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    noisy, clean = _data()
    tr, w0 = _trainer(str(tmp_path), noisy, clean, WORLD * B_LOCAL, device)
    objs, sd, nto = _steps(tr, [(noisy, clean)] * STEPS, device)
    assert nto == 0
    # the objective is a mean over samples and nothing couples them across shards (even local batch: the drop-band groups
    # agree): mean of the shard objectives = global objective, step after step, at the bf16 tolerance of the step tests
    assert np.allclose(got["objs"], objs, rtol=2e-3, atol=1e-5), (got["objs"], objs)
    # weights after two Adam steps (lr 1e-4: an element moves by <= ~2.05e-4): Adam normalises, so where a gradient is near
    # zero the bf16 noise of two different kernel plans may flip an update; held as (i) the movement vectors agree in
    # direction, (ii) few elements differ by more than a quarter of one update, (iii) none by more than the two runs can move
    mv_a = torch.cat([(got["sd"][k] - w0[k].float()).reshape(-1) for k in sd])
    mv_b = torch.cat([(sd[k] - w0[k].float()).reshape(-1) for k in sd])
    assert float(mv_b.abs().max()) > 1e-4                                 # two updates really happened
    cos = float(torch.dot(mv_a, mv_b) / (mv_a.norm() * mv_b.norm()))
    d = (mv_a - mv_b).abs()
    assert cos > 0.97, cos
    assert float((d > 2.5e-5).float().mean()) < 0.05, float((d > 2.5e-5).float().mean())
    assert float(d.max()) < 4.5e-4, float(d.max())
