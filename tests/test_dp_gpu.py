"""GPU: the data-parallel branch of NPPCAudioTrainer.train_step on the real kernels -- two ranks (gloo, both on cuda:0, the
tiny fixture so that no cooperative kernel is involved) must end up with the weights of one process on the global batch:
sharded sampler, range hooks inside backward, bucketed exchange, 1/W in the Adam scale, pipelined update."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from golden_util import load, waves
from oracle import weights as W

pytestmark = pytest.mark.gpu
NAME = "g0_tiny_g1"          # G = 1: a local batch of 2 satisfies B > G


def _trainer(c, tmp, noisy, clean, batch_size):
    from nppc_audio.nppc_model import NPPCModelConfig
    from nppc_audio.trainer import NPPCAudioTrainer, NPPCAudioTrainerConfig
    spec = W.nppc_spec(c["K"], num_freqs=c["F"], sb_neighbors=c["sbn"], sb_hidden=c["sbh"])
    wts = {k: torch.from_numpy(v) for k, v in W.make_weights(spec, c["seed"]).items()}
    pre = "pretrained_restoration_model."
    ck = os.path.join(tmp, f"restorer_{os.getpid()}.tar")
    torch.save({"model": {k[len(pre):]: v for k, v in wts.items() if k.startswith(pre)}}, ck)
    common = dict(num_freqs=c["F"], sb_num_neighbors=c["sbn"], sb_model_hidden_size=c["sbh"], precision="fp32")
    mc = NPPCModelConfig(
        pretrained_restoration_model_configuration=dict(common, num_groups_in_drop_band=c["G_rest"]),
        pretrained_restoration_model_path=ck,
        audio_pc_wrapper_configuration=dict(multi_direction_configuration=dict(common, num_groups_in_drop_band=c["G_pc"],
                                                                               n_directions=c["K"])),
        stft_configuration=dict(nfft=c["nfft"], hop_length=c["hop"], win_length=c["nfft"]), device="cuda")
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
    tr.nppc_model.to("cuda")
    return tr


def _run(tr, loader_batches, pipeline):
    tr.pipeline_update = pipeline
    tr.step = 500
    objs = []
    for b in loader_batches:
        _, obj, _ = tr.train_step(tuple(x.cuda() for x in b))
        objs.append(float(obj))
    tr.flush()
    torch.cuda.synchronize()
    return objs, {k: v.detach().cpu().clone() for k, v in tr.nppc_model.audio_pc_wrapper.state_dict().items()}


def _worker(rank, world, port, tmp, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    z, meta = load(NAME)
    c = meta["config"]
    noisy, clean = (torch.from_numpy(a) for a in waves(z, meta))
    tr = _trainer(c, tmp, noisy, clean, c["B"])               # GLOBAL batch size in the config, like the reference yaml
    assert tr.world == world and tr.rank == rank
    batches = list(tr.dataloader) * 2                         # this rank's contiguous shard of the one global minibatch, twice
    assert batches[0][0].shape[0] == c["B"] // world
    objs, sd = _run(tr, batches, pipeline=(rank >= 0))
    o = torch.tensor(objs, dtype=torch.float64)
    dist.all_reduce(o)
    if rank == 0:
        torch.save({"objs": (o / world).tolist(), "sd": sd}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_the_real_kernels_equal_one_process_on_the_global_batch(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "dp.pt")
    mp.spawn(_worker, args=(2, port, str(tmp_path), out), nprocs=2, join=True)
    got = torch.load(out)
    z, meta = load(NAME)
    c = meta["config"]
    noisy, clean = (torch.from_numpy(a) for a in waves(z, meta))
    tr = _trainer(c, str(tmp_path), noisy, clean, c["B"])
    objs, sd = _run(tr, [(noisy, clean)] * 2, pipeline=False)
    # the objective is a mean over samples and nothing couples them (G = 1: the same bins for every sample): the mean of the
    # two shard objectives is the global objective, step after step (the second step sees the first update)
    assert np.allclose(got["objs"], objs, rtol=0, atol=2e-6), (got["objs"], objs)
    for k, v in sd.items():
        d = float((got["sd"][k].float() - v.float()).abs().max())
        assert d < 2e-6, (k, d)                               # two Adam steps of 1e-4: summation order only
    moved = max(float((sd[k] - torch.from_numpy(wv)).abs().max()) for k, wv in
                ((k, W.make_weights({"audio_pc_wrapper." + k: tuple(v.shape)}, c["seed"])["audio_pc_wrapper." + k]) for k, v in list(sd.items())[:5]))
    assert moved > 1e-5
