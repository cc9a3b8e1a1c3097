"""CPU, world_size 2 (gloo): the data-parallel recipe of NPPCAudioTrainer -- contiguous equal shards of the global
minibatch (trainer.ShardedBatchSampler), the bucketed sum all-reduce of the flat gradient with the 1/W handed to the
optimizer's gradient scale (dp.FlatGradientReducer: the SAME object train_step drives from the engine's range hooks) --
reproduces the single-process global-batch objective and gradients.  The arithmetic is the CPU oracle (the HIP model
needs a GPU).  Also: bench.py --gpus N starts its own ranks and propagates a failing rank's exit code."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import nppc_ref as R
from oracle import weights as W

CFG = dict(F=33, nfft=64, hop=32, sbn=3, sbh=16, K=3, B=8, L=640, G_rest=1, G_pc=2, seed=11)


def _objective_and_flat_grad(noisy, clean, P):
    c = CFG
    train = [k for k in P if k.startswith("audio_pc_wrapper")]
    for k in train:
        P[k].requires_grad_(True)
    _, obj, _ = R.nppc_step(noisy, clean, P, c["K"], 500, stft=(c["nfft"], c["hop"], c["nfft"]), g_rest=c["G_rest"],
                            g_pc=c["G_pc"], sb_neighbors=c["sbn"])
    gs = torch.autograd.grad(obj, [P[k] for k in train])
    return float(obj), torch.cat([g.reshape(-1) for g in gs])


def _weights():
    c = CFG
    spec = W.nppc_spec(c["K"], num_freqs=c["F"], sb_neighbors=c["sbn"], sb_hidden=c["sbh"])
    return {k: torch.from_numpy(v) for k, v in W.make_weights(spec, c["seed"]).items()}


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from nppc_audio.trainer import ShardedBatchSampler
    noisy, clean = (torch.from_numpy(a) for a in W.synth_batch(CFG["B"], CFG["L"]))
    idx = next(iter(ShardedBatchSampler(CFG["B"], CFG["B"], world, rank, shuffle=False)))
    obj, g = _objective_and_flat_grad(noisy[idx], clean[idx], _weights())
    # what NPPCAudioTrainer.train_step does: the engine reports final ranges from inside backward (here: the tail of
    # the buffer first, then a middle range, like the sub-band segment and TCN blocks 7..4), `finish` covers the rest
    from nppc_audio.dp import FlatGradientReducer
    red = FlatGradientReducer()
    assert red.world == world
    n = g.numel()
    red.range_ready(g, n - n // 5, n)
    red.range_ready(g, n // 3, n // 2)
    g *= red.finish(g)                                   # the returned 1/W is what train_step hands to Adam's grad scale
    o = torch.tensor([obj], dtype=torch.float64)
    dist.all_reduce(o)
    if rank == 0:
        torch.save({"g": g, "obj": float(o) / world, "idx": idx}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_dp_equals_global_batch(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "dp.pt")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    assert got["idx"] == [0, 1, 2, 3]
    noisy, clean = (torch.from_numpy(a) for a in W.synth_batch(CFG["B"], CFG["L"]))
    obj, g = _objective_and_flat_grad(noisy, clean, _weights())
    # no op couples samples (per-sample norms, GroupNorm(1,.), per-sequence LSTM) and the objective is a mean over
    # samples, so equal shards reproduce the global batch up to fp32 summation order
    assert abs(got["obj"] - obj) < 1e-5
    assert float((got["g"] - g).abs().max()) < 2e-3 * float(g.abs().max())
    cos = float((got["g"].double() @ g.double()) / (got["g"].double().norm() * g.double().norm()))
    assert cos > 0.99999


def _bench(*args):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(root, "bench.py"), *args], env=env, capture_output=True, text=True,
                          timeout=600)


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no external launcher: the ranks come up (gloo rehearsal of the exchange on CPU),
    rank 0 prints ONE JSON line, exit code 0"""
    import json
    r = _bench("--gpus", "2", "--dp-selftest")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out == {"dp_selftest": True, "n_gpus": 2, "scale": 0.5}


def test_bench_exits_nonzero_when_a_rank_fails():
    r = _bench("--gpus", "2", "--dp-selftest", "--dp-selftest-fail-rank", "1")
    assert r.returncode != 0
