#!/usr/bin/env python3
"""Headline benchmark: NPPC K=5 FullSubNet+ train step, spectrogram-frames/s on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one full train step of BASELINE config C2 on one minibatch of synthetic 16 kHz noisy speech that is
already resident in HBM: noisy+clean STFT, frozen FullSubNet+ restorer forward, cIRM build/apply, K=5 direction net
forward, Gram-Schmidt, NPPC loss, full backward, gradient all-reduce (N > 1), Adam.  Weak scaling: every rank runs
the C2 per-GPU batch (32 x 4 s), rank r on clips [32r, 32r+32).  Weights: seeded random init (no checkpoint offline).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "generative-audio_amd"))

BATCH, SECONDS, SR, K_DIRS = 32, 4, 16000, 5
NFFT, HOP = 512, 256
PEAK_BF16_TFLOPS = 2500.0     # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)
PEAK_F32_TFLOPS = 157.3


def build_trainer(precision, rank, world, batch, length, n_dirs=K_DIRS):
    from nppc_audio.data import SyntheticNoisySpeech
    from nppc_audio.fullsubnet import FullSubNet_Plus, FullSubNetPlusConfig
    from nppc_audio.trainer import NPPCAudioTrainer, NPPCAudioTrainerConfig
    torch.manual_seed(0)                                         # identical init on every rank
    rest_cfg = dict(num_groups_in_drop_band=1, precision=precision)
    tmp = tempfile.mkdtemp(prefix=f"nppc_bench_r{rank}_")
    ck = os.path.join(tmp, "restorer.tar")
    torch.save({"model": FullSubNet_Plus(FullSubNetPlusConfig(**rest_cfg)).state_dict()}, ck)
    cfg = NPPCAudioTrainerConfig(
        nppc_model_configuration=dict(
            pretrained_restoration_model_configuration=rest_cfg, pretrained_restoration_model_path=ck,
            audio_pc_wrapper_configuration=dict(multi_direction_configuration=dict(
                num_groups_in_drop_band=2, n_directions=n_dirs, precision=precision)),
            stft_configuration=dict(nfft=NFFT, hop_length=HOP, win_length=NFFT), device="cuda"),
        data_configuration=dict(data_path=".", dataset=dict(clean_path=".", noisy_path=".")),
        data_loader_configuration=dict(batch_size=batch * world, num_workers=0, pin_memory=False, shuffle=False),
        optimizer_configuration=dict(type="Adam", args=dict(lr=1e-4, betas=[0.9, 0.999], eps=1e-8, weight_decay=0)),
        device="cuda")
    ds = SyntheticNoisySpeech(batch * world, length)
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):          # the trainer prints like the reference; stdout is ONE JSON line
        tr = NPPCAudioTrainer(cfg, dataset=ds)
    dev = torch.device("cuda", torch.cuda.current_device())
    # this rank's shard of the (single) global minibatch, uploaded once: inputs are HBM-resident for the timed region
    clips = [ds[rank * batch + i] for i in range(batch)]
    noisy = torch.stack([c[0] for c in clips]).to(dev)
    clean = torch.stack([c[1] for c in clips]).to(dev)
    return tr, (noisy, clean)


def host_cores():
    """CPU threads this process may really use: affinity mask, capped at the GPU box's per-GPU share (16)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def pmc_traffic(dom_name):
    """HBM bytes per launch of the dominant kernel (FETCH_SIZE x2 per the gfx950 note + WRITE_SIZE) from the committed
    rocprofv3 counter passes of this same command (profiles/r01_bench_c2_bf16_pmc_traffic.csv, tools/summarize_pmc.py):
    counters cannot be collected from inside the timed process, so the figure is null when that file is absent or the
    run is not the default C2 configuration it was measured on."""
    import csv
    path = os.path.join(ROOT, "profiles", "r01_bench_c2_bf16_pmc_traffic.csv")
    key = {"lstm2_bwd_coop_ksplit": "lstm2_coop_bwd2_kernel", "lstm2_fwd_coop_g2[N=8224": "2, 5, 64, false",
           "lstm2_fwd_coop_g2[N=4096": "2, 2, 64, true"}
    pat = next((v for k, v in key.items() if dom_name.startswith(k)), None)
    if pat is None or not os.path.exists(path) or "N=8224" not in dom_name and "N=4096" not in dom_name:
        return None, "no committed PMC pass for this kernel / configuration"
    with open(path) as f:
        for r in csv.DictReader(f):
            if pat in r["kernel"]:
                mb = float(r["fetch_MB_per_launch_x2_corrected"]) + float(r["write_MB_per_launch"])
                return mb * 1048576.0, "bytes per launch, profiles/r01_bench_c2_bf16_pmc_traffic.csv (separate --pmc passes)"
    return None, "kernel not in the committed PMC summary"


def cpu_baseline():
    """Reference-shaped CPU path (the oracle restatement, pinned to the reference by tests/golden) on a bounded
    sample: B=4 x 4 s, K=5, G_pc=2, one full train step (forward x [2 restorer, 3 STFT] + backward + Adam)."""
    sys.path.insert(0, ROOT)
    from oracle import nppc_ref as R
    from oracle import weights as W
    torch.set_num_threads(host_cores())
    B, L = 4, SECONDS * SR
    spec = W.nppc_spec(K_DIRS)
    P = {k: torch.from_numpy(v) for k, v in W.make_weights(spec, 7).items()}
    train = {k: v.requires_grad_(True) for k, v in P.items() if k.startswith("audio_pc_wrapper")}
    noisy, clean = (torch.from_numpy(a) for a in W.synth_batch(B, L))
    state = {}

    def one(step, n, c):
        _, obj, _ = R.nppc_step(n, c, P, K_DIRS, step, g_rest=1, g_pc=2, reference_shaped=True)
        names = list(train)
        gs = torch.autograd.grad(obj, [train[k] for k in names])
        with torch.no_grad():
            R.adam_step(train, dict(zip(names, gs)), state, step + 1)

    one(0, noisy[:, :4096], clean[:, :4096])                  # thread-pool / allocator warm-up on a 0.25 s crop
    t0 = time.perf_counter()
    one(1, noisy, clean)
    dt = time.perf_counter() - t0
    frames = B * (1 + L // HOP)
    return {"value": frames / dt, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"1 train step, B={B} x {SECONDS} s, K={K_DIRS}, G_pc=2, fp32, reference-shaped "
                      f"(2x restorer, 3x STFT), {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--seconds", type=float, default=SECONDS)
    ap.add_argument("--dirs", type=int, default=K_DIRS, help="number of PC directions K (config C5 uses 8)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} needs torch.distributed.run with --nproc-per-node {a.gpus} (WORLD_SIZE={world})")
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local))

    from nppc_audio import ops_lstm
    length = int(a.seconds * SR)
    tr, batch = build_trainer(a.precision, rank, world, a.batch, length, a.dirs)
    frames = a.batch * (1 + length // HOP)

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    if rank == 0:
        log(f"model built; {a.warmup} warm-up + {a.steps} timed steps, {frames} frames/step/GPU")
    for _ in range(a.warmup):
        tr.train_step(batch)
    sync()
    if rank == 0:
        log("warm-up done")
    ops_lstm.PROFILE = []
    t0 = time.perf_counter()
    for _ in range(a.steps):
        _, objective, _ = tr.train_step(batch)
    sync()
    dt = time.perf_counter() - t0
    prof, ops_lstm.PROFILE = ops_lstm.PROFILE, None
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt)
    if rank != 0:
        if world > 1:
            torch.distributed.destroy_process_group()
        return

    # dominant kernel: the fused 2-layer LSTM recurrence (forward instances + the backward recurrence)
    per = {}
    for label, e0, e1 in prof:
        per.setdefault(label, []).append(e0.elapsed_time(e1))
    kern = {}
    H_, I_ = 384, 34
    for (name, train, N, Tn, mt), ms in per.items():
        avg = sum(ms) / len(ms)
        flops = N * Tn * (2 * 4 * H_ * (I_ + H_) + 2 * 4 * H_ * (2 * H_))      # algorithmic: both layers' gate GEMMs
        kern[f"{name}{'_train' if train and name.endswith('fwd') else ''}[N={N},T'={Tn},mtile={mt}]"] = dict(
            ms=avg, tflops=flops / (avg * 1e-3) / 1e12)
    dom_name, dom = max(kern.items(), key=lambda kv: kv[1]["ms"]) if kern else ("none", dict(ms=0.0, tflops=0.0))
    peak = PEAK_BF16_TFLOPS if a.precision == "bf16" else PEAK_F32_TFLOPS
    out = {
        "metric": "spectrogram-frames/sec, NPPC K=5 FullSubNet+ train step",
        "value": world * frames * a.steps / dt,
        "unit": "frames/s",
        "per_gpu": frames * a.steps / dt,
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": 1e3 * dt / a.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16" if a.precision == "bf16" else "f32", "data": "synthetic",
        "config": {"workload": f"{'C2' if (a.batch, a.seconds, a.dirs) == (BATCH, SECONDS, K_DIRS) else ('C5' if (a.batch, a.seconds, a.dirs) == (8, 30, 8) else 'custom')}: speech-enhancement NPPC, FullSubNet+ restorer + K={a.dirs} direction net, "
                               f"batch={a.batch}x{a.seconds:g}s@16kHz per GPU, STFT {NFFT}/{HOP}, G_rest=1 G_pc=2, "
                               f"full train step (fwd+loss+bwd+Adam{'+RCCL all-reduce' if world > 1 else ''})",
                   "global_batch": a.batch * world, "frames_per_step_per_gpu": frames,
                   "parallelism": f"dp{world}", "objective_last": float(objective)},
        "roofline": {"bound": "mfma", "kernel": dom_name, "achieved": dom["tflops"], "peak": peak, "unit": "TFLOP/s",
                     "frac": dom["tflops"] / peak, "avg_ms": dom["ms"], "traffic": None,
                     "all_lstm_kernels": kern},
    }
    log(f"timed region done: {1e3 * dt / a.steps:.1f} ms/step")
    tr_bytes, tr_src = pmc_traffic(dom_name)
    out["roofline"]["traffic"] = tr_bytes
    out["roofline"]["traffic_note"] = tr_src
    from nppc_audio import ops_lstm
    nto = ops_lstm.coop_timeouts()
    if nto:      # a bounded hand-off spin gave up: the step's results are wrong, so is any number measured on them
        raise RuntimeError(f"{nto} cooperative LSTM hand-off time-outs during the run: measurement invalid")
    out["lstm_handoff_timeouts"] = 0
    if world == 1 and not a.no_cpu_baseline:
        log("timing the CPU baseline (bounded sample)")
        out["cpu_baseline"] = cpu_baseline()
        out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
    print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
