#!/usr/bin/env python3
"""Config C3 of BASELINE.json: inpainting NPPC train step (U-Net restorer + K=5 direction U-Net), batch 32 x 4 s.

Same JSON line as bench.py (which stays on the headline config C2); single GPU ("replicas only": the direction
U-Net's train-mode BatchNorm and the batch-global input statistics couple the samples).
One step = log-magnitude preprocessing, frozen restorer forward (eval BatchNorm folded), direction U-Net forward
(batch-stat BatchNorm), masking, Gram-Schmidt, loss, full backward, clip_grad_norm_ + Adam.  Inputs (STFT pairs +
frame mask) are synthetic and HBM-resident.
"""
import argparse
import contextlib
import json
import os
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "generative-audio_amd"))
sys.path.insert(0, ROOT)
NFFT, HOP, K_DIRS = 255, 128, 5
PEAK = {"bf16": 2500.0, "fp32": 157.3}
# rocprofv3 kernel name of each HIP-event family (unet_engine.PROFILE kinds) and the committed counter passes of this command
ROCPROF_NAME = {"conv_fwd": "conv_tiled_kernel", "conv_bwd_data": "conv_tiled_kernel", "conv_wgrad": "gemm_tn_tiled_kernel"}
PMC_CSV = os.path.join(ROOT, "profiles", "r04_bench_c3_bf16_pmc_traffic.csv")
PMC_STAMP = os.path.join(ROOT, "profiles", "r04_bench_c3_bf16_pmc_stamp.json")


def log(msg):
    print(f"[bench-c3 {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def synth(B, F, T, dev):
    """STFT-like pairs: complex Gaussian with a 1/f-ish spectral tilt, -25 dBFS-ish scale; a 13-frame gap (100 ms)"""
    g = torch.Generator(device="cpu").manual_seed(1234)
    tilt = (1.0 / (1.0 + torch.arange(F, dtype=torch.float32) / 8.0))[None, None, :, None]
    clean = torch.randn(B, 2, F, T, generator=g) * tilt * 2.0
    mask = torch.ones(B, T)
    for i in range(B):
        s0 = int(torch.randint(20, T - 40, (1,), generator=g))
        mask[i, s0:s0 + 13] = 0
    return (clean * mask[:, None, None, :]).to(dev), mask.to(dev), clean.to(dev)


def build(precision, B, F, T):
    from nppc_audio.inpainting.networks.unet import UNet, UNetConfig
    from nppc_audio.inpainting.trainer.nppc_trainer import NPPCAudioInpaintingTrainer, NPPCAudioInpaintingTrainerConfig
    torch.manual_seed(0)
    tmp = tempfile.mkdtemp(prefix="nppc_bench_c3_")
    ck = os.path.join(tmp, "restorer.pt")
    torch.save({"model_state_dict": UNet(UNetConfig(in_channels=1, out_channels=1, dropout=0.2)).state_dict()}, ck)
    cfg = NPPCAudioInpaintingTrainerConfig(
        nppc_model_configuration=dict(
            pretrained_restoration_model_configuration=dict(in_channels=1, out_channels=1, dropout=0.2, precision=precision),
            pretrained_restoration_model_path=ck,
            audio_pc_wrapper_configuration=dict(n_dirs=K_DIRS, model_configuration=dict(in_channels=2, out_channels=K_DIRS,
                                                                                         precision=precision)),
            device="cuda"),
        data_configuration=dict(clean_path=".", stft_configuration=dict(nfft=NFFT, hop_length=HOP, win_length=NFFT)),
        dataloader_configuration=dict(batch_size=B, num_workers=0, pin_memory=False, shuffle=False),
        optimizer_configuration=dict(type="Adam", args=dict(lr=1e-4, betas=[0.5, 0.999])), device="cuda")

    class One(torch.utils.data.Dataset):
        def __len__(self):
            return B

        def __getitem__(self, i):
            raise IndexError

    with contextlib.redirect_stdout(sys.stderr):
        tr = NPPCAudioInpaintingTrainer(cfg, dataset=One())
        tr.nppc_model.to("cuda")
    return tr


def cpu_baseline(F, T):
    from oracle import inpaint_ref as R
    from oracle import weights as W
    n = min(16, len(os.sched_getaffinity(0)))
    torch.set_num_threads(n)
    B = 4
    P = {k: torch.from_numpy(v) if hasattr(v, "shape") and v.shape else torch.tensor(v) for k, v in
         W.make_weights(W.inpainting_spec(K_DIRS), 7).items()}
    train = {k: v.requires_grad_(True) for k, v in P.items()
             if k.startswith("pc_wrapper.") and v.is_floating_point() and "running_" not in k}
    masked, mask, clean = (t.cpu() for t in synth(B, F, T, "cpu"))
    state = {}

    def one(step, sl):
        stats = {}
        _, obj, _ = R.inpaint_step(masked[..., sl], mask[:, sl], clean[..., sl], P, step, reference_shaped=True,
                                   new_stats=stats)
        gs = dict(zip(train, torch.autograd.grad(obj, list(train.values()))))
        coef, _ = R.clip_coef(gs.values())
        with torch.no_grad():
            R.adam_step(train, {k: g * coef for k, g in gs.items()}, state, step + 1)

    one(0, slice(0, 64))
    t0 = time.perf_counter()
    one(1, slice(0, T))
    dt = time.perf_counter() - t0
    return {"value": B * T / dt, "unit": "frames/s", "cores": n, "kind": "port",
            "sample": f"1 train step, B={B} x 4 s (F={F}, T={T}), K={K_DIRS}, fp32, reference-shaped (2x restorer), {dt:.1f} s"}


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--seconds", type=float, default=4.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args(argv)
    from nppc_audio import unet_engine
    L = int(a.seconds * 16000)
    F, T = NFFT // 2 + 1, 1 + (L + 2 * (NFFT // 2) - NFFT) // HOP
    torch.cuda.set_device(0)
    tr = build(a.precision, a.batch, F, T)
    batch = synth(a.batch, F, T, "cuda")
    log(f"model built; F={F} T={T}; {a.warmup} warm-up + {a.steps} timed steps")
    for _ in range(a.warmup):
        tr.train_step(batch)
    torch.cuda.synchronize()
    log("warm-up done")
    unet_engine.PROFILE = []
    t0 = time.perf_counter()
    for _ in range(a.steps):
        _, objective, _ = tr.train_step(batch)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof, unet_engine.PROFILE = unet_engine.PROFILE, None
    agg = {}
    for kind, flops, e0, e1 in prof:
        d = agg.setdefault(kind, [0.0, 0.0, 0])
        d[0] += flops
        d[1] += e0.elapsed_time(e1)
        d[2] += 1
    kern = {k: dict(launches_per_step=v[2] // a.steps, ms_per_step=v[1] / a.steps, tflops=v[0] / (v[1] * 1e-3) / 1e12)
            for k, v in agg.items()}
    dom_name, dom = max(kern.items(), key=lambda kv: kv[1]["ms_per_step"])
    frames = a.batch * T
    ms_step = 1e3 * dt / a.steps
    flops_step = sum(v[0] for v in agg.values()) / a.steps
    import bench
    wl = "C3" if (a.batch, a.seconds, a.precision) == (32, 4.0, "bf16") else "custom"
    traffic, traffic_note = bench.pmc_traffic(ROCPROF_NAME.get(dom_name), wl, PMC_CSV, PMC_STAMP, expect="C3", aggregate=True)
    out = {
        "metric": "spectrogram-frames/sec, inpainting NPPC K=5 U-Net train step",
        "value": frames * a.steps / dt, "unit": "frames/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "bf16" if a.precision == "bf16" else "f32", "data": "synthetic",
        "config": {"workload": f"C3: audio-inpainting NPPC, U-Net restorer + K={K_DIRS} direction U-Net, batch={a.batch}x"
                               f"{a.seconds:g}s@16kHz, STFT {NFFT}/{HOP} (F={F}, T={T}), full train step "
                               f"(fwd+loss+bwd+clip+Adam)", "frames_per_step": frames, "objective_last": float(objective)},
        "roofline": {"bound": "mfma", "kernel": dom_name, "achieved": dom["tflops"], "peak": PEAK[a.precision],
                     "unit": "TFLOP/s", "frac": dom["tflops"] / PEAK[a.precision], "traffic": traffic,
                     "traffic_note": traffic_note, "conv_families": kern,
                     "step": {"algorithmic_flop": flops_step, "achieved": flops_step / (ms_step * 1e-3) / 1e12,
                              "frac": flops_step / (ms_step * 1e-3) / 1e12 / PEAK[a.precision],
                              "note": "convolution FLOPs only (2*MACs of every conv launch of the step) / whole-step time"}},
    }
    log(f"timed region done: {1e3 * dt / a.steps:.1f} ms/step")
    if not a.no_cpu_baseline:
        log("timing the CPU baseline (bounded sample)")
        out["cpu_baseline"] = cpu_baseline(F, T)
        out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
