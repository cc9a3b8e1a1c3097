#!/usr/bin/env python3
"""Generate the inpainting golden fixtures (tests/golden/inp_*.npz) by RUNNING THE REFERENCE (CPU, fp32).

Same rules as make_goldens.py: runs only in the build container, imports the reference from
/root/reference where it lies (with the same empty placeholder modules for the absent third-party
packages), writes data only.  Weights come from oracle/weights.py (seeded numpy) and are loaded through
the reference's own `_load_from_local` / `load_state_dict` paths.

Usage:  python tests/golden/make_goldens_inpainting.py
"""
import json
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, HERE)

from oracle import weights as W  # noqa: E402
from make_goldens import install_placeholders, to_t  # noqa: E402


class MemDataset(torch.utils.data.Dataset):
    def __init__(self, cfg):
        pass

    def __len__(self):
        return 1

    def __getitem__(self, i):
        raise IndexError


def small(a, n=4096):
    return np.ascontiguousarray(a).reshape(-1)[:n].copy()


GRAD_NAMES = [
    "pc_wrapper.net.inc.conv.conv.0.weight", "pc_wrapper.net.inc.conv.conv.0.bias",
    "pc_wrapper.net.inc.conv.conv.1.weight", "pc_wrapper.net.inc.conv.conv.1.bias",
    "pc_wrapper.net.inc.conv.conv.3.weight", "pc_wrapper.net.inc.conv.conv.4.weight",
    "pc_wrapper.net.down1.mpconv.1.conv.0.weight", "pc_wrapper.net.down2.mpconv.1.conv.3.weight",
    "pc_wrapper.net.down2.mpconv.1.conv.4.bias", "pc_wrapper.net.down3.mpconv.1.conv.0.weight",
    "pc_wrapper.net.down4.mpconv.1.conv.3.weight", "pc_wrapper.net.down4.mpconv.1.conv.4.weight",
    "pc_wrapper.net.up1.conv.conv.0.weight", "pc_wrapper.net.up1.conv.conv.1.bias",
    "pc_wrapper.net.up2.conv.conv.3.weight", "pc_wrapper.net.up3.conv.conv.0.weight",
    "pc_wrapper.net.up4.conv.conv.0.weight", "pc_wrapper.net.up4.conv.conv.3.bias",
    "pc_wrapper.net.up4.conv.conv.4.weight", "pc_wrapper.net.outc.conv.weight", "pc_wrapper.net.outc.conv.bias",
]
STAT_NAMES = [
    "pc_wrapper.net.inc.conv.conv.1.running_mean", "pc_wrapper.net.inc.conv.conv.1.running_var",
    "pc_wrapper.net.down4.mpconv.1.conv.4.running_mean", "pc_wrapper.net.down4.mpconv.1.conv.4.running_var",
    "pc_wrapper.net.up4.conv.conv.4.running_var", "pc_wrapper.net.up2.conv.conv.1.num_batches_tracked",
]


def run_config(name, c, out_dir):
    import utils as ref_utils
    from nppc_audio.inpainting.trainer import nppc_trainer as ref_tr
    from nppc_audio.inpainting.nppc.pc_wrapper import gram_schmidt_to_spec_mag

    torch.manual_seed(0)
    K, B, T = c["K"], c["B"], c["T"]
    spec = W.inpainting_spec(K)
    wts = W.make_weights(spec, c["seed"])
    masked, mask, clean = W.synth_inpaint_batch(B, T, c["nfft"], c["hop"])

    tmp = tempfile.mkdtemp()
    ck = os.path.join(tmp, "restorer.pt")
    pre = "pretrained_restoration_model.net."
    torch.save({"model_state_dict": to_t({k[len(pre):]: v for k, v in wts.items() if k.startswith(pre)})}, ck)
    ref_tr.AudioInpaintingDataset = MemDataset
    cfg = ref_tr.NPPCAudioInpaintingTrainerConfig(
        nppc_model_configuration=dict(
            pretrained_restoration_model_configuration=dict(in_channels=1, out_channels=1, dropout=0.2),
            pretrained_restoration_model_path=ck,
            audio_pc_wrapper_configuration=dict(n_dirs=K, model_configuration=dict(in_channels=2, out_channels=K)),
            device="cpu"),
        data_configuration=dict(clean_path=".", stft_configuration=dict(nfft=c["nfft"], hop_length=c["hop"],
                                                                         win_length=c["nfft"]), use_vad=False),
        dataloader_configuration=dict(batch_size=B, num_workers=0, pin_memory=False, shuffle=False),
        optimizer_configuration=dict(type="Adam", args=dict(lr=1e-4, betas=[0.5, 0.999])),
        device="cpu")
    tr = ref_tr.NPPCAudioInpaintingTrainer(cfg)
    model = tr.nppc_model
    sd = model.state_dict()
    assert list(sd.keys()) == list(spec.keys()), "state-dict names/order differ from oracle/weights.py spec"
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(spec[k]), (k, v.shape, spec[k])
    model.load_state_dict(to_t(wts), strict=True)
    # the reference leaves the direction U-Net in train mode and the restorer in eval mode
    assert model.pc_wrapper.training and not model.pretrained_restoration_model.training

    out = {"masked_spec": masked, "mask_frames": mask, "clean_spec": clean}
    meta = {"config": c, "n_tensors": len(spec), "n_elems": int(sum(np.prod(s) for s in spec.values()))}
    tm, tk, tc = torch.from_numpy(masked), torch.from_numpy(mask), torch.from_numpy(clean)

    cn, mk, mn, mean, std = ref_utils.preprocess_data(tc, tm, tk, plot_mean_std=True)
    out["clean_norm"], out["masked_norm"] = cn.numpy(), mn.numpy()
    meta["mean"], meta["std"] = float(mean), float(std)
    out["pred_norm"] = model.get_pred_spec_mag_norm(mn, mk).numpy()

    # intermediates of the direction U-Net (first train-mode forward)
    caps = {}
    net = model.pc_wrapper.net

    def cap(tag):
        def f(mod, inp, outp):            # must return None: a hook's return value replaces the output
            if tag not in caps:
                caps[tag] = outp.detach().clone()
        return f

    hs = [net.inc.conv.conv[0].register_forward_hook(cap("inc.raw0")), net.inc.register_forward_hook(cap("x1")),
          net.down4.register_forward_hook(cap("x5")), net.up1.register_forward_hook(cap("u1")),
          net.up4.register_forward_hook(cap("u4")), net.outc.register_forward_hook(cap("pc_raw"))]

    params = dict(model.named_parameters())
    objs = {}
    for tag, st in (("g0", 0), ("g500", 500)):
        model.load_state_dict(to_t(wts), strict=True)
        tr.optimizer = torch.optim.Adam(model.parameters(), lr=1e-4, betas=(0.5, 0.999))
        tr.step = st
        _, obj, log = tr.base_step((tm, tk, tc))
        tr.optimizer.zero_grad()
        obj.backward()
        meta[tag + ".objective"] = float(obj)
        if tag == "g0":
            for h in hs:
                h.remove()
            for k, v in caps.items():
                a = v.numpy()
                out["tap." + k] = a if a.size <= 300000 else a[:, : min(a.shape[1], 8)]
                meta["tap." + k + ".shape"] = list(a.shape)
                meta["tap." + k + ".abs_sum"] = float(np.abs(a.astype(np.float64)).sum())
            for k in ("w_mat", "err_norm", "err_proj", "w_norms", "reconst_err", "second_moment_mse", "objective"):
                out["log." + k] = log[k].numpy()
            for s2 in (0, 250, 375, 500):
                tr.step = s2
                objs[str(s2)] = float(tr._calculate_final_objective(log["reconst_err"], log["second_moment_mse"]))
            tr.step = st
            meta["objective_at_step"] = objs
            bn = model.state_dict()
            for n in STAT_NAMES:
                out["bn1." + n] = small(bn[n].numpy())
        gn = {}
        for n in GRAD_NAMES:
            g = params[n].grad.numpy()
            out[f"{tag}.grad.{n}"] = small(g)
            gn[n] = [float(np.abs(g).max()), float(np.sqrt((g.astype(np.float64) ** 2).sum()))]
        meta[tag + ".grad_absmax_l2"] = gn
        tot = sum(float((p.grad.double() ** 2).sum()) for p in params.values() if p.grad is not None)
        meta[tag + ".grad_total_l2"] = float(np.sqrt(tot))
        meta[tag + ".n_params_with_grad"] = int(sum(p.grad is not None for p in params.values()))
        if tag == "g500":
            tn = torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0)
            meta["clip.total_norm"] = float(tn)
            tr.optimizer.step()
            for n in GRAD_NAMES[::2]:
                out[f"adam1.{n}"] = small(params[n].detach().numpy())
            tr.step = 501
            _, obj2, _ = tr.base_step((tm, tk, tc))
            tr.optimizer.zero_grad()
            obj2.backward()
            torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0)
            tr.optimizer.step()
            meta["adam1.objective_next"] = float(obj2)
            for n in GRAD_NAMES[::2]:
                out[f"adam2.{n}"] = small(params[n].detach().numpy())

    rng = np.random.Generator(np.random.PCG64(78))
    xg = rng.standard_normal((3, 4, 7, 11)).astype(np.float32)
    xg[1, 2] = xg[1, 0] * 0.7 + 1e-3 * xg[1, 2]
    out["gs.in"] = xg
    out["gs.out"] = gram_schmidt_to_spec_mag(torch.from_numpy(xg)).numpy()

    np.savez_compressed(os.path.join(out_dir, name + ".npz"), **out)
    with open(os.path.join(out_dir, name + ".json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    sz = os.path.getsize(os.path.join(out_dir, name + ".npz")) / 1e6
    print(f"[{name}] wrote {len(out)} arrays, {sz:.2f} MB; objective@0={objs['0']:.6f} "
          f"grad_l2@500={meta['g500.grad_total_l2']:.4g}")


CONFIGS = {
    # small spectrogram, odd sizes (floor-mode pooling, pad-to-skip on both axes); the U-Net widths are fixed
    "inp_tiny": dict(nfft=63, hop=32, T=37, K=3, B=3, seed=41),
    # reference yaml shape (nfft 255 / hop 128 -> F=128, K=5) at reduced batch and length
    "inp_c3s": dict(nfft=255, hop=128, T=101, K=5, B=2, seed=42),
}


def main():
    install_placeholders()
    sys.path.insert(0, REF)
    torch.set_num_threads(8)
    for name in (sys.argv[1:] or list(CONFIGS)):
        run_config(name, CONFIGS[name], HERE)


if __name__ == "__main__":
    main()
