#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE (CPU, fp32).

Runs only in the build container, where /root/reference exists; the fixtures it writes
(inputs + expected outputs, .npz) are what travels.  No reference source is copied: the
reference is imported from where it lies.  Seven third-party packages the reference imports
but never uses on this path are absent offline (omegaconf, librosa, torchaudio, wandb,
soundfile, torchvision, pesq/pystoi ...) and one in-repo file (speech_enhance/utils/logger.py)
is missing from the reference tree itself, so empty placeholder modules are registered for
them before the import.

Weights: oracle/weights.py (seeded numpy) -> written into the reference modules through
their own load_state_dict / utils.preload_model code paths.

Usage:  python tests/golden/make_goldens.py            # writes tests/golden/*.npz, *.json
"""
import json
import os
import sys
import tempfile
import types
import zlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)

from oracle import weights as W  # noqa: E402


def _placeholder(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__path__ = []
    sys.modules[name] = m
    return m


def install_placeholders():
    class _Dummy:
        def __init__(self, *a, **k):
            pass

    _placeholder("omegaconf", ListConfig=type("ListConfig", (), {}),
                 DictConfig=type("DictConfig", (), {}), OmegaConf=_Dummy)
    _placeholder("hydra", main=lambda *a, **k: (lambda f: f))
    for n in ("librosa", "torchaudio", "wandb", "soundfile", "line_profiler", "torchinfo",
              "torchvision", "torchvision.utils", "torchvision.transforms",
              "torchvision.datasets", "toml", "GPUtil", "colorful", "whisper", "seaborn"):
        _placeholder(n)
    sys.modules["torchinfo"].summary = lambda *a, **k: None
    sys.modules["line_profiler"].LineProfiler = _Dummy
    _placeholder("pesq", pesq=lambda *a, **k: 0.0)
    _placeholder("pystoi", stoi=lambda *a, **k: 0.0)
    _placeholder("pystoi.stoi", stoi=lambda *a, **k: 0.0)
    # file missing from the reference tree itself (imported by fullsubnet_plus.py:13)
    _placeholder("FullSubNet_plus.speech_enhance.utils.logger", log=print, init=lambda *a, **k: None)


def to_t(d):
    return {k: torch.from_numpy(v.copy()) for k, v in d.items()}


def crc(a):
    return int(zlib.crc32(np.ascontiguousarray(a).tobytes()))


class MemDataset(torch.utils.data.Dataset):
    """In-memory stand-in for the wav-folder dataset (needs files + torchaudio)."""
    data = None

    def __init__(self, cfg):
        pass

    def __len__(self):
        return self.data[0].shape[0]

    def __getitem__(self, i):
        return self.data[0][i], self.data[1][i]


def model_cfg_dict(c):
    common = dict(num_freqs=c["F"], sb_num_neighbors=c["sbn"], fb_num_neighbors=0, look_ahead=2,
                  sequence_model="LSTM", fb_model_hidden_size=c["fbh"], sb_model_hidden_size=c["sbh"],
                  kersize=[3, 5, 10])
    return dict(
        pretrained_restoration_model_configuration=dict(common, num_groups_in_drop_band=c["G_rest"]),
        audio_pc_wrapper_configuration=dict(multi_direction_configuration=dict(
            common, num_groups_in_drop_band=c["G_pc"], n_directions=c["K"])),
        stft_configuration=dict(nfft=c["nfft"], hop_length=c["hop"], win_length=c["nfft"]),
        device="cpu")


def run_config(name, c, out_dir):
    import utils as ref_utils
    from nppc_audio import trainer as ref_trainer
    from nppc_audio.pc_wrapper import gram_schmidt_to_crm
    from FullSubNet_plus.speech_enhance.audio_zen.acoustics.mask import decompress_cIRM
    from FullSubNet_plus.speech_enhance.audio_zen.acoustics.feature import drop_band

    torch.manual_seed(0)
    F, K, B, L = c["F"], c["K"], c["B"], c["L"]
    kw = dict(num_freqs=F, sb_neighbors=c["sbn"], sb_hidden=c["sbh"])
    spec = W.nppc_spec(K, **kw)
    wts = W.make_weights(spec, c["seed"])
    noisy, clean = W.synth_batch(B, L)
    if c.get("scale_wave"):
        noisy, clean = noisy * c["scale_wave"], clean * c["scale_wave"]

    # restorer checkpoint through the reference's own loader (utils.py:82-104)
    tmp = tempfile.mkdtemp()
    ck = os.path.join(tmp, "restorer.tar")
    pre = "pretrained_restoration_model."
    torch.save({"model": to_t({k[len(pre):]: v for k, v in wts.items() if k.startswith(pre)})}, ck)
    mc = model_cfg_dict(c)
    mc["pretrained_restoration_model_path"] = ck
    MemDataset.data = (torch.from_numpy(noisy), torch.from_numpy(clean))
    ref_trainer.AudioDataset = MemDataset
    cfg = ref_trainer.NPPCAudioTrainerConfig(
        nppc_model_configuration=mc,
        data_configuration=dict(data_path=".", dataset=dict(clean_path=".", noisy_path=".")),
        data_loader_configuration=dict(batch_size=B, num_workers=0, pin_memory=False, shuffle=False),
        optimizer_configuration=dict(type="Adam", args=dict(lr=1e-4, betas=[0.9, 0.999], eps=1e-8,
                                                           weight_decay=0)),
        device="cpu")
    tr = ref_trainer.NPPCAudioTrainer(cfg)
    model = tr.nppc_model
    sd = model.state_dict()
    assert list(sd.keys()) == list(spec.keys()), "state-dict names/order differ from oracle/weights.py spec"
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(spec[k]), (k, v.shape, spec[k])
    model.load_state_dict(to_t(wts), strict=True)

    lite = bool(c.get("lite"))
    meta = {"config": c, "n_tensors": len(spec), "n_elems": int(sum(np.prod(s) for s in spec.values()))}
    if lite:
        # long clips: the waveforms are regenerated by the tests (oracle/weights.synth_batch, checked by CRC) and maps
        # larger than 64 K elements are stored on a strided (bin, frame) lattice; the per-sample loss terms, which
        # depend on every element, stay complete
        out = {}
        meta["noisy_crc"], meta["clean_crc"] = crc(noisy), crc(clean)
        meta["lattice"] = [5, 7]
    else:
        out = {"noisy": noisy, "clean": clean}

    def keep(a):
        a = np.asarray(a)
        if lite and a.ndim >= 2 and a.size > 65536:
            return np.ascontiguousarray(a[..., ::5, ::7])
        return a

    # ---- intermediates via forward hooks -------------------------------------------------
    caps = {}

    def hook(tag, store_in=False):
        def f(mod, inp, outp):
            lst = caps.setdefault(tag, [])
            lst.append((inp[0].detach().clone() if store_in else None, outp.detach().clone()))
        return f

    rest = model.pretrained_restoration_model
    pc = model.audio_pc_wrapper.net
    hs = []
    for nm, net in (() if lite else (("rest", rest), ("pc", pc))):
        hs.append(net.channel_attention.register_forward_hook(hook(nm + ".att_mag", True)))
        hs.append(net.channel_attention_real.register_forward_hook(hook(nm + ".att_real", True)))
        hs.append(net.channel_attention_imag.register_forward_hook(hook(nm + ".att_imag", True)))
        hs.append(net.fb_model.register_forward_hook(hook(nm + ".fb_mag")))
        hs.append(net.fb_model_real.register_forward_hook(hook(nm + ".fb_real")))
        hs.append(net.fb_model_imag.register_forward_hook(hook(nm + ".fb_imag")))
        hs.append(net.fb_model.sequence_model[0].register_forward_hook(hook(nm + ".tcn0_mag")))
        hs.append(net.sb_model.register_forward_hook(hook(nm + ".sb", True)))
        hs.append(net.sb_model.sequence_model.register_forward_hook(
            lambda m, i, o, nm=nm: caps.setdefault(nm + ".lstm_out", []).append((None, o[0].detach().clone()))))

    tnoisy = torch.from_numpy(noisy)
    tclean = torch.from_numpy(clean)
    mag, re, im = ref_utils.prepare_input_from_waveform(tnoisy, c["nfft"], c["hop"], c["nfft"], "cpu")
    out["noisy_mag"], out["noisy_real"], out["noisy_imag"] = keep(mag.numpy()), keep(re.numpy()), keep(im.numpy())

    tr.step = 0
    reconst_err, objective, log = tr.base_step((tnoisy, tclean))
    for h in hs:
        h.remove()

    # restorer ran twice (forward + get_pred_crm) -> keep the first capture of each
    for tag, lst in caps.items():
        i0, o0 = lst[0]
        if tag.endswith(".sb"):
            a = i0.numpy()
            out[tag + ".in_head"] = a[: min(8, a.shape[0])]
            out[tag + ".in_tail"] = a[-min(8, a.shape[0]):]
            meta[tag + ".in_shape"] = list(a.shape)
            meta[tag + ".in_sum"] = float(a.astype(np.float64).sum())
            meta[tag + ".in_abs_sum"] = float(np.abs(a.astype(np.float64)).sum())
            o = o0.numpy()
            out[tag + ".out_head"] = o[: min(64, o.shape[0])]
            meta[tag + ".out_shape"] = list(o.shape)
        elif tag.endswith(".lstm_out"):
            o = o0.numpy()
            out[tag + "_head"] = o[: min(4, o.shape[0])]
        elif ".att_" in tag:
            out[tag + ".in"] = i0.numpy()
            out[tag + ".out"] = o0.numpy()
            if len(lst) > 1 and tag.startswith("pc."):
                out[tag + ".in_enh"] = lst[1][0].numpy()
                out[tag + ".out_enh"] = lst[1][1].numpy()
        else:
            out[tag] = o0.numpy()

    with torch.no_grad():
        pred_full = model.get_pred_crm(tnoisy)                     # compressed, not drop-banded
        out["pred_crm_full"] = keep(pred_full.numpy())
        dec = decompress_cIRM(pred_full.permute(0, 2, 3, 1))
        if not lite:
            out["pred_crm_decompressed"] = dec.numpy()
        emag, ere, eim = ref_utils.crm_to_stft_components(dec, re, im)
        out["enh_mag"], out["enh_real"], out["enh_imag"] = keep(emag.numpy()), keep(ere.numpy()), keep(eim.numpy())
        raw = model.audio_pc_wrapper.net(mag, re, im, emag.unsqueeze(1), ere.unsqueeze(1), eim.unsqueeze(1))
        out["pc_raw"] = keep(raw.numpy())
        gt, pred = tr._get_true_and_pred_crm(tclean, model, tnoisy, c["G_pc"])
        assert np.array_equal(log["pred_crm"].numpy(), pred.numpy())
        out["gt_crm"], out["pred_crm"] = keep(gt.numpy()), keep(pred.numpy())

    for k in ("w_mat", "err_norm", "err_proj_mag", "w_norms", "reconst_err", "second_moment_mse", "objective"):
        out["log." + k] = keep(log[k].numpy())
    out["log.err_proj_re"] = log["err_proj"].real.numpy()
    out["log.err_proj_im"] = log["err_proj"].imag.numpy()

    objs = {}
    for st in (0, 250, 375, 500):
        tr.step = st
        objs[str(st)] = float(tr._calculate_final_objective(log["reconst_err"], log["second_moment_mse"]))
    meta["objective_at_step"] = objs

    # ---- gradients + two Adam steps at lambda = 1 (step 500) and at step 0 ------------------
    grad_names = [
        "audio_pc_wrapper.net.channel_attention.smallConv1d.0.weight",
        "audio_pc_wrapper.net.channel_attention_real.largeConv1d.0.bias",
        "audio_pc_wrapper.net.channel_attention.feature_concate_fc.weight",
        "audio_pc_wrapper.net.channel_attention_imag.fc1.weight",
        "audio_pc_wrapper.net.channel_attention.fc2.bias",
        "audio_pc_wrapper.net.fb_model.sequence_model.0.conv1x1.weight",
        "audio_pc_wrapper.net.fb_model.sequence_model.0.conv1x1.bias",
        "audio_pc_wrapper.net.fb_model.sequence_model.3.prelu1.weight",
        "audio_pc_wrapper.net.fb_model_real.sequence_model.2.norm1.weight",
        "audio_pc_wrapper.net.fb_model_real.sequence_model.2.norm2.bias",
        "audio_pc_wrapper.net.fb_model_imag.sequence_model.7.depthwise_conv.weight",
        "audio_pc_wrapper.net.fb_model_imag.sequence_model.7.depthwise_conv.bias",
        "audio_pc_wrapper.net.fb_model.sequence_model.7.sconv.weight",
        "audio_pc_wrapper.net.fb_model.sequence_model.5.prelu2.weight",
        "audio_pc_wrapper.net.fb_model_real.fc_output_layer.weight",
        "audio_pc_wrapper.net.fb_model.fc_output_layer.bias",
        "audio_pc_wrapper.net.sb_model.sequence_model.weight_ih_l0",
        "audio_pc_wrapper.net.sb_model.sequence_model.weight_hh_l0",
        "audio_pc_wrapper.net.sb_model.sequence_model.bias_ih_l0",
        "audio_pc_wrapper.net.sb_model.sequence_model.weight_ih_l1",
        "audio_pc_wrapper.net.sb_model.sequence_model.weight_hh_l1",
        "audio_pc_wrapper.net.sb_model.sequence_model.bias_hh_l1",
        "audio_pc_wrapper.net.sb_model.fc_output_layer.weight",
        "audio_pc_wrapper.net.sb_model.fc_output_layer.bias",
    ]
    params = dict(model.named_parameters())

    def small(a, n=4096):
        a = a.reshape(-1)
        return a[:n].copy()

    for tag, st in (("g0", 0), ("g500", 500)):
        model.load_state_dict(to_t(wts), strict=True)
        tr.optimizer = torch.optim.Adam(model.parameters(), lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0)
        tr.step = st
        _, obj, _ = tr.base_step((tnoisy, tclean))
        tr.optimizer.zero_grad()
        obj.backward()
        meta[tag + ".objective"] = float(obj)
        gn = {}
        for n in grad_names:
            g = params[n].grad.numpy()
            out[f"{tag}.grad.{n}"] = small(g)
            gn[n] = [float(np.abs(g).max()), float(np.sqrt((g.astype(np.float64) ** 2).sum()))]
        meta[tag + ".grad_absmax_l2"] = gn
        tot = 0.0
        for n, p in params.items():
            if p.grad is not None:
                tot += float((p.grad.double() ** 2).sum())
        meta[tag + ".grad_total_l2"] = float(np.sqrt(tot))
        meta[tag + ".n_params_with_grad"] = int(sum(p.grad is not None for p in params.values()))
        if tag == "g500":
            tr.optimizer.step()
            for n in grad_names[::3]:
                out[f"adam1.{n}"] = small(params[n].detach().numpy())
            tr.step = 501
            _, obj2, _ = tr.base_step((tnoisy, tclean))
            tr.optimizer.zero_grad()
            obj2.backward()
            tr.optimizer.step()
            meta["adam1.objective_next"] = float(obj2)
            for n in grad_names[::3]:
                out[f"adam2.{n}"] = small(params[n].detach().numpy())

    # ---- standalone pieces ---------------------------------------------------------------------
    if not lite:
        rng = np.random.Generator(np.random.PCG64(77))
        xg = rng.standard_normal((3, 4, 2, 9, 11)).astype(np.float32)
        xg[1, 2] = xg[1, 0] * 0.7 + 1e-3 * xg[1, 2]            # nearly collinear case
        out["gs.in"] = xg
        out["gs.out"] = gram_schmidt_to_crm(torch.from_numpy(xg)).numpy()
        db_in = torch.arange(6 * 2 * 9 * 3, dtype=torch.float32).reshape(6, 2, 9, 3)
        out["dropband.in"] = db_in.numpy()
        for g in (2, 3):
            out[f"dropband.out{g}"] = drop_band(db_in, g).numpy()

    np.savez_compressed(os.path.join(out_dir, name + ".npz"), **out)
    with open(os.path.join(out_dir, name + ".json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    sz = os.path.getsize(os.path.join(out_dir, name + ".npz")) / 1e6
    print(f"[{name}] wrote {len(out)} arrays, {sz:.2f} MB; objective@0={objs['0']:.6f}")


CONFIGS = {
    # tiny nets: every code path in seconds (drop-band G=2 on the direction net)
    "g0_tiny": dict(F=33, nfft=64, hop=32, sbn=3, fbh=32, sbh=16, K=3, B=4, L=1024, G_rest=1, G_pc=2, seed=11),
    "g0_tiny_g1": dict(F=33, nfft=64, hop=32, sbn=3, fbh=32, sbh=16, K=3, B=4, L=1024, G_rest=1, G_pc=1, seed=12),
    # BASELINE config 1: full-size nets, B=2 x 1 s, K=2, no drop-band
    "g1_c1": dict(F=257, nfft=512, hop=256, sbn=15, fbh=512, sbh=384, K=2, B=2, L=16000, G_rest=1, G_pc=1, seed=21),
    # reference training yaml shape at reduced batch/length: K=5, direction net G=2
    "g2_k5": dict(F=257, nfft=512, hop=256, sbn=15, fbh=512, sbh=384, K=5, B=4, L=8192, G_rest=1, G_pc=2, seed=31),
    # BASELINE config 5's code paths at a size the reference finishes in minutes: 6 s clips (T' = 378 > 253), K = 8
    # (16 head outputs), odd batch 3 with G_pc = 2 (384 direction-net sequences: the small-N cooperative plans)
    "g3_long": dict(F=257, nfft=512, hop=256, sbn=15, fbh=512, sbh=384, K=8, B=3, L=96000, G_rest=1, G_pc=2, seed=41,
                    lite=True),
}


def main():
    install_placeholders()
    sys.path.insert(0, REF)
    torch.set_num_threads(8)
    which = sys.argv[1:] or list(CONFIGS)
    for name in which:
        run_config(name, CONFIGS[name], HERE)


if __name__ == "__main__":
    main()
