"""K-head principal-component direction net (trainable) on the MI355X kernels.

Mirrors nppc_audio/networks.py:9-163 of the reference: `MultiDirectionConfig`,
`MultiDirectionFullSubNet_Plus(config).forward(noisy_mag, noisy_real, noisy_imag, enhanced_mag,
enhanced_real, enhanced_imag) -> [B, 2K, F', T]`, same module tree / state-dict names.
Forward and backward both run in libnppc_hip.so (engine.FSNEngine); autograd sees ONE function
whose inputs are the net's parameters, so `objective.backward()` and any torch optimizer work.
"""
from typing import Optional

import torch

from . import _hip as H
from .fullsubnet import FullSubNet_Plus, FullSubNetPlusConfig, _SeqModel


class MultiDirectionConfig(FullSubNetPlusConfig):
    n_directions: int = 4  # Number of output CRMs for uncertainty

    def make_instance(self):
        return MultiDirectionFullSubNet_Plus(self)


class _DirectionNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, n_maps_tensors, *tensors):
        maps = tensors[:n_maps_tensors]
        eng = net.engine()
        out = eng.forward(list(maps), train=True)
        ctx.gen = eng.last_train["gen"]
        ctx.net = net
        ctx.n_maps_tensors = n_maps_tensors
        return out

    @staticmethod
    def backward(ctx, dout):
        eng = ctx.net.engine()
        eng.backward(dout.contiguous().float(), gen=ctx.gen)
        grads = [None] * ctx.n_maps_tensors
        if ctx.net.flat_grad_only:
            # the caller (NPPCAudioTrainer.train_step) consumes the flat gradient buffer directly: one all-reduce, one
            # Adam launch; handing 340 views to autograd would only make it copy each of them into .grad
            return (None, None, *grads, *([None] * len(eng.fp.named)))
        for n, p in ctx.net.named_parameters():
            grads.append(eng.fp.gview(n) if p.requires_grad else None)
        return (None, None, *grads)


class MultiDirectionFullSubNet_Plus(FullSubNet_Plus):
    n_maps = 2

    def __init__(self, config: Optional[MultiDirectionConfig] = None):
        if config is None:
            config = MultiDirectionConfig()
        config.output_size = 2 * config.n_directions   # networks.py:23 (mutates the config, like the reference)
        super().__init__(config)
        self.n_directions = config.n_directions
        self.flat_grad_only = False
        F = self.num_freqs
        # the full-band models are rebuilt for the concatenated (noisy ++ enhanced) input, networks.py:32-61
        self.fb_model = _SeqModel(2 * F, F, config.fb_model_hidden_size, "TCN")
        self.fb_model_real = _SeqModel(2 * F, F, config.fb_model_hidden_size, "TCN")
        self.fb_model_imag = _SeqModel(2 * F, F, config.fb_model_hidden_size, "TCN")

    def forward(self, noisy_mag, noisy_real, noisy_imag, enhanced_mag=None, enhanced_real=None, enhanced_imag=None):
        """six [B,1,F,T] maps -> [B, 2*n_directions, F', T]"""
        maps = [noisy_mag, noisy_real, noisy_imag, enhanced_mag, enhanced_real, enhanced_imag]
        if any(m is None for m in maps):
            raise TypeError("MultiDirectionFullSubNet_Plus needs the three enhanced maps (networks.py:83-85 pads them)")
        H.require_gpu()
        params = [p for _, p in self.named_parameters()]
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            return _DirectionNetFn.apply(self, 6, *[m.detach() for m in maps], *params)
        return self.engine().forward([m.detach() for m in maps], train=False)
