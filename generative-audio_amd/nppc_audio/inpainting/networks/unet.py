"""U-Net restorer / direction net of the inpainting path on the MI355X kernels.

Mirrors nppc_audio/inpainting/networks/unet.py:131-142 (UNetConfig), :247-290 (UNet) and :293-313
(RestorationWrapper) plus tmp_utils.py:8-99: same constructor arguments, module tree and state-dict
names (conv / BatchNorm parameters AND running buffers), so reference checkpoints load unchanged.
The modules only hold parameters; every forward / backward runs in libnppc_hip.so through
`unet_engine.UNetEngine`.
"""
import pydantic
import torch
import torch.nn as nn

from ... import _hip as H
from ...unet_engine import UNetEngine  # noqa: F401  (re-exported for tests)


class UNetConfig(pydantic.BaseModel):
    in_channels: int = 1
    out_channels: int = 1
    dropout: float = 0.0
    precision: str = "bf16"      # MI355X build only: "bf16" operands / fp32 accumulate, or "fp32" exact-f32 MFMA


def _double_conv(in_ch, out_ch, dropout):
    # tmp_utils.double_conv: Sequential indices 0 conv, 1 BN, 2 act, 3 conv, 4 BN, 5 act(, 6 dropout);
    # only the parameterised entries (and the nn.Dropout flag-holder utils.enable_dropout looks for) exist here,
    # under the same indices
    d = nn.ModuleDict({"0": nn.Conv2d(in_ch, out_ch, 3, padding=1), "1": nn.BatchNorm2d(out_ch),
                       "3": nn.Conv2d(out_ch, out_ch, 3, padding=1), "4": nn.BatchNorm2d(out_ch)})
    if dropout:
        d["6"] = nn.Dropout(dropout)
    d.dropout_p = float(dropout)
    return d


class _Holder(nn.Module):
    """parameter container reproducing one level of the reference's attribute path"""

    def __init__(self, **children):
        super().__init__()
        for k, v in children.items():
            self.add_module(k, v)


def _inconv(i, o):
    return _Holder(conv=_Holder(conv=_double_conv(i, o, 0)))


def _down(i, o, dropout=0):
    h = _Holder()
    h.mpconv = nn.ModuleDict({"1": _Holder(conv=_double_conv(i, o, dropout))})
    return h


def _up(i, o, dropout=0):
    return _Holder(conv=_Holder(conv=_double_conv(i, o, dropout)))


class UNet(nn.Module):
    def __init__(self, config: UNetConfig):
        super().__init__()
        if config.precision not in ("bf16", "fp32"):
            raise ValueError("precision must be 'bf16' or 'fp32'")
        self.config = config
        d = config.dropout
        self.inc = _inconv(config.in_channels, 64)
        self.down1 = _down(64, 128)
        self.down2 = _down(128, 256)
        self.down3 = _down(256, 512, d)
        self.down4 = _down(512, 512, d)
        self.up1 = _up(1024, 256, d)
        self.up2 = _up(512, 128, d)
        self.up3 = _up(256, 64)
        self.up4 = _up(128, 64)
        self.outc = _Holder(conv=nn.Conv2d(64, config.out_channels, 1))
        self._engine = None
        self.flat_grad_only = False
        self.dropout_seed = int(torch.initial_seed()) & 0x7FFFFFFFFFFFFFFF   # MC-dropout stream (Philox key)
        self.dropout_pass = 0                                                # advanced by every stochastic forward
        self.dropout_tap = None                                              # tests: {} collects the keep bits

    def _prec(self):
        return H.PREC_BF16 if self.config.precision == "bf16" else H.PREC_F32

    def engine(self):
        H.require_gpu()
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("UNet (MI355X build) runs on a HIP device only; call .to('cuda') first")
        e = self._engine
        if e is None or e.dev != dev or not self._flat_is_live():
            e = self._engine = UNetEngine(self, self.config.in_channels, self.config.out_channels, self._prec(),
                                          trainable=any(p.requires_grad for p in self.parameters()))
        return e

    def _flat_is_live(self):
        fp = self._engine.fp
        for n, p in self.named_parameters():
            if p.data_ptr() != fp.flat.data_ptr() + 4 * fp.off[n][0]:
                return False
        return all(b.device == self._engine.dev for b in self.buffers())

    def _dropout_state(self):
        """None, or the Philox stream of this stochastic pass when any nn.Dropout of the net is in train mode
        (utils.enable_dropout, utils.py:334-338; restoration_model.train() in base_step2, nppc_trainer.py:268)."""
        p = self.config.dropout
        if not p or not any(isinstance(m, nn.Dropout) and m.training for m in self.modules()):
            return None
        if torch.is_grad_enabled() and any(q.requires_grad for q in self.parameters()):
            raise NotImplementedError(
                "UNet (MI355X build): dropout is implemented for MC-dropout inference (no_grad / frozen weights); the "
                "training path configures the direction U-Net with dropout 0 (config_nppc.yaml:33-36)")
        st = dict(p=p, seed=self.dropout_seed, pass_id=self.dropout_pass, tap=self.dropout_tap)
        self.dropout_pass += 1
        return st

    def run(self, maps, map_bstride, shape, mask, out, out_pstride, xin=None, xin_bstride=0):
        """engine entry used by the wrappers: see UNetEngine.forward"""
        drop = self._dropout_state()
        train = self.training
        self.engine().forward(shape, maps, map_bstride, mask, out, out_pstride, xin, xin_bstride, train=train, dropout=drop)

    def forward(self, x):
        """[B, in_channels, F, T] -> [B, out_channels, F, T] (unet.py:277-290)"""
        H.require_gpu()
        x = x.contiguous().float()
        B, C, F, T = x.shape
        assert C == self.config.in_channels
        ones = torch.zeros(B, T, dtype=torch.float32, device=x.device)          # mask 0 -> plain U-Net output
        maps = [x[:, c].contiguous() for c in range(C)]
        if torch.is_grad_enabled() and self.training and any(p.requires_grad for p in self.parameters()):
            return _UNetFn.apply(self, (B, F, T), ones, maps, *[p for _, p in self.named_parameters()])
        out = torch.empty(B, self.config.out_channels, F, T, dtype=torch.float32, device=x.device)
        self.run(maps, F * T, (B, F, T), ones, out, F * T)
        return out


class _UNetFn(torch.autograd.Function):
    """autograd sees ONE function whose differentiable inputs are the net's parameters"""

    @staticmethod
    def forward(ctx, net, shape, mask, maps, *params):
        B, F, T = shape
        K = net.config.out_channels
        out = torch.empty(B, K, F, T, dtype=torch.float32, device=mask.device)
        net.run(maps, F * T, shape, mask, out, F * T)
        ctx.net = net
        return out

    @staticmethod
    def backward(ctx, dout):
        net = ctx.net
        eng = net.engine()
        d = dout.contiguous().float()
        eng.backward(d, d.shape[2] * d.shape[3])
        if net.flat_grad_only:
            return (None, None, None, None, *([None] * len(eng.fp.named)))
        grads = [eng.fp.gview(n) if p.requires_grad else None for n, p in net.named_parameters()]
        return (None, None, None, None, *grads)


class RestorationWrapper(nn.Module):
    def __init__(self, base_net: UNet):
        super().__init__()
        self.net = base_net

    def forward(self, x_in: torch.Tensor, mask: torch.Tensor):
        """x_in [B,1,F,T] (single-channel input, the NPPC path's case), mask [B,1,F,T] (1 = known) ->
        x_in * mask + net(x_in) * (1 - mask)   (unet.py:299-312).  Inference only."""
        H.require_gpu()
        if x_in.shape[1] != 1 or self.net.config.out_channels != 1:
            raise NotImplementedError("RestorationWrapper (MI355X build) implements the single-channel case of the NPPC path")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise RuntimeError("RestorationWrapper (MI355X build) is forward-only; wrap the call in torch.no_grad()")
        x = x_in.contiguous().float()
        B, _, F, T = x.shape
        m = mask[:, 0, 0, :].contiguous().float()
        out = torch.empty(B, 1, F, T, dtype=torch.float32, device=x.device)
        self.net.run([x], F * T, (B, F, T), m, out, F * T, xin=x, xin_bstride=F * T)
        return out
