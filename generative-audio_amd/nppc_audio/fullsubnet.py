"""FullSubNet+ (frozen restorer) on the MI355X kernels.

Mirrors the public surface of the reference model
(FullSubNet_plus/speech_enhance/fullsubnet_plus/model/fullsubnet_plus.py:18-230): the pydantic
`FullSubNetPlusConfig`, the module tree (so state_dict() has the reference's 340 names/shapes and
checkpoints load strictly) and `forward(noisy_mag, noisy_real, noisy_imag) -> [B, 2, F', T]`.
The torch.nn sub-modules below are PARAMETER CONTAINERS only: nothing here calls their forward;
all arithmetic runs in libnppc_hip.so through `engine.FSNEngine`.
"""
from typing import List, Optional

import pydantic
import torch
import torch.nn as nn

from . import _hip as H
from .engine import FSNEngine, FlatParams, TCN_DILATIONS, TCN_HIDDEN


class FullSubNetPlusConfig(pydantic.BaseModel):
    num_freqs: int = 257
    look_ahead: int = 2
    sequence_model: str = "LSTM"
    sb_num_neighbors: int = 15
    fb_num_neighbors: int = 0
    fb_output_activate_function: str = "ReLU"
    sb_output_activate_function: bool = False
    fb_model_hidden_size: int = 512
    sb_model_hidden_size: int = 384
    channel_attention_model: str = "TSSE"
    norm_type: str = "offline_laplace_norm"
    num_groups_in_drop_band: int = 1
    output_size: int = 2
    subband_num: int = 1
    kersize: List[int] = pydantic.Field(default_factory=lambda: [3, 5, 10])
    weight_init: bool = False

    @pydantic.field_validator("kersize", mode="before")
    @classmethod
    def _listify_kersize(cls, v):
        if isinstance(v, (list, tuple)) or type(v).__name__ == "ListConfig":
            return list(v)
        raise ValueError(f"kersize must be a list of integers, got {type(v).__name__}")

    # NPPC-audio (MI355X build) extension: arithmetic of the HIP path.  "bf16" = bf16 MFMA operands and
    # saved activations with fp32 accumulation/state; "fp32" = exact-f32 MFMA everywhere (parity mode).
    precision: str = "bf16"


class _TSSE(nn.Module):
    """parameter container: ChannelTimeSenseSELayer (attention_model.py:43-76)"""

    def __init__(self, c, kersize):
        super().__init__()
        self.smallConv1d = nn.Sequential(nn.Conv1d(c, c, kersize[0], groups=c), nn.AdaptiveAvgPool1d(1), nn.ReLU())
        self.middleConv1d = nn.Sequential(nn.Conv1d(c, c, kersize[1], groups=c), nn.AdaptiveAvgPool1d(1), nn.ReLU())
        self.largeConv1d = nn.Sequential(nn.Conv1d(c, c, kersize[2], groups=c), nn.AdaptiveAvgPool1d(1), nn.ReLU())
        self.feature_concate_fc = nn.Linear(3, 1, bias=True)
        self.fc1 = nn.Linear(c, c // 2, bias=True)
        self.fc2 = nn.Linear(c // 2, c, bias=True)


class _TCNBlock(nn.Module):
    """parameter container: TCNBlock (causal_conv.py:67-81)"""

    def __init__(self, c, dilation, hidden=TCN_HIDDEN):
        super().__init__()
        self.conv1x1 = nn.Conv1d(c, hidden, 1)
        self.prelu1 = nn.PReLU()
        self.norm1 = nn.GroupNorm(1, hidden, eps=1e-8)
        self.depthwise_conv = nn.Conv1d(hidden, hidden, 3, groups=hidden, padding=dilation, dilation=dilation)
        self.prelu2 = nn.PReLU()
        self.norm2 = nn.GroupNorm(1, hidden, eps=1e-8)
        self.sconv = nn.Conv1d(hidden, c, 1)


class _SeqModel(nn.Module):
    """parameter container: SequenceModel (sequence_model.py:5-96), "TCN" or "LSTM" flavour"""

    def __init__(self, input_size, output_size, hidden_size, kind):
        super().__init__()
        if kind == "TCN":
            self.sequence_model = nn.Sequential(*[_TCNBlock(input_size, d) for d in TCN_DILATIONS], nn.ReLU())
            self.fc_output_layer = nn.Linear(input_size, output_size)
        elif kind == "LSTM":
            self.sequence_model = nn.LSTM(input_size=input_size, hidden_size=hidden_size, num_layers=2,
                                          batch_first=True, bidirectional=False)
            self.fc_output_layer = nn.Linear(hidden_size, output_size)
        else:
            raise NotImplementedError(f"Not implemented {kind}")


def _check_supported(cfg):
    if cfg.sequence_model != "LSTM":
        raise NotImplementedError("the MI355X sub-band kernel implements sequence_model='LSTM' (the reference yaml)")
    if cfg.channel_attention_model != "TSSE":
        raise NotImplementedError(f"Not implemented channel attention model {cfg.channel_attention_model}")
    if cfg.norm_type != "offline_laplace_norm":
        raise NotImplementedError("You must set up a type of Norm: only offline_laplace_norm is built for MI355X")
    if cfg.subband_num != 1 or cfg.fb_num_neighbors != 0:
        raise NotImplementedError("subband_num=1 and fb_num_neighbors=0 are the configurations built for MI355X")
    if cfg.fb_output_activate_function != "ReLU" or cfg.sb_output_activate_function:
        raise NotImplementedError("fb ReLU / no sub-band output activation are the configurations built for MI355X")
    if cfg.precision not in ("bf16", "fp32"):
        raise ValueError("precision must be 'bf16' or 'fp32'")


class FullSubNet_Plus(nn.Module):
    n_maps = 1

    def __init__(self, config: Optional[FullSubNetPlusConfig] = None):
        super().__init__()
        if config is None:
            config = FullSubNetPlusConfig()
        _check_supported(config)
        self.config = config
        self.num_freqs = config.num_freqs
        self.look_ahead = config.look_ahead
        self.sb_num_neighbors = config.sb_num_neighbors
        self.fb_num_neighbors = config.fb_num_neighbors
        self.num_groups_in_drop_band = config.num_groups_in_drop_band
        self.output_size = config.output_size
        self.kersize = list(config.kersize)
        F = self.num_freqs
        # same construction order as the reference -> same RNG stream -> same default init under a seed
        self.channel_attention = _TSSE(F, self.kersize)
        self.channel_attention_real = _TSSE(F, self.kersize)
        self.channel_attention_imag = _TSSE(F, self.kersize)
        self.fb_model = _SeqModel(F, F, config.fb_model_hidden_size, "TCN")
        self.fb_model_real = _SeqModel(F, F, config.fb_model_hidden_size, "TCN")
        self.fb_model_imag = _SeqModel(F, F, config.fb_model_hidden_size, "TCN")
        self.sb_model = _SeqModel((self.sb_num_neighbors * 2 + 1) + 3 * (self.fb_num_neighbors * 2 + 1),
                                  self.output_size, config.sb_model_hidden_size, config.sequence_model)
        self._engine = None
        self._flat = None

    # -- HIP engine plumbing ---------------------------------------------------------------------
    def _prec(self):
        return H.PREC_BF16 if self.config.precision == "bf16" else H.PREC_F32

    def engine(self):
        H.require_gpu()
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise RuntimeError("FullSubNet_Plus (MI355X build) runs on a HIP device only; call .to('cuda') first")
        if self._engine is None or self._flat.flat.device != dev or not self._flat_is_live():
            self._flat = FlatParams(self, dev)
            self._engine = FSNEngine(self._flat, num_freqs=self.num_freqs, n_maps=self.n_maps,
                                     out_size=self.output_size, sb_neighbors=self.sb_num_neighbors,
                                     look_ahead=self.look_ahead, sb_hidden=self.config.sb_model_hidden_size,
                                     groups=self.num_groups_in_drop_band, kersize=self.kersize, prec=self._prec(),
                                     trainable=any(p.requires_grad for p in self.parameters()))
        return self._engine

    def _flat_is_live(self):
        # .to()/.float()/load with assign can re-home parameters; detect that and re-flatten
        for n, p in self.named_parameters():
            o, _ = self._flat.off[n]
            if p.data_ptr() != self._flat.flat.data_ptr() + 4 * o:
                return False
        return True

    def forward(self, noisy_mag, noisy_real, noisy_imag):
        """[B,1,F,T] x3 -> compressed cIRM [B,2,F',T] (fullsubnet_plus.py:143-230).  Inference only:
        the NPPC path always runs this net frozen under no_grad (nppc_model.py:94-95)."""
        assert noisy_mag.dim() == 4
        assert noisy_mag.shape[1] == 1, f"{self.__class__.__name__} takes the mag feature as inputs."
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise RuntimeError("FullSubNet_Plus (MI355X build) is forward-only; wrap the call in torch.no_grad() "
                               "(training this net is outside the NPPC hot path)")
        return self.engine().forward([noisy_mag, noisy_real, noisy_imag], train=False)
