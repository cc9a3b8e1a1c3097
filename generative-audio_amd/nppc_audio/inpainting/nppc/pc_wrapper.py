"""Direction-net wrapper of the inpainting path: mirrors nppc_audio/inpainting/nppc/pc_wrapper.py:43-87."""
import pydantic
import torch
import torch.nn as nn

from ... import _hip as H
from ...pc_ops import gram_schmidt_to_spec_mag  # noqa: F401  (same free function name as the reference)
from ..networks.unet import UNet, UNetConfig, _UNetFn


class AudioInpaintingPCWrapperConfig(pydantic.BaseModel):
    model_configuration: UNetConfig
    n_dirs: int


class AudioInpaintingPCWrapper(nn.Module):
    def __init__(self, pc_wrapper_config: AudioInpaintingPCWrapperConfig):
        super().__init__()
        self.config = pc_wrapper_config
        self.net = UNet(self.config.model_configuration)

    def forward(self, mag_spec: torch.Tensor, mask: torch.Tensor):
        """mag_spec [B,2,F,T] (masked ++ restored), mask [B,1,F,T] (1 = known) -> w_mat [B,n_dirs,F,T]:
        U-Net directions, zeroed on the known frames, Gram-Schmidt (pc_wrapper.py:75-87)."""
        H.require_gpu()
        x = mag_spec.contiguous().float()
        B, C, F, T = x.shape
        m = mask[:, 0, 0, :].contiguous().float()
        maps = [x[:, c].contiguous() for c in range(C)]
        net = self.net
        if torch.is_grad_enabled() and net.training and any(p.requires_grad for p in net.parameters()):
            raw = _UNetFn.apply(net, (B, F, T), m, maps, *[p for _, p in net.named_parameters()])
        else:
            raw = torch.empty(B, net.config.out_channels, F, T, dtype=torch.float32, device=x.device)
            net.run(maps, F * T, (B, F, T), m, raw, F * T)
        return gram_schmidt_to_spec_mag(raw)
