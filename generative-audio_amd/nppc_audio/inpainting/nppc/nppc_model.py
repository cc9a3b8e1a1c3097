"""Inpainting NPPC model: mirrors nppc_audio/inpainting/nppc/nppc_model.py:24-159 (local-checkpoint path;
the wandb artifact download, :52-98, needs the network and is out of scope -- pass
`pretrained_restoration_model_path`)."""
from pathlib import Path
from typing import Literal, Optional

import pydantic
import torch
import torch.nn as nn

from ..networks.unet import RestorationWrapper, UNet, UNetConfig
from .pc_wrapper import AudioInpaintingPCWrapper, AudioInpaintingPCWrapperConfig


class WandbConfig(pydantic.BaseModel):
    entity: str = "kfirc-tel-aviv-university"
    project: str = "generative-audio"
    artifact_name: str
    artifact_version: str = "latest"
    checkpoint_filename: str = "checkpoint_final.pt"


class NPPCModelConfig(pydantic.BaseModel):
    pretrained_restoration_model_configuration: UNetConfig
    pretrained_restoration_model_path: Optional[str] = None
    wandb_config: Optional[WandbConfig] = None
    audio_pc_wrapper_configuration: AudioInpaintingPCWrapperConfig
    device: Literal['cpu', 'cuda'] = 'cuda'


class NPPCModel(nn.Module):
    def __init__(self, config: NPPCModelConfig):
        super().__init__()
        self.config = config
        self.device = config.device
        if config.device == 'cuda':
            self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        if config.wandb_config:
            raise NotImplementedError("loading the restorer from a wandb artifact needs the network; "
                                      "download it and pass pretrained_restoration_model_path")
        elif config.pretrained_restoration_model_path:
            self._load_from_local()
        else:
            raise ValueError("Either wandb_config or pretrained_restoration_model_path must be provided")
        self.pc_wrapper = AudioInpaintingPCWrapper(self.config.audio_pc_wrapper_configuration)
        self.pc_wrapper.to(self.device)
        self._memo = None

    def _load_from_local(self):
        """Restorer weights from a local checkpoint file (the reference's local branch, nppc_model.py:100-117):
        the file holds {'model_state_dict': <UNet state dict>}, loaded strictly; the wrapped net is frozen in eval
        mode.  Any failure (missing file, missing key, shape mismatch) surfaces as a RuntimeError naming the path."""
        path = Path(self.config.pretrained_restoration_model_path).absolute()
        unet = UNet(self.config.pretrained_restoration_model_configuration)
        try:
            state = torch.load(path, map_location="cpu")["model_state_dict"]
            unet.load_state_dict(state)
        except Exception as err:
            raise RuntimeError(f"could not load the restoration U-Net from {path}: {err}") from err
        self.pretrained_restoration_model = RestorationWrapper(unet.to(self.device)).eval()
        print(f"restoration U-Net loaded from {path}")

    def train(self, mode: bool = True):
        # nn.Module.train() would flip the frozen restorer into train mode; the reference never calls it on the
        # model during training (its loop method shadows it), so the restorer stays in eval mode here too
        super().train(mode)
        self.pretrained_restoration_model.eval()
        return self

    def forward(self, masked_spec_mag_norm: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
        """[B,1,F,T] x 2 -> w_mat [B,n_dirs,F,T]   (nppc_model.py:119-145)"""
        pred = self.get_pred_spec_mag_norm(masked_spec_mag_norm, mask, reuse=False)
        return self.pc_wrapper(torch.cat((masked_spec_mag_norm, pred), dim=1), mask)

    def get_pred_spec_mag_norm(self, masked_spec_mag_log, mask, reuse=True):
        """frozen restorer under no_grad (nppc_model.py:147-159).  The reference evaluates it twice per step on the
        same tensors (forward, then base_step :358); `forward` always computes and leaves the result for a following
        call on the SAME tensor objects (the memo holds them, so identity + version is a safe key)."""
        m = self._memo
        if (reuse and m is not None and m[0] is masked_spec_mag_log and m[1] == masked_spec_mag_log._version
                and m[2] is mask and m[3] == mask._version):
            return m[4]
        with torch.no_grad():
            pred = self.pretrained_restoration_model(masked_spec_mag_log, mask)
        self._memo = (masked_spec_mag_log, masked_spec_mag_log._version, mask, mask._version, pred)
        return pred
