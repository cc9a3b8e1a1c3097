"""Reference surface nppc_audio/nppc_model.py:13-132 on the HIP kernels.

NPPCModel = frozen FullSubNet+ restorer + AudioPCWrapper.  The reference executes the restorer twice
and the noisy STFT three times per train step (nppc_model.py:88,95,122,129; trainer.py:354); both are
deterministic, so this build memoises them per input tensor: `forward` stores the compressed cIRM and
`get_pred_crm` on the same waveform returns it without re-running the kernels.
"""
from pathlib import Path
from typing import Literal, Union

import pydantic
import torch
import torch.nn as nn

from . import _hip as H
from . import ops
from .fullsubnet import FullSubNet_Plus, FullSubNetPlusConfig
from .pc_wrapper import AudioPCWrapper, AudioPCWrapperConfig


class StftConfig(pydantic.BaseModel):      # utils.py:14-17
    nfft: int = 512
    hop_length: int = 256
    win_length: int = 512


def preload_model(model_path: Union[Path, str], model: FullSubNet_Plus) -> FullSubNet_Plus:
    """utils.py:82-98: '*.tar' checkpoint = {"model": state_dict}, loaded strict=False."""
    model_path = Path(model_path).expanduser().absolute()
    assert model_path.exists(), f"The file {model_path.as_posix()} is not exist. please check path."
    ck = torch.load(model_path.as_posix(), map_location="cpu")
    model.load_state_dict(ck["model"], strict=False)
    return model


def load_pretrained_model(model_path, model_config: FullSubNetPlusConfig) -> FullSubNet_Plus:
    return preload_model(model_path, FullSubNet_Plus(model_config))


class NPPCModelConfig(pydantic.BaseModel):
    pretrained_restoration_model_configuration: FullSubNetPlusConfig
    pretrained_restoration_model_path: str
    audio_pc_wrapper_configuration: AudioPCWrapperConfig
    stft_configuration: StftConfig
    device: Literal['cpu', 'cuda'] = 'cuda'

    def make_instance(self):
        return NPPCModel(self)


class NPPCModel(nn.Module):
    def __init__(self, config: NPPCModelConfig):
        super().__init__()
        self.config = config
        self.pretrained_restoration_model = load_pretrained_model(config.pretrained_restoration_model_path,
                                                                  config.pretrained_restoration_model_configuration)
        self.device = config.device
        if config.device == 'cuda':
            self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.pretrained_restoration_model.to(self.device)
        self.pretrained_restoration_model.eval()
        self.audio_pc_wrapper = AudioPCWrapper(config.audio_pc_wrapper_configuration)
        self.audio_pc_wrapper.to(self.device)
        self._memo = None

    # -- shared front end ------------------------------------------------------------------------
    def _front(self, noisy_waveform, reuse=True):
        """STFT + frozen restorer.  `forward` always recomputes (reuse=False) and leaves the result for the
        `get_pred_crm` / gt-mask calls that follow on the SAME tensor within the step."""
        H.require_gpu()
        # the memo holds the input tensor itself: identity + version is then a safe key (an address alone could be
        # recycled by the caching allocator for the next batch)
        m = self._memo
        if reuse and m is not None and m[0] is noisy_waveform and m[1] == noisy_waveform._version:
            return m[2]
        st = self.config.stft_configuration
        if st.win_length != st.nfft:
            raise NotImplementedError("win_length == nfft is the STFT configuration built for MI355X")
        mag, re, im = ops.stft(noisy_waveform, st.nfft, st.hop_length)
        with torch.no_grad():
            pred_crm = self.pretrained_restoration_model(mag[:, None], re[:, None], im[:, None])
        out = dict(mag=mag, re=re, im=im, pred_crm=pred_crm)
        self._memo = (noisy_waveform, noisy_waveform._version, out)
        return out

    def forward(self, noisy_waveform: torch.Tensor) -> torch.Tensor:
        """[B, L] -> w_mat [B, n_dirs, 2, F', T] (nppc_model.py:58-115)"""
        f = self._front(noisy_waveform, reuse=False)
        # decompress + the reference's swapped mask application (conj(mask) * noisy), utils.py:241-249
        _, emag, ere, eim = ops.cirm_decompress_apply_conj(f["pred_crm"], f["re"], f["im"])
        return self.audio_pc_wrapper(f["mag"][:, None], f["re"][:, None], f["im"][:, None],
                                     emag[:, None], ere[:, None], eim[:, None])

    def get_pred_crm(self, noisy_waveform: torch.Tensor) -> torch.Tensor:
        """compressed, un-drop-banded restorer output [B,2,F,T] (nppc_model.py:117-132)"""
        return self._front(noisy_waveform)["pred_crm"]
