"""Reference surface nppc_audio/pc_wrapper.py:8-106 on the HIP kernels."""
import pydantic
import torch.nn as nn

from .networks import MultiDirectionConfig, MultiDirectionFullSubNet_Plus
from .pc_ops import gram_schmidt_to_crm  # noqa: F401  (re-exported: same free function as the reference)


class AudioPCWrapperConfig(pydantic.BaseModel):
    multi_direction_configuration: MultiDirectionConfig

    def make_instance(self):
        return AudioPCWrapper(self)


class AudioPCWrapper(nn.Module):
    def __init__(self, audio_pc_wrapper_config: AudioPCWrapperConfig):
        super().__init__()
        self.net = MultiDirectionFullSubNet_Plus(audio_pc_wrapper_config.multi_direction_configuration)
        self.n_dirs = self.net.n_directions

    def forward(self, noisy_mag, noisy_real, noisy_imag, enhanced_mag=None, enhanced_real=None, enhanced_imag=None):
        """six [B,1,F,T] maps -> orthogonalised directions w_mat [B, n_dirs, 2, F', T] (pc_wrapper.py:70-106)"""
        crm = self.net(noisy_mag, noisy_real, noisy_imag, enhanced_mag, enhanced_real, enhanced_imag)
        batch_size, _, freq_bins, time_steps = crm.shape
        crm = crm.reshape(batch_size, self.n_dirs, 2, freq_bins, time_steps)
        return gram_schmidt_to_crm(crm)
