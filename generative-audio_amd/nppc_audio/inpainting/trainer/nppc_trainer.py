"""Inpainting NPPC trainer on the MI355X kernels: mirrors nppc_audio/inpainting/trainer/nppc_trainer.py
(NPPCAudioInpaintingTrainerConfig :28-46, NPPCAudioInpaintingTrainer.__init__ :49-93, train :115-166,
base_step :338-385, save_checkpoint :604-618, _calculate_final_objective :680-687).

Differences, deliberately: the frozen restorer runs once per step (the reference runs it twice on the same input);
clip_grad_norm_ + Adam run as sum-of-squares + ONE fused kernel over the flat gradient with the clip coefficient
computed on the device (no host round trip); wandb logging, plotting, the MC-dropout variants (base_step2) and the
LibriSpeech/VAD dataset are outside the hot path.
"""
import os
from datetime import datetime
from pathlib import Path
from typing import List, Optional, Union

import pydantic
import torch
import torch.nn as nn
import torch.optim as optim

from ... import _hip as H
from ...data import DataLoaderConfig
from ...nppc_model import StftConfig
from ...pc_ops import NPPCLoss, planes, second_moment_weight
from ...trainer import FlatAdamStepper, HipAdam, LoopLoader, OptimizerConfig
from ..nppc.nppc_model import NPPCModel, NPPCModelConfig
from ..utils import preprocess_data


class AudioInpaintingConfig(pydantic.BaseModel):
    """dataset/audio_dataset_inpainting.py:60-83 (the fields; the LibriSpeech loader itself is out of scope)"""
    clean_path: Union[str, Path]
    sample_rate: int = 16000
    missing_length_seconds: float = 0.128
    missing_start_seconds: Optional[float] = None
    missing_end_seconds: Optional[float] = None
    sub_sample_length_seconds: float = 3.0
    target_dB_FS: float = -25.0
    target_dB_FS_floating_value: float = 0.0
    stft_configuration: StftConfig
    use_vad: bool = False
    seed: Optional[int] = None
    is_random_sub_sample: bool = True


class NPPCAudioInpaintingTrainerConfig(pydantic.BaseModel):
    nppc_model_configuration: NPPCModelConfig
    data_configuration: AudioInpaintingConfig
    dataloader_configuration: DataLoaderConfig
    optimizer_configuration: OptimizerConfig
    device: str = "cuda"
    save_interval: int = 10
    log_interval: int = 100
    second_moment_loss_lambda: float = 1.0
    second_moment_loss_grace: int = 500
    max_grad_norm: float = 1.0
    use_wandb: bool = False
    wandb_project_name: Optional[str] = "generative-audio"
    wandb_run_name: Optional[str] = None
    wandb_tags: Optional[List[str]] = None
    wandb_artifact_name: str = "nppc_inpainting_model"


def inpainting_base_step(model, batch, step, grace, lam_cfg):
    """nppc_trainer.py:338-385: (masked_spec [B,2,F,T], mask [B,T], clean_spec [B,2,F,T]) ->
    (reconst_err [B], objective [], log)."""
    masked_spec, mask, clean_spec = batch
    clean_norm, mask4, masked_norm = preprocess_data(clean_spec, masked_spec, mask)
    w_mat = model(masked_norm, mask4)                                   # [B, n_dirs, F, T]
    pred = model.get_pred_spec_mag_norm(masked_norm, mask4)             # memoised: the restorer ran inside model()
    B, K, F, T = w_mat.shape
    lam = second_moment_weight(step, grace, lam_cfg)
    # real vectors = complex vectors with a zero imaginary plane; eps 1e-6 inside the norms (:355,:363)
    reconst_err, objective, err_norm, pr, _, _, w_norms, sm = NPPCLoss.apply(
        planes(w_mat), planes(clean_norm).view(B, 2, F, T), planes(pred).view(B, 2, F, T), lam, 1e-6, 1)
    log = {
        'w_mat': w_mat.detach(),
        'err_norm': err_norm.detach(),
        'err_proj': pr.detach(),
        'w_norms': w_norms.detach(),
        'reconst_err': reconst_err.detach(),
        'second_moment_mse': sm.detach(),
        'objective': objective.detach(),
    }
    return reconst_err, objective, log


class NPPCAudioInpaintingTrainer(nn.Module):
    def __init__(self, config: NPPCAudioInpaintingTrainerConfig, dataset=None):
        super().__init__()
        self.config = config
        if config.use_wandb:
            raise NotImplementedError("wandb logging is outside the MI355X hot path build (no network)")
        self.nppc_model = NPPCModel(self.config.nppc_model_configuration)
        self.device = self.config.device
        if dataset is None:
            raise ValueError("pass a dataset yielding (stft_masked [2,F,T], mask_frames [T], stft_clean [2,F,T]) items; "
                             "the LibriSpeech/VAD loader of the reference is outside the hot path")
        print(f"Total sample pairs in dataset: {len(dataset)}")
        dl = config.dataloader_configuration
        self.dataloader = torch.utils.data.DataLoader(dataset, batch_size=dl.batch_size, shuffle=dl.shuffle,
                                                      num_workers=dl.num_workers, pin_memory=dl.pin_memory)
        self.step = 0
        okind = config.optimizer_configuration.type
        if okind == "Adam":
            self.optimizer = HipAdam(self.nppc_model.parameters(), **config.optimizer_configuration.args)
        else:
            self.optimizer = getattr(optim, okind)(self.nppc_model.parameters(), **config.optimizer_configuration.args)
        self._flat_adam = None
        self._sumsq = None

    # ---------------------------------------------------------------------------------- reference API
    def base_step(self, batch):
        return inpainting_base_step(self.nppc_model, batch, self.step, self.config.second_moment_loss_grace,
                                    self.config.second_moment_loss_lambda)

    def base_step2(self, batch, n_mc_samples=50):
        """nppc_trainer.py:244-336: the alternative target -- the NPPC directions are fitted to the MC-dropout + PCA
        components of the restorer (50 stochastic passes with the WHOLE restorer in train mode, as the reference's
        `restoration_model.train()` does: BatchNorm uses batch statistics and its running buffers move)."""
        from ..mc_baseline import PairProjectionLoss, calculate_unet_baseline
        masked_spec, mask, clean_spec = batch
        clean_norm, mask4, masked_norm = preprocess_data(clean_spec, masked_spec, mask)
        w_mat = self.nppc_model(masked_norm, mask4)                      # [B, n_dirs, F, T]
        restoration_model = self.nppc_model.pretrained_restoration_model
        restoration_model.train()
        try:
            mc = calculate_unet_baseline(restoration_model, masked_norm, mask4, n_mc_samples=n_mc_samples,
                                         n_components=w_mat.shape[1])
        finally:
            restoration_model.eval()
        w_mc, singular_values = mc['scaled_principal_components'], mc['singular_vals']
        lam = second_moment_weight(self.step, self.config.second_moment_loss_grace, self.config.second_moment_loss_lambda)
        reconst_err, objective, proj, w_norms, second_moment_mse = PairProjectionLoss.apply(w_mat, w_mc, singular_values, lam)
        log = {
            'w_mat': w_mat.detach(),
            'w_mc': w_mc.detach(),
            'proj_W_mc_on_W_nppc': proj.detach(),
            'w_norms': w_norms.detach(),
            'reconst_err': reconst_err.detach(),
            'second_moment_mse': second_moment_mse.detach(),
            'objective': objective.detach(),
        }
        return reconst_err, objective, log

    def _calculate_final_objective(self, reconst_err, second_moment_mse):
        lam = second_moment_weight(self.step, self.config.second_moment_loss_grace, self.config.second_moment_loss_lambda)
        return reconst_err.mean() + lam * second_moment_mse.mean()

    # ---------------------------------------------------------------------------------- one optimisation step
    def train_step(self, batch):
        """forward + loss + backward + clip_grad_norm_(max_grad_norm) + optimizer step (nppc_trainer.py:145-154)"""
        net = self.nppc_model.pc_wrapper.net
        fast = isinstance(self.optimizer, HipAdam)
        net.flat_grad_only = fast
        try:
            reconst_err, objective, log = self.base_step(batch)
            self.optimizer.zero_grad()
            objective.backward()
        finally:
            net.flat_grad_only = False
        if fast:
            eng = net.engine()
            gflat = eng.fp.grad
            if self._sumsq is None:
                self._sumsq = torch.zeros(1, dtype=torch.float64, device=gflat.device)
            self._sumsq.zero_()
            H.call("nppc_sumsq", gflat, gflat.numel(), self._sumsq, H.stream())
            if self._flat_adam is None or self._flat_adam.eng is not eng:
                self._flat_adam = FlatAdamStepper(self.optimizer, eng)
            self._flat_adam.step(gflat, 1.0, clip=(self._sumsq, float(self.config.max_grad_norm)))
        else:
            torch.nn.utils.clip_grad_norm_(self.nppc_model.parameters(), max_norm=self.config.max_grad_norm)
            self.optimizer.step()
        self.step += 1
        return reconst_err, objective, log

    def train(self, n_steps=None, n_epochs=None, checkpoint_dir="checkpoints", save_flag=True, val_dataloader=None,
              log_every=None):
        """training loop (the name shadows nn.Module.train exactly like the reference, nppc_trainer.py:115)"""
        os.makedirs(checkpoint_dir, exist_ok=True)
        loop_loader = LoopLoader(dataloader=self.dataloader, n_steps=n_steps, n_epochs=n_epochs)
        log_every = log_every or self.config.log_interval
        for it, batch in enumerate(loop_loader):
            masked_spec, mask_frames, clean_spec = batch[:3]
            batch = (masked_spec.to(self.device), mask_frames.to(self.device), clean_spec.to(self.device))
            reconst_err, objective, log_dict = self.train_step(batch)
            if it % log_every == 0 or it + 1 == len(loop_loader):
                print(f'step {self.step}: Objective: {objective.item():.4f} | '
                      f'Second Moment MSE: {log_dict["second_moment_mse"].mean().item():.4f} | '
                      f'Reconstract Error: {reconst_err.mean().item():.4f}')
        if save_flag:
            timestamp = datetime.now().strftime("%Y%m%d_%H%M%S")
            self.save_checkpoint(os.path.join(checkpoint_dir, f"checkpoint_final_{timestamp}.pt"))

    def save_checkpoint(self, checkpoint_path):
        checkpoint = {
            'model_state_dict': self.nppc_model.state_dict(),
            'optimizer_state_dict': self.optimizer.state_dict(),
            'step': self.step,
        }
        os.makedirs(os.path.dirname(checkpoint_path) or ".", exist_ok=True)
        torch.save(checkpoint, checkpoint_path)
        print(f"Checkpoint saved to {checkpoint_path}")
