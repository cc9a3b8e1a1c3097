"""On-device batch synthesis for the inpainting step: what
AudioInpaintingDataset.__getitem__ (dataset/audio_dataset_inpainting.py:254-327) computes per clip on the CPU
(dBFS normalisation, gap mask, centred STFT, time_to_spec_mask, frame-masked STFT), batched in HBM with three kernels."""
import torch

from .. import _hip as H


def time_to_spec_mask(mask_time, T_frames, waveform_length=None, win_length=255, hop_length=128, center=True):
    """AudioInpaintingDataset.time_to_spec_mask (:223-251) for a batch of sample masks [B, L] (the reference takes
    [1, L]): frame t -> 1.0 iff every sample of its clamped window is 1, 0.0 for an empty window."""
    H.require_gpu()
    mask_time = mask_time.contiguous().float()
    assert mask_time.dim() == 2, "mask_time should be [B, T] shape."
    if waveform_length is not None and int(waveform_length) < mask_time.shape[1]:
        mask_time = mask_time[:, :int(waveform_length)].contiguous()
    B, L = mask_time.shape
    out = torch.empty(B, T_frames, dtype=torch.float32, device=mask_time.device)
    H.call("nppc_time_to_spec_mask", mask_time, out, B, L, win_length, hop_length, int(center), T_frames, H.stream())
    return out


def audio_to_stft(waveform, nfft=255, hop_length=128, win_length=None):
    """utils.audio_to_stft (utils.py:150-175): [B, L] -> [B, 2, F, T] real/imag pair, any nfft = win_length <= 512."""
    H.require_gpu()
    if waveform.dim() == 1:
        waveform = waveform[None]
    if win_length not in (None, nfft):
        raise NotImplementedError("win_length != nfft")
    waveform = waveform.contiguous().float()
    B, L = waveform.shape
    spec = torch.empty(B, 2, nfft // 2 + 1, 1 + L // hop_length, dtype=torch.float32, device=waveform.device)
    H.call("nppc_stft_pair", waveform, None, spec, None, B, L, nfft, hop_length, H.stream())
    return spec


def inpainting_batch_on_device(clean_audio, gap_start, gap_end, nfft=255, hop_length=128, target_dB_FS=-25.0,
                               normalize=True):
    """clean_audio [B, L] (device), gap_start / gap_end [B] int sample indices ->
    (stft_masked [B,2,F,T], mask_frames [B,T], stft_clean [B,2,F,T], masked_audio [B,1,L]): the order of
    AudioInpaintingSample.get_training_tuple (:38-40), ready for NPPCAudioInpaintingTrainer.base_step."""
    H.require_gpu()
    clean_audio = clean_audio.contiguous().float()
    B, L = clean_audio.shape
    dev = clean_audio.device
    g0 = torch.as_tensor(gap_start, dtype=torch.int32, device=dev).contiguous()
    g1 = torch.as_tensor(gap_end, dtype=torch.int32, device=dev).contiguous()
    assert g0.shape == (B,) and g1.shape == (B,)
    T, F = 1 + L // hop_length, nfft // 2 + 1
    s = H.stream()
    audio = torch.empty_like(clean_audio)
    masked_audio = torch.empty(B, 1, L, dtype=torch.float32, device=dev)
    mask_frames = torch.empty(B, T, dtype=torch.float32, device=dev)
    H.call("nppc_inpaint_prepare", clean_audio, g0, g1, int(bool(normalize)), float(target_dB_FS), audio, masked_audio,
           mask_frames, B, L, nfft, hop_length, T, s)
    stft_clean = torch.empty(B, 2, F, T, dtype=torch.float32, device=dev)
    stft_masked = torch.empty_like(stft_clean)
    H.call("nppc_stft_pair", audio, mask_frames, stft_clean, stft_masked, B, L, nfft, hop_length, s)
    return stft_masked, mask_frames, stft_clean, masked_audio
