"""Inpainting sibling of the NPPC-audio path (reference: nppc_audio/inpainting/), on the MI355X kernels.
Same import paths as the reference: inpainting.networks.unet, inpainting.nppc.{pc_wrapper,nppc_model},
inpainting.trainer.nppc_trainer."""
