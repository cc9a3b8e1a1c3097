/* C ABI of libnppc_hip.so -- the gfx950 kernels behind the NPPC-audio train step.
 *
 * The reference (kfirc1503/generative-audio) has no FFI: its hot path is stock PyTorch ops called
 * from Python.  Each entry point below replaces the ATen call sites cited
 * beside it (file:line under the reference root).  Conventions:
 *   - plain device pointers + sizes, no torch types; every tensor is borrowed for the launch,
 *     nothing is allocated or freed here; outputs/workspaces are pre-allocated by the caller
 *   - `stream` is a hipStream_t; launches are asynchronous and never synchronise the device
 *   - return 0 on success, non-zero (NPPC_E*) on bad arguments / unsupported shape / launch failure
 *   - prec: 0 = bf16 MFMA operands + bf16 saved activations, fp32 accumulate/state
 *           1 = fp32 everywhere (exact-f32 MFMA) -- the parity mode
 */
#ifndef NPPC_HIP_H
#define NPPC_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

/* ---- sub-band sequence model: nn.LSTM(I,H,2) --------------------------------------------------
 * reference: audio_zen/model/module/sequence_model.py:30-37 (construction), :113-123 (forward) */
int nppc_lstm2_packed_elems(int I, int H, long* n1, long* n2, int* kx);
int nppc_lstm2_pack_weights(int prec, const float* w_ih0, const float* w_hh0, const float* b_ih0, const float* b_hh0,
                            const float* w_ih1, const float* w_hh1, const float* b_ih1, const float* b_hh1, int I, int H,
                            void* wp1, void* wp2, float* bias1, float* bias2, void* stream);
/* x [Tn][N][kx]; h2 [Tn][N][H] time-major; when train also c1,c2 [Tn][N][H], g1,g2 [Tn][N][H][4] (i,g,f,o) and the
 * transposed hidden states h1T,h2T [H][Tn*N] that the weight-gradient GEMMs consume. */
int nppc_lstm2_fwd(int prec, int train, int mtile, const void* x, const void* wp1, const void* wp2, const float* bias1,
                   const float* bias2, void* h2, void* h1T, void* h2T, void* g1, void* g2, void* c1, void* c2, long N,
                   int Tn, int I, int H, long Np, void* stream);

#ifdef __cplusplus
}
#endif
#endif
