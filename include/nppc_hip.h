/* C ABI of libnppc_hip.so -- the gfx950 (MI355X / CDNA4) kernels behind the NPPC-audio train step.
 *
 * The reference (kfirc1503/generative-audio) has no FFI: its hot path is stock PyTorch ops called from Python
 * (SURVEY.md section 8b).  Each entry point below replaces the ATen call sites cited beside it (file:line under
 * the reference root).  Conventions:
 *   - plain device pointers + sizes, no torch types; every tensor is borrowed for the launch, nothing is
 *     allocated or freed here; outputs / workspaces are pre-allocated by the caller
 *   - `stream` is a hipStream_t; launches are asynchronous and never synchronise the device
 *   - return 0 on success, non-zero on bad arguments (1) / launch failure (2) / unsupported shape (3)
 *   - prec: 0 = bf16 MFMA operands + bf16 saved activations, fp32 accumulate / cell state / statistics
 *           1 = fp32 everywhere (exact-f32 MFMA) -- the parity mode
 *   - activations of the full-band nets are time-major [batch][Tp][ld] (Tp = T' rounded up to 128, ld = channels
 *     rounded up to 64, padding zero); sub-band tensors are time-major [T'][N][...], N = B*F' sequences
 */
#ifndef NPPC_HIP_H
#define NPPC_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

/* ---- signal front end ------------------------------------------------------------------------------------
 * torch.stft(center, periodic hann, onesided): utils.py:107-147, nppc_audio/trainer.py:349-355 */
int nppc_stft(const float* wave, float* re, float* im, float* mag /*nullable*/, int B, int L, int nfft, int hop,
              void* stream);
/* drop_band: audio_zen/acoustics/feature.py:254-285 */
int nppc_dropband(const float* in, float* out, int B, int C, int F, int T, int G, void* stream);
/* build_complex_ideal_ratio_mask + compress_cIRM (+ drop_band of trainer.py:359-362): audio_zen/acoustics/mask.py:24-54 */
int nppc_cirm_build_compress(const float* nr, const float* ni, const float* cr, const float* ci, float* out, int B, int F,
                             int T, int G, float eps, void* stream);
/* decompress_cIRM (mask.py:57-60) + crm_to_stft_components (utils.py:241-249 -> :75-79, conj(mask)*noisy) */
int nppc_cirm_decompress_apply_conj(const float* crm, const float* nr, const float* ni, float* dec /*nullable*/,
                                    float* emag, float* ere, float* eim, int B, int F, int T, void* stream);

/* the same with the true product mask*noisy: utils.model_outputs_to_waveforms (utils.py:37-58), crm_to_spectogram (:252-256) */
int nppc_cirm_decompress_apply(const float* crm, const float* nr, const float* ni, float* dec /*nullable*/, float* emag,
                               float* ere, float* eim, int B, int F, int T, void* stream);
/* torch.istft(center, periodic hann, length=L): utils.py:60-70, nppc_audio/validator.py:136-143 */
int nppc_istft(const float* re, const float* im, float* out, int B, int T, int nfft, int hop, int L, void* stream);

/* ---- full-band front: offline_laplace_norm + ChannelTimeSenseSELayer ---------------------------------------
 * audio_zen/model/base_model.py:210-224, audio_zen/model/module/attention_model.py:43-98,
 * fullsubnet_plus.py:158-185, nppc_audio/networks.py:80-112 */
int nppc_rowsum(const float* x, double* sums, long R, int T, void* stream);
/* the same for ALL input maps of a net at once (maps: host array of nmaps = 3 or 6 device pointers, map j = m*3 + z with
 * z the mag / real / imag branch and m the noisy / enhanced call): three launches (row sums, attention, scale + transpose)
 * instead of three per map.  rowsum [nmaps][B][C]; scale and the saved tensors [3][nm][B][..]; parameters of branch z sit
 * z * sW elements behind the pointers given; X0 [3][B][Tp][ld] with branch stride sY, map j -> branch z, columns m*C.. */
int nppc_tsse_fwd_maps(int prec, const float* const* maps, int nmaps, double* rowsum, const float* cw0, const float* cb0,
                       const float* cw1, const float* cb1, const float* cw2, const float* cb2, int ks0, int ks1, int ks2,
                       const float* fcw, const float* fcb, const float* w1, const float* b1, const float* w2, const float* b2, long sW,
                       float* scale, float* ns, float* pre, float* sq, float* h1, float* sg, void* X0, long sY, int B, int C, int T,
                       int look_ahead, int Tp, int ld, void* stream);
/* ... and its backward (parameter gradients only: the maps are data), three launches, NO atomics: every gradient element
 * has one writer and a fixed summation order (samples, then maps, in index order), so repeated runs are bit-identical.
 * ws: nppc_tsse_bwd_ws_elems(nmaps, ..) floats (per map: dsg | da2 | da1 | per-sample conv / fc contributions) */
int nppc_tsse_bwd_ws_elems(int nmaps, int B, int C, int ks0, int ks1, int ks2, long* elems);
int nppc_tsse_bwd_maps(int prec, const void* dX0, long sY, const float* const* maps, int nmaps, const double* rowsum,
                       const float* cw0, const float* cw1, const float* cw2, int ks0, int ks1, int ks2, const float* fcw,
                       const float* w1, const float* w2, long sW, const float* ns, const float* pre, const float* sq,
                       const float* h1, const float* sg, float* ws, float* g_cw0, float* g_cb0, float* g_cw1, float* g_cb1,
                       float* g_cw2, float* g_cb2, float* g_fcw, float* g_fcb, float* g_w1, float* g_b1, float* g_w2, float* g_b2,
                       int B, int C, int T, int look_ahead, int Tp, int ld, void* stream);
int nppc_tsse_fwd(const float* x, const double* rowsum, const float* cw0, const float* cb0, const float* cw1,
                  const float* cb1, const float* cw2, const float* cb2, int ks0, int ks1, int ks2, const float* fcw,
                  const float* fcb, const float* w1, const float* b1, const float* w2, const float* b2, float* scale,
                  float* ns, float* pre, float* sq, float* h1, float* sg, int B, int C, int T, int look_ahead,
                  void* stream);
int nppc_tsse_bwd(int prec, const void* dX0, const float* x, const double* rowsum, const float* cw0, const float* cw1,
                  const float* cw2, int ks0, int ks1, int ks2, const float* fcw, const float* w1, const float* w2,
                  const float* ns, const float* pre, const float* sq, const float* h1, const float* sg, float* dsg_ws, /* nppc_tsse_bwd_ws_elems(1, ..) floats */
                  float* g_cw0, float* g_cb0, float* g_cw1, float* g_cb1, float* g_cw2, float* g_cb2, float* g_fcw,
                  float* g_fcb, float* g_w1, float* g_b1, float* g_w2, float* g_b2, int B, int C, int T, int look_ahead, int Tp,
                  int ld, int coff, void* stream);
int nppc_scale_transpose(int prec, const float* x, const float* scale /*nullable*/, void* y, int B, int C, int T, int Tp,
                         int ld, int coff, void* stream);

/* ---- full-band TCN stack: TCNBlock x8 + Linear + ReLU ------------------------------------------------------
 * audio_zen/model/module/causal_conv.py:67-108, audio_zen/model/module/sequence_model.py:47-58,106-112
 * nppc_gemm_nt: C[R][N] = A[R][K] * B[N][K]^T (+ epilogue); epi: 0 plain, 1 bias+PReLU+GroupNorm statistics,
 * 2 bias+residual, 3 bias+ReLU, 4 fp32 output (split-K slabs), 5 mask by (res > 0).  ksplit > 1 splits K. */
int nppc_gemm_nt(int prec, int epi, const void* A, long lda, long sA, const void* B, long ldb, long sB, void* C, long ldc,
                 long sC, const float* bias, long sBias, const void* res, long ldres, long sRes, const float* slope,
                 long sSlope, double* stats, long sStats, int R, int N, int K, int Tp, int Tv, int Nv, int relu_in,
                 int batch, int ksplit, void* stream);
/* long-K weight-gradient product: C_slab[z][M][N] (fp32) = A[M][K/ksplit slice z] * B[N][same slice]^T, LDS-staged
 * 128x128 tiles; M, N multiples of 128, K a multiple of (64 bf16 | 32 fp32) * ksplit */
int nppc_gemm_nt_splitk(int prec, const void* A, long lda, const void* B, long ldb, float* C, long ldc, int M, int N, long K,
                        int ksplit, void* stream);
/* the same product on ROW-major operands (no transposed copies): C_slab[z][M][N] = A[rows z][M]^T * B[rows z][N], bf16,
 * LDS tiles read back with ds_read_b64_tr_b16; M % 128 == 0, N % 64 == 0, R % (64*ksplit) == 0 */
int nppc_gemm_tn_splitk(const void* A, long lda, const void* B, long ldb, float* C, long ldc, int M, int N, long R, int ksplit,
                        void* stream);
/* the same plus rowsum[z][m] = sum over the rows of slice z of A[r][m] ([ksplit][M] fp32): the bias gradient of an LSTM
 * layer out of the weight-gradient product that reads the same gate gradients (LDS-DMA kernel shapes only: M % 256 == 0,
 * N % 128 == 0, (R / ksplit) % 64 == 0, at least 256 workgroups; NPPC_EUNSUPPORTED otherwise) */
int nppc_gemm_tn_splitk_rowsum(const void* A, long lda, const void* B, long ldb, float* C, long ldc, int M, int N, long R,
                               int ksplit, float* rowsum, void* stream);
/* two B operands side by side behind ONE pass over A: C_slab[z][m][0 .. N1) = A^T . B1, C_slab[z][m][N1 .. N1 + N2) = A^T . B2
 * over the rows of K slice z (the weight gradients of one LSTM layer, W_ih | W_hh, share their gate gradients: torch's
 * autograd of nn.LSTM, audio_zen/model/module/sequence_model.py:113-123, computes them as separate products).  rowsum
 * optional ([ksplit][M], as above).  bf16; M % 256 == 0, N1 % 192 == 0, N2 == 64 or N2 % 192 == 0, (R / ksplit) % 64 == 0,
 * ldc >= N1 + N2, at least 256 workgroups; NPPC_EUNSUPPORTED otherwise (run the products separately) */
int nppc_gemm_tn_splitk2(const void* A, long lda, const void* B1, long ldb1, int N1, const void* B2, long ldb2, int N2, float* C,
                         long ldc, int M, long R, int ksplit, float* rowsum, void* stream);
int nppc_gemm_tn_splitk_batched(const void* A, long lda, long sA, const void* B, long ldb, long sB, float* C, long ldc, long sC,
                                int M, int N, long R, int ksplit, int batch, void* stream);
int nppc_gemm_tn_splitk_taps(const void* A, long lda, const void* B, long ldb, float* C, long ldc, int M, int N, long R,
                             int ksplit, int Wp, int shift_a, void* stream);
int nppc_pack_matrix(int prec, const float* src, void* dst, int N, int K, int Npad, int ldd, int transpose, void* stream);
/* the same for an n_a x n_b grid of equally shaped matrices at constant strides (elements) in ONE launch: the 8 TCN blocks
 * x 3 full-band branches of a FullSubNet+ keep their parameters at constant offsets in the flat parameter buffer */
int nppc_pack_matrix_batched(int prec, const float* src, void* dst, int N, int K, int Npad, int ldd, int transpose, int n_a,
                             int n_b, long src_stride_a, long src_stride_b, long dst_stride_a, long dst_stride_b, void* stream);
int nppc_tcn_dwconv(int prec, const void* in, void* out, const double* st1, double* st2, const float* gamma,
                    const float* beta, const float* wd, const float* bd, const float* slope2, int B, int Cc, int ld, int Tp,
                    int Tv, int dil, float eps, long sAct, long sSt, long sP, int batch, void* stream);
int nppc_tcn_gn_apply(int prec, const void* in, void* out, const double* st, const float* gamma, const float* beta, int B,
                      int Cc, int ld, int Tp, int Tv, float eps, long sAct, long sSt, long sP, int batch, void* stream);
/* backward of GroupNorm(1,C)+PReLU and of the depthwise dilated conv (autograd of causal_conv.py:96-108) */
int nppc_tcn_gn_bwd(int prec, const void* dA, const void* y, const double* st, const float* gamma, const float* slope,
                    double* S, void* dpre, float* dgamma, float* dbeta, float* dslope, int B, int Cc, int Tp, int Tv, float eps,
                    long sAct, long sSt, long sP, int batch, void* stream);
int nppc_tcn_dwconv_bwd(int prec, const void* du, const void* y1, const double* st1, const float* gamma, const float* beta,
                        const float* wd, void* dz, float* dwd, float* dbd, int B, int Cc, int Tp, int Tv, int dil, float eps,
                        long sAct, long sSt, long sP, int batch, void* stream);
/* fused backward of a TCNBlock's middle (causal_conv.py:98-106 in reverse: GroupNorm-2, PReLU-2, depthwise dilated conv,
 * GroupNorm-1, PReLU-1): dA = gradient of GN2's output -> dpre1 = gradient of conv1x1's output, every parameter gradient of
 * those stages and the conv1x1 bias gradient, in ONE reduce + ONE apply launch (csrc/tcn_bwd.hip).  y1 / y2: the saved PReLU
 * outputs, st1 / st2 their GroupNorm (sum, sumsq); S: [batch][B][Cc/64][8] fp64 workspace for the channel groups' shares of the per-sample sums (no initial state; no atomics: the apply pass adds the shares in index order); part:
 * nppc_tcn_mid_bwd_part_elems(B, Cc, Tp, batch, &n) -> n floats of workspace for per-workgroup partial sums (a third, tiny launch adds
 * them to the gradients in a fixed order: no atomics anywhere, repeated runs are bit-identical); a2 (nullable) receives GN2(y2), the operand of the sconv weight
 * gradient.  Gradients are ACCUMULATED into their destinations.  colpart (nullable): the tile column sums that
 * nppc_gemm_nt_colsum left for the block's upstream gradient, [batch][cp_tiles][cp_ld] -> the finishing launch also writes the
 * sconv bias gradient dbias2[z*sP + c] = sum over the tiles, c < cp_cols (sconv.bias: causal_conv.py:107). */
int nppc_tcn_mid_bwd_part_elems(int B, int Cc, int Tp, int batch, long* n);
int nppc_tcn_mid_bwd(int prec, const void* dA, const void* y2, const void* y1, const double* st1, const double* st2, double* S,
                     float* part,
                     const float* gamma1, const float* beta1, const float* gamma2, const float* beta2, const float* wd,
                     const float* slope1, const float* slope2, void* a2, void* dpre1, float* dgamma2, float* dbeta2,
                     float* dgamma1, float* dbeta1, float* dwd, float* dbd, float* dslope1, float* dslope2, float* dbias1,
                     const float* colpart, int cp_tiles, int cp_ld, int cp_cols, float* dbias2, int B,
                     int Cc, int Tp, int Tv, int dil, float eps, long sAct, long sSt, long sP, int batch, int finish_now,
                     void* stream);
/* finish_now = 0 defers the finishing launch: one nppc_tcn_mid_bwd_finish then adds up the partial rows of `nblk` blocks at
 * once (block k: part + k * partL, colpart + k * colpartL, gradients k * sL elements behind the pointers given -- the blocks of
 * the TCN stack sit at a constant stride in the flat buffer).  Round 3 ran eight 5-us finishing launches per step, each of
 * which waited up to 0.5 ms for CUs beside the weight-gradient GEMMs of the side queue. */
int nppc_tcn_mid_bwd_finish(const float* part, long partL, const float* colpart, long colpartL, int cp_tiles, int cp_ld,
                            int cp_cols, float* dgamma2, float* dbeta2, float* dgamma1, float* dbeta1, float* dwd, float* dbd,
                            float* dslope1, float* dslope2, float* dbias1, float* dbias2, int B, int Cc, long sP, long sL,
                            int batch, int nblk, void* stream);
/* sconv of a TCNBlock with the GroupNorm in front of it (norm2, causal_conv.py:104-106) folded into the product:
 *   C = rstd_b * (A Wg^T) - mean_b * rstd_b * v + u + res,  A = the un-normalised depthwise output, stats = its per-sample
 * (sum, sumsq), cnt = elements per sample; nppc_tcn_pack_sconv builds Wg[n][k] = gamma[k] W[n][k], v[n] = sum_k Wg[n][k],
 * u[n] = sum_k beta[k] W[n][k] + bias[n] for n_a x n_b equally shaped blocks at constant parameter strides */
int nppc_gemm_nt_gn(int prec, const void* A, long lda, long sA, const void* Wg, long ldb, long sB, void* C, long ldc, long sC,
                    const float* u, const float* v, long sUV, const void* res, long ldres, long sRes, const double* stats,
                    long sStats, double cnt, float eps, int R, int N, int K, int Tp, int Tv, int Nv, int batch, void* stream);
/* nppc_gemm_nt with epi EPI_RESIDUAL or EPI_MASK_POS (LDS-staged shapes only: K % 64 == 0 in bf16 / % 32 in fp32) that also
 * leaves colpart[z][R/128][N] (fp32) = the column sums of every 128-row tile of the stored output.  The output is the upstream
 * gradient of the next 1x1 convolution down the backward chain (causal_conv.py:107, fullsubnet_plus.py fc_output_layer), whose
 * bias gradient is the sum of these partials over the tiles (nppc_tcn_mid_bwd adds them up): no pass of its own over C. */
int nppc_gemm_nt_colsum(int prec, int epi, const void* A, long lda, long sA, const void* B, long ldb, long sB, void* C, long ldc,
                        long sC, const float* bias, long sBias, const void* res, long ldres, long sRes, int R, int N, int K,
                        int Tp, int Tv, int Nv, int batch, float* colpart, void* stream);
int nppc_tcn_pack_sconv(int prec, const float* W, const float* gamma, const float* beta, const float* bias, void* Wg, float* u,
                        float* v, int N, int K, int Npad, int ldd, int n_a, int n_b, long src_stride_a, long src_stride_b,
                        long dst_stride_a, long dst_stride_b, void* stream);
int nppc_transpose(int prec, const void* in, void* out, int rows, int cols, long ld_in, long ld_out, long sIn, long sOut,
                   int relu, int batch, void* stream);
/* out[z][c] += sum_r M[z][r][c] (bias gradients), two launches, NO atomics: row blocks leave partial sums in `scratch`
 * (nppc_colsum_scratch_elems floats, caller-owned: one scratch per concurrently running call) and a finishing launch adds them
 * in a fixed order, so repeated runs are bit-identical */
int nppc_colsum_scratch_elems(int rows, int cols, int batch, long* n);
int nppc_colsum(int prec, const void* M, float* out, int rows, int cols, long ld, long sM, long sOut, int batch,
                float* scratch, long scratch_elems, void* stream);
int nppc_reduce_slabs_t(const float* slabs, int S, long slab_stride, long ld, float* dst, long dst_ld, int rows, int ncols,
                        long sSlab, long sDst, int batch, void* stream);
int nppc_reduce_slabs(const float* slabs, int S, long slab_stride, long ld, float* dst, long dst_ld, int rows, int col0,
                      int ncols, int permH, int accumulate, long sSlab, long sDst, int batch, void* stream);

/* ---- sub-band stage: unfold + concat + laplace norm + drop_band + LSTM input layout, output head ------------
 * audio_zen/model/base_model.py:15-46, fullsubnet_plus.py:188-230, nppc_audio/networks.py:115-161,
 * sequence_model.py:118-123 (fc_output_layer) */
/* work: caller-owned, 2*B 8-byte words (B fp64 partial sums + B arrival counters), zeroed ONCE by the caller; the kernel
 * re-arms it, so the same workspace serves every later launch and launches on different workspaces may overlap */
int nppc_subband_mean(int prec, const void* src, int ldS, const void* fb, int ldF, long strideFb, const float* mult,
                      float* scale, double* work, int B, int F, int Tp, int Tv, int nfeat, void* stream);
int nppc_subband_stage(int prec, const void* src, int ldS, const void* fb, int ldF, long strideFb, const float* scale,
                       void* x, int B, int F, int Tp, int Tv, int nb, int G, int KX, int ones_col, void* stream);
int nppc_subband_stage_bwd(int prec, const void* dx, const void* x, const void* fb, const float* scale, double* D,
                           void* dpre, int B, int F, int Tp, int Tv, int ldF, long strideFb, int nb, int G, int KX,
                           void* stream);
int nppc_sb_head(int prec, const void* h2, const void* wh, const float* bias, float* out, long Nseq, int Tn, int la,
                 int Hd, int O, int Fo, void* stream);
int nppc_sb_head_bwd(int prec, const float* dout, const void* whT, const void* h2, void* dh2, float* dWh, float* dbh,
                     long Nseq, int Tn, int la, int Hd, int O, int Fo, void* stream);

/* ---- sub-band sequence model: nn.LSTM(I,H,2) -----------------------------------------------------------------
 * audio_zen/model/module/sequence_model.py:30-37 (construction), :113-123 (forward) and its autograd */
int nppc_lstm2_packed_elems(int I, int H, long* n1, long* n2, int* kx);
int nppc_lstm2_pack_weights(int prec, const float* w_ih0, const float* w_hh0, const float* b_ih0, const float* b_hh0,
                            const float* w_ih1, const float* w_hh1, const float* b_ih1, const float* b_hh1, int I, int H,
                            void* wp1, void* wp2, float* bias1, float* bias2, void* stream);
/* x [Tn][N][kx]; h2 [Tn][N][H] time-major; when train also h1, c1, c2 [Tn][N][H] and g1, g2 [Tn][N][H][4] (i,g,f,o) */
int nppc_lstm2_fwd(int prec, int train, int mtile, const void* x, const void* wp1, const void* wp2, const float* bias1,
                   const float* bias2, void* h2, void* h1, void* g1, void* g2, void* c1, void* c2, long N, int Tn, int I,
                   int H, void* stream);
/* Cooperative forward: G workgroups (CUs) share a tile of 16*mtile sequences and split the hidden units, each streaming
 * 1/G of the weights; h slices cross CUs through `xch` with bounded-spin epoch flags.  `flags` (caller-owned, at least
 * clusters*2*G + 4 words, zero-initialised ONCE by the caller) = the epoch words, which every launcher of this family
 * zeroes itself, followed by the STICKY hand-off time-out counter flags[clusters*2*G]: a workgroup whose bounded spin
 * gives up adds 1 to it and carries on with wrong numbers; no launcher ever clears it, so a host read at any later time
 * sees a time-out of any earlier launch on this flag block.  The caller clears it (a 4-byte memset) after handling it.
 * Same tensor contract as nppc_lstm2_fwd. */
int nppc_lstm2_coop_plan(int prec, int train, long N, int H, int n_cu, int* G, int* mtile, int* clusters);
int nppc_lstm2_fwd_coop(int prec, int train, int G, int mtile, const void* x, const void* wp1, const void* wp2,
                        const float* bias1, const float* bias2, void* h2, void* h1, void* g1, void* g2, void* c1, void* c2,
                        void* xch, long xch_bytes, unsigned* flags, long N, int Tn, int I, int H, void* stream);
/* the same with the output head fused (SequenceModel.forward's fc_output_layer, sequence_model.py:119-123): the CU pair
 * also leaves hpart [2][Tn][N][O] fp32 = per-CU partial sums of h2[t][n][:] . whp[o][:] (whp [16][H], rows >= O zero);
 * nppc_sb_head_finalize adds them and the bias and writes out[bo][o][fo][t - la] like nppc_sb_head (O <= 16).
 * Inference (train = 0): h2 is not stored at all (h2 .. c2 may be null); training keeps the saved state. */
int nppc_lstm2_fwd_coop_head(int prec, int train, int mtile, const void* x, const void* wp1, const void* wp2,
                             const float* bias1, const float* bias2, void* h2, void* h1, void* g1, void* g2, void* c1, void* c2,
                             void* xch, long xch_bytes, unsigned* flags, long N, int Tn, int I, int H, const void* whp,
                             float* hpart, int O, int x_ld /* row stride of x in elements: 0 or 64 = [Tn][N][64]; a packed width
                             (multiple of 8, I < x_ld <= 64): only x_ld columns per row are fetched (the frozen restorer's input,
                             staged 40 wide: -37 % of its bytes) */, void* stream);
int nppc_sb_head_finalize(const float* hpart, int G, const float* bias, float* out, long Nseq, int Tn, int la, int O, int Fo,
                          void* stream);
/* Weight-STATIONARY forward of the same LSTM (sequence_model.py:113-123; bf16, H = 384, I <= 64; any N, the last 32-sequence chunk may be ragged): clusters of
 * 12 CUs keep all weights in registers for the whole launch and exchange the hidden state instead (csrc/lstm_ws.hip).
 * nppc_lstm2_ws_plan: clusters == 0 -> not applicable.  wp1 / wp2: nppc_lstm2_ws_pack of the fp32 weights.
 * train = 0: h1 / h2 are [2][N][H] exchange rings and g / c may be null; train = 1: the saved state of nppc_lstm2_fwd
 * (h1 / h2 [Tn][N][H] double as the exchange medium).  cst: clusters * nch_max * 2048 floats of scratch; flags: clusters *
 * nch_max * 16 + 4 u32, the last 4 behind the epochs hold the sticky time-out counter (never cleared here).
 * whp != null: hpart [Tn][N][O] fp32 = h2[t][n][:] . whp[o][:] (finish with nppc_sb_head_finalize, G = 1). */
int nppc_lstm2_ws_plan(int prec, long N, int H, int I, int n_cu, int* clusters, int* nch_max);
int nppc_lstm2_ws_packed_elems(long* n1, long* n2);
int nppc_lstm2_ws_pack(const float* w_ih0, const float* w_hh0, const float* w_ih1, const float* w_hh1, int I, void* wp1, void* wp2,
                       void* stream);
int nppc_lstm2_fwd_ws(int train, const void* x, const void* wp1, const void* wp2, const float* bias1, const float* bias2, void* h1,
                      void* h2, void* g1, void* g2, void* c1, void* c2, float* cst, unsigned* flags, const void* whp, float* hpart,
                      int O, long N, int Tn, int clusters, int nch_max, void* stream);
/* cooperative backward (bf16, H = 384): CU pairs share 32 sequences, each owns half the hidden units / output columns */
int nppc_lstm2_coop_bwd_packed_elems(long* n);
int nppc_lstm2_coop_bwd_pack(const float* w_ih0, const float* w_hh0, const float* w_ih1, const float* w_hh1, int I, void* wb1,
                             void* wb2, void* stream);
int nppc_lstm2_bwd_coop(const void* g1, const void* g2, const void* c1, const void* c2, const void* dh2, const void* wb1,
                        const void* wb2, void* dx, void* dg1, void* dg2, void* xch, long xch_bytes, unsigned* flags, long N,
                        int Tn, int n_cu, void* stream);
/* K-split variant of the cooperative backward (each CU multiplies its own gate-gradient half with all output columns and
 * the pair exchanges bf16 partial sums, accumulator to accumulator: 38 KB per step instead of 96 KB); same tensor contract,
 * except flags: ceil(N / 32) * 48 + 4 u32 ([cluster][layer 2][CU 2][wave 12] epochs -- every wave hands off its own tiles --
 * followed by the 4 sticky time-out words); xch: ceil(N / 32) * 2 * 2 * 2 * 32 * 384 bf16 */
int nppc_lstm2_coop_bwd2_packed_elems(long* n);
int nppc_lstm2_coop_bwd2_pack(const float* w_ih0, const float* w_hh0, const float* w_ih1, const float* w_hh1, int I, void* wb1,
                              void* wb2, void* stream);
int nppc_lstm2_bwd_coop2(const void* g1, const void* g2, const void* c1, const void* c2, const void* dh2, const void* wb1,
                         const void* wb2, void* dx, void* dg1, void* dg2, void* xch, long xch_bytes, unsigned* flags, long N,
                         int Tn, int n_cu, void* stream);
/* the same with the head backward fused: instead of dh2 the kernel takes the dY rows dyt [Tn][N][16] bf16
 * (nppc_head_dy_gather: dout [B'][O][Fo][Tn - la] -> rows, zero beyond O and before the look-ahead) and the packed head
 * weights whT [H][32] ([u][o]) and forms d h2 += dY . Wh itself; nppc_sb_head_bwd_w is the weight / bias half of
 * nppc_sb_head_bwd (no dh2 output). */
int nppc_lstm2_bwd_coop2_head(const void* g1, const void* g2, const void* c1, const void* c2, const void* dyt, const void* whT,
                              const void* wb1, const void* wb2, void* dx, void* dg1, void* dg2, void* xch, long xch_bytes,
                              unsigned* flags, long N, int Tn, int n_cu, void* stream);
/* The same K-split backward on FOUR-CU clusters of 64 sequences (each CU owns 96 hidden units and streams a quarter of the weight
 * fragments per step; three bf16 partial-sum shipments per layer and step, accumulator to accumulator, per-wave epochs).  Same
 * tensor contract as nppc_lstm2_bwd_coop2 / _head: dh2 != null, or dh2 == null with the fused head's dyt + whT.  Needs
 * ceil(N / 64) * 4 <= n_cu.  nppc_lstm2_coop_bwd4_sizes: elements of each packed weight buffer (nppc_lstm2_coop_bwd4_pack),
 * bytes of xch and u32 words of flags (epochs + the 4 sticky time-out words) for N sequences. */
int nppc_lstm2_coop_bwd4_sizes(long N, long* packed_elems, long* xch_bytes, long* flag_words);
int nppc_lstm2_coop_bwd4_pack(const float* w_ih0, const float* w_hh0, const float* w_ih1, const float* w_hh1, int I, void* wb1,
                              void* wb2, void* stream);
int nppc_lstm2_bwd_coop4(const void* g1, const void* g2, const void* c1, const void* c2, const void* dh2, const void* dyt,
                         const void* whT, const void* wb1, const void* wb2, void* dx, void* dg1, void* dg2, void* xch,
                         long xch_bytes, unsigned* flags, long N, int Tn, int n_cu, void* stream);
int nppc_head_dy_gather(const float* dout, void* dyt, long Nseq, int Tn, int la, int O, int Fo, void* stream);
int nppc_sb_head_bwd_w(int prec, const float* dout, const void* h2, float* dWh, float* dbh, long Nseq, int Tn, int la, int Hd,
                       int O, int Fo, void* stream);
int nppc_lstm2_bwd_packed_elems(int I, int H, long* n1, long* n2);
int nppc_lstm2_pack_weights_bwd(int prec, const float* w_ih0, const float* w_hh0, const float* w_ih1, const float* w_hh1,
                                int I, int H, void* wb1, void* wb2, void* stream);
/* dh2 [Tn][N][H] -> dx [Tn][N][kx], gate gradients dg1/dg2 [Tn][N][4H] (column k = unit*4 + gate in i,g,f,o order) */
int nppc_lstm2_bwd(int prec, const void* g1, const void* g2, const void* c1, const void* c2, const void* dh2,
                   const void* wb1, const void* wb2, void* dx, void* dg1, void* dg2, long N, int Tn, int I, int H,
                   void* stream);

/* ---- Gram-Schmidt on the K complex directions + NPPC loss ----------------------------------------------------
 * nppc_audio/pc_wrapper.py:8-44 (gram_schmidt_to_crm), nppc_audio/trainer.py:259-317 (base_step) */
int nppc_gram(const float* a, const float* b_or_null, const float* gt, const float* pred, double* out, int B, int K,
              long N, void* stream);
int nppc_combine(const float* a, const double* M1, const float* b, const double* M2, const float* gt, const float* pred,
                 float* out, int B, int K, long N, void* stream);
int nppc_gs_solve(const double* G, double* C, double* Ch, int B, int K, int KV, void* stream);
int nppc_gs_bwd_solve(const double* G, const double* P, const double* Ch, double* D, int B, int K, int KV, void* stream);
int nppc_loss_solve(const double* G, float* err_norm, float* proj_re, float* proj_im, float* proj_mag, float* w_norms,
                    float* reconst, float* sm, double* coefA, double* coefE, int B, int K, void* stream);
int nppc_loss_solve_eps(const double* G, float* err_norm, float* proj_re, float* proj_im, float* proj_mag, float* w_norms,
                        float* reconst, float* sm, double* coefA, double* coefE, int B, int K, double eps, int eps_in_norms,
                        void* stream);
/* the same, which also leaves the step's objective = mean_b reconst + lam * mean_{b,i} sm (trainer.py:300-304) in objective[0]:
 * one workgroup, B <= 1024 (NPPC_EUNSUPPORTED beyond) */
int nppc_loss_solve_obj(const double* G, float* err_norm, float* proj_re, float* proj_im, float* proj_mag, float* w_norms,
                        float* reconst, float* sm, double* coefA, double* coefE, int B, int K, double eps, int eps_in_norms,
                        float lam, float* objective, void* stream);
int nppc_loss_bwd_coef(const double* coefA, const double* coefE, const float* grec, float gobj_over_B, float gsm, double* M1,
                       int B, int K, void* stream);
/* the same with the upstream gradient of the objective read from DEVICE memory (gobj, one float): the coefficients are
 * (gobj * inv_B, gobj * sm_weight); autograd's backward then never reads a scalar back to the host */
int nppc_loss_bwd_coef_dev(const double* coefA, const double* coefE, const float* grec, const float* gobj, float inv_B,
                           float sm_weight, double* M1, int B, int K, void* stream);

/* ---- optimizer: torch.optim.Adam (nppc_audio/trainer.py:64-69,102-104) -------------------------------------- */
int nppc_adam_step(float* p, const float* g, float* m, float* v, long n, double lr, double b1, double b2, double eps,
                   double wd, int step, double gscale, void* stream);
/* the same, guarded: `guards` = device array of n_guards device pointers to the sticky hand-off time-out counters of the
 * cooperative LSTM launches; if any is non-zero the update is skipped (p, m, v untouched) and *poison (nullable, the step's
 * objective) becomes NaN -- wrong numbers never reach the weights between two host checks of the counters */
int nppc_adam_step_guarded(float* p, const float* g, float* m, float* v, long n, double lr, double b1, double b2, double eps,
                           double wd, int step, double gscale, const void* guards, int n_guards, float* poison, void* stream);

/* ---- on-device batch synthesis (dataset/audio_dataset.py:92-152: dBFS normalisation, SNR mix, clip guard) -------
 * target_item (nullable): per-clip normalisation level [B] in dBFS (the reference's target_dB_FS_floating_value > 0
 * draws one per item, :94-101); null = target_dbfs for every clip */
int nppc_mix_snr(const float* clean, const float* noise, const float* snr_db, float target_dbfs, const float* target_item,
                 float* noisy_out, float* clean_out, int B, int L, void* stream);

/* inpainting batch synthesis (dataset/audio_dataset_inpainting.py __getitem__ :291-313), one launch per batch:
 * _normalize_audio (:154-168, when do_norm), the gap mask of _create_random_mask (:170-181: zeros on
 * [gap_start[b], gap_end[b])) applied to the audio (masked_out, nullable), and time_to_spec_mask (:223-251, centred)
 * for that gap -> mask_frames [B, T]. */
int nppc_inpaint_prepare(const float* clean, const int* gap_start, const int* gap_end, int do_norm, float target_dbfs,
                         float* clean_out /*nullable*/, float* masked_out /*nullable*/, float* mask_frames, int B, int L,
                         int win, int hop, int T, void* stream);
/* time_to_spec_mask (:223-251) for an arbitrary sample mask [B, L]: frame = 1 iff every sample of its clamped window is 1 */
int nppc_time_to_spec_mask(const float* mask_time, float* mask_frames, int B, int L, int win, int hop, int center, int T,
                           void* stream);
/* utils.audio_to_stft (utils.py:150-175) for any nfft = win_length <= 512: [B, L] -> spec [B, 2, F, T] (T = 1 + L / hop);
 * masked_spec (nullable) = spec * mask_frames[b, t] (audio_dataset_inpainting.py:307-310) */
int nppc_stft_pair(const float* wave, const float* mask_frames /*nullable*/, float* spec, float* masked_spec /*nullable*/,
                   int B, int L, int nfft, int hop, void* stream);

/* ---- MC-dropout + PCA baseline (utils.py:334-648) -----------------------------------------------
 * nn.Dropout(p) (tmp_utils.py:28-29) in place on channels [0, C) of a haloed NHWC activation X [rows][ld]:
 * keep bit = Philox4x32-10(seed; row, channel / 4, stream_id) >= p * 2^32, kept values scaled by 1 / (1 - p);
 * keep_out (nullable) [rows][C] u8 receives the bits. */
int nppc_dropout(int prec, void* X, long ld, long rows, int C, float p, long seed, int stream_id,
                 unsigned char* keep_out /*nullable*/, void* stream);
/* compute_pca_sklearn_batch (utils.py:393-496) for all items at once: X [K][B][D] fp32 (K <= 60 samples per item) ->
 * mean [B][D], comps [B][n][D] (unit, largest-magnitude entry positive), scaled = comps * singular value,
 * svals [B][n], weights = svals / sum(svals) [B][n]; work = *elems of nppc_pca_work_elems doubles. */
int nppc_pca_work_elems(int K, int B, int n, long* elems);
int nppc_pca_batch(const float* X, int K, int B, int D, int n, float* mean, float* comps, float* scaled, float* svals,
                   float* weights, double* work, void* stream);

/* base_step2's projection loss (inpainting/trainer/nppc_trainer.py:285-323): rows w, m [B*K][N] (NPPC directions, scaled MC
 * components), sv [B][K] -> proj, w_norms (= |w| + eps) [B][K], reconst, second [B]; sums [B*K*3] / coef [B*K*3] doubles are
 * workspace kept for nppc_pair_loss_bwd: dw = d(sum_b grec[b] * reconst_b + g_rec_all * sum_b reconst_b
 * + g_sm_all * sum_b second_b) / dw. */
int nppc_pair_loss(const float* w, const float* m, const float* sv, double* sums, float* proj, float* w_norms, float* reconst,
                   float* second, double* coef, int B, int K, long N, double eps, void* stream);
int nppc_pair_loss_bwd(const float* w, const float* m, const double* coef, const float* grec /*nullable*/, float g_rec_all,
                       float g_sm_all, float* dw, int B, int K, long N, void* stream);

/* compute_metrics (inpainting/validator/validator_nppc_model.py:742-828): out [3][N] = {pred - clean, the same on the gap
 * only (mask == 0), (mean - clean) on the gap only}; nppc_rows_gram: G [Ra][Rb] (fp64) = A [Ra][N] . Bm [Rb][N]^T */
int nppc_metric_rows(const float* pred, const float* clean, const float* mean, const float* mask, float* out, long N,
                     void* stream);
int nppc_rows_gram(const float* A, int Ra, const float* Bm, int Rb, long N, double* G, void* stream);

/* clip_grad_norm_(max_norm) + Adam without a host round trip (inpainting/trainer/nppc_trainer.py:149-154):
 * nppc_sumsq accumulates sum(g^2) into a zeroed device double, nppc_adam_step_clip reads it. */
int nppc_sumsq(const float* g, long n, double* out, void* stream);
int nppc_adam_step_clip(float* p, const float* g, float* m, float* v, long n, double lr, double b1, double b2, double eps,
                        double wd, int step, double gscale, const double* sumsq, double max_norm, void* stream);

/* ---- inpainting sibling: U-Net on haloed NHWC activations ------------------------------------------------------
 * nppc_audio/inpainting/networks/unet.py:247-313, tmp_utils.py:8-99 (conv3x3 + BatchNorm2d + LeakyReLU(0.2),
 * MaxPool2d(2), bilinear x2 align_corners + pad + cat, conv1x1), utils.py:273-306 (log-magnitude, batch-global
 * mean / unbiased std), inpainting/nppc/pc_wrapper.py:75-84 and unet.py:299-312 (mask blending).
 * Activations: X[(b*(H+2)+y)*(W+2)+x][ld], zero halo, zero guard rows; see csrc/unet.hip. */
int nppc_logmag(const float* spec, float* out, long out_bstride, int B, long FT, double* st, void* stream);
int nppc_standardize(float* a, float* b_or_null, long bstride, int B, long FT, const double* st, float* mean_std, void* stream);
int nppc_unet_stage_map(int prec, const float* src, long src_bstride, void* dst, long ld, int c, int B, int H, int W,
                        void* stream);
int nppc_conv_pack(int prec, const float* w, void* wf, void* wb, int Cout, int Cin, int ksize, int Np, int Cinp, int Mb,
                   int Coutp, void* stream);
int nppc_conv_fwd(int prec, const void* A, long lda, const void* Wp, void* C, long ldc, const float* bias, const float* scale,
                  const float* shift, float slope, int B, int H, int W, int Cin, int Cout, int Np, int ksize, void* stream);
/* the same convolution without the folded-BatchNorm epilogue, which also leaves the column sums (sum, sum of squares) of every
 * 128-row tile of its STORED output in stat_part [ceil(B (H+2) (W+2) / 128)][2][Np] (fp32): the batch statistics of the
 * train-mode BatchNorm that follows (tmp_utils.py:8-37) come from nppc_bn_stats_from_parts instead of from a pass over the
 * tensor (nppc_bn_stats).  Tiled-kernel shapes only (Cin % 64 == 0 in bf16, % 32 in fp32; NPPC_EUNSUPPORTED otherwise).
 * nppc_bn_stats_from_parts: st[c] = sum x, st[C + c] = sum x^2 (fp64, plain stores: no initial state, fixed summation order);
 * scratch: 2 * C * 128 doubles. */
int nppc_conv_fwd_stats(int prec, const void* A, long lda, const void* Wp, void* C, long ldc, const float* bias, int B, int H, int W,
                        int Cin, int Cout, int Np, int ksize, float* stat_part, void* stream);
int nppc_bn_stats_from_parts(const float* stat_part, int B, int H, int W, int Np, int C, double* st, double* scratch, void* stream);
int nppc_conv_wgrad(int prec, const void* dY, long lddy, const void* X, long ldx, float* slabs, int M, int N, int B, int H, int W,
                    int ksize, int ksplit, void* stream);
int nppc_conv_wgrad_transposed(int M, int N);
int nppc_conv_wgrad_reduce(int prec, const float* slabs, int ksplit, int M, int N, float* dW, int Cout, int Cin, int ksize,
                           void* stream);
/* The thin ends of the U-Net (unet.py:247-262 `inc`: conv3x3 from 1-2 channels; `outc`: conv1x1 64 -> K <= 8; tmp_utils.py:8-37):
 * memory-bound products that the MFMA path computes 32-64 channels wide.  Direct kernels, same haloed NHWC tensors and the same
 * results as nppc_conv_fwd / nppc_conv_wgrad + _reduce / the transposed nppc_conv_fwd, but w / dW are the fp32 parameter
 * tensors in torch layout ([Cout][Cin][kh][kw], no packing) and the products accumulate in fp32.  part: workspace of
 * nppc_conv_thin_part_elems floats.  3x3: Cin 1 or 2, Cout % 8 == 0, <= 64; 1x1: Cin == 64, K <= 8. */
int nppc_conv_thin_part_elems(long* n);
int nppc_conv3x3_thin_fwd(int prec, const void* A, long lda, const float* w, const float* bias, const float* scale,
                          const float* shift, float slope, void* C, long ldc, int B, int H, int W, int Cin, int Cout, void* stream);
int nppc_conv3x3_thin_wgrad(int prec, const void* dY, long lddy, const void* X, long ldx, float* part, float* dW, int B, int H,
                            int W, int Cin, int Cout, void* stream);
int nppc_conv1x1_thin_fwd(int prec, const void* X, long ldx, const float* w, const float* bias, void* C, long ldc, int B, int H,
                          int W, int Cin, int K, void* stream);
int nppc_conv1x1_thin_bwd_data(int prec, const void* dY, long lddy, const float* w, void* dX, long lddx, int B, int H, int W,
                               int Cin, int K, void* stream);
int nppc_conv1x1_thin_wgrad(int prec, const void* dY, long lddy, const void* X, long ldx, float* part, float* dW, int B, int H,
                            int W, int Cin, int K, void* stream);
int nppc_bn_stats(int prec, const void* X, long ld, long P, int C, double* st, void* stream);
int nppc_bn_finalize(const double* st, const float* gamma, const float* beta, float* rmean, float* rvar, float* ss, int C,
                     double n, float eps, float momentum, int train, void* stream);
int nppc_bn_act(int prec, const void* X, long ldx, void* Y, long ldy, const float* ss, int C, int B, int H, int W, float slope,
                void* stream);
/* (Y is accepted and not read since round 4: the LeakyReLU mask is recomputed from X with bn_act's own expression) */
int nppc_bn_bwd(int prec, const void* dyA, long ldA, const void* dyB, long ldB, const void* Y, long ldy, const void* X, long ldx,
                const float* ss, double* S, void* dX, long lddx, float* dgamma, float* dbeta, int C, int B, int H, int W,
                float slope, void* stream);
int nppc_maxpool2(int prec, const void* X, long ldx, void* Y, long ldy, unsigned char* idx, int C, int B, int H, int W,
                  void* stream);
int nppc_maxpool2_bwd(int prec, const void* dY, long ldy, const unsigned char* idx, void* dX, long ldx, int C, int B, int H, int W,
                      void* stream);
int nppc_upsample2(int prec, const void* X, long ldx, void* Y, long ldy, int C, int B, int Hi, int Wi, int Ht, int Wt,
                   void* stream);
int nppc_upsample2_bwd(int prec, const void* dY, long ldy, void* dX, long ldx, int C, int B, int Hi, int Wi, int Ht, int Wt,
                       void* stream);
int nppc_unet_out(int prec, const void* raw, long ld, const float* mask, const float* xin, long xin_bstride, float* out,
                  long out_pstride, int K, int B, int H, int W, int mode, void* stream);
int nppc_unet_out_bwd(int prec, const float* dout, long dout_pstride, const float* mask, void* draw, long ld, int K, int B,
                      int H, int W, void* stream);

#ifdef __cplusplus
}
#endif
#endif
