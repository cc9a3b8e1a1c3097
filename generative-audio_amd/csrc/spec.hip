// Full-band front: offline laplace norm (per-sample mean), TSSE channel attention, and the
// scale + transpose into the time-major activation layout the TCN GEMMs consume.
// Reference: audio_zen/model/base_model.py:210-224 (offline_laplace_norm),
//            audio_zen/model/module/attention_model.py:43-98 (ChannelTimeSenseSELayer),
//            fullsubnet_plus.py:158-185 / nppc_audio/networks.py:80-112 (pad look_ahead, norm, attention, concat).
#include "common.h"
#include "nppc_hip.h"

namespace {

// one wave per row: sums[r] = sum_t x[r][t]   (fp64 accumulate: the real/imag maps have |mean| << |x|)
__global__ __launch_bounds__(256) void rowsum_kernel(const float* __restrict__ x, double* __restrict__ sums, long R, int T) {
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  const int lane = threadIdx.x & 63;
  const float* p = x + (size_t)r * T;
  double s = 0.0;
  for (int t = lane; t < T; t += 64) s += (double)p[t];
  s = wave_sum(s);
  if (lane == 0) sums[r] = s;
}

struct TsseW {
  const float* cw[3];   // depthwise conv weights [C][ks]
  const float* cb[3];   // [C]
  int ks[3];
  const float* fcw;     // feature_concate_fc.weight [1][3]
  const float* fcb;     // [1]
  const float* w1;      // fc1 [C2][C]
  const float* b1;
  const float* w2;      // fc2 [C][C2]
  const float* b2;
};

// One workgroup per sample.  x [B][C][T] is the UNPADDED map; the padded length is Tp = T + la (zeros).
// mean_t(conv_ks(norm x))[c] = bias + ns/(Tp-ks+1) * sum_k w[c][k] * (rowsum - prefix(k) - suffix(ks-1-k)).
// Outputs: scale[b][c] = ns_b * s[b][c];  saved for backward: ns[B], pre[B][C][3], sq[B][C], h1[B][C2], sg[B][C].
constexpr int TSSE_MAXC = 1024;
__global__ __launch_bounds__(256) void tsse_fwd_kernel(const float* __restrict__ x, const double* __restrict__ rowsum,
                                                       TsseW w, float* __restrict__ scale, float* __restrict__ ns_out,
                                                       float* __restrict__ pre_out, float* __restrict__ sq_out,
                                                       float* __restrict__ h1_out, float* __restrict__ sg_out, int C, int C2,
                                                       int T, int la) {
  __shared__ double red[4];
  __shared__ float sq[TSSE_MAXC];
  __shared__ float h1[TSSE_MAXC / 2];
  __shared__ float ns_s;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int Tp = T + la;
  double part = 0.0;
  for (int c = tid; c < C; c += 256) part += rowsum[(size_t)b * C + c];
  part = wave_sum(part);
  if ((tid & 63) == 0) red[tid >> 6] = part;
  __syncthreads();
  if (tid == 0) {
    const float mu = (float)((red[0] + red[1] + red[2] + red[3]) / ((double)C * Tp));
    ns_s = 1.0f / (mu + 1e-5f);
    if (ns_out) ns_out[b] = ns_s;
  }
  __syncthreads();
  const float ns = ns_s;
  for (int c = tid; c < C; c += 256) {
    const float* xr = x + ((size_t)b * C + c) * T;
    const double tot = rowsum[(size_t)b * C + c];
    float acc = w.fcb[0];
    for (int i = 0; i < 3; ++i) {
      const int ks = w.ks[i];
      const int Lout = Tp - ks + 1;
      double conv = 0.0;
      for (int k = 0; k < ks; ++k) {
        double pre = 0.0, suf = 0.0;
        for (int t = 0; t < k; ++t) pre += (t < T) ? (double)xr[t] : 0.0;
        for (int m = 0; m < ks - 1 - k; ++m) {
          const int t = Tp - 1 - m;
          suf += (t < T && t >= 0) ? (double)xr[t] : 0.0;
        }
        conv += (double)w.cw[i][c * ks + k] * (tot - pre - suf);
      }
      const float pv = w.cb[i][c] + ns * (float)(conv / Lout);
      if (pre_out) pre_out[((size_t)b * C + c) * 3 + i] = pv;
      acc += w.fcw[i] * fmaxf(pv, 0.f);
    }
    sq[c] = acc;
    if (sq_out) sq_out[(size_t)b * C + c] = acc;
  }
  __syncthreads();
  for (int j = tid; j < C2; j += 256) {
    float a = w.b1[j];
    const float* wr = w.w1 + (size_t)j * C;
    for (int c = 0; c < C; ++c) a += wr[c] * sq[c];
    a = fmaxf(a, 0.f);
    h1[j] = a;
    if (h1_out) h1_out[(size_t)b * C2 + j] = a;
  }
  __syncthreads();
  for (int c = tid; c < C; c += 256) {
    float a = w.b2[c];
    const float* wr = w.w2 + (size_t)c * C2;
    for (int j = 0; j < C2; ++j) a += wr[j] * h1[j];
    const float sg = 1.0f / (1.0f + expf(-a));
    if (sg_out) sg_out[(size_t)b * C + c] = sg;
    scale[(size_t)b * C + c] = ns * sg;
  }
}

// y[b][t][coff + c] = x[b][c][t] * scale[b][c]   for t < T  (rows T..Tp-1 and pad columns stay as the caller left them: zero)
template <typename T>
__global__ __launch_bounds__(256) void scale_transpose_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                              T* __restrict__ y, int C, int Tn, int Tp, int ld, int coff) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, c0 = blockIdx.y * 32, t0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, t = t0 + tx;
    float v = 0.f;
    if (c < C && t < Tn) {
      v = x[((size_t)b * C + c) * Tn + t];
      if (scale) v *= scale[(size_t)b * C + c];
    }
    tile[i][tx] = v;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int t = t0 + i, c = c0 + tx;
    if (t < Tn && c < C) y[((size_t)b * Tp + t) * ld + coff + c] = from_f32<T>(tile[tx][i]);
  }
}

}  // namespace

extern "C" {

int nppc_rowsum(const float* x, double* sums, long R, int T, void* stream) {
  if (!x || !sums || R <= 0 || T <= 0) return NPPC_EBADARG;
  hipLaunchKernelGGL(rowsum_kernel, dim3(ceil_div(R, 4)), dim3(256), 0, (hipStream_t)stream, x, sums, R, T);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

int nppc_tsse_fwd(const float* x, const double* rowsum, const float* cw0, const float* cb0, const float* cw1,
                  const float* cb1, const float* cw2, const float* cb2, int ks0, int ks1, int ks2, const float* fcw,
                  const float* fcb, const float* w1, const float* b1, const float* w2, const float* b2, float* scale,
                  float* ns, float* pre, float* sq, float* h1, float* sg, int B, int C, int T, int look_ahead,
                  void* stream) {
  if (!x || !rowsum || !scale || B <= 0 || C <= 0 || C > TSSE_MAXC) return NPPC_EBADARG;
  if (ks0 > T || ks1 > T || ks2 > T) return NPPC_EUNSUPPORTED;
  TsseW w{{cw0, cw1, cw2}, {cb0, cb1, cb2}, {ks0, ks1, ks2}, fcw, fcb, w1, b1, w2, b2};
  hipLaunchKernelGGL(tsse_fwd_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, x, rowsum, w, scale, ns, pre, sq, h1, sg,
                     C, C / 2, T, look_ahead);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

int nppc_scale_transpose(int prec, const float* x, const float* scale, void* y, int B, int C, int T, int Tp, int ld,
                         int coff, void* stream) {
  if (!x || !y || B <= 0 || coff + C > ld || T > Tp) return NPPC_EBADARG;
  dim3 grid(ceil_div(T, 32), ceil_div(C, 32), B);
  if (prec == NPPC_PREC_BF16)
    hipLaunchKernelGGL(scale_transpose_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, x, scale, (bf16_t*)y, C, T,
                       Tp, ld, coff);
  else if (prec == NPPC_PREC_F32)
    hipLaunchKernelGGL(scale_transpose_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, x, scale, (float*)y, C, T, Tp,
                       ld, coff);
  else
    return NPPC_EBADARG;
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

}  // extern "C"
