// Sub-band stage: unfold (N neighbour bins, reflect pad) + concat with the three full-band outputs
// + offline laplace norm + drop-band + re-layout into the time-major LSTM input, and the LSTM output head
// (Linear(H -> O) + [B*F',O,T'] -> [B,O,F',T] re-layout with the look-ahead crop).
// Reference: audio_zen/model/base_model.py:15-46 (unfold), fullsubnet_plus.py:188-230,
//            nppc_audio/networks.py:115-161, audio_zen/acoustics/feature.py:254-285 (drop_band),
//            sequence_model.py:118-123 (fc_output_layer on the LSTM output).
#include "common.h"
#include "nppc_hip.h"

namespace {

__device__ __forceinline__ int reflect_idx(int i, int F) {
  if (i < 0) i = -i;
  if (i >= F) i = 2 * (F - 1) - i;
  return i;
}

// scale[b] = 1 / (mean over (F, 2n+1+3, T') of the concatenated sub-band input + 1e-5)
// sum of the unfolded part = sum_f mult[f] * src[f]  (mult = how many windows contain bin f after reflection)
// grid (SM_CHUNKS, B): every workgroup sums a slice of the frames; the last one to arrive for sample b (arrival counter)
// turns the fp64 total into the scale and re-arms the accumulators (one workgroup per sample left 224 CUs idle).
// Accumulators and counters live in the CALLER's workspace `work` (2*B 8-byte words: B fp64 sums, then B counters,
// zero-initialised once by the caller), so launches on different workspaces may overlap freely.
constexpr int SM_CHUNKS = 16;

template <typename T>
__global__ __launch_bounds__(256) void subband_mean_kernel(const T* __restrict__ src, int ldS, const T* __restrict__ fb,
                                                           int ldF, long strideFb, const float* __restrict__ mult,
                                                           float* __restrict__ scale, double* __restrict__ work, int F,
                                                           int Tp, int Tv, int nfeat) {
  __shared__ double red[4];
  double* g_sm_acc = work;
  unsigned* g_sm_cnt = reinterpret_cast<unsigned*>(work + gridDim.y);
  const int b = blockIdx.y, tid = threadIdx.x;
  const int tper = (Tv + gridDim.x - 1) / gridDim.x;
  const int ta = blockIdx.x * tper, tb = ta + tper < Tv ? ta + tper : Tv;
  double s = 0.0;
  const long total = (long)(tb > ta ? tb - ta : 0) * F;
  for (long e = tid; e < total; e += 256) {
    const int t = ta + (int)(e / F), f = (int)(e % F);
    const size_t row = (size_t)b * Tp + t;
    float v = mult[f] * to_f32<T>(src[row * ldS + f]);
    v += to_f32<T>(fb[row * ldF + f]) + to_f32<T>(fb[strideFb + row * ldF + f]) + to_f32<T>(fb[2 * strideFb + row * ldF + f]);
    s += (double)v;
  }
  s = wave_sum(s);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  if (tid == 0) {
    atomicAdd(&g_sm_acc[b], red[0] + red[1] + red[2] + red[3]);
    __threadfence();
    if (atomicAdd(&g_sm_cnt[2 * b], 1u) == gridDim.x - 1) {
      const double tot = atomicAdd(&g_sm_acc[b], 0.0);
      const float mu = (float)(tot / ((double)F * nfeat * Tv));
      scale[b] = 1.0f / (mu + 1e-5f);
      g_sm_acc[b] = 0.0;
      g_sm_cnt[2 * b] = 0u;
      __threadfence();
    }
  }
}

// x_tm[t][n][j],  n = bo*Fo + fo (drop-band order), j < 2n+1: scale*src[refl(f+j-nb)], then 3 full-band values, then 0 pad
template <typename T>
__global__ __launch_bounds__(256) void subband_stage_kernel(const T* __restrict__ src, int ldS, const T* __restrict__ fb,
                                                            int ldF, long strideFb, const float* __restrict__ scale,
                                                            T* __restrict__ x, int B, int F, int Tp, int nb, int G, int Fo,
                                                            int KX, long Nseq, int ones_col) {
  extern __shared__ float sm[];   // [4][F]
  const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const size_t row = (size_t)b * Tp + t;
  for (int f = tid; f < F; f += 256) {
    sm[f] = to_f32<T>(src[row * ldS + f]);
    sm[F + f] = to_f32<T>(fb[row * ldF + f]);
    sm[2 * F + f] = to_f32<T>(fb[strideFb + row * ldF + f]);
    sm[3 * F + f] = to_f32<T>(fb[2 * strideFb + row * ldF + f]);
  }
  __syncthreads();
  const float sc = scale[b];
  int g = 0, bo = b;
  if (G > 1) {
    g = b % G;
    int start = 0;
    for (int gg = 0; gg < g; ++gg) start += (B - gg + G - 1) / G;
    bo = start + b / G;
  }
  const int W = 2 * nb + 1;
  const int cpr = KX / 8;
  for (int e = tid; e < Fo * cpr; e += 256) {
    const int fo = e / cpr, j0 = (e % cpr) * 8;
    const int f = G > 1 ? g + fo * G : fo;
    T v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int j = j0 + i;
      float val = 0.f;
      if (j < W) val = sm[reflect_idx(f + j - nb, F)] * sc;
      else if (j < W + 3) val = sm[(j - W + 1) * F + f] * sc;
      else if (j == W + 3 && ones_col) val = 1.f;   // first padded column: multiplies zero weights in the forward, and
                                                     // makes its column of dgates^T * x the bias gradient in the backward
      v[i] = from_f32<T>(val);
    }
    T* po = x + ((size_t)t * Nseq + (size_t)bo * Fo + fo) * KX + j0;
#pragma unroll
    for (int i = 0; i < 8; ++i) po[i] = v[i];
  }
}

// out[bo][o][fo][t - la] = bias[o] + sum_u h2[t][n][u] * Wh[o][u],  n = bo*Fo + fo, t >= la
// One wave = 16 rows (t fixed, 16 consecutive n) x 16 outputs; K = Hd in steps of 32.
template <typename T>
__global__ __launch_bounds__(256) void head_kernel(const T* __restrict__ h2, const T* __restrict__ wh /*[Opad][Hd]*/,
                                                   const float* __restrict__ bias, float* __restrict__ out, long Nseq,
                                                   int Tn, int la, int Hd, int O, int Fo) {
  typedef typename Frag<T>::type frag;
  const int lane = threadIdx.x & 63, n = lane & 15, q = lane >> 4;
  const long tiles_per_t = (Nseq + 15) / 16;
  const long tile = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tile >= tiles_per_t * (Tn - la)) return;
  const int t = la + (int)(tile / tiles_per_t);
  const long n0 = (tile % tiles_per_t) * 16;
  long arow = n0 + n;
  if (arow >= Nseq) arow = Nseq - 1;   // clamp: duplicated rows are never stored
  const T* ap = h2 + ((size_t)t * Nseq + arow) * Hd + 8 * q;
  const int To = Tn - la;
  for (int oc = 0; oc < O; oc += 16) {
    const T* bp = wh + (size_t)(oc + n) * Hd + 8 * q;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < Hd; k += 32) acc = mma16(load_frag<T>(ap + k), load_frag<T>(bp + k), acc);
    const int o = oc + n;
    if (o < O) {
      const float bv = bias[o];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const long nn = n0 + 4 * q + j;
        if (nn < Nseq) {
          const long bo = nn / Fo, fo = nn % Fo;
          out[((bo * O + o) * Fo + fo) * To + (t - la)] = acc[j] + bv;
        }
      }
    }
  }
}

// generic fallback (any Hd): one thread per output element
template <typename T>
__global__ void head_scalar_kernel(const T* __restrict__ h2, const T* __restrict__ wh, const float* __restrict__ bias,
                                   float* __restrict__ out, long Nseq, int Tn, int la, int Hd, int O, int Fo) {
  const int To = Tn - la;
  const long total = Nseq * O * To;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int to = (int)(e % To);
    long r = e / To;
    const long fo = r % Fo; r /= Fo;
    const int o = (int)(r % O);
    const long bo = r / O;
    const T* hp = h2 + ((size_t)(to + la) * Nseq + bo * Fo + fo) * Hd;
    float a = bias[o];
    for (int u = 0; u < Hd; ++u) a += to_f32<T>(hp[u]) * to_f32<T>(wh[(size_t)o * Hd + u]);
    out[e] = a;
  }
}

// out[bo][o][fo][t - la] = bias[o] + sum_g hpart[g][t][n][o]   (partial sums of the head fused into the cooperative LSTM
// forward, one per CU of a pair).  32 frames x ns sequences (ns * O <= 128 columns) per workgroup through an LDS tile:
// reads run along (n, o), writes along t.
constexpr int HF_COLS = 128;
__global__ __launch_bounds__(256) void head_finalize_kernel(const float* __restrict__ hpart, int G, const float* __restrict__ bias,
                                                            float* __restrict__ out, long Nseq, int Tn, int la, int O, int Fo,
                                                            int ns) {
  __shared__ float tile[32][HF_COLS + 1];
  const long n0 = (long)blockIdx.x * ns;
  const int t0 = la + blockIdx.y * 32, To = Tn - la, w = ns * O;
  for (int e = threadIdx.x; e < 32 * w; e += 256) {
    const int tt = e / w, c = e % w;                         // c = n_local * O + o
    const long nn = n0 + c / O;
    float v = 0.f;
    if (t0 + tt < Tn && nn < Nseq)
      for (int g = 0; g < G; ++g) v += hpart[(((size_t)g * Tn + t0 + tt) * Nseq + n0) * O + c];
    tile[tt][c] = v;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 32 * w; e += 256) {
    const int tt = e % 32, c = e / 32;
    const long nn = n0 + c / O;
    const int o = c % O;
    if (t0 + tt < Tn && nn < Nseq) {
      const long bo = nn / Fo, fo = nn % Fo;
      out[((bo * O + o) * Fo + fo) * To + (t0 + tt - la)] = tile[tt][c] + bias[o];
    }
  }
}

}  // namespace

extern "C" {

int nppc_sb_head_finalize(const float* hpart, int G, const float* bias, float* out, long Nseq, int Tn, int la, int O, int Fo,
                          void* stream) {
  if (!hpart || !bias || !out || G < 1 || Nseq <= 0 || Tn <= la || O < 1 || Nseq % Fo) return NPPC_EBADARG;
  if (O > 16) return NPPC_EUNSUPPORTED;
  const int ns = HF_COLS / O;
  hipLaunchKernelGGL(head_finalize_kernel, dim3(ceil_div(Nseq, (long)ns), ceil_div(Tn - la, 32)), dim3(256), 0,
                     (hipStream_t)stream, hpart, G, bias, out, Nseq, Tn, la, O, Fo, ns);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}


int nppc_subband_mean(int prec, const void* src, int ldS, const void* fb, int ldF, long strideFb, const float* mult,
                      float* scale, double* work, int B, int F, int Tp, int Tv, int nfeat, void* stream) {
  if (!src || !fb || !mult || !scale || !work || B <= 0) return NPPC_EBADARG;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(Tv < SM_CHUNKS ? Tv : SM_CHUNKS, B);
  if (prec == NPPC_PREC_BF16)
    hipLaunchKernelGGL(subband_mean_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)src, ldS, (const bf16_t*)fb, ldF,
                       strideFb, mult, scale, work, F, Tp, Tv, nfeat);
  else
    hipLaunchKernelGGL(subband_mean_kernel<float>, grid, dim3(256), 0, s, (const float*)src, ldS, (const float*)fb, ldF,
                       strideFb, mult, scale, work, F, Tp, Tv, nfeat);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

int nppc_subband_stage(int prec, const void* src, int ldS, const void* fb, int ldF, long strideFb, const float* scale,
                       void* x, int B, int F, int Tp, int Tv, int nb, int G, int KX, int ones_col, void* stream) {
  if (!src || !fb || !scale || !x || B <= 0 || G < 1 || 2 * nb + 4 + (ones_col ? 1 : 0) > KX || KX % 8) return NPPC_EBADARG;
  if (B > 1 && !(B > G)) return NPPC_EBADARG;
  const int Geff = B > 1 ? G : 1;                    // the reference applies drop_band only when batch > 1
  const int Fo = Geff <= 1 ? F : (F - F % Geff) / Geff;
  const long Nseq = (long)B * Fo;
  dim3 grid(Tv, B);
  const size_t sm = (size_t)4 * F * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
  if (prec == NPPC_PREC_BF16)
    hipLaunchKernelGGL(subband_stage_kernel<bf16_t>, grid, dim3(256), sm, s, (const bf16_t*)src, ldS, (const bf16_t*)fb, ldF,
                       strideFb, scale, (bf16_t*)x, B, F, Tp, nb, Geff, Fo, KX, Nseq, ones_col);
  else
    hipLaunchKernelGGL(subband_stage_kernel<float>, grid, dim3(256), sm, s, (const float*)src, ldS, (const float*)fb, ldF,
                       strideFb, scale, (float*)x, B, F, Tp, nb, Geff, Fo, KX, Nseq, ones_col);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

int nppc_sb_head(int prec, const void* h2, const void* wh, const float* bias, float* out, long Nseq, int Tn, int la,
                 int Hd, int O, int Fo, void* stream) {
  if (!h2 || !wh || !bias || !out || Nseq <= 0 || Tn <= la || O < 1 || Nseq % Fo) return NPPC_EBADARG;
  const long tiles = ((Nseq + 15) / 16) * (Tn - la);
  hipStream_t s = (hipStream_t)stream;
  if (Hd % 32) {
    const long total = Nseq * O * (Tn - la);
    const int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    if (prec == NPPC_PREC_BF16)
      hipLaunchKernelGGL(head_scalar_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)h2, (const bf16_t*)wh, bias,
                         out, Nseq, Tn, la, Hd, O, Fo);
    else
      hipLaunchKernelGGL(head_scalar_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)h2, (const float*)wh, bias, out,
                         Nseq, Tn, la, Hd, O, Fo);
    NPPC_CHECK_LAUNCH();
    return NPPC_OK;
  }
  if (prec == NPPC_PREC_BF16)
    hipLaunchKernelGGL(head_kernel<bf16_t>, dim3(ceil_div(tiles, 4)), dim3(256), 0, s, (const bf16_t*)h2, (const bf16_t*)wh,
                       bias, out, Nseq, Tn, la, Hd, O, Fo);
  else
    hipLaunchKernelGGL(head_kernel<float>, dim3(ceil_div(tiles, 4)), dim3(256), 0, s, (const float*)h2, (const float*)wh, bias,
                       out, Nseq, Tn, la, Hd, O, Fo);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

}  // extern "C"
