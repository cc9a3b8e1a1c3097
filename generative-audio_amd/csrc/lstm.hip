// Sub-band sequence model: fused 2-layer LSTM recurrence for B*F' independent sequences.
// Replaces nn.LSTM(34,384,num_layers=2,batch_first) inside SequenceModel
// (reference: FullSubNet_plus/speech_enhance/audio_zen/model/module/sequence_model.py:30-37,113-123).
//
// Design (gfx950): one workgroup owns M = 16*MT sequences for ALL time steps and BOTH layers.
//   * activations [x_t | h1 | h2] live in LDS as MFMA A-operands (double-buffered, 1 barrier/step)
//   * each wave owns 16*UB hidden units and all four gates of them -> the LSTM cell update happens
//     in the accumulator registers; cell state c never leaves registers
//   * weights are pre-packed into per-wave MFMA B-fragment order and streamed L2 -> VGPR
//     (1 KiB fully coalesced per wave-load); nothing but h/c/gates (train) goes to HBM
// Layout of every [Tn][N][...] tensor is time-major so one (step, tile) block is contiguous.
#include "common.h"
#include "nppc_hip.h"

namespace {

struct LstmFwdArgs {
  const void* x;      // [Tn][N][KX]  T
  const void* wp1;    // packed layer-1 weights
  const void* wp2;    // packed layer-2 weights
  const float* bias1; // [4][H]  b_ih + b_hh
  const float* bias2;
  void* h2;           // [Tn][N][H]  T   (always written)
  void* h1;           // [Tn][N][H]  T   (train: layer-1 hidden state, input of the weight-gradient GEMMs)
  void* h2T_unused;
  void* g1;           // [Tn][N][H][4] T (train; post-activation, order i,g,f,o)
  void* g2;
  void* c1;           // [Tn][N][H]  T   (train)
  void* c2;
  long N;
  int Tn;
  int KX;
  long unused_np;
};

// One gate PAIR (gp = 0: i,g   gp = 1: f,o) of one layer:  acc[ub][s][mt] += A[16mt.., koff..] * Wpair
// A is one contiguous K range of the LDS row (see the layout in lstm2_fwd_kernel).
// Packed B fragments of this wave/pair: w_lane + (kk*UB*2 + ub*2 + s)*512.  nk is even.
template <typename T, int UB, int MT, int RS, int DEPTH>
__device__ __forceinline__ void pair_gemm(f32x4 (&acc)[UB][2][MT], const T* a_lane, int nk, __amdgpu_buffer_rsrc_t wr,
                                          int wave_boff /*uniform byte offset of this wave's (pair) stream*/, int lane) {
  typedef typename Frag<T>::type frag;
  constexpr int FB = 512 * (int)sizeof(T);  // bytes per packed fragment
  auto loadb = [&](frag(&b)[UB * 2], int kk) {
#pragma unroll
    for (int i = 0; i < UB * 2; ++i) b[i] = BFrag<T>::load(wr, lane, wave_boff + (kk * UB * 2 + i) * FB);
  };
  auto compute = [&](const frag(&b)[UB * 2], int kk) {
    frag af[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) af[mt] = load_frag<T>(a_lane + 16 * mt * RS + 32 * kk);
#pragma unroll
    for (int ub = 0; ub < UB; ++ub)
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[ub][s][mt] = mma16(af[mt], b[ub * 2 + s], acc[ub][s][mt]);
  };
  // DEPTH-stage register ring: DEPTH-1 k-steps of weight fragments are in flight while one is consumed.  The stream
  // is bound by how many bytes a CU keeps in flight towards L2, so depth (not issue rate) sets the speed.
  frag b[DEPTH][UB * 2];
#pragma unroll
  for (int d = 0; d < DEPTH - 1; ++d)
    if (d < nk) loadb(b[d], d);
#pragma unroll 1
  for (int kk = 0; kk < nk; kk += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int kl = kk + d + DEPTH - 1;
      if (kl < nk) loadb(b[(d + DEPTH - 1) % DEPTH], kl);
      if (kk + d < nk) compute(b[d], kk + d);
    }
  }
}

template <typename T> __device__ __forceinline__ void store_pair(T* p, float a, float b);
template <> __device__ __forceinline__ void store_pair<float>(float* p, float a, float b) {
  *reinterpret_cast<float2*>(p) = make_float2(a, b);
}
template <> __device__ __forceinline__ void store_pair<bf16_t>(bf16_t* p, float a, float b) {
  *reinterpret_cast<uint32_t*>(p) = (uint32_t)f2bf(a) | ((uint32_t)f2bf(b) << 16);
}

template <typename T> __device__ __forceinline__ void store_gates(T* p, float a, float b, float c, float d);
template <> __device__ __forceinline__ void store_gates<float>(float* p, float a, float b, float c, float d) {
  *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
}
template <> __device__ __forceinline__ void store_gates<bf16_t>(bf16_t* p, float a, float b, float c, float d) {
  uint2 v;
  v.x = (uint32_t)f2bf(a) | ((uint32_t)f2bf(b) << 16);
  v.y = (uint32_t)f2bf(c) | ((uint32_t)f2bf(d) << 16);
  *reinterpret_cast<uint2*>(p) = v;
}

// One LSTM layer step for this wave's 16*UB units x M rows.  Two passes over K keep only two gate
// accumulators live: pass 0 -> i,g (ig = sigmoid(i)*tanh(g)), pass 1 -> f,o -> c, h.
// Saved gate layout per (t,row,unit): [i, g, f, o] (post-activation).
// Addressing: element (row 16mt+4q+j, unit ubase+16ub+n) = ebase + (16mt+j)*H + 16ub with
// ebase = (t*N + row0 + 4q)*H + ubase + n, so every per-element offset is a compile-time constant.
template <typename T, int UB, int MT, int RS, int H, bool SAVE, bool WRITE_H, int DEPTH>
__device__ __forceinline__ void layer_step(const float (&bias)[UB][4], f32x4 (&c)[UB][MT], const T* a_lane, int nk,
                                           __amdgpu_buffer_rsrc_t wr, int wave_boff, int pair_bstride, int lane,
                                           T* lds_h /* + 4q*RS + ubase+n */,
                                           long rbase /* row0 + 4q */, long N, size_t ebase, T* gates_out, T* c_out,
                                           T* h_out) {
  f32x4 ig[UB][MT];
  f32x4 sv_i[SAVE ? UB : 1][SAVE ? MT : 1], sv_g[SAVE ? UB : 1][SAVE ? MT : 1];   // post-activation i, g kept for one store
  {
    f32x4 acc[UB][2][MT];
#pragma unroll
    for (int ub = 0; ub < UB; ++ub)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        acc[ub][0][mt] = acc[ub][1][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    pair_gemm<T, UB, MT, RS, DEPTH>(acc, a_lane, nk, wr, wave_boff, lane);
#pragma unroll
    for (int ub = 0; ub < UB; ++ub)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float iv = sigmoid_f(acc[ub][0][mt][j] + bias[ub][0]);
          const float gv = tanh_f(acc[ub][1][mt][j] + bias[ub][2]);
          ig[ub][mt][j] = iv * gv;
          if (SAVE) {
            sv_i[ub][mt][j] = iv;
            sv_g[ub][mt][j] = gv;
          }
        }
  }
  f32x4 acc[UB][2][MT];
#pragma unroll
  for (int ub = 0; ub < UB; ++ub)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      acc[ub][0][mt] = acc[ub][1][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  pair_gemm<T, UB, MT, RS, DEPTH>(acc, a_lane, nk, wr, wave_boff + pair_bstride, lane);
#pragma unroll
  for (int ub = 0; ub < UB; ++ub)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float fv = sigmoid_f(acc[ub][0][mt][j] + bias[ub][1]);
        const float ov = sigmoid_f(acc[ub][1][mt][j] + bias[ub][3]);
        const float cn = fv * c[ub][mt][j] + ig[ub][mt][j];
        c[ub][mt][j] = cn;
        const float hn = ov * tanh_f(cn);
        lds_h[(16 * mt + j) * RS + 16 * ub] = from_f32<T>(hn);
        if ((SAVE || WRITE_H) && rbase + 16 * mt + j < N) {
          const size_t e = ebase + (16 * mt + j) * H + 16 * ub;
          if (WRITE_H) h_out[e] = from_f32<T>(hn);
          if (SAVE) {
            c_out[e] = from_f32<T>(cn);
            store_gates<T>(gates_out + e * 4, sv_i[ub][mt][j], sv_g[ub][mt][j], fv, ov);
          }
        }
      }
}

// LDS row layout (elements of T), two halves so that every GEMM reads ONE contiguous K range:
//   half0 = [X0 (KX) | H1a (HP) | H2b (HP)]   half1 = [X1 (KX) | H1b (HP) | H2a (HP)]
// step parity p:  layer 1 reads half_p[0 : KX+HP]         writes h1_t into half_{1-p}.H1
//                 layer 2 reads half_{1-p}[KX : KX+2HP]   writes h2_t into half_p.H2
// so one barrier per step orders everything (see the comment at the end of the loop).
template <typename T, int UB, int NW, int KX, int MT, bool TRAIN>
__global__ __launch_bounds__(NW * 64) void lstm2_fwd_kernel(LstmFwdArgs a) {
  constexpr int H = 16 * UB * NW;
  constexpr int HP = (H + 31) / 32 * 32;
  constexpr int M = 16 * MT;
  constexpr int NT = NW * 64;
  constexpr int VEC = 16 / (int)sizeof(T);  // elements per 16-byte chunk
  constexpr int HALF = KX + 2 * HP;
  constexpr int RS = 2 * HALF + VEC;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* lds = reinterpret_cast<T*>(smem_raw);

  const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long row0 = (long)blockIdx.x * M;
  const long N = a.N;
  const int ubase_n = wave * 16 * UB + n;

  for (int i = tid; i < M * RS; i += NT) lds[i] = from_f32<T>(0.f);

  float b1[UB][4], b2[UB][4];
#pragma unroll
  for (int ub = 0; ub < UB; ++ub)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      b1[ub][g] = a.bias1[g * H + ubase_n + 16 * ub];
      b2[ub][g] = a.bias2[g * H + ubase_n + 16 * ub];
    }
  f32x4 c1[UB][MT], c2[UB][MT];
#pragma unroll
  for (int ub = 0; ub < UB; ++ub)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) c1[ub][mt] = c2[ub][mt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const T* xg = reinterpret_cast<const T*>(a.x);
  constexpr int cpr = KX / VEC;  // 16-byte chunks per row of x
  constexpr int nchunk = M * cpr;
  constexpr int XCH = (nchunk + NT - 1) / NT;  // x-tile chunks per thread
  __syncthreads();
  for (int ch = tid; ch < nchunk; ch += NT) {  // x_0 -> X0
    const int r = ch / cpr, cc = ch % cpr;
    if (row0 + r < N)
      *reinterpret_cast<uint4*>(lds + r * RS + cc * VEC) =
          *reinterpret_cast<const uint4*>(xg + ((size_t)(row0 + r)) * KX + cc * VEC);
  }
  __syncthreads();

  constexpr int nk1 = (KX + HP) / 32, nk2 = 2 * HP / 32;
  // weight-fragment prefetch depth: as deep as the register budget of the variant allows
  constexpr int DEPTH = (sizeof(T) == 2 && MT == 1) ? 4 : 2;
  constexpr int ps1 = NW * nk1 * UB * 2 * 512 * (int)sizeof(T);  // gate-pair stride (bytes)
  constexpr int ps2 = NW * nk2 * UB * 2 * 512 * (int)sizeof(T);
  const __amdgpu_buffer_rsrc_t wr1 = make_rsrc(a.wp1, 2u * ps1);
  const __amdgpu_buffer_rsrc_t wr2 = make_rsrc(a.wp2, 2u * ps2);
  const int wb1 = wave * (nk1 * UB * 2 * 512 * (int)sizeof(T));
  const int wb2 = wave * (nk2 * UB * 2 * 512 * (int)sizeof(T));
  const T* a_lane = lds + n * RS + 8 * q;
  T* hw_lane = lds + 4 * q * RS + ubase_n;
  const long rbase = row0 + 4 * q;
  T* h1o = reinterpret_cast<T*>(a.h1);
  T* h2o = reinterpret_cast<T*>(a.h2);
  T* g1o = reinterpret_cast<T*>(a.g1);
  T* g2o = reinterpret_cast<T*>(a.g2);
  T* c1o = reinterpret_cast<T*>(a.c1);
  T* c2o = reinterpret_cast<T*>(a.c2);

#pragma unroll 1
  for (int t = 0; t < a.Tn; ++t) {
    const int p = t & 1;
    const size_t ebase = ((size_t)t * N + rbase) * H + ubase_n;
    const int cur = p * HALF, oth = (1 - p) * HALF;

    // prefetch x_{t+1} into registers; written to LDS after the layer-1 GEMM
    uint4 xr[XCH];
    const bool more = t + 1 < a.Tn;
    if (more) {
#pragma unroll
      for (int u = 0; u < XCH; ++u) {
        const int ch = tid + u * NT;
        xr[u] = make_uint4(0, 0, 0, 0);
        if (ch < nchunk) {
          const int r = ch / cpr, cc = ch % cpr;
          if (row0 + r < N)
            xr[u] = *reinterpret_cast<const uint4*>(xg + ((size_t)(t + 1) * N + row0 + r) * KX + cc * VEC);
        }
      }
    }
    // layer 1: [x_t | h1_{t-1}]  ->  h1_t into the other half
    layer_step<T, UB, MT, RS, H, TRAIN, TRAIN, DEPTH>(b1, c1, a_lane + cur, nk1, wr1, wb1, ps1, lane, hw_lane + oth + KX, rbase, N, ebase,
                                               g1o, c1o, h1o);
    if (more) {
#pragma unroll
      for (int u = 0; u < XCH; ++u) {
        const int ch = tid + u * NT;
        if (ch < nchunk) {
          const int r = ch / cpr, cc = ch % cpr;
          *reinterpret_cast<uint4*>(lds + r * RS + oth + cc * VEC) = xr[u];
        }
      }
    }
    __syncthreads();
    // layer 2: [h1_t | h2_{t-1}] (other half)  ->  h2_t into this half
    layer_step<T, UB, MT, RS, H, TRAIN, true, DEPTH>(b2, c2, a_lane + oth + KX, nk2, wr2, wb2, ps2, lane, hw_lane + cur + KX + HP, rbase, N,
                                              ebase, g2o, c2o, h2o);
    // One barrier per step suffices: what step t+1 writes before its barrier (h1 -> half_p.H1,
    // x_{t+2} -> half_p.X, and after it h2 -> half_{1-p}.H2) was last READ before the barrier above
    // (layer 1 of step t) or in layer 2 of step t, which every wave finishes before it can reach
    // the barrier of step t+1; what step t+1 reads was written before a barrier it has passed.
  }
}

// ---------------------------------------------------------------- weight packing
// Packed element (gate pair gp, wave w, k-step kk, unit-block ub, s, lane l, j):
//   gate = {i,g | f,o}[gp][s]  (torch row blocks i=0,f=1,g=2,o=3);  column = gate*H + w*16*UB + 16*ub + (l&15)
//   k = 32*kk + 8*(l>>4) + j;  k < K0 -> W_ih[col][k] (0 for k >= I);  else W_hh[col][k-K0] (0 beyond H)
template <typename T>
__global__ void lstm_pack_kernel(const float* __restrict__ w_ih, const float* __restrict__ w_hh, T* __restrict__ out,
                                 int I, int H, int K0 /*padded width of the ih segment*/, int HP, int UB, int NW) {
  const int nk = (K0 + HP) / 32;
  const size_t per_pair = (size_t)NW * nk * UB * 2 * 512;
  const size_t total = 2 * per_pair;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int gp = (int)(e / per_pair);
    size_t r = e % per_pair;
    const int j = r & 7;
    const int l = (r >> 3) & 63;
    size_t f = r >> 9;
    const int sidx = f & 1; f >>= 1;
    const int ub = f % UB; f /= UB;
    const int kk = f % nk;
    const int w = (int)(f / nk);
    const int gate = gp == 0 ? (sidx == 0 ? 0 : 2) : (sidx == 0 ? 1 : 3);
    const int col = gate * H + w * 16 * UB + 16 * ub + (l & 15);
    const int k = 32 * kk + 8 * (l >> 4) + j;
    float v = 0.f;
    if (k < K0) {
      if (k < I) v = w_ih[(size_t)col * I + k];
    } else {
      const int kh = k - K0;
      if (kh < H) v = w_hh[(size_t)col * H + kh];
    }
    out[e] = from_f32<T>(v);
  }
}

__global__ void bias_sum_kernel(const float* a, const float* b, float* o, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) o[i] = a[i] + b[i];
}

struct HCfg { int UB, NW; };
// x segment width: multiple of 32 and such that the layer-1 K-step count (KX+HP)/32 is even
static int kx_for(int I, int H) {
  int kx = round_up(I, 32);
  if (((kx + round_up(H, 32)) / 32) & 1) kx += 32;
  return kx;
}
static bool hidden_cfg(int H, HCfg* c) {
  if (H == 384) { *c = {3, 8}; return true; }
  if (H == 16) { *c = {1, 1}; return true; }
  return false;
}

template <typename T, int UB, int NW, int KX, int MT, bool TRAIN>
static int launch_fwd(const LstmFwdArgs& a, hipStream_t s) {
  constexpr int H = 16 * UB * NW, HP = (H + 31) / 32 * 32, M = 16 * MT;
  constexpr int RS = 2 * (KX + 2 * HP) + 16 / (int)sizeof(T);
  constexpr size_t smem = (size_t)M * RS * sizeof(T);
  static_assert(smem <= 160 * 1024, "LDS budget");
  if (a.KX != KX) return NPPC_EUNSUPPORTED;
  auto k = lstm2_fwd_kernel<T, UB, NW, KX, MT, TRAIN>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) !=
      hipSuccess)
    return NPPC_ELAUNCH;
  const int grid = ceil_div(a.N, M);
  hipLaunchKernelGGL(k, dim3(grid), dim3(NW * 64), smem, s, a);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

template <typename T, int UB, int NW, int KX>
static int dispatch_fwd(const LstmFwdArgs& a, int mt, int train, hipStream_t s) {
  if (mt == 1) return train ? launch_fwd<T, UB, NW, KX, 1, true>(a, s) : launch_fwd<T, UB, NW, KX, 1, false>(a, s);
  if constexpr (sizeof(T) == 2) {
    if (mt == 2) return train ? launch_fwd<T, UB, NW, KX, 2, true>(a, s) : launch_fwd<T, UB, NW, KX, 2, false>(a, s);
    if (mt == 3 && !train) return launch_fwd<T, UB, NW, KX, 3, false>(a, s);
  }
  return NPPC_EUNSUPPORTED;
}

}  // namespace

extern "C" {

int nppc_lstm2_packed_elems(int I, int H, long* n1, long* n2, int* kx) {
  HCfg c;
  if (!hidden_cfg(H, &c) || I < 1) return NPPC_EUNSUPPORTED;
  const int KX = kx_for(I, H), HP = round_up(H, 32);
  *kx = KX;
  *n1 = (long)c.NW * ((KX + HP) / 32) * c.UB * 4 * 512;
  *n2 = (long)c.NW * ((2 * HP) / 32) * c.UB * 4 * 512;
  return NPPC_OK;
}

int nppc_lstm2_pack_weights(int prec, const float* w_ih0, const float* w_hh0, const float* b_ih0, const float* b_hh0,
                            const float* w_ih1, const float* w_hh1, const float* b_ih1, const float* b_hh1, int I, int H,
                            void* wp1, void* wp2, float* bias1, float* bias2, void* stream) {
  HCfg c;
  if (!hidden_cfg(H, &c)) return NPPC_EUNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const int KX = kx_for(I, H), HP = round_up(H, 32);
  const int blk = 256, grid = 512;
  if (prec == NPPC_PREC_BF16) {
    hipLaunchKernelGGL(lstm_pack_kernel<bf16_t>, dim3(grid), dim3(blk), 0, s, w_ih0, w_hh0, (bf16_t*)wp1, I, H, KX, HP,
                       c.UB, c.NW);
    hipLaunchKernelGGL(lstm_pack_kernel<bf16_t>, dim3(grid), dim3(blk), 0, s, w_ih1, w_hh1, (bf16_t*)wp2, H, H, HP, HP,
                       c.UB, c.NW);
  } else if (prec == NPPC_PREC_F32) {
    hipLaunchKernelGGL(lstm_pack_kernel<float>, dim3(grid), dim3(blk), 0, s, w_ih0, w_hh0, (float*)wp1, I, H, KX, HP,
                       c.UB, c.NW);
    hipLaunchKernelGGL(lstm_pack_kernel<float>, dim3(grid), dim3(blk), 0, s, w_ih1, w_hh1, (float*)wp2, H, H, HP, HP,
                       c.UB, c.NW);
  } else {
    return NPPC_EBADARG;
  }
  hipLaunchKernelGGL(bias_sum_kernel, dim3(ceil_div(4 * H, 256)), dim3(256), 0, s, b_ih0, b_hh0, bias1, 4 * H);
  hipLaunchKernelGGL(bias_sum_kernel, dim3(ceil_div(4 * H, 256)), dim3(256), 0, s, b_ih1, b_hh1, bias2, 4 * H);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

int nppc_lstm2_fwd(int prec, int train, int mtile, const void* x, const void* wp1, const void* wp2, const float* bias1,
                   const float* bias2, void* h2, void* h1, void* g1, void* g2, void* c1, void* c2, long N, int Tn, int I,
                   int H, void* stream) {
  HCfg c;
  if (!hidden_cfg(H, &c)) return NPPC_EUNSUPPORTED;
  if (N <= 0 || Tn <= 0 || !x || !wp1 || !wp2 || !h2) return NPPC_EBADARG;
  if (train && (!h1 || !g1 || !g2 || !c1 || !c2)) return NPPC_EBADARG;
  LstmFwdArgs a{x, wp1, wp2, bias1, bias2, h2, h1, nullptr, g1, g2, c1, c2, N, Tn, kx_for(I, H), N};
  hipStream_t s = (hipStream_t)stream;
  if (prec == NPPC_PREC_BF16) {
    if (H == 384) return dispatch_fwd<bf16_t, 3, 8, 64>(a, mtile, train, s);
    return dispatch_fwd<bf16_t, 1, 1, 32>(a, mtile, train, s);
  } else if (prec == NPPC_PREC_F32) {
    if (H == 384) return dispatch_fwd<float, 3, 8, 64>(a, mtile, train, s);
    return dispatch_fwd<float, 1, 1, 32>(a, mtile, train, s);
  }
  return NPPC_EBADARG;
}

}  // extern "C"

// =====================================================================================================
// Backward recurrence (training, direction net).  One workgroup = 16 sequences, all steps (descending),
// both layers.  Per step and layer:
//   P: cell backward on (row, unit) pairs -> dgates tile [16][4H] (order per unit: i,g,f,o) into LDS (MFMA A operand)
//   G: [16][4H] x Wcat[4H][Kin] on MFMA -> d(input) | d(h_prev); the tile is also written TRANSPOSED to HBM
//      (dgT [4H][Tn*N]) so the weight-gradient GEMMs read a K-contiguous operand.
// Backward weights are packed per wave like the forward ones (B fragment: column = input feature, k = u*4+gate).
namespace {

struct LstmBwdArgs {
  const void* g1; const void* g2;   // [Tn][N][H][4]
  const void* c1; const void* c2;   // [Tn][N][H]
  const void* dh2;                  // [Tn][N][H]
  const void* wb1; const void* wb2; // packed backward weights
  void* dx;                         // [Tn][N][KX]
  void* dg1; void* dg2;             // [Tn][N][4H]  gate gradients, column k = unit*4 + gate (i,g,f,o)
  long N; int Tn;
};

template <typename T, int TPW>
__device__ __forceinline__ void bwd_gemm(f32x4 (&acc)[TPW], const T* a_lane, int nk, __amdgpu_buffer_rsrc_t wr, int wave_boff,
                                         int lane) {
  typedef typename Frag<T>::type frag;
  constexpr int FB = 512 * (int)sizeof(T);
  auto loadb = [&](frag(&b)[TPW], int kk) {
#pragma unroll
    for (int i = 0; i < TPW; ++i) b[i] = BFrag<T>::load(wr, lane, wave_boff + (kk * TPW + i) * FB);
  };
  auto compute = [&](const frag(&b)[TPW], int kk) {
    const frag af = load_frag<T>(a_lane + 32 * kk);
#pragma unroll
    for (int i = 0; i < TPW; ++i) acc[i] = mma16(af, b[i], acc[i]);
  };
  frag b0[TPW], b1[TPW];
  loadb(b0, 0);
#pragma unroll 1
  for (int kk = 0; kk < nk - 2; kk += 2) {
    loadb(b1, kk + 1);
    compute(b0, kk);
    loadb(b0, kk + 2);
    compute(b1, kk + 1);
  }
  loadb(b1, nk - 1);
  compute(b0, nk - 2);
  compute(b1, nk - 1);
}

template <typename T, int UB, int NW, int KX>
__global__ __launch_bounds__(NW * 64) void lstm2_bwd_kernel(LstmBwdArgs a) {
  constexpr int H = 16 * UB * NW;
  constexpr int HP = (H + 31) / 32 * 32;
  constexpr int NT = NW * 64;
  constexpr int M = 16;
  constexpr int TPR = NT / M;          // threads per row in the cell-backward phase
  constexpr int UPT = H / TPR;         // units per thread
  constexpr int K4 = 4 * H;            // gate columns
  constexpr int VEC = 16 / (int)sizeof(T);
  constexpr int RSA = K4 + VEC;        // A tile row stride (elements)
  constexpr int NT2 = 2 * HP / 16, NT1 = (KX + HP) / 16;
  constexpr int TPW2 = (NT2 + NW - 1) / NW, TPW1 = (NT1 + NW - 1) / NW;
  constexpr int nk = K4 / 32;
  static_assert(H % TPR == 0 && nk % 2 == 0, "shape");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* Abuf = reinterpret_cast<T*>(smem_raw);                                   // [16][RSA]
  float* dh1buf = reinterpret_cast<float*>(smem_raw + (size_t)M * RSA * sizeof(T));  // [16][HP]
  float* dhrec2 = dh1buf + M * HP;                                            // [16][HP]

  const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long row0 = (long)blockIdx.x * M;
  const long N = a.N;
  for (int i = tid; i < 2 * M * HP; i += NT) dh1buf[i] = 0.f;
  for (int i = tid; i < M * RSA; i += NT) Abuf[i] = from_f32<T>(0.f);

  const int prow = tid / TPR, u0 = (tid % TPR) * UPT;
  const bool prow_ok = row0 + prow < N;
  float dc1[UPT], dc2[UPT];
#pragma unroll
  for (int i = 0; i < UPT; ++i) dc1[i] = dc2[i] = 0.f;

  const T* g1 = reinterpret_cast<const T*>(a.g1);
  const T* g2 = reinterpret_cast<const T*>(a.g2);
  const T* c1 = reinterpret_cast<const T*>(a.c1);
  const T* c2 = reinterpret_cast<const T*>(a.c2);
  const T* dh2 = reinterpret_cast<const T*>(a.dh2);
  T* dx = reinterpret_cast<T*>(a.dx);
  T* dg1T = reinterpret_cast<T*>(a.dg1);
  T* dg2T = reinterpret_cast<T*>(a.dg2);
  const __amdgpu_buffer_rsrc_t wr1 = make_rsrc(a.wb1, (unsigned)(NW * nk * TPW1 * 512 * sizeof(T)));
  const __amdgpu_buffer_rsrc_t wr2 = make_rsrc(a.wb2, (unsigned)(NW * nk * TPW2 * 512 * sizeof(T)));
  const int wb1 = wave * (nk * TPW1 * 512 * (int)sizeof(T));
  const int wb2 = wave * (nk * TPW2 * 512 * (int)sizeof(T));
  const T* a_lane = Abuf + n * RSA + 8 * q;
  __syncthreads();

  // Saved state of one (layer, step) for this thread's (row, UPT units), fetched one phase AHEAD (during the previous
  // GEMM) so the cell-backward phase never waits on HBM.
  struct Saved {
    T g[UPT * 4];   // i,g,f,o per unit
    T ct[UPT];
    T cp[UPT];
    T dh[UPT];      // external gradient into h (layer 2 only)
  };
  auto fetch = [&](Saved& sv, const T* gs, const T* cs, const T* dh_ext, int t) {
    if (!prow_ok || t < 0) return;
    const size_t e = ((size_t)t * N + row0 + prow) * H + u0;
    constexpr int GB = UPT * 4 * (int)sizeof(T), CB = UPT * (int)sizeof(T);
    const char* gp = reinterpret_cast<const char*>(gs + e * 4);
    const char* cp_ = reinterpret_cast<const char*>(cs + e);
    if constexpr (GB % 16 == 0 && CB % 8 == 0) {
#pragma unroll
      for (int i = 0; i < GB / 16; ++i) reinterpret_cast<uint4*>(sv.g)[i] = reinterpret_cast<const uint4*>(gp)[i];
#pragma unroll
      for (int i = 0; i < CB / 8; ++i) reinterpret_cast<uint2*>(sv.ct)[i] = reinterpret_cast<const uint2*>(cp_)[i];
      if (t > 0) {
        const char* pp = reinterpret_cast<const char*>(cs + e - (size_t)N * H);
#pragma unroll
        for (int i = 0; i < CB / 8; ++i) reinterpret_cast<uint2*>(sv.cp)[i] = reinterpret_cast<const uint2*>(pp)[i];
      }
      if (dh_ext) {
        const char* dp = reinterpret_cast<const char*>(dh_ext + e);
#pragma unroll
        for (int i = 0; i < CB / 8; ++i) reinterpret_cast<uint2*>(sv.dh)[i] = reinterpret_cast<const uint2*>(dp)[i];
      }
    } else {
#pragma unroll
      for (int i = 0; i < UPT * 4; ++i) sv.g[i] = gs[e * 4 + i];
#pragma unroll
      for (int i = 0; i < UPT; ++i) {
        sv.ct[i] = cs[e + i];
        if (t > 0) sv.cp[i] = cs[e - (size_t)N * H + i];
        if (dh_ext) sv.dh[i] = dh_ext[e + i];
      }
    }
  };
  // cell backward for this thread's (row, units); dh_tot comes from LDS (+ the external grad for layer 2)
  auto cell_bwd = [&](const Saved& sv, const float* dh_lds, bool has_ext, float (&dc)[UPT], int t) {
#pragma unroll
    for (int i = 0; i < UPT; ++i) {
      float di = 0.f, dg = 0.f, df = 0.f, dO = 0.f;
      if (prow_ok) {
        const float iv = to_f32<T>(sv.g[i * 4 + 0]), gv = to_f32<T>(sv.g[i * 4 + 1]);
        const float fv = to_f32<T>(sv.g[i * 4 + 2]), ov = to_f32<T>(sv.g[i * 4 + 3]);
        const float ct = to_f32<T>(sv.ct[i]);
        const float cp = t > 0 ? to_f32<T>(sv.cp[i]) : 0.f;
        float dh = dh_lds[prow * HP + u0 + i];
        if (has_ext) dh += to_f32<T>(sv.dh[i]);
        const float tc = tanh_f(ct);
        const float dct = dh * ov * (1.f - tc * tc) + dc[i];
        dO = dh * tc * ov * (1.f - ov);
        di = dct * gv * iv * (1.f - iv);
        dg = dct * iv * (1.f - gv * gv);
        df = dct * cp * fv * (1.f - fv);
        dc[i] = dct * fv;
      }
      T* ap = Abuf + prow * RSA + (u0 + i) * 4;
      store_gates<T>(ap, di, dg, df, dO);
    }
  };
  // row-major copy of the dgates tile for the weight-gradient GEMMs: dg[(t*N + row)][4H] (fully coalesced rows)
  auto write_T = [&](T* dg, int t) {
    constexpr int CPR = K4 / VEC;                                      // 16-byte chunks per row
    for (int ch = tid; ch < M * CPR; ch += NT) {
      const int r = ch / CPR, cc = ch % CPR;
      if (row0 + r < N)
        *reinterpret_cast<uint4*>(dg + ((size_t)t * N + row0 + r) * K4 + cc * VEC) =
            *reinterpret_cast<const uint4*>(Abuf + r * RSA + cc * VEC);
    }
  };

  Saved sv2, sv1;
  fetch(sv2, g2, c2, dh2, a.Tn - 1);
#pragma unroll 1
  for (int t = a.Tn - 1; t >= 0; --t) {
    // ---- layer 2
    cell_bwd(sv2, dhrec2, true, dc2, t);
    __syncthreads();
    fetch(sv1, g1, c1, nullptr, t);          // layer-1 state of this step: lands during the layer-2 GEMM
    {
      f32x4 acc[TPW2];
#pragma unroll
      for (int i = 0; i < TPW2; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      bwd_gemm<T, TPW2>(acc, a_lane, nk, wr2, wb2, lane);
      write_T(dg2T, t);
#pragma unroll
      for (int i = 0; i < TPW2; ++i) {
        const int col = 16 * (wave * TPW2 + i) + n;
        if (wave * TPW2 + i < NT2) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int r = 4 * q + j;
            if (col < HP) dh1buf[r * HP + col] += acc[i][j];      // d h1_t from layer 2 (+ recurrent part already there)
            else dhrec2[r * HP + col - HP] = acc[i][j];           // d h2_{t-1}
          }
        }
      }
    }
    __syncthreads();
    // ---- layer 1
    cell_bwd(sv1, dh1buf, false, dc1, t);
    __syncthreads();
    fetch(sv2, g2, c2, dh2, t - 1);          // layer-2 state of the next (earlier) step: lands during the layer-1 GEMM
    {
      f32x4 acc[TPW1];
#pragma unroll
      for (int i = 0; i < TPW1; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      bwd_gemm<T, TPW1>(acc, a_lane, nk, wr1, wb1, lane);
      write_T(dg1T, t);
#pragma unroll
      for (int i = 0; i < TPW1; ++i) {
        const int col = 16 * (wave * TPW1 + i) + n;
        if (wave * TPW1 + i < NT1) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int r = 4 * q + j;
            if (col < KX) {
              if (row0 + r < N) dx[((size_t)t * N + row0 + r) * KX + col] = from_f32<T>(acc[i][j]);
            } else {
              dh1buf[r * HP + col - KX] = acc[i][j];              // d h1_{t-1} (recurrent)
            }
          }
        }
      }
    }
    __syncthreads();
  }
}

// packed backward weights: element (wave w, kk, tile i, lane l, j):
//   col = 16*(w*TPW + i) + (l&15)  (input feature: layer 1: [x (K0) | h (HP)], layer 2: [h1 (HP) | h2 (HP)])
//   k = 32*kk + 8*(l>>4) + j = u*4 + g', g' in saved order (i,g,f,o) -> torch gate block tg = {0,2,1,3}[g']
//   value = col < K0 ? W_ih[tg*H+u][col] (col < I) : W_hh[tg*H+u][col-K0] (col-K0 < H), 0 elsewhere
template <typename T>
__global__ void lstm_pack_bwd_kernel(const float* __restrict__ w_ih, const float* __restrict__ w_hh, T* __restrict__ out,
                                     int I, int H, int K0, int HP, int NW, int TPW) {
  const int nk = 4 * H / 32;
  const int ntiles = (K0 + HP) / 16;
  const size_t total = (size_t)NW * nk * TPW * 512;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int j = e & 7;
    const int l = (e >> 3) & 63;
    size_t f = e >> 9;
    const int i = f % TPW; f /= TPW;
    const int kk = f % nk;
    const int w = (int)(f / nk);
    const int tile = w * TPW + i;
    const int col = 16 * tile + (l & 15);
    const int k = 32 * kk + 8 * (l >> 4) + j;
    const int u = k >> 2, gp = k & 3;
    const int tg = gp == 0 ? 0 : (gp == 1 ? 2 : (gp == 2 ? 1 : 3));
    float v = 0.f;
    if (tile < ntiles) {
      if (col < K0) {
        if (col < I) v = w_ih[(size_t)(tg * H + u) * I + col];
      } else if (col - K0 < H) {
        v = w_hh[(size_t)(tg * H + u) * H + (col - K0)];
      }
    }
    out[e] = from_f32<T>(v);
  }
}

template <typename T, int UB, int NW, int KX>
static int launch_bwd(const LstmBwdArgs& a, hipStream_t s) {
  constexpr int H = 16 * UB * NW, HP = (H + 31) / 32 * 32;
  constexpr size_t smem = (size_t)16 * (4 * H + 16 / sizeof(T)) * sizeof(T) + (size_t)2 * 16 * HP * sizeof(float);
  static_assert(smem <= 160 * 1024, "LDS budget");
  auto k = lstm2_bwd_kernel<T, UB, NW, KX>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) !=
      hipSuccess)
    return NPPC_ELAUNCH;
  hipLaunchKernelGGL(k, dim3(ceil_div(a.N, 16)), dim3(NW * 64), smem, s, a);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

}  // namespace

extern "C" {

int nppc_lstm2_bwd_packed_elems(int I, int H, long* n1, long* n2) {
  HCfg c;
  if (!hidden_cfg(H, &c)) return NPPC_EUNSUPPORTED;
  const int KX = kx_for(I, H), HP = round_up(H, 32);
  const int nk = 4 * H / 32;
  const int tpw1 = ((KX + HP) / 16 + c.NW - 1) / c.NW, tpw2 = (2 * HP / 16 + c.NW - 1) / c.NW;
  *n1 = (long)c.NW * nk * tpw1 * 512;
  *n2 = (long)c.NW * nk * tpw2 * 512;
  return NPPC_OK;
}

int nppc_lstm2_pack_weights_bwd(int prec, const float* w_ih0, const float* w_hh0, const float* w_ih1, const float* w_hh1,
                                int I, int H, void* wb1, void* wb2, void* stream) {
  HCfg c;
  if (!hidden_cfg(H, &c)) return NPPC_EUNSUPPORTED;
  const int KX = kx_for(I, H), HP = round_up(H, 32);
  const int tpw1 = ((KX + HP) / 16 + c.NW - 1) / c.NW, tpw2 = (2 * HP / 16 + c.NW - 1) / c.NW;
  hipStream_t s = (hipStream_t)stream;
  if (prec == NPPC_PREC_BF16) {
    hipLaunchKernelGGL(lstm_pack_bwd_kernel<bf16_t>, dim3(512), dim3(256), 0, s, w_ih0, w_hh0, (bf16_t*)wb1, I, H, KX, HP,
                       c.NW, tpw1);
    hipLaunchKernelGGL(lstm_pack_bwd_kernel<bf16_t>, dim3(512), dim3(256), 0, s, w_ih1, w_hh1, (bf16_t*)wb2, H, H, HP, HP,
                       c.NW, tpw2);
  } else if (prec == NPPC_PREC_F32) {
    hipLaunchKernelGGL(lstm_pack_bwd_kernel<float>, dim3(512), dim3(256), 0, s, w_ih0, w_hh0, (float*)wb1, I, H, KX, HP, c.NW,
                       tpw1);
    hipLaunchKernelGGL(lstm_pack_bwd_kernel<float>, dim3(512), dim3(256), 0, s, w_ih1, w_hh1, (float*)wb2, H, H, HP, HP, c.NW,
                       tpw2);
  } else {
    return NPPC_EBADARG;
  }
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

int nppc_lstm2_bwd(int prec, const void* g1, const void* g2, const void* c1, const void* c2, const void* dh2,
                   const void* wb1, const void* wb2, void* dx, void* dg1, void* dg2, long N, int Tn, int I, int H,
                   void* stream) {
  HCfg c;
  if (!hidden_cfg(H, &c)) return NPPC_EUNSUPPORTED;
  if (!g1 || !g2 || !c1 || !c2 || !dh2 || !wb1 || !wb2 || !dx || !dg1 || !dg2 || N <= 0 || Tn <= 0) return NPPC_EBADARG;
  LstmBwdArgs a{g1, g2, c1, c2, dh2, wb1, wb2, dx, dg1, dg2, N, Tn};
  hipStream_t s = (hipStream_t)stream;
  if (prec == NPPC_PREC_BF16) {
    if (H == 384) return launch_bwd<bf16_t, 3, 8, 64>(a, s);
    return launch_bwd<bf16_t, 1, 1, 32>(a, s);
  } else if (prec == NPPC_PREC_F32) {
    if (H == 384) return launch_bwd<float, 3, 8, 64>(a, s);
    return launch_bwd<float, 1, 1, 32>(a, s);
  }
  return NPPC_EBADARG;
}

}  // extern "C"
