// Cooperative ("cluster") variant of the fused 2-layer LSTM forward: G workgroups (one per CU) share one tile of
// MC = 16*MT sequences and split the hidden units, so every CU streams only 1/G of the packed weights per step --
// the single-workgroup kernel in lstm.hip is bound by that stream (~116 GB/s per CU from L2), not by MFMA.
// Per layer and step each CU publishes its slice of h (MC x H/G values, 12 KB) to its partners through global memory.
//
// Hand-off protocol (cdna_hip_programming.md, Guideline 16, recipe R1 / MI355X_MICROARCH visibility table row 1):
//   producer: payload stored write-through (sc1, 16-byte buffer stores) -> every storing wave s_waitcnt vmcnt(0) ->
//             workgroup barrier -> ONE lane stores the epoch flag (relaxed, agent scope)
//   consumer: ONE wave polls the partners' flags (relaxed agent loads, bounded spin) -> workgroup barrier ->
//             every wave reads the payload with sc1 buffer loads
// Correctness never depends on placement or dispatch order; all G workgroups of a cluster must be co-resident, which
// the host guarantees by launching <= one workgroup per CU (grid <= CU count, LDS request > half a CU).
// Every spin is bounded: on timeout a word in the flag block is set and the workgroup carries on (wrong numbers, no hang).
#include "common.h"
#include "nppc_hip.h"

namespace {

typedef __attribute__((address_space(1))) unsigned int gu32;

struct CoopArgs {
  const void* x;      // [Tn][N][KX]
  const void* wp1;    // packed like lstm.hip (pair-major, 8 "waves" x 48 units): re-indexed here per 16-unit block
  const void* wp2;
  const float* bias1;
  const float* bias2;
  void* h2;           // [Tn][N][H]
  void* h1T; void* h2T; void* g1; void* g2; void* c1; void* c2;   // train
  void* xch;          // [clusters][2 layers][2 parities][G][MC][HC]  exchange slices
  unsigned* flags;    // [clusters][2 layers][G] epochs, then 1 timeout word; zeroed before every launch
  long N; int Tn; long Np; int clusters;
};

constexpr unsigned SPIN_LIMIT = 1u << 22;

__device__ __forceinline__ void store_sc1_b128(__amdgpu_buffer_rsrc_t r, int off, u32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, 16);   // aux 16 = sc1 (write-through)
}
__device__ __forceinline__ u32x4 load_sc1_b128(__amdgpu_buffer_rsrc_t r, int off) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16);
}

// weight fragments of unit block `ublk` (16 units) for gate pair `gp`, k-step kk of a layer packed by
// lstm_pack_kernel (layout [gp][wave8][kk][ub3][s2][lane][8], wave8 = ublk/3, ub3 = ublk%3)
template <typename T>
__device__ __forceinline__ int frag_boff(int gp, int ublk, int kk, int s, int nk) {
  const int w8 = ublk / 3, ub3 = ublk % 3;
  return ((((gp * 8 + w8) * nk + kk) * 3 + ub3) * 2 + s) * 512 * (int)sizeof(T);
}

template <typename T, int MT, int RS, int DEPTH>
__device__ __forceinline__ void coop_gemm(f32x4 (&acc)[2][MT], const T* a_lane /* row n, col 8q of the segment */, int k0, int k1,
                                          int koff /* packed k-step of segment start */, int nk, __amdgpu_buffer_rsrc_t wr, int gp,
                                          int ublk, int lane) {
  typedef typename Frag<T>::type frag;
  auto loadb = [&](frag(&b)[2], int kk) {
#pragma unroll
    for (int s = 0; s < 2; ++s) b[s] = BFrag<T>::load(wr, lane, frag_boff<T>(gp, ublk, koff + kk, s, nk));
  };
  auto compute = [&](const frag(&b)[2], int kk) {
    frag af[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) af[mt] = load_frag<T>(a_lane + 16 * mt * RS + 32 * kk);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[s][mt] = mma16(af[mt], b[s], acc[s][mt]);
  };
  frag b[DEPTH][2];
  const int n = k1 - k0;
#pragma unroll
  for (int d = 0; d < DEPTH - 1; ++d)
    if (d < n) loadb(b[d], k0 + d);
#pragma unroll 1
  for (int kk = 0; kk < n; kk += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int kl = kk + d + DEPTH - 1;
      if (kl < n) loadb(b[(d + DEPTH - 1) % DEPTH], k0 + kl);
      if (kk + d < n) compute(b[d], k0 + kk + d);
    }
  }
}

template <typename T> __device__ __forceinline__ void store4(T* p, float a, float b, float c, float d);
template <> __device__ __forceinline__ void store4<float>(float* p, float a, float b, float c, float d) {
  *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
}
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, float a, float b, float c, float d) {
  uint2 v;
  v.x = (uint32_t)f2bf(a) | ((uint32_t)f2bf(b) << 16);
  v.y = (uint32_t)f2bf(c) | ((uint32_t)f2bf(d) << 16);
  *reinterpret_cast<uint2*>(p) = v;
}

// LDS row (elements): [X0 (KX) | X1 (KX) | H1 (H) | H2 (H)] + pad.  H = 384 = 12 k-steps.
template <typename T, int G, int MT, int KX, bool TRAIN>
__global__ __launch_bounds__((384 / G / 16) * 64) void lstm2_coop_fwd_kernel(CoopArgs a) {
  constexpr int H = 384, HC = H / G, NW = HC / 16, NT = NW * 64, MC = 16 * MT;
  constexpr int VEC = 16 / (int)sizeof(T);
  constexpr int RS = 2 * KX + 2 * H + VEC;
  constexpr int OX0 = 0, OX1 = KX, OH1 = 2 * KX, OH2 = 2 * KX + H;
  constexpr int NKX = KX / 32, NKH = H / 32;
  constexpr int nk1 = NKX + NKH, nk2 = 2 * NKH;
  constexpr int DEPTH = (MT <= 2 || G == 4) ? 4 : 2;              // weight-fragment ring depth the register budget allows
  constexpr int SLICE = MC * HC;                                  // elements of one CU's h slice
  constexpr int SLICE_CH = SLICE * (int)sizeof(T) / 16;           // 16-byte chunks
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* lds = reinterpret_cast<T*>(smem_raw);

  const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cluster = blockIdx.x / G, cu = blockIdx.x % G;
  const int ublk = cu * NW + wave;                                // global 16-unit block of this wave
  const int unit_n = ublk * 16 + n;
  const long row0 = (long)cluster * MC;
  const long N = a.N;
  const long rbase = row0 + 4 * q;

  for (int i = tid; i < MC * RS; i += NT) lds[i] = from_f32<T>(0.f);
  float b1[4], b2[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    b1[g] = a.bias1[g * H + unit_n];
    b2[g] = a.bias2[g * H + unit_n];
  }
  f32x4 c1[MT], c2[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) c1[mt] = c2[mt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const T* xg = reinterpret_cast<const T*>(a.x);
  constexpr int cpr = KX / VEC, nchunk = MC * cpr;
  constexpr int XCH = (nchunk + NT - 1) / NT;
  __syncthreads();
  for (int ch = tid; ch < nchunk; ch += NT) {
    const int r = ch / cpr, cc = ch % cpr;
    if (row0 + r < N)
      *reinterpret_cast<uint4*>(lds + r * RS + OX0 + cc * VEC) =
          *reinterpret_cast<const uint4*>(xg + ((size_t)(row0 + r)) * KX + cc * VEC);
  }
  __syncthreads();

  constexpr int ps1 = 8 * nk1 * 3 * 2 * 512 * (int)sizeof(T), ps2 = 8 * nk2 * 3 * 2 * 512 * (int)sizeof(T);
  const __amdgpu_buffer_rsrc_t wr1 = make_rsrc(a.wp1, 2u * ps1);
  const __amdgpu_buffer_rsrc_t wr2 = make_rsrc(a.wp2, 2u * ps2);
  // exchange block of this cluster: [layer][parity][G][MC*HC]
  const size_t xch_cluster = (size_t)cluster * 2 * 2 * G * SLICE;
  const __amdgpu_buffer_rsrc_t xr =
      make_rsrc(reinterpret_cast<const T*>(a.xch) + xch_cluster, (unsigned)(2 * 2 * G * SLICE * sizeof(T)));
  gu32* flags = (gu32*)(a.flags + (size_t)cluster * 2 * G);
  gu32* tmo = (gu32*)(a.flags + (size_t)a.clusters * 2 * G);

  const T* a_lane = lds + n * RS + 8 * q;
  T* hw_lane = lds + 4 * q * RS + unit_n;                         // + OH1/OH2 + (16mt+j)*RS : own unit column
  T* h1T = reinterpret_cast<T*>(a.h1T);
  T* h2T = reinterpret_cast<T*>(a.h2T);
  T* h2o = reinterpret_cast<T*>(a.h2);
  T* g1o = reinterpret_cast<T*>(a.g1);
  T* g2o = reinterpret_cast<T*>(a.g2);
  T* c1o = reinterpret_cast<T*>(a.c1);
  T* c2o = reinterpret_cast<T*>(a.c2);
  const size_t Rtot = (size_t)a.Tn * a.Np;

  // publish this CU's h slice (already in LDS columns [hoff + cu*HC, +HC)) of layer `layer` for epoch `ep`
  auto publish = [&](int layer, int hoff, int ep) {
    const int par = ep & 1;
    const int base = ((layer * 2 + par) * G + cu) * SLICE * (int)sizeof(T);
    for (int ch = tid; ch < SLICE_CH; ch += NT) {
      const int r = ch / (HC / VEC), cc = ch % (HC / VEC);
      const u32x4 v = *reinterpret_cast<const u32x4*>(lds + r * RS + hoff + cu * HC + cc * VEC);
      store_sc1_b128(xr, base + ch * 16, v);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // every storing wave drains its write-through stores
    __syncthreads();
    if (tid == 0) __hip_atomic_store(flags + layer * G + cu, (unsigned)ep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  // wait for the partners' slices of (layer, epoch) and copy them into LDS
  auto consume = [&](int layer, int hoff, int ep) {
    if (wave == 0) {
      if (lane < G && lane != cu) {
        unsigned spins = 0;
        while (__hip_atomic_load(flags + layer * G + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)ep) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > SPIN_LIMIT) {
            __hip_atomic_store(tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");     // compiler ordering only: payload loads are sc1
    }
    __syncthreads();
    const int par = ep & 1;
#pragma unroll 1
    for (int p = 0; p < G; ++p) {
      if (p == cu) continue;
      const int base = ((layer * 2 + par) * G + p) * SLICE * (int)sizeof(T);
      for (int ch = tid; ch < SLICE_CH; ch += NT) {
        const int r = ch / (HC / VEC), cc = ch % (HC / VEC);
        const u32x4 v = load_sc1_b128(xr, base + ch * 16);
        *reinterpret_cast<u32x4*>(lds + r * RS + hoff + p * HC + cc * VEC) = v;
      }
    }
  };

#pragma unroll 1
  for (int t = 0; t < a.Tn; ++t) {
    const int p = t & 1;
    const int ep = t + 1;
    const size_t ebase = ((size_t)t * N + rbase) * H + unit_n;
    const size_t tbase = (size_t)unit_n * Rtot + (size_t)t * a.Np + rbase;
    uint4 xrg[XCH];
    const bool more = t + 1 < a.Tn;
    if (more) {
#pragma unroll
      for (int u = 0; u < XCH; ++u) {
        const int ch = tid + u * NT;
        xrg[u] = make_uint4(0, 0, 0, 0);
        if (ch < nchunk) {
          const int r = ch / cpr, cc = ch % cpr;
          if (row0 + r < N) xrg[u] = *reinterpret_cast<const uint4*>(xg + ((size_t)(t + 1) * N + row0 + r) * KX + cc * VEC);
        }
      }
    }
    // ================= layer 1: [x_t | h1_{t-1}] =================
    float iv1[TRAIN ? MT : 1][4], gv1[TRAIN ? MT : 1][4];
    f32x4 ig[MT];
    {
      f32x4 acc[2][MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[0][mt] = acc[1][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
      coop_gemm<T, MT, RS, DEPTH>(acc, a_lane + (p ? OX1 : OX0), 0, NKX, 0, nk1, wr1, 0, ublk, lane);
      coop_gemm<T, MT, RS, DEPTH>(acc, a_lane + OH1, 0, NKH, NKX, nk1, wr1, 0, ublk, lane);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float iv = sigmoid_f(acc[0][mt][j] + b1[0]);
          const float gv = tanh_f(acc[1][mt][j] + b1[2]);
          ig[mt][j] = iv * gv;
          if (TRAIN) { iv1[mt][j] = iv; gv1[mt][j] = gv; }
        }
    }
    float hn1[MT][4];
    {
      f32x4 acc[2][MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[0][mt] = acc[1][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
      coop_gemm<T, MT, RS, DEPTH>(acc, a_lane + (p ? OX1 : OX0), 0, NKX, 0, nk1, wr1, 1, ublk, lane);
      coop_gemm<T, MT, RS, DEPTH>(acc, a_lane + OH1, 0, NKH, NKX, nk1, wr1, 1, ublk, lane);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float fv = sigmoid_f(acc[0][mt][j] + b1[1]);
          const float ov = sigmoid_f(acc[1][mt][j] + b1[3]);
          const float cn = fv * c1[mt][j] + ig[mt][j];
          c1[mt][j] = cn;
          hn1[mt][j] = ov * tanh_f(cn);
          if (TRAIN && rbase + 16 * mt + j < N) {
            const size_t e = ebase + (size_t)(16 * mt + j) * H;
            c1o[e] = from_f32<T>(cn);
            store4<T>(g1o + e * 4, iv1[mt][j], gv1[mt][j], fv, ov);
            h1T[tbase + 16 * mt + j] = from_f32<T>(hn1[mt][j]);
          }
        }
    }
    // the partners' h2_{t-1} (published at the end of the previous step) -> LDS; nobody reads H2 during layer 1
    if (t > 0) consume(1, OH2, ep - 1);
    __syncthreads();                                              // (1) all waves done reading h1_{t-1} and x_t
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int j = 0; j < 4; ++j) hw_lane[OH1 + (16 * mt + j) * RS] = from_f32<T>(hn1[mt][j]);
    if (more) {
#pragma unroll
      for (int u = 0; u < XCH; ++u) {
        const int ch = tid + u * NT;
        if (ch < nchunk) {
          const int r = ch / cpr, cc = ch % cpr;
          *reinterpret_cast<uint4*>(lds + r * RS + (p ? OX0 : OX1) + cc * VEC) = xrg[u];
        }
      }
    }
    __syncthreads();                                              // (2a) own h1_t slice complete in LDS
    publish(0, OH1, ep);
    // ================= layer 2 =================
    // gate pair (i,g): the h2_{t-1} half first -- it does not need the partners' h1_t, so it hides the hand-off latency
    f32x4 ig2[MT];
    float iv2[TRAIN ? MT : 1][4], gv2[TRAIN ? MT : 1][4];
    {
      f32x4 acc[2][MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[0][mt] = acc[1][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
      coop_gemm<T, MT, RS, DEPTH>(acc, a_lane + OH2, 0, NKH, NKH, nk2, wr2, 0, ublk, lane);
      consume(0, OH1, ep);
      __syncthreads();                                            // (2c) full h1_t in LDS
      coop_gemm<T, MT, RS, DEPTH>(acc, a_lane + OH1, 0, NKH, 0, nk2, wr2, 0, ublk, lane);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float iv = sigmoid_f(acc[0][mt][j] + b2[0]);
          const float gv = tanh_f(acc[1][mt][j] + b2[2]);
          ig2[mt][j] = iv * gv;
          if (TRAIN) { iv2[mt][j] = iv; gv2[mt][j] = gv; }
        }
    }
    float hn2[MT][4];
    {
      // gate pair (f,o): [h1_t | h2_{t-1}] is one contiguous K range of the LDS row
      f32x4 acc[2][MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[0][mt] = acc[1][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
      coop_gemm<T, MT, RS, DEPTH>(acc, a_lane + OH1, 0, 2 * NKH, 0, nk2, wr2, 1, ublk, lane);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float fv = sigmoid_f(acc[0][mt][j] + b2[1]);
          const float ov = sigmoid_f(acc[1][mt][j] + b2[3]);
          const float cn = fv * c2[mt][j] + ig2[mt][j];
          c2[mt][j] = cn;
          hn2[mt][j] = ov * tanh_f(cn);
          if (rbase + 16 * mt + j < N) {
            const size_t e = ebase + (size_t)(16 * mt + j) * H;
            h2o[e] = from_f32<T>(hn2[mt][j]);
            if (TRAIN) {
              c2o[e] = from_f32<T>(cn);
              store4<T>(g2o + e * 4, iv2[mt][j], gv2[mt][j], fv, ov);
              h2T[tbase + 16 * mt + j] = from_f32<T>(hn2[mt][j]);
            }
          }
        }
    }
    __syncthreads();                                              // (3) all waves done reading h2_{t-1} / h1_t
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int j = 0; j < 4; ++j) hw_lane[OH2 + (16 * mt + j) * RS] = from_f32<T>(hn2[mt][j]);
    __syncthreads();                                              // own h2_t slice complete in LDS
    if (more) publish(1, OH2, ep);
  }
}

template <typename T, int G, int MT, bool TRAIN>
static int launch_coop(CoopArgs a, size_t xch_bytes, hipStream_t s) {
  constexpr int KX = 64, H = 384, MC = 16 * MT;
  constexpr int RS = 2 * KX + 2 * H + 16 / (int)sizeof(T);
  size_t smem = (size_t)MC * RS * sizeof(T);
  if (smem < 84 * 1024) smem = 84 * 1024;            // > half a CU's LDS: at most one workgroup per CU
  if (smem > 160 * 1024) return NPPC_EUNSUPPORTED;
  const long need = (long)a.clusters * 2 * 2 * G * MC * (H / G) * sizeof(T);
  if ((long)xch_bytes < need) return NPPC_EBADARG;
  auto k = lstm2_coop_fwd_kernel<T, G, MT, KX, TRAIN>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) !=
      hipSuccess)
    return NPPC_ELAUNCH;
  if (hipMemsetAsync(a.flags, 0, ((size_t)a.clusters * 2 * G + 4) * sizeof(unsigned), s) != hipSuccess) return NPPC_ELAUNCH;
  hipLaunchKernelGGL(k, dim3(a.clusters * G), dim3((H / G / 16) * 64), smem, s, a);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

}  // namespace

extern "C" {

// Rows per cluster and cluster count the cooperative kernel would use for N sequences; 0 clusters = not applicable.
int nppc_lstm2_coop_plan(int prec, int train, long N, int H, int n_cu, int* G, int* mtile, int* clusters) {
  *G = 0; *mtile = 0; *clusters = 0;
  if (prec != NPPC_PREC_BF16 || H != 384 || N <= 0) return NPPC_OK;
  // candidates (G, mtile), most CUs first: the stream per CU shrinks with G, the rows per cluster grow with mtile
  const int cand[3][2] = {{2, 2}, {4, 4}, {2, 5}};   // measured at N = 4096: G = 2 beats G = 4 (6 waves leave two SIMDs idle)
  for (int oi = 0; oi < 3; ++oi) {
    const int g = cand[oi][0], mt = cand[oi][1];
    if (train && mt == 5) continue;
    const long cl = (N + 16 * mt - 1) / (16 * mt);
    if (cl * g <= n_cu && (cl * g * 2 > n_cu || oi == 2)) { *G = g; *mtile = mt; *clusters = (int)cl; return NPPC_OK; }
  }
  return NPPC_OK;
}

// flags: (clusters*2*G + 4) u32; xch: clusters*2*2*G*MC*(H/G) elements.  Same tensor contract as nppc_lstm2_fwd.
int nppc_lstm2_fwd_coop(int prec, int train, int G, int mtile, const void* x, const void* wp1, const void* wp2,
                        const float* bias1, const float* bias2, void* h2, void* h1T, void* h2T, void* g1, void* g2, void* c1,
                        void* c2, void* xch, long xch_bytes, unsigned* flags, long N, int Tn, int I, int H, long Np,
                        void* stream) {
  if (prec != NPPC_PREC_BF16 || H != 384 || (G != 2 && G != 4)) return NPPC_EUNSUPPORTED;
  if (!x || !wp1 || !wp2 || !h2 || !xch || !flags || N <= 0 || Tn <= 0 || I > 64) return NPPC_EBADARG;
  if (train && (!h1T || !h2T || !g1 || !g2 || !c1 || !c2 || Np < N)) return NPPC_EBADARG;
  const int MC = 16 * mtile;
  CoopArgs a{x, wp1, wp2, bias1, bias2, h2, h1T, h2T, g1, g2, c1, c2, xch, flags, N, Tn, Np, (int)((N + MC - 1) / MC)};
  hipStream_t s = (hipStream_t)stream;
  if (G == 4) {
    if (mtile != 4) return NPPC_EUNSUPPORTED;
    return train ? launch_coop<bf16_t, 4, 4, true>(a, (size_t)xch_bytes, s) : launch_coop<bf16_t, 4, 4, false>(a, (size_t)xch_bytes, s);
  }
  if (train) {
    if (mtile == 2) return launch_coop<bf16_t, 2, 2, true>(a, (size_t)xch_bytes, s);
    return NPPC_EUNSUPPORTED;
  }
  if (mtile == 2) return launch_coop<bf16_t, 2, 2, false>(a, (size_t)xch_bytes, s);
  if (mtile == 5) return launch_coop<bf16_t, 2, 5, false>(a, (size_t)xch_bytes, s);
  return NPPC_EUNSUPPORTED;
}

}  // extern "C"
