// Full-band sequence model: 8 x TCNBlock + Linear + ReLU on time-major activations.
// Reference: audio_zen/model/module/causal_conv.py:67-108 (TCNBlock: conv1x1, PReLU, GroupNorm(1,512,eps=1e-8),
// depthwise dilated conv k=3, PReLU, GroupNorm, sconv, skip), sequence_model.py:47-58,106-112.
//
// Layout: activations are [B][Tp][ld] (frames x channels, channels contiguous, Tp = T' rounded up to 128,
// ld = channels rounded up to 64, padding zero) so every 1x1 conv / Linear is one row-major NT GEMM
//   Y[R][N] = X[R][K] * W[N][K]^T,  R = B*Tp
// on bf16 (or exact-f32) MFMA 16x16 tiles, with bias / PReLU / GroupNorm statistics / residual fused in the epilogue.
#include <cstdlib>
#include "common.h"
#include "nppc_hip.h"

namespace {

enum { EPI_PLAIN = 0, EPI_PRELU_STATS = 1, EPI_RESIDUAL = 2, EPI_RELU = 3, EPI_PLAIN_F32 = 4, EPI_MASK_POS = 5, EPI_RESIDUAL_GN = 6 };
// EPI_RESIDUAL_GN: the GroupNorm in front of the product is folded into it.  With a = (y - mean_b) * rstd_b * gamma + beta,
//   a W^T = rstd_b * (y Wg^T) - mean_b * rstd_b * v + u,   Wg[n][k] = gamma[k] W[n][k],  v[n] = sum_k Wg[n][k],  u[n] = sum_k beta[k] W[n][k] + bias[n]
// so the GEMM reads the un-normalised y (B operand = Wg, `bias` = u, `gnv` = v, `stats` = the per-sample (sum, sumsq) of y)
// and the normalised activation is never written (nppc_tcn_pack_sconv builds Wg, u, v).

struct GemmArgs {
  const void* A; long lda; long strideA;   // [R][lda]   (strides in elements, per batch z)
  const void* B; long ldb; long strideB;   // [N][ldb]
  void* C; long ldc; long strideC;         // [R][ldc]
  const float* bias; long strideBias;      // [N] or null
  const void* res; long ldres; long strideRes;  // residual [R][ldres] (EPI_RESIDUAL)
  const float* slope; long strideSlope;    // PReLU slope (1 value)
  double* stats; long strideStats;         // [R/Tp][2] (sum, sumsq) (EPI_PRELU_STATS)
  int R, N, K;                             // R % 128 == 0, N % 64 == 0, K % 32 == 0
  int Tp, Tv;                              // rows with (row % Tp) >= Tv are padding: forced to zero
  int Nv;                                  // columns >= Nv are padding: forced to zero
  int relu_in;                             // apply ReLU to A on load (TCN trailing nn.ReLU before the Linear)
  int ksplit;                              // >1: blockIdx.z = batch*ksplit + s; slice s covers K elements [s*K, (s+1)*K)
  const float* gnv; long strideGnv;        // EPI_RESIDUAL_GN: v[N]; `stats` then holds the GroupNorm (sum, sumsq) per sample
  double gcnt; float geps;                 // ... elements per sample, epsilon
  int staged;                              // LDS kernels, residual / mask epilogues: coalesced epilogue through LDS (ldc, ldres % 8 == 0)
  float* colpart; long strideColpart;      // optional (LDS kernels, EPI_RESIDUAL / EPI_MASK_POS): column sums of each 128-row tile of
                                           // the OUTPUT, [z][R/128][N] fp32 -- the bias gradient of the layer that consumes C as its
                                           // upstream gradient, without a pass of its own over C
};

template <typename T> __device__ __forceinline__ typename Frag<T>::type relu_frag(typename Frag<T>::type f);
template <> __device__ __forceinline__ bf16x8 relu_frag<bf16_t>(bf16x8 f) {
  u32x4 v = __builtin_bit_cast(u32x4, f);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    unsigned x = v[i];
    if (x & 0x8000u) x &= 0xffff0000u;
    if (x & 0x80000000u) x &= 0x0000ffffu;
    v[i] = x;
  }
  return __builtin_bit_cast(bf16x8, v);
}
template <> __device__ __forceinline__ f32x8 relu_frag<float>(f32x8 f) {
#pragma unroll
  for (int i = 0; i < 8; ++i) f[i] = fmaxf(f[i], 0.f);
  return f;
}

// 256 threads = 4 waves stacked over rows; wave tile 32 x 64, block tile 128 x 64.
template <typename T, int EPI>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmArgs g) {
  typedef typename Frag<T>::type frag;
  const int ks = g.ksplit > 1 ? g.ksplit : 1;
  const int z = blockIdx.z / ks, ksl = blockIdx.z % ks;
  const T* A = reinterpret_cast<const T*>(g.A) + (size_t)z * g.strideA + (size_t)ksl * g.K;
  const T* B = reinterpret_cast<const T*>(g.B) + (size_t)z * g.strideB + (size_t)ksl * g.K;
  T* C = reinterpret_cast<T*>(g.C) + (size_t)blockIdx.z * g.strideC;
  float* Cf = reinterpret_cast<float*>(g.C) + (size_t)blockIdx.z * g.strideC;   // EPI_PLAIN_F32: fp32 output (split-K slabs)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, q = lane >> 4;
  const int r0 = blockIdx.x * 128 + wave * 32, c0 = blockIdx.y * 64;
  const T* ap = A + (size_t)(r0 + n) * g.lda + 8 * q;
  const T* bp = B + (size_t)(c0 + n) * g.ldb + 8 * q;
  f32x4 acc[2][4];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nk = g.K / 32;
  frag a0[2], b0[4], a1[2], b1[4];
  auto ld = [&](frag(&a)[2], frag(&b)[4], int kk) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) a[mi] = load_frag<T>(ap + (size_t)16 * mi * g.lda + 32 * kk);
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) b[ni] = load_frag<T>(bp + (size_t)16 * ni * g.ldb + 32 * kk);
  };
  auto mm = [&](frag(&a)[2], frag(&b)[4]) {
    if (g.relu_in) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) a[mi] = relu_frag<T>(a[mi]);
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = mma16(a[mi], b[ni], acc[mi][ni]);
  };
  ld(a0, b0, 0);
  int kk = 0;
#pragma unroll 1
  for (; kk + 2 < nk; kk += 2) {
    ld(a1, b1, kk + 1);
    mm(a0, b0);
    ld(a0, b0, kk + 2);
    mm(a1, b1);
  }
  if (kk + 1 < nk) {
    ld(a1, b1, kk + 1);
    mm(a0, b0);
    mm(a1, b1);
  } else {
    mm(a0, b0);
  }

  // ---- epilogue: element (row r0+16mi+4q+j, col c0+16ni+n)
  const float* bias = g.bias ? g.bias + (size_t)z * g.strideBias : nullptr;
  float slope = 0.f;
  if (EPI == EPI_PRELU_STATS) slope = g.slope[(size_t)z * g.strideSlope];
  float gn_rstd = 1.f, gn_mr = 0.f;        // EPI_RESIDUAL_GN: one sample per 128-row tile (Tp % 128 == 0)
  const float* gnv = nullptr;
  if (EPI == EPI_RESIDUAL_GN) {
    const double* st = g.stats + (size_t)z * g.strideStats + (size_t)((blockIdx.x * 128) / g.Tp) * 2;
    const double m = st[0] / g.gcnt, var = st[1] / g.gcnt - m * m;
    gn_rstd = (float)(1.0 / sqrt((var > 0 ? var : 0) + (double)g.geps));
    gn_mr = (float)m * gn_rstd;
    gnv = g.gnv + (size_t)z * g.strideGnv;
  }
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    const int col = c0 + 16 * ni + n;
    const bool cvalid = col < g.Nv;
    float bv = (bias && cvalid) ? bias[col] : 0.f;
    if (EPI == EPI_RESIDUAL_GN && cvalid) bv -= gn_mr * gnv[col];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = r0 + 16 * mi + 4 * q + j;
        const bool valid = cvalid && (row % g.Tp) < g.Tv;
        float v = (EPI == EPI_RESIDUAL_GN ? gn_rstd * acc[mi][ni][j] : acc[mi][ni][j]) + bv;
        if (EPI == EPI_PRELU_STATS) v = v > 0.f ? v : slope * v;
        if (EPI == EPI_RELU) v = fmaxf(v, 0.f);
        if (EPI == EPI_RESIDUAL || EPI == EPI_RESIDUAL_GN) {
          const T* res = reinterpret_cast<const T*>(g.res) + (size_t)z * g.strideRes;
          v += to_f32<T>(res[(size_t)row * g.ldres + col]);
        }
        if (EPI == EPI_MASK_POS) {
          const T* res = reinterpret_cast<const T*>(g.res) + (size_t)z * g.strideRes;
          if (!(to_f32<T>(res[(size_t)row * g.ldres + col]) > 0.f)) v = 0.f;
        }
        if (!valid) v = 0.f;
        if (EPI == EPI_PLAIN_F32) {
          Cf[(size_t)row * g.ldc + col] = v;
          continue;
        }
        const T o = from_f32<T>(v);
        C[(size_t)row * g.ldc + col] = o;
        if (EPI == EPI_PRELU_STATS) {
          const float vo = to_f32<T>(o);   // statistics of the STORED (rounded) activations
          s1 += vo;
          s2 += vo * vo;
        }
      }
  }
  if (EPI == EPI_PRELU_STATS) {
    double d1 = wave_sum((double)s1), d2 = wave_sum((double)s2);
    if (lane == 0) {
      double* st = g.stats + (size_t)z * g.strideStats + (size_t)(r0 / g.Tp) * 2;
      atomicAdd(st, d1);
      atomicAdd(st + 1, d2);
    }
  }
}

// LDS-staged variant of gemm_nt_kernel (same arguments and epilogues) for K-slices that are multiples of the 128-byte
// stage (64 bf16 / 32 f32): 256 threads = 2 x 2 waves, block tile 128 x BN, double-buffered XOR-swizzled LDS with register
// staging, one barrier per stage.  Operand tiles are read from HBM/L2 once per workgroup instead of once per wave.
template <typename T, int EPI, int BN>
__global__ __launch_bounds__(256, 2) void gemm_nt_lds_kernel(GemmArgs g) {
  typedef typename Frag<T>::type frag;
  constexpr int EPC = 16 / (int)sizeof(T);
  constexpr int BK = 8 * EPC;
  constexpr int KS = BK / 32;
  constexpr int NJ = BN / 32;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2][(128 + BN) * 128];
  __shared__ double red[2][4];
  const int ks_ = g.ksplit > 1 ? g.ksplit : 1;
  const int z = blockIdx.z / ks_, ksl = blockIdx.z % ks_;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, q = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.x * 128, n0 = blockIdx.y * BN;
  const int srow = tid >> 3, sc = tid & 7;
  const T* ga = reinterpret_cast<const T*>(g.A) + (size_t)z * g.strideA + (size_t)ksl * g.K + (long)(m0 + srow) * g.lda + sc * EPC;
  const T* gb = reinterpret_cast<const T*>(g.B) + (size_t)z * g.strideB + (size_t)ksl * g.K + (long)(n0 + srow) * g.ldb + sc * EPC;
  const int soff = srow * 128 + ((sc ^ (srow & 7)) << 4);
  // diagnostic builds (tools/diag/nt_variants.py; timing only): NT_DIAG_NOLOOP = one K stage instead of K / BK (what the launch,
  // the first stage's latency and the epilogue cost without the main loop), NT_DIAG_NOEPI = no epilogue (one store per lane).
  // Measured at the TCN shapes (profiles/r04_nt_variants.txt): 38 us = 12 us launch + first stage, 14 us for the other eight
  // stages (one stage of prefetch: a stage takes the memory latency, 4-6 us would be MFMA-bound), 12-15 us epilogue.  Two
  // register stages of prefetch were tried again in round 4 (named registers, unconditional clamped loads): hipcc still
  // drains the queue (vmcnt(0)) in front of the LDS stores and spills at 256 VGPRs -- a deeper pipeline here needs the
  // inline-asm LDS-DMA ring of gemm_tn_dma_kernel, not built.
#ifdef NT_DIAG_NOLOOP
  const int nstage = 1;
#else
  const int nstage = g.K / BK;
#endif
  uint4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
#define NGLOAD(k0)                                                            \
  {                                                                           \
    ra0 = *reinterpret_cast<const uint4*>(ga + (k0));                         \
    ra1 = *reinterpret_cast<const uint4*>(ga + 32 * g.lda + (k0));            \
    ra2 = *reinterpret_cast<const uint4*>(ga + 64 * g.lda + (k0));            \
    ra3 = *reinterpret_cast<const uint4*>(ga + 96 * g.lda + (k0));            \
    rb0 = *reinterpret_cast<const uint4*>(gb + (k0));                         \
    rb1 = *reinterpret_cast<const uint4*>(gb + 32 * g.ldb + (k0));            \
    if constexpr (NJ == 4) {                                                  \
      rb2 = *reinterpret_cast<const uint4*>(gb + 64 * g.ldb + (k0));          \
      rb3 = *reinterpret_cast<const uint4*>(gb + 96 * g.ldb + (k0));          \
    }                                                                         \
  }
#define NLSTORE(buf)                                                          \
  {                                                                           \
    *reinterpret_cast<uint4*>(&lds[buf][soff]) = ra0;                         \
    *reinterpret_cast<uint4*>(&lds[buf][soff + 32 * 128]) = ra1;              \
    *reinterpret_cast<uint4*>(&lds[buf][soff + 64 * 128]) = ra2;              \
    *reinterpret_cast<uint4*>(&lds[buf][soff + 96 * 128]) = ra3;              \
    *reinterpret_cast<uint4*>(&lds[buf][128 * 128 + soff]) = rb0;             \
    *reinterpret_cast<uint4*>(&lds[buf][128 * 128 + soff + 32 * 128]) = rb1;  \
    if constexpr (NJ == 4) {                                                  \
      *reinterpret_cast<uint4*>(&lds[buf][128 * 128 + soff + 64 * 128]) = rb2; \
      *reinterpret_cast<uint4*>(&lds[buf][128 * 128 + soff + 96 * 128]) = rb3; \
    }                                                                         \
  }
  auto lfrag = [&](const unsigned char* base, int row, int ks) -> frag {
    if constexpr (sizeof(T) == 2) {
      const int c = 4 * ks + q;
      return *reinterpret_cast<const frag*>(base + row * 128 + ((c ^ (row & 7)) << 4));
    } else {
      const int c0 = 2 * q, c1 = 2 * q + 1;
      const float4 lo = *reinterpret_cast<const float4*>(base + row * 128 + ((c0 ^ (row & 7)) << 4));
      const float4 hi = *reinterpret_cast<const float4*>(base + row * 128 + ((c1 ^ (row & 7)) << 4));
      frag f;
      f[0] = lo.x; f[1] = lo.y; f[2] = lo.z; f[3] = lo.w; f[4] = hi.x; f[5] = hi.y; f[6] = hi.z; f[7] = hi.w;
      return f;
    }
  };
  f32x4 acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  NGLOAD(0)
  NLSTORE(0)
  __syncthreads();
  for (int s = 0; s < nstage; ++s) {
    const int buf = s & 1;
    if (s + 1 < nstage) NGLOAD((s + 1) * BK)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      frag af[4], bf[NJ];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        af[i] = lfrag(lds[buf], wm * 64 + 16 * i + n, ks);
        if (g.relu_in) af[i] = relu_frag<T>(af[i]);
      }
#pragma unroll
      for (int j = 0; j < NJ; ++j) bf[j] = lfrag(lds[buf] + 128 * 128, wn * (BN / 2) + 16 * j + n, ks);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = mma16(af[i], bf[j], acc[i][j]);
    }
    if (s + 1 < nstage) NLSTORE(buf ^ 1)
    __syncthreads();
  }
#undef NGLOAD
#undef NLSTORE
#ifdef NT_DIAG_NOEPI
  {
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < NJ; ++j) v += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    reinterpret_cast<T*>(g.C)[(size_t)blockIdx.z * g.strideC + (size_t)(m0 + (tid >> 1)) * g.ldc + n0 + (tid & 1)] = from_f32<T>(v);
    return;
  }
#endif
  // ---- epilogue (identical to gemm_nt_kernel): element (row m0 + wm*64 + 16 i + 4 q + r, col n0 + wn*BN/2 + 16 j + n)
  T* C = reinterpret_cast<T*>(g.C) + (size_t)blockIdx.z * g.strideC;
  float* Cf = reinterpret_cast<float*>(g.C) + (size_t)blockIdx.z * g.strideC;
  const float* bias = g.bias ? g.bias + (size_t)z * g.strideBias : nullptr;
  float slope = 0.f;
  if (EPI == EPI_PRELU_STATS) slope = g.slope[(size_t)z * g.strideSlope];
  float gn_rstd = 1.f, gn_mr = 0.f;        // EPI_RESIDUAL_GN: one sample per 128-row tile (Tp % 128 == 0)
  const float* gnv = nullptr;
  if (EPI == EPI_RESIDUAL_GN) {
    const double* st = g.stats + (size_t)z * g.strideStats + (size_t)(m0 / g.Tp) * 2;
    const double m = st[0] / g.gcnt, var = st[1] / g.gcnt - m * m;
    gn_rstd = (float)(1.0 / sqrt((var > 0 ? var : 0) + (double)g.geps));
    gn_mr = (float)m * gn_rstd;
    gnv = g.gnv + (size_t)z * g.strideGnv;
  }
  constexpr bool HAS_RES = EPI == EPI_RESIDUAL || EPI == EPI_RESIDUAL_GN || EPI == EPI_MASK_POS;
  // round 4: the plain / PReLU + statistics / ReLU epilogues take the same coalesced route (conv1x1 forward, dA2 of the backward,
  // the fc layer: 64 two-byte stores per lane were 12-15 of a launch's 38 us, profiles/r04_nt_variants.txt)
  // (bf16 only: in fp32 the compiler contracts the sum of squares differently in the two layouts, 4e-9 relative -- the parity mode
  // keeps the numbers of rounds 1-3 bit for bit)
  constexpr bool STAGED_PLAIN = sizeof(T) == 2 && (EPI == EPI_PLAIN || EPI == EPI_PRELU_STATS || EPI == EPI_RELU);
  if constexpr (HAS_RES || STAGED_PLAIN) {
    if (g.staged) {
      // Coalesced epilogue.  In the accumulator layout a lane owns 2-byte elements of 32 different (row, column) pairs: the
      // residual came in as 32 dependent 2-byte gathers per lane AFTER the K loop (19 of the sconv product's 59 us) and the
      // output left the same way.  Here the tile goes through the LDS operand buffers (free after the loop) as fp32, row stride
      // BN + 4 words (the four row groups of a wave land on four different bank quarters), and comes back as 16-byte chunks of
      // a row: ONE 16-byte residual load and ONE 16-byte store per 8 outputs.
      constexpr int LD = BN + 4, VE = 8, CPR = BN / VE;
      constexpr int HALVES = (128 * LD * 4 <= (int)sizeof(lds)) ? 1 : 2;   // BN = 128: two passes of 64 rows
      constexpr int ROWS = 128 / HALVES;
      float* ot = reinterpret_cast<float*>(&lds[0][0]);
      [[maybe_unused]] const T* res = HAS_RES ? reinterpret_cast<const T*>(g.res) + (size_t)z * g.strideRes : nullptr;
      float csum[VE];
      float t1 = 0.f, t2 = 0.f;                 // EPI_PRELU_STATS: sums of the stored (rounded) activations
#pragma unroll
      for (int e = 0; e < VE; ++e) csum[e] = 0.f;
      for (int h = 0; h < HALVES; ++h) {
        if (HALVES == 1 || wm == h) {
#pragma unroll
          for (int j = 0; j < NJ; ++j) {
            const int col = n0 + wn * (BN / 2) + 16 * j + n;
            const bool cvalid = col < g.Nv;
            float bv = (bias && cvalid) ? bias[col] : 0.f;
            if (EPI == EPI_RESIDUAL_GN && cvalid) bv -= gn_mr * gnv[col];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                float pv = (EPI == EPI_RESIDUAL_GN ? gn_rstd * acc[i][j][r] : acc[i][j][r]) + bv;
                if (EPI == EPI_PRELU_STATS) pv = pv > 0.f ? pv : slope * pv;
                if (EPI == EPI_RELU) pv = fmaxf(pv, 0.f);
                ot[((HALVES == 1 ? wm * 64 : 0) + 16 * i + 4 * q + r) * LD + wn * (BN / 2) + 16 * j + n] = pv;
                if (EPI == EPI_PRELU_STATS) {
                  // statistics of the stored (rounded, masked) activations, summed per lane in the accumulator layout and in
                  // the order of the unstaged epilogue: the GroupNorm statistics stay bit-identical to rounds 1-3
                  const int row = m0 + wm * 64 + 16 * i + 4 * q + r;
                  const float vo = (cvalid && (row % g.Tp) < g.Tv) ? to_f32<T>(from_f32<T>(pv)) : 0.f;
                  t1 += vo;
                  t2 += vo * vo;
                }
              }
          }
        }
        __syncthreads();
        for (int c = tid; c < ROWS * CPR; c += 256) {
          const int rl = c / CPR, cc = (c % CPR) * VE;
          const int row = m0 + h * ROWS * (HALVES - 1) + rl, col = n0 + cc;
          const bool rvalid = (row % g.Tp) < g.Tv;
          float v[VE], rv[VE];
          const float4 lo = *reinterpret_cast<const float4*>(ot + rl * LD + cc), hi = *reinterpret_cast<const float4*>(ot + rl * LD + cc + 4);
          v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
          if constexpr (HAS_RES) load8<T>(res + (size_t)row * g.ldres + col, rv);
#pragma unroll
          for (int e = 0; e < VE; ++e) {
            float x = v[e];
            if constexpr (HAS_RES) {
              if (EPI == EPI_MASK_POS) { if (!(rv[e] > 0.f)) x = 0.f; }
              else x += rv[e];
            }
            if (!(rvalid && col + e < g.Nv)) x = 0.f;
            v[e] = x;
            csum[e] += to_f32<T>(from_f32<T>(x));
          }
          store8<T>(reinterpret_cast<T*>(g.C) + (size_t)blockIdx.z * g.strideC + (size_t)row * g.ldc + col, v);
        }
        if (HALVES == 2) __syncthreads();
      }
      if (EPI == EPI_PRELU_STATS) {               // one sample per 128-row tile (Tp % 128 == 0): one atomic pair per workgroup
        const double d1 = wave_sum((double)t1), d2 = wave_sum((double)t2);
        __syncthreads();                          // (the tile image in LDS is dead; `red` is separate, but keep the phases apart)
        if (lane == 0) { red[0][wave] = d1; red[1][wave] = d2; }
        __syncthreads();
        if (tid == 0) {
          double* st = g.stats + (size_t)z * g.strideStats + (size_t)(m0 / g.Tp) * 2;
          atomicAdd(st, red[0][0] + red[0][1] + red[0][2] + red[0][3]);
          atomicAdd(st + 1, red[1][0] + red[1][1] + red[1][2] + red[1][3]);
        }
      }
      if (HAS_RES && EPI != EPI_RESIDUAL_GN && g.colpart) {
        // column sums of the stored tile: a thread's chunk column is the same in every pass (256 % CPR == 0); lanes with the
        // same chunk are CPR apart in a wave, the four waves meet in LDS
        __shared__ float cpl[4][BN];
#pragma unroll
        for (int e = 0; e < VE; ++e) {
          float cv = csum[e];
#pragma unroll
          for (int m = CPR; m < 64; m <<= 1) cv += __shfl_xor(cv, m);
          if (lane < CPR) cpl[wave][lane * VE + e] = cv;
        }
        __syncthreads();
        if (tid < BN)
          g.colpart[(size_t)z * g.strideColpart + (size_t)blockIdx.x * g.N + n0 + tid] =
              cpl[0][tid] + cpl[1][tid] + cpl[2][tid] + cpl[3][tid];
      }
      return;
    }
  }
  float s1 = 0.f, s2 = 0.f;
  float cs[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    cs[j] = 0.f;
    const int col = n0 + wn * (BN / 2) + 16 * j + n;
    const bool cvalid = col < g.Nv;
    float bv = (bias && cvalid) ? bias[col] : 0.f;
    if (EPI == EPI_RESIDUAL_GN && cvalid) bv -= gn_mr * gnv[col];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm * 64 + 16 * i + 4 * q + r;
        const bool valid = cvalid && (row % g.Tp) < g.Tv;
        float v = (EPI == EPI_RESIDUAL_GN ? gn_rstd * acc[i][j][r] : acc[i][j][r]) + bv;
        if (EPI == EPI_PRELU_STATS) v = v > 0.f ? v : slope * v;
        if (EPI == EPI_RELU) v = fmaxf(v, 0.f);
        if (EPI == EPI_RESIDUAL || EPI == EPI_RESIDUAL_GN) {
          const T* res = reinterpret_cast<const T*>(g.res) + (size_t)z * g.strideRes;
          v += to_f32<T>(res[(size_t)row * g.ldres + col]);
        }
        if (EPI == EPI_MASK_POS) {
          const T* res = reinterpret_cast<const T*>(g.res) + (size_t)z * g.strideRes;
          if (!(to_f32<T>(res[(size_t)row * g.ldres + col]) > 0.f)) v = 0.f;
        }
        if (!valid) v = 0.f;
        if (EPI == EPI_PLAIN_F32) {
          Cf[(size_t)row * g.ldc + col] = v;
          continue;
        }
        const T o = from_f32<T>(v);
        C[(size_t)row * g.ldc + col] = o;
        if (EPI == EPI_RESIDUAL || EPI == EPI_MASK_POS) cs[j] += to_f32<T>(o);
        if (EPI == EPI_PRELU_STATS) {
          const float vo = to_f32<T>(o);
          s1 += vo;
          s2 += vo * vo;
        }
      }
  }
  if constexpr (EPI == EPI_RESIDUAL || EPI == EPI_MASK_POS) {
    if (g.colpart) {                         // column sums of this 128-row output tile (as stored): 16 rows per lane -> 64 per
      __shared__ float cpl[2][BN];           // wave (lanes n, n+16, n+32, n+48) -> the two waves along M -> one store per column
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        float c = cs[j];
        c += __shfl_xor(c, 16);
        c += __shfl_xor(c, 32);
        if (q == 0) cpl[wm][wn * (BN / 2) + 16 * j + n] = c;
      }
      __syncthreads();
      if (tid < BN)
        g.colpart[(size_t)z * g.strideColpart + (size_t)blockIdx.x * g.N + n0 + tid] = cpl[0][tid] + cpl[1][tid];
    }
  }
  if (EPI == EPI_PRELU_STATS) {
    // one sample per 128-row tile (Tp % 128 == 0): one atomic pair per workgroup
    const double d1 = wave_sum((double)s1), d2 = wave_sum((double)s2);
    if (lane == 0) { red[0][wave] = d1; red[1][wave] = d2; }
    __syncthreads();
    if (tid == 0) {
      double* st = g.stats + (size_t)z * g.strideStats + (size_t)(m0 / g.Tp) * 2;
      atomicAdd(st, red[0][0] + red[0][1] + red[0][2] + red[0][3]);
      atomicAdd(st + 1, red[1][0] + red[1][1] + red[1][2] + red[1][3]);
    }
  }
}

// ---------------------------------------------------------------- depthwise stage
// in  = y1 [B][Tp][ld] (post-PReLU1), GroupNorm-1 statistics (sum, sumsq over the C*Tv valid elements)
// z   = GN1(y1) on valid frames, 0 outside (conv zero padding)
// out = PReLU2( bias[c] + sum_k w[c][k] * z[t + (k-1)*dil][c] )   (+ GroupNorm-2 statistics of out)
constexpr int DW_FRAMES = 32;   // frames per workgroup of the depthwise stage

template <typename T>
__global__ __launch_bounds__(256) void dwconv_kernel(const T* __restrict__ in, T* __restrict__ out,
                                                     const double* __restrict__ st1, double* __restrict__ st2,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     const float* __restrict__ wd, const float* __restrict__ bd,
                                                     const float* __restrict__ slope2, int Cc, int ld, int Tp, int Tv, int dil,
                                                     float eps, long strideAct, long strideSt, long strideP) {
  __shared__ double red[2][4];
  const int z = blockIdx.z, b = blockIdx.y;
  in += (size_t)z * strideAct;
  out += (size_t)z * strideAct;
  st1 += (size_t)z * strideSt;
  st2 += (size_t)z * strideSt;
  gamma += (size_t)z * strideP; beta += (size_t)z * strideP; bd += (size_t)z * strideP; wd += (size_t)z * strideP;
  const float a2 = slope2[(size_t)z * strideP];
  const double cnt = (double)Cc * Tv;
  const double m = st1[b * 2] / cnt;
  const double var = st1[b * 2 + 1] / cnt - m * m;
  const float mean = (float)m, rstd = (float)(1.0 / sqrt((var > 0 ? var : 0) + (double)eps));
  const int cpr = Cc / 8;        // 8-channel chunks per frame; host guarantees cpr <= 256
  const int rpi = 256 / cpr;     // frames per iteration
  const int tl = threadIdx.x / cpr, c8 = (threadIdx.x % cpr) * 8;
  float s1 = 0.f, s2 = 0.f;
  if (tl < rpi) {
    float g8[8], be8[8], b8[8], w8[3][8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      g8[i] = gamma[c8 + i] * rstd;
      be8[i] = beta[c8 + i] - mean * g8[i];
      b8[i] = bd[c8 + i];
#pragma unroll
      for (int k = 0; k < 3; ++k) w8[k][i] = wd[(c8 + i) * 3 + k];
    }
    const int t1 = min((blockIdx.x + 1) * DW_FRAMES, Tp);
    for (int t = blockIdx.x * DW_FRAMES + tl; t < t1; t += rpi) {
      float o[8];
      if (t < Tv) {
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = b8[i];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const int ts = t + (k - 1) * dil;
          if (ts >= 0 && ts < Tv) {
            float v[8];
            load8<T>(in + ((size_t)b * Tp + ts) * ld + c8, v);
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] += w8[k][i] * (v[i] * g8[i] + be8[i]);
          }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          o[i] = o[i] > 0.f ? o[i] : a2 * o[i];
          const float vf = to_f32<T>(from_f32<T>(o[i]));      // statistics of the STORED (rounded) activations
          s1 += vf;
          s2 += vf * vf;
        }
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = 0.f;
      }
      store8<T>(out + ((size_t)b * Tp + t) * ld + c8, o);
    }
  }
  const double d1 = wave_sum((double)s1), d2 = wave_sum((double)s2);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = d1; red[1][threadIdx.x >> 6] = d2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(st2 + b * 2, red[0][0] + red[0][1] + red[0][2] + red[0][3]);
    atomicAdd(st2 + b * 2 + 1, red[1][0] + red[1][1] + red[1][2] + red[1][3]);
  }
}

// y = GN(x) = (x - mean) * rstd * gamma + beta on valid frames (padding stays 0)
template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ in, T* __restrict__ out,
                                                       const double* __restrict__ st, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, int Cc, int ld, int Tp, int Tv,
                                                       float eps, long strideAct, long strideSt, long strideP) {
  const int z = blockIdx.z, b = blockIdx.y;
  in += (size_t)z * strideAct;
  out += (size_t)z * strideAct;
  st += (size_t)z * strideSt;
  gamma += (size_t)z * strideP; beta += (size_t)z * strideP;
  const double cnt = (double)Cc * Tv;
  const double m = st[b * 2] / cnt;
  const double var = st[b * 2 + 1] / cnt - m * m;
  const float mean = (float)m, rstd = (float)(1.0 / sqrt((var > 0 ? var : 0) + (double)eps));
  const int cpr = Cc / 8;
  const long total = (long)Tv * cpr;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int t = (int)(e / cpr), c8 = (int)(e % cpr) * 8;
    float v[8];
    load8<T>(in + ((size_t)b * Tp + t) * ld + c8, v);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (v[i] - mean) * (gamma[c8 + i] * rstd) + beta[c8 + i];
    store8<T>(out + ((size_t)b * Tp + t) * ld + c8, v);
  }
}

// dst[n][k] (ld = ldd, zero padded) = src[n][k] * (colscale ? colscale[k] : 1), n < N, k < K;  T = bf16 / f32
// blockIdx.y = matrix (a, b) of an n_a x n_b grid of equally shaped matrices with constant strides (TCN layers x branches)
template <typename T>
__global__ void pack_matrix_kernel(const float* __restrict__ src, T* __restrict__ dst, int N, int K, int Npad, int ldd,
                                   int transpose, int n_b, long ssa, long ssb, long dsa, long dsb) {
  src += (size_t)(blockIdx.y / n_b) * ssa + (size_t)(blockIdx.y % n_b) * ssb;
  dst += (size_t)(blockIdx.y / n_b) * dsa + (size_t)(blockIdx.y % n_b) * dsb;
  const long total = (long)Npad * ldd;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int nn = (int)(e / ldd), k = (int)(e % ldd);
    float v = 0.f;
    if (!transpose) {
      if (nn < N && k < K) v = src[(size_t)nn * K + k];
    } else {  // dst[n][k] = src[k][n], src is [K][N]
      if (nn < N && k < K) v = src[(size_t)k * N + nn];
    }
    dst[e] = from_f32<T>(v);
  }
}

// Wg[n][k] = gamma[k] * W[n][k] (packed like nppc_pack_matrix: [Npad][ldd], zero padded), v[n] = sum_k Wg[n][k] as stored
// (rounded), u[n] = sum_k beta[k] * W[n][k] + bias[n]; for n_a x n_b equally shaped blocks at constant parameter strides
// (TCN blocks x branches); u, v laid out [n_a][n_b][Npad]
template <typename T>
__global__ __launch_bounds__(64) void pack_sconv_kernel(const float* __restrict__ W, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, const float* __restrict__ bias,
                                                        T* __restrict__ Wg, float* __restrict__ u, float* __restrict__ v, int N,
                                                        int K, int Npad, int ldd, int n_b, long ssa, long ssb, long dsa, long dsb) {
  const int nn = blockIdx.x, ab = blockIdx.y;
  const size_t so = (size_t)(ab / n_b) * ssa + (size_t)(ab % n_b) * ssb;
  T* dst = Wg + (size_t)(ab / n_b) * dsa + (size_t)(ab % n_b) * dsb + (size_t)nn * ldd;
  float su = 0.f, sv = 0.f;
  for (int k = threadIdx.x; k < ldd; k += 64) {
    float wg = 0.f;
    if (nn < N && k < K) {
      const float w = W[so + (size_t)nn * K + k];
      wg = gamma[so + k] * w;
      su += beta[so + k] * w;
    }
    const T o = from_f32<T>(wg);
    dst[k] = o;
    sv += to_f32<T>(o);
  }
  su = wave_sum(su);
  sv = wave_sum(sv);
  if (threadIdx.x == 0) {
    u[(size_t)ab * Npad + nn] = nn < N ? su + bias[so + nn] : 0.f;
    v[(size_t)ab * Npad + nn] = sv;
  }
}

}  // namespace

// NPPC_NT_STAGED=0: residual / mask epilogues in the accumulator layout (A/B switch for the coalesced epilogue)
static int nt_staged() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("NPPC_NT_STAGED");
    v = (e && e[0] == '0') ? 0 : 1;
  }
  return v;
}

// NPPC_NT_STAGED_PLAIN=0: the plain / PReLU / ReLU epilogues in the accumulator layout, as in rounds 1-3 (A/B switch)
static int nt_staged_plain() {                       // (read per call: tests/test_tcn_gpu.py compares the two epilogues in one process)
  const char* e = getenv("NPPC_NT_STAGED_PLAIN");
  return (e && e[0] == '0') ? 0 : 1;
}

static int launch_nt(int prec, int epi, const void* A, long lda, long sA, const void* B, long ldb, long sB, void* C, long ldc,
                     long sC, const float* bias, long sBias, const void* res, long ldres, long sRes, const float* slope,
                     long sSlope, double* stats, long sStats, int R, int N, int K, int Tp, int Tv, int Nv, int relu_in,
                     int batch, int ksplit, float* colpart, void* stream) {
  if (!A || !B || !C || R <= 0 || N <= 0 || K <= 0 || batch <= 0) return NPPC_EBADARG;
  if (R % 128 || N % 64 || K % 32 || Tp <= 0 || (Tp % 128) || lda % 8 || ldb % 8) return NPPC_EUNSUPPORTED;
  if (epi == EPI_PRELU_STATS && (!slope || !stats)) return NPPC_EBADARG;
  if ((epi == EPI_RESIDUAL || epi == EPI_MASK_POS) && !res) return NPPC_EBADARG;
  if (epi == EPI_RESIDUAL_GN) return NPPC_EBADARG;            // has its own entry point (nppc_gemm_nt_gn)
  if (ksplit < 1) ksplit = 1;
  if (K % (32 * ksplit)) return NPPC_EUNSUPPORTED;
  GemmArgs g{A, lda, sA, B, ldb, sB, C, ldc, sC, bias, sBias, res, ldres, sRes, slope, sSlope, stats, sStats,
             R, N, K / ksplit, Tp, Tv, Nv, relu_in, ksplit, nullptr, 0, 1.0, 0.f,
             ((res ? (ldres % 8 == 0 && nt_staged())
                   : ((epi == EPI_PLAIN || epi == EPI_PRELU_STATS || epi == EPI_RELU) && ksplit == 1 && nt_staged_plain())) &&
              ldc % 8 == 0) ? 1 : 0,
             colpart, (long)(R / 128) * N};
  hipStream_t s = (hipStream_t)stream;
  const int bk = prec == NPPC_PREC_BF16 ? 64 : 32;
  const int lds_path = ((K / ksplit) % bk == 0) ? (N % 128 == 0 ? 128 : 64) : 0;
  // the tile column sums come out of the LDS kernels' epilogue for the two epilogues that produce an upstream gradient
  if (colpart && (lds_path == 0 || ksplit != 1 || (epi != EPI_RESIDUAL && epi != EPI_MASK_POS))) return NPPC_EUNSUPPORTED;
  dim3 grid(R / 128, lds_path == 128 ? N / 128 : N / 64, batch * ksplit);
#define LAUNCH(TT, E)                                                                              \
  do {                                                                                             \
    if (lds_path == 128) hipLaunchKernelGGL((gemm_nt_lds_kernel<TT, E, 128>), grid, dim3(256), 0, s, g); \
    else if (lds_path == 64) hipLaunchKernelGGL((gemm_nt_lds_kernel<TT, E, 64>), grid, dim3(256), 0, s, g); \
    else hipLaunchKernelGGL((gemm_nt_kernel<TT, E>), grid, dim3(256), 0, s, g);                    \
  } while (0)
  if (prec == NPPC_PREC_BF16) {
    switch (epi) {
      case EPI_PLAIN: LAUNCH(bf16_t, EPI_PLAIN); break;
      case EPI_PRELU_STATS: LAUNCH(bf16_t, EPI_PRELU_STATS); break;
      case EPI_RESIDUAL: LAUNCH(bf16_t, EPI_RESIDUAL); break;
      case EPI_RELU: LAUNCH(bf16_t, EPI_RELU); break;
      case EPI_PLAIN_F32: LAUNCH(bf16_t, EPI_PLAIN_F32); break;
      case EPI_MASK_POS: LAUNCH(bf16_t, EPI_MASK_POS); break;
      default: return NPPC_EBADARG;
    }
  } else if (prec == NPPC_PREC_F32) {
    switch (epi) {
      case EPI_PLAIN: LAUNCH(float, EPI_PLAIN); break;
      case EPI_PRELU_STATS: LAUNCH(float, EPI_PRELU_STATS); break;
      case EPI_RESIDUAL: LAUNCH(float, EPI_RESIDUAL); break;
      case EPI_RELU: LAUNCH(float, EPI_RELU); break;
      case EPI_PLAIN_F32: LAUNCH(float, EPI_PLAIN_F32); break;
      case EPI_MASK_POS: LAUNCH(float, EPI_MASK_POS); break;
      default: return NPPC_EBADARG;
    }
  } else {
    return NPPC_EBADARG;
  }
#undef LAUNCH
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

extern "C" {

int nppc_gemm_nt(int prec, int epi, const void* A, long lda, long sA, const void* B, long ldb, long sB, void* C, long ldc,
                 long sC, const float* bias, long sBias, const void* res, long ldres, long sRes, const float* slope,
                 long sSlope, double* stats, long sStats, int R, int N, int K, int Tp, int Tv, int Nv, int relu_in,
                 int batch, int ksplit, void* stream) {
  return launch_nt(prec, epi, A, lda, sA, B, ldb, sB, C, ldc, sC, bias, sBias, res, ldres, sRes, slope, sSlope, stats, sStats, R, N,
                   K, Tp, Tv, Nv, relu_in, batch, ksplit, nullptr, stream);
}

// nppc_gemm_nt (EPI_RESIDUAL or EPI_MASK_POS, LDS-staged shapes: K a multiple of 64 (bf16) / 32 (fp32), no K split) that also
// leaves colpart[z][R/128][N] = the column sums of every 128-row tile of the stored output: C is the upstream gradient of a
// 1x1 convolution further down the backward chain, whose bias gradient is the sum of these partials over the tiles
// (nppc_tcn_mid_bwd adds them up) -- no pass of its own over C.
int nppc_gemm_nt_colsum(int prec, int epi, const void* A, long lda, long sA, const void* B, long ldb, long sB, void* C, long ldc,
                        long sC, const float* bias, long sBias, const void* res, long ldres, long sRes, int R, int N, int K,
                        int Tp, int Tv, int Nv, int batch, float* colpart, void* stream) {
  if (!colpart) return NPPC_EBADARG;
  return launch_nt(prec, epi, A, lda, sA, B, ldb, sB, C, ldc, sC, bias, sBias, res, ldres, sRes, nullptr, 0, nullptr, 0, R, N, K,
                   Tp, Tv, Nv, 0, batch, 1, colpart, stream);
}

// sconv with the GroupNorm in front of it folded in (EPI_RESIDUAL_GN above):
//   C = rstd_b * (A Wg^T) - mean_b * rstd_b * v + u + res,   A = the un-normalised depthwise output, stats = its (sum, sumsq)
int nppc_gemm_nt_gn(int prec, const void* A, long lda, long sA, const void* Wg, long ldb, long sB, void* C, long ldc, long sC,
                    const float* u, const float* v, long sUV, const void* res, long ldres, long sRes, const double* stats,
                    long sStats, double cnt, float eps, int R, int N, int K, int Tp, int Tv, int Nv, int batch, void* stream) {
  if (!A || !Wg || !C || !u || !v || !res || !stats || R <= 0 || N <= 0 || K <= 0 || batch <= 0 || cnt <= 0) return NPPC_EBADARG;
  if (R % 128 || N % 64 || K % 32 || Tp <= 0 || (Tp % 128) || lda % 8 || ldb % 8) return NPPC_EUNSUPPORTED;
  GemmArgs g{A, lda, sA, Wg, ldb, sB, C, ldc, sC, u, sUV, res, ldres, sRes, nullptr, 0, const_cast<double*>(stats), sStats,
             R, N, K, Tp, Tv, Nv, 0, 1, v, sUV, cnt, eps, (ldc % 8 == 0 && ldres % 8 == 0 && nt_staged()) ? 1 : 0, nullptr, 0};
  hipStream_t s = (hipStream_t)stream;
  const int bk = prec == NPPC_PREC_BF16 ? 64 : 32;
  const int lds_path = (K % bk == 0) ? (N % 128 == 0 ? 128 : 64) : 0;
  dim3 grid(R / 128, lds_path == 128 ? N / 128 : N / 64, batch);
#define LAUNCH_GN(TT)                                                                                             \
  do {                                                                                                            \
    if (lds_path == 128) hipLaunchKernelGGL((gemm_nt_lds_kernel<TT, EPI_RESIDUAL_GN, 128>), grid, dim3(256), 0, s, g);   \
    else if (lds_path == 64) hipLaunchKernelGGL((gemm_nt_lds_kernel<TT, EPI_RESIDUAL_GN, 64>), grid, dim3(256), 0, s, g); \
    else hipLaunchKernelGGL((gemm_nt_kernel<TT, EPI_RESIDUAL_GN>), grid, dim3(256), 0, s, g);                      \
  } while (0)
  if (prec == NPPC_PREC_BF16) LAUNCH_GN(bf16_t);
  else if (prec == NPPC_PREC_F32) LAUNCH_GN(float);
  else return NPPC_EBADARG;
#undef LAUNCH_GN
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

int nppc_tcn_pack_sconv(int prec, const float* W, const float* gamma, const float* beta, const float* bias, void* Wg, float* u,
                        float* v, int N, int K, int Npad, int ldd, int n_a, int n_b, long src_stride_a, long src_stride_b,
                        long dst_stride_a, long dst_stride_b, void* stream) {
  if (!W || !gamma || !beta || !bias || !Wg || !u || !v || N > Npad || K > ldd || n_a <= 0 || n_b <= 0 || (long)n_a * n_b > 65535)
    return NPPC_EBADARG;
  hipStream_t s = (hipStream_t)stream;
  if (prec == NPPC_PREC_BF16)
    hipLaunchKernelGGL(pack_sconv_kernel<bf16_t>, dim3(Npad, n_a * n_b), dim3(64), 0, s, W, gamma, beta, bias, (bf16_t*)Wg, u, v, N,
                       K, Npad, ldd, n_b, src_stride_a, src_stride_b, dst_stride_a, dst_stride_b);
  else
    hipLaunchKernelGGL(pack_sconv_kernel<float>, dim3(Npad, n_a * n_b), dim3(64), 0, s, W, gamma, beta, bias, (float*)Wg, u, v, N, K,
                       Npad, ldd, n_b, src_stride_a, src_stride_b, dst_stride_a, dst_stride_b);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

int nppc_tcn_dwconv(int prec, const void* in, void* out, const double* st1, double* st2, const float* gamma,
                    const float* beta, const float* wd, const float* bd, const float* slope2, int B, int Cc, int ld, int Tp,
                    int Tv, int dil, float eps, long sAct, long sSt, long sP, int batch, void* stream) {
  if (!in || !out || !st1 || !st2 || Cc % 8 || Cc / 8 > 256) return NPPC_EBADARG;
  dim3 grid(ceil_div(Tp, DW_FRAMES), B, batch);
  hipStream_t s = (hipStream_t)stream;
  if (prec == NPPC_PREC_BF16)
    hipLaunchKernelGGL(dwconv_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)in, (bf16_t*)out, st1, st2, gamma, beta,
                       wd, bd, slope2, Cc, ld, Tp, Tv, dil, eps, sAct, sSt, sP);
  else
    hipLaunchKernelGGL(dwconv_kernel<float>, grid, dim3(256), 0, s, (const float*)in, (float*)out, st1, st2, gamma, beta, wd,
                       bd, slope2, Cc, ld, Tp, Tv, dil, eps, sAct, sSt, sP);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

int nppc_tcn_gn_apply(int prec, const void* in, void* out, const double* st, const float* gamma, const float* beta, int B,
                      int Cc, int ld, int Tp, int Tv, float eps, long sAct, long sSt, long sP, int batch, void* stream) {
  if (!in || !out || !st || Cc % 8) return NPPC_EBADARG;
  dim3 grid(ceil_div((long)Tv * (Cc / 8), 256), B, batch);
  hipStream_t s = (hipStream_t)stream;
  if (prec == NPPC_PREC_BF16)
    hipLaunchKernelGGL(gn_apply_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)in, (bf16_t*)out, st, gamma, beta, Cc,
                       ld, Tp, Tv, eps, sAct, sSt, sP);
  else
    hipLaunchKernelGGL(gn_apply_kernel<float>, grid, dim3(256), 0, s, (const float*)in, (float*)out, st, gamma, beta, Cc, ld,
                       Tp, Tv, eps, sAct, sSt, sP);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

int nppc_pack_matrix(int prec, const float* src, void* dst, int N, int K, int Npad, int ldd, int transpose, void* stream) {
  if (!src || !dst || N > Npad || K > ldd) return NPPC_EBADARG;
  const long total = (long)Npad * ldd;
  const int grid = (int)((total + 255) / 256 > 1024 ? 1024 : (total + 255) / 256);
  hipStream_t s = (hipStream_t)stream;
  if (prec == NPPC_PREC_BF16)
    hipLaunchKernelGGL(pack_matrix_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, src, (bf16_t*)dst, N, K, Npad, ldd, transpose, 1,
                       0L, 0L, 0L, 0L);
  else
    hipLaunchKernelGGL(pack_matrix_kernel<float>, dim3(grid), dim3(256), 0, s, src, (float*)dst, N, K, Npad, ldd, transpose, 1, 0L,
                       0L, 0L, 0L);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

int nppc_pack_matrix_batched(int prec, const float* src, void* dst, int N, int K, int Npad, int ldd, int transpose, int n_a,
                             int n_b, long src_stride_a, long src_stride_b, long dst_stride_a, long dst_stride_b, void* stream) {
  if (!src || !dst || N > Npad || K > ldd || n_a <= 0 || n_b <= 0 || (long)n_a * n_b > 65535) return NPPC_EBADARG;
  const long total = (long)Npad * ldd;
  const int gx = (int)((total + 255) / 256 > 256 ? 256 : (total + 255) / 256);
  hipStream_t s = (hipStream_t)stream;
  if (prec == NPPC_PREC_BF16)
    hipLaunchKernelGGL(pack_matrix_kernel<bf16_t>, dim3(gx, n_a * n_b), dim3(256), 0, s, src, (bf16_t*)dst, N, K, Npad, ldd,
                       transpose, n_b, src_stride_a, src_stride_b, dst_stride_a, dst_stride_b);
  else
    hipLaunchKernelGGL(pack_matrix_kernel<float>, dim3(gx, n_a * n_b), dim3(256), 0, s, src, (float*)dst, N, K, Npad, ldd,
                       transpose, n_b, src_stride_a, src_stride_b, dst_stride_a, dst_stride_b);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

}  // extern "C"

// =====================================================================================================
// LDS-staged NT GEMM for the long-K weight-gradient products (K = T'*N ~ 1e6, split over blockIdx.z):
//   C_slab[z][M][N] (fp32) = A[M][Kz] * B[N][Kz]^T,   A, B K-contiguous.
// 256 threads = 2x2 waves, block tile 128x128, 128-byte K slices per stage (64 bf16 / 32 f32), double-buffered LDS,
// register staging (issue the next stage's global loads before the MFMAs, write them to LDS after), one barrier
// per stage.  LDS rows are 128 B = 8 chunks of 16 B stored at slot (chunk ^ (row & 7)): both the staging
// ds_write_b128 and the fragment ds_read_b128 are bank-conflict free.
namespace {

template <typename T>
__global__ __launch_bounds__(256, 2) void gemm_nt_tiled_kernel(const T* __restrict__ A, long lda, const T* __restrict__ B,
                                                               long ldb, float* __restrict__ C, long ldc, long slab_stride,
                                                               long Kz) {
  typedef typename Frag<T>::type frag;
  constexpr int EPC = 16 / (int)sizeof(T);      // elements per 16-byte chunk
  constexpr int BK = 8 * EPC;                   // elements per 128-byte stage row
  constexpr int KS = BK / 32;                   // 32-wide MFMA k-steps per stage
  __shared__ __attribute__((aligned(16))) unsigned char lds[2][2][128 * 128];   // [buffer][A|B][row*128 B]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, q = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  const long m0 = (long)blockIdx.x * 128, n0 = (long)blockIdx.y * 128;
  const long kbase = (long)blockIdx.z * Kz;
  const T* Ab = A + m0 * lda + kbase;
  const T* Bb = B + n0 * ldb + kbase;
  float* Cz = C + (size_t)blockIdx.z * slab_stride;

  // per-thread staging slots: chunk (row, c) of the 128 x 8-chunk stage, i = 0..3
  const int srow = tid >> 3, sc = tid & 7;                       // rows srow + 32*i
  const T* ga = Ab + (long)srow * lda + sc * EPC;
  const T* gb = Bb + (long)srow * ldb + sc * EPC;
  const int soff = srow * 128 + ((sc ^ (srow & 7)) << 4);        // (srow + 32*i) & 7 == srow & 7
  uint4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
#define GLOAD(k0)                                                              \
  ra0 = *reinterpret_cast<const uint4*>(ga + (k0));                            \
  ra1 = *reinterpret_cast<const uint4*>(ga + 32 * lda + (k0));                 \
  ra2 = *reinterpret_cast<const uint4*>(ga + 64 * lda + (k0));                 \
  ra3 = *reinterpret_cast<const uint4*>(ga + 96 * lda + (k0));                 \
  rb0 = *reinterpret_cast<const uint4*>(gb + (k0));                            \
  rb1 = *reinterpret_cast<const uint4*>(gb + 32 * ldb + (k0));                 \
  rb2 = *reinterpret_cast<const uint4*>(gb + 64 * ldb + (k0));                 \
  rb3 = *reinterpret_cast<const uint4*>(gb + 96 * ldb + (k0));
#define LSTORE(buf)                                                            \
  *reinterpret_cast<uint4*>(&lds[buf][0][soff]) = ra0;                         \
  *reinterpret_cast<uint4*>(&lds[buf][0][soff + 32 * 128]) = ra1;              \
  *reinterpret_cast<uint4*>(&lds[buf][0][soff + 64 * 128]) = ra2;              \
  *reinterpret_cast<uint4*>(&lds[buf][0][soff + 96 * 128]) = ra3;              \
  *reinterpret_cast<uint4*>(&lds[buf][1][soff]) = rb0;                         \
  *reinterpret_cast<uint4*>(&lds[buf][1][soff + 32 * 128]) = rb1;              \
  *reinterpret_cast<uint4*>(&lds[buf][1][soff + 64 * 128]) = rb2;              \
  *reinterpret_cast<uint4*>(&lds[buf][1][soff + 96 * 128]) = rb3;
  auto lfrag = [&](const unsigned char* base, int row, int ks) -> frag {
    if constexpr (sizeof(T) == 2) {
      const int c = 4 * ks + q;
      return *reinterpret_cast<const frag*>(base + row * 128 + ((c ^ (row & 7)) << 4));
    } else {
      const int c0 = 2 * q, c1 = 2 * q + 1;   // KS == 1: the 32-float row is the whole k-step
      const float4 lo = *reinterpret_cast<const float4*>(base + row * 128 + ((c0 ^ (row & 7)) << 4));
      const float4 hi = *reinterpret_cast<const float4*>(base + row * 128 + ((c1 ^ (row & 7)) << 4));
      frag f;
      f[0] = lo.x; f[1] = lo.y; f[2] = lo.z; f[3] = lo.w; f[4] = hi.x; f[5] = hi.y; f[6] = hi.z; f[7] = hi.w;
      return f;
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const long nstage = Kz / BK;
  GLOAD(0)
  LSTORE(0)
  __syncthreads();
  for (long s = 0; s < nstage; ++s) {
    const int buf = (int)(s & 1);
    if (s + 1 < nstage) { GLOAD((s + 1) * BK) }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      frag af[4], bf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = lfrag(lds[buf][0], wm * 64 + 16 * i + n, ks);
#pragma unroll
      for (int j = 0; j < 4; ++j) bf[j] = lfrag(lds[buf][1], wn * 64 + 16 * j + n, ks);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mma16(af[i], bf[j], acc[i][j]);
    }
    if (s + 1 < nstage) { LSTORE(buf ^ 1) }
    __syncthreads();
  }
#undef GLOAD
#undef LSTORE
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long row = m0 + wm * 64 + 16 * i + 4 * q + r;
        const long col = n0 + wn * 64 + 16 * j + n;
        Cz[row * ldc + col] = acc[i][j][r];
      }
}

}  // namespace

extern "C" int nppc_gemm_nt_splitk(int prec, const void* A, long lda, const void* B, long ldb, float* C, long ldc, int M, int N,
                                   long K, int ksplit, void* stream) {
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || ksplit < 1) return NPPC_EBADARG;
  const int bk = prec == NPPC_PREC_BF16 ? 64 : 32;
  if (M % 128 || N % 128 || K % ((long)bk * ksplit) || lda % 8 || ldb % 8) return NPPC_EUNSUPPORTED;
  dim3 grid(M / 128, N / 128, ksplit);
  hipStream_t s = (hipStream_t)stream;
  if (prec == NPPC_PREC_BF16)
    hipLaunchKernelGGL(gemm_nt_tiled_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)A, lda, (const bf16_t*)B, ldb, C, ldc,
                       (long)M * ldc, K / ksplit);
  else if (prec == NPPC_PREC_F32)
    hipLaunchKernelGGL(gemm_nt_tiled_kernel<float>, grid, dim3(256), 0, s, (const float*)A, lda, (const float*)B, ldb, C, ldc,
                       (long)M * ldc, K / ksplit);
  else
    return NPPC_EBADARG;
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

// =====================================================================================================
// TN split-K GEMM for weight gradients on row-major operands (no transposed copies anywhere):
//   C_slab[z][m][n] (fp32) = sum_{r in slice z} A[r][m] * B[r][n],   A [R][lda], B [R][ldb] bf16, rows = (t, sequence).
// Tiles are staged row-major in LDS ([64 r][128 cols]); MFMA fragments need 8 consecutive r per lane for a fixed
// column, which ds_read_b64_tr_b16 delivers (a 4x16 block read column-major per 16-lane group, guide T10).
namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4;

// BM = 128 (4 waves, two workgroups per CU) or 256 (8 waves, one per CU: 0.75x the operand bytes per FLOP through the
// CU's 64 B/clk global-load path, which is what bounds the 128 x 128 tile)
template <int BN, int BM = 128>
__global__ __launch_bounds__(2 * BM, BM == 128 ? 2 : 1) void gemm_tn_tiled_kernel(const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ B,
                                                               long ldb, float* __restrict__ C, long ldc, long slab_stride,
                                                               long Rz, int ntap, int Wp, int shift_a, int nzb, long bsA,
                                                               long bsB, long bsC) {
  constexpr int BR = 64, NT = 2 * BM;
  constexpr int RSA = BM + 8, RSB = BN + 8;                 // LDS row strides (elements): 16-byte aligned rows
  constexpr int WN = BN / 2;                                // columns per wave (BM/64 x 2 waves)
  constexpr int NJ = WN / 16;
  __shared__ __attribute__((aligned(16))) bf16_t lds[2][BR * RSA + BR * RSB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i16 = lane & 15, qq = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs (each with its own L2) in linear id order.
  // All output tiles of one K slice read the same operand rows, so they are mapped to ONE XCD (slice = f(id % 8)):
  // the slice is fetched from HBM once and the (M/128)*(N/BN)-fold re-reads hit that XCD's L2.
  unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (gridDim.z % 8 == 0) {
    const unsigned tiles = gridDim.x * gridDim.y;
    const unsigned lin = bx + gridDim.x * (by + gridDim.y * bz);
    const unsigned xcd = lin & 7, j = lin >> 3;
    bz = xcd + 8 * (j / tiles);
    const unsigned t = j % tiles;
    bx = t % gridDim.x;
    by = t / gridDim.x;
  }
  // batches (independent problems, e.g. the three full-band branches): bz = batch * nzb + (tap * ksplit + slice)
  const unsigned bb = bz / nzb;
  bz %= nzb;
  A += (size_t)bb * bsA;
  B += (size_t)bb * bsB;
  C += (size_t)bb * bsC;
  const long m0 = (long)bx * BM, n0 = (long)by * BN;
  // bz = tap * ksplit + slice: tap t reads B (or A, shift_a) shifted by (t/3 - 1) * Wp + (t%3 - 1) rows (3x3 convolution
  // weight gradient: nine row-shifted products in one launch); ntap == 1 is the plain split-K GEMM
  const int ksl = nzb / ntap, tap = bz / ksl;
  const long rbase = (long)(bz % ksl) * Rz;
  const long boff = ntap == 9 ? (long)(tap / 3 - 1) * Wp + (tap % 3 - 1) : 0;
  const bf16_t* Ab = A + (rbase + (shift_a ? boff : 0)) * lda + m0;
  const bf16_t* Bb = B + (rbase + (shift_a ? 0 : boff)) * ldb + n0;
  float* Cz = C + (size_t)bz * slab_stride;

  // staging: A tile 64 rows x 16 chunks (16 B) = 1024 chunks -> 4 per thread; B tile 64 x BN/8 chunks
  constexpr int ACH = BM / 8, BCH = BN / 8;
  constexpr int NA = BR * ACH / NT, NB = BR * BCH / NT;
  static_assert(NA == 4 && (NB == 4 || NB == 2 || NB == 1), "staging slots are named registers");
  // chunk (row, c) of slot i: idx = tid + 256*i.  A: row = idx/16 -> rows (tid>>4) + 16*i, c = tid & 15.
  const int arow = tid / ACH, ac = tid % ACH, brow = tid / BCH, bc = tid % BCH;
  constexpr int ASTEP = NT / ACH, BSTEP = NT / BCH;                                      // rows between slots
  const bf16_t* gA = Ab + (long)arow * lda + ac * 8;
  const bf16_t* gB = Bb + (long)brow * ldb + bc * 8;
  const int sA = arow * RSA + ac * 8, sB = BR * RSA + brow * RSB + bc * 8;
  uint4 ra0, ra1, ra2, ra3, rb0, rb1 = {}, rb2 = {}, rb3 = {};
#define TN_GLOAD(r0)                                                            \
  ra0 = *reinterpret_cast<const uint4*>(gA + ((r0) + 0 * ASTEP) * lda);         \
  ra1 = *reinterpret_cast<const uint4*>(gA + ((r0) + 1 * ASTEP) * lda);         \
  ra2 = *reinterpret_cast<const uint4*>(gA + ((r0) + 2 * ASTEP) * lda);         \
  ra3 = *reinterpret_cast<const uint4*>(gA + ((r0) + 3 * ASTEP) * lda);         \
  rb0 = *reinterpret_cast<const uint4*>(gB + ((r0) + 0 * BSTEP) * ldb);         \
  if (NB >= 2) rb1 = *reinterpret_cast<const uint4*>(gB + ((r0) + 1 * BSTEP) * ldb); \
  if (NB == 4) {                                                                \
    rb2 = *reinterpret_cast<const uint4*>(gB + ((r0) + 2 * BSTEP) * ldb);       \
    rb3 = *reinterpret_cast<const uint4*>(gB + ((r0) + 3 * BSTEP) * ldb);       \
  }
#define TN_LSTORE(buf)                                                          \
  *reinterpret_cast<uint4*>(&lds[buf][sA + 0 * ASTEP * RSA]) = ra0;             \
  *reinterpret_cast<uint4*>(&lds[buf][sA + 1 * ASTEP * RSA]) = ra1;             \
  *reinterpret_cast<uint4*>(&lds[buf][sA + 2 * ASTEP * RSA]) = ra2;             \
  *reinterpret_cast<uint4*>(&lds[buf][sA + 3 * ASTEP * RSA]) = ra3;             \
  *reinterpret_cast<uint4*>(&lds[buf][sB + 0 * BSTEP * RSB]) = rb0;             \
  if (NB >= 2) *reinterpret_cast<uint4*>(&lds[buf][sB + 1 * BSTEP * RSB]) = rb1; \
  if (NB == 4) {                                                                \
    *reinterpret_cast<uint4*>(&lds[buf][sB + 2 * BSTEP * RSB]) = rb2;           \
    *reinterpret_cast<uint4*>(&lds[buf][sB + 3 * BSTEP * RSB]) = rb3;           \
  }
  // transposed fragment: 8 consecutive r (starting r0 + 8*qq) of column (col0 + lane&15)
  auto tfrag = [&](const bf16_t* tile, int rs, int r0, int col0) -> bf16x8 {
    const int q4 = i16 >> 2, p4 = i16 & 3;                  // lane 4q+p of the group supplies row q, columns 4p..4p+3
    const bf16_t* base = tile + (r0 + 8 * qq + q4) * rs + col0 + 4 * p4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + 4 * rs));
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  };

  f32x4 acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const long nstage = Rz / BR;
  TN_GLOAD(0)
  TN_LSTORE(0)
  __syncthreads();
  for (long s = 0; s < nstage; ++s) {
    const int buf = (int)(s & 1);
    if (s + 1 < nstage) { TN_GLOAD((s + 1) * BR) }
    const bf16_t* ta = lds[buf];
    const bf16_t* tb = lds[buf] + BR * RSA;
#pragma unroll
    for (int ks = 0; ks < BR / 32; ++ks) {
      bf16x8 af[4], bfr[NJ];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = tfrag(ta, RSA, 32 * ks, wm * 64 + 16 * i);
#pragma unroll
      for (int j = 0; j < NJ; ++j) bfr[j] = tfrag(tb, RSB, 32 * ks, wn * WN + 16 * j);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = mma16(af[i], bfr[j], acc[i][j]);
    }
    if (s + 1 < nstage) { TN_LSTORE(buf ^ 1) }
    __syncthreads();
  }
#undef TN_GLOAD
#undef TN_LSTORE
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long row = m0 + wm * 64 + 16 * i + 4 * qq + r;
        const long col = n0 + wn * WN + 16 * j + i16;
        Cz[row * ldc + col] = acc[i][j][r];
      }
}

}  // namespace

namespace {
// =====================================================================================================
// LDS-DMA ring variant of the TN split-K GEMM (256 x 128 output tile, 8 waves as 4 x 2, 64 x 64 per wave).
// Why: the register-staged kernel above keeps ONE stage of global loads in flight per workgroup (issued at the top of a
// stage, written to LDS at its end), so every stage waits out most of an HBM round trip: 711 TFLOP/s = 0.28 of the MFMA
// peak on the LSTM weight-gradient shapes.  Here the operand rows stream global -> LDS directly (`global_load_lds_dwordx4`,
// no staging registers, no ds_write pass) into a ring of DR_STAGES stages of DR_BR K-rows, DR_STAGES - 1 stages ahead of the
// MFMAs; a stage is retired with a COUNTED s_waitcnt vmcnt (never 0 in the loop) and ONE raw s_barrier (a __syncthreads()
// would drain the whole ring: hipcc emits vmcnt(0) in front of it while an LDS-DMA is in flight).
// LDS image of a stage: A rows [BR][256] bf16 (512 B) then B rows [BR][BN], each row linear but with its 32-byte chunks
// rotated by a per-row key (dr_key below): the 8 rows a 32-lane half of ds_read_b64_tr_b16 touches then sit on 8 different
// 32-byte bank groups (un-swizzled they all hit the same 8 banks).  The DMA writes LDS lane-linearly, so the swizzle is
// applied to each lane's SOURCE address (cdna_hip_programming.md, LDS-DMA caveat).
// Two geometries (template): <BN 128, 64-row stages, 3-stage ring> and <BN 192, 32-row stages, 5-stage ring>.  The kernel is
// bound by bytes in flight per CU / memory latency (~800 TFLOP/s at 87 FLOP per operand byte with ~100 KB in flight), so
// the 256 x 192 tile (110 FLOP/B; N = 384 = 2 tiles, 6 x 2 x 64 slices = exactly 3 rounds of 256 workgroups) is the one
// the LSTM weight gradients use.
constexpr int DR_BM = 256;

// LDS row swizzle: 32-byte chunk c of row r is stored at chunk (c + key(r)) mod (row bytes / 32), key(r) = (r & 3) | ((r >> 3) & 1) << 2:
// the 8 rows one 32-lane half of ds_read_b64_tr_b16 touches (r0 .. r0+3 and r0+8 .. r0+11) get 8 different keys -> 8 different
// 32-byte bank groups (a rotation, not an XOR, so that it also works for the 12-chunk rows of the 192-column tile)
// (rows of fewer than 8 chunks -- the 64-column tile -- keep the low bits only: two rows of a half then share a bank group)
__device__ __forceinline__ int dr_key(int row, int nch) {
  const int k = (row & 3) | (((row >> 3) & 1) << 2);
  return nch >= 8 ? k : (k & (nch - 1));
}
__device__ __forceinline__ int dr_phys(int bytes_in_row, int key, int nch) {      // logical byte offset -> stored byte offset
  const int c = bytes_in_row >> 5;
  int pc = c + key;
  if (pc >= nch) pc -= nch;
  return (pc << 5) | (bytes_in_row & 31);
}
__device__ __forceinline__ int dr_logical(int bytes_in_row, int key, int nch) {   // stored byte offset -> logical byte offset
  const int pc = bytes_in_row >> 5;
  int c = pc - key;
  if (c < 0) c += nch;
  return (c << 5) | (bytes_in_row & 31);
}

// One 256 x BN output tile of one K slice: A rows [rbaseA, rbaseA + Rz) x columns [m0, m0 + 256), B rows [rbaseB, ...) x
// columns [n0, n0 + BN) -> Cz[m0 + ..][c0 + ..] (c0: the tile's first column in the slab, = n0 unless B is one of several
// sources side by side); `rs` != nullptr: also the row sums of the A tile -> rs[m0 + ..].
template <int BM, int BN, int BR, int STAGES>
__device__ __forceinline__ void dma_tile(const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ B, long ldb,
                                         float* __restrict__ Cz, long ldc, long m0, long n0, long c0, long rbaseA, long rbaseB,
                                         long Rz, float* __restrict__ rs) {
  constexpr int KS = BR / 32;                                   // 32-deep MFMA k-steps per stage
  constexpr int A_BYTES = BR * BM * 2, B_BYTES = BR * BN * 2, STAGE_BYTES = A_BYTES + B_BYTES;
  constexpr int MI = BM / 64;                                   // 16-row tiles per wave (4 waves across M)
  // TN_DMA_PRODUCERS = 4: only the second wave of every SIMD (waves 4-7) requests operands, twice the pieces each: behind the stage
  // barrier the first waves go straight to their matrix instructions (round 4, from the convolution ring kernel: csrc/unet.hip).
  // LSTM weight-gradient shape alone 1.365 -> 1.315 ms, step -0.11 ms in an alternating A/B (profiles/r04_tn_producers_ab.txt);
  // 8 = every wave requests its share (rounds 2-3).
#ifndef TN_DMA_PRODUCERS
#define TN_DMA_PRODUCERS 4
#endif
  constexpr int NPW = TN_DMA_PRODUCERS;
  constexpr int NA = A_BYTES / 1024 / NPW;                      // whole 1 KB DMA pieces per requesting wave and stage (A)
  constexpr int BW = B_BYTES / NPW;                             // B bytes per requesting wave and stage: NBF whole pieces + one half piece
  constexpr int NBF = BW / 1024, NBH = (BW % 1024) / 512;
  static_assert(A_BYTES % (1024 * NPW) == 0 && BW % 512 == 0 && NBH <= 1 && (NPW == 4 || NPW == 8), "DMA pieces must divide evenly over the requesting waves");
  constexpr int NDMA = NA + NBF + NBH;                          // DMA instructions per wave and stage (the vmcnt unit)
  constexpr int NJ = BN / 32;                                   // 16-column tiles per wave (2 waves across N)
  constexpr int ACH = BM * 2 / 32, BCH = BN * 2 / 32;           // 32-byte chunks per row
  extern __shared__ __attribute__((aligned(1024))) unsigned char dr_lds[];
  typedef __attribute__((address_space(3))) void lds_void;
  const int tid = threadIdx.x, lane = tid & 63, i16 = lane & 15, qq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const bool producer = wave >= 8 - NPW;
  const int pwv = producer ? wave - (8 - NPW) : 0;

  // ---- this wave's DMA pieces of a stage.  A: pieces NA*w + h, 1 KB = 2 rows of 512 B.  B: bytes [w*BW, (w+1)*BW) of the
  // stage's B image (rows of BN*2 bytes back to back): NBF whole pieces and, if BW is not a multiple of 1 KB, one half piece
  // issued by lanes 0-31 only (the LDS destination of a lane is piece base + 16 * lane, masked lanes write nothing)
  const unsigned char* srcA[NA];
  const unsigned char* srcB[NBF + NBH];
#pragma unroll
  for (int h = 0; h < NA; ++h) {
    const int o = (NA * pwv + h) * 1024 + lane * 16;            // byte offset inside the stage's A image (rows of BM*2 bytes)
    const int row = o / (BM * 2);
    const int p = dr_logical(o % (BM * 2), dr_key(row, ACH), ACH);
    srcA[h] = reinterpret_cast<const unsigned char*>(A + (rbaseA + row) * lda + m0) + p;
  }
#pragma unroll
  for (int h = 0; h < NBF + NBH; ++h) {
    const int o = pwv * BW + h * 1024 + lane * 16;              // byte offset inside the stage's B image
    const int row = o / (BN * 2);
    const int p = dr_logical(o % (BN * 2), dr_key(row, BCH), BCH);
    srcB[h] = reinterpret_cast<const unsigned char*>(B + (rbaseB + row) * ldb + n0) + p;
  }
  const long strA = (long)BR * lda * 2, strB = (long)BR * ldb * 2;        // bytes per stage
  // The DMAs go out through inline asm: issued with the builtin, hipcc treats every later ds_read as a possible reader of
  // the pending LDS write and puts s_waitcnt vmcnt(0) in front of each stage's first fragment read, which drains the ring
  // (the stage issued a moment earlier included) -- the pipeline this kernel exists for.  Hidden from the compiler, their
  // completion is counted by hand below (cdna_hip_programming.md section 5.7 item 1: an LDS-DMA has no VGPR destination,
  // so it is register-safe).  M0 carries the wave-uniform LDS byte address of each piece.
  const unsigned lds_base = (unsigned)(unsigned long)(lds_void*)dr_lds;
  auto dma = [&](const unsigned char* src, unsigned ldst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(ldst) : "memory");
  };
  auto fill = [&](int slot, long st) {
    if (!producer) return;
    const unsigned l = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)slot * STAGE_BYTES);
#pragma unroll
    for (int h = 0; h < NA; ++h) dma(srcA[h] + st * strA, l + (NA * pwv + h) * 1024);
#pragma unroll
    for (int h = 0; h < NBF; ++h) dma(srcB[h] + st * strB, l + A_BYTES + pwv * BW + h * 1024);
    if constexpr (NBH != 0) {
      if (lane < 32) dma(srcB[NBF] + st * strB, l + A_BYTES + pwv * BW + NBF * 1024);
    }
  };
  // transposed fragment of k-rows [32 ks + 8 qq, + 8) x 16 columns starting at col0 (ds_read_b64_tr_b16, two 4-row halves)
  const int q4 = i16 >> 2, p4 = i16 & 3;
  auto tfrag = [&](const unsigned char* tile, int rowbytes, int nch, int ks, int col0) -> bf16x8 {
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const int r_lo = 32 * ks + 8 * qq + q4, r_hi = r_lo + 4;
    const int cb = col0 * 2 + 8 * p4;
    const unsigned char* lo_p = tile + r_lo * rowbytes + dr_phys(cb, dr_key(r_lo, nch), nch);
    const unsigned char* hi_p = tile + r_hi * rowbytes + dr_phys(cb, dr_key(r_hi, nch), nch);
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lo_p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(hi_p));
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
  };

  f32x4 acc[MI][NJ];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // rowsum (optional): sum_r A[r][m] over the K slice -- the A fragments times an all-ones B fragment, by the waves that
  // hold column tile 0 (a few extra MFMAs): a bias gradient comes out of the weight-gradient product that reads the same
  // gate gradients, instead of out of a second pass over them
  const bool do_rowsum = rs != nullptr && wn == 0;
  f32x4 racc[MI];
#pragma unroll
  for (int i = 0; i < MI; ++i) racc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  typedef __attribute__((ext_vector_type(8))) short ones_t;
  const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_t{0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80});

  const long nstage = Rz / BR;
#pragma unroll
  for (int s0 = 0; s0 < STAGES - 1; ++s0)
    if (s0 < nstage) fill(s0, s0);
  int slot = 0;
  for (long st = 0; st < nstage; ++st) {
    // stage st has landed once at most NDMA * (stages issued after it) of this wave's youngest DMAs are outstanding
    const long younger = nstage - 1 - st < STAGES - 2 ? nstage - 1 - st : STAGES - 2;
    static_assert(STAGES <= 5, "the counted waits below cover rings of up to 5 stages");
    static_assert(3 * NDMA <= 63, "vmcnt is a six-bit count");
    if (producer) {                              // (the other waves learn it from the barrier)
      if (younger >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NDMA) : "memory");
      else if (younger == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NDMA) : "memory");
      else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_barrier" ::: "memory");      // every wave's pieces of stage st landed; everyone is done reading stage st - 1
    if (st + STAGES - 1 < nstage) {
      int fs = slot - 1;
      if (fs < 0) fs += STAGES;
      fill(fs, st + STAGES - 1);                // into the slot stage st - 1 occupied
    }
    const unsigned char* ta = dr_lds + (size_t)slot * STAGE_BYTES;
    const unsigned char* tb = ta + A_BYTES;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bf16x8 af[MI], bfr[NJ];
#pragma unroll
      for (int i = 0; i < MI; ++i) af[i] = tfrag(ta, BM * 2, ACH, ks, wm * (BM / 4) + 16 * i);
#pragma unroll
      for (int j = 0; j < NJ; ++j) bfr[j] = tfrag(tb, BN * 2, BCH, ks, wn * (BN / 2) + 16 * j);
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = mma16(af[i], bfr[j], acc[i][j]);
      if (do_rowsum) {
#pragma unroll
        for (int i = 0; i < MI; ++i) racc[i] = mma16(af[i], ones, racc[i]);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this wave's fragment reads are complete before it can reach the next barrier
    slot = slot + 1 == STAGES ? 0 : slot + 1;
  }
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long row = m0 + wm * (BM / 4) + 16 * i + 4 * qq + r;
        const long col = c0 + wn * (BN / 2) + 16 * j + i16;
        Cz[row * ldc + col] = acc[i][j][r];
      }
  if (do_rowsum && i16 == 0) {                      // every column of racc holds the row sum: lane column 0 stores it
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) rs[m0 + wm * (BM / 4) + 16 * i + 4 * qq + r] = racc[i][r];
  }
}

// all output tiles of one K slice on ONE XCD (see gemm_tn_tiled_kernel): remaps (bx, by, bz) when gridDim.z % 8 == 0
__device__ __forceinline__ void dr_xcd_remap(unsigned& bx, unsigned& by, unsigned& bz) {
  if (gridDim.z % 8 == 0) {
    const unsigned tiles = gridDim.x * gridDim.y;
    const unsigned lin = bx + gridDim.x * (by + gridDim.y * bz);
    const unsigned xcd = lin & 7, j = lin >> 3;
    bz = xcd + 8 * (j / tiles);
    const unsigned t = j % tiles;
    bx = t % gridDim.x;
    by = t / gridDim.x;
  }
}

template <int BN, int BR, int STAGES>
__global__ __launch_bounds__(512, 1) void gemm_tn_dma_kernel(const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ B,
                                                             long ldb, float* __restrict__ C, long ldc, long slab_stride, long Rz,
                                                             int nzb, long bsA, long bsB, long bsC, float* __restrict__ rowsum,
                                                             int ntap, int Wp, int shift_a) {
  unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  dr_xcd_remap(bx, by, bz);
  const unsigned bb = bz / nzb;
  bz %= nzb;
  A += (size_t)bb * bsA;
  B += (size_t)bb * bsB;
  C += (size_t)bb * bsC;
  // bz = tap * ksplit + slice (ntap == 9: the nine row-shifted products of a 3x3 convolution weight gradient, see
  // gemm_tn_tiled_kernel; the operands carry guard rows for the negative shifts)
  const int ksl = nzb / ntap, tap = bz / ksl;
  const long boff = ntap == 9 ? (long)(tap / 3 - 1) * Wp + (tap % 3 - 1) : 0;
  const long rbase = (long)(bz % ksl) * Rz;
  float* rs = rowsum != nullptr && by == 0 ? rowsum + ((size_t)bb * nzb + bz) * ((size_t)gridDim.x * DR_BM) : nullptr;
  dma_tile<DR_BM, BN, BR, STAGES>(A, lda, B, ldb, C + (size_t)bz * slab_stride, ldc, (long)bx * DR_BM, (long)by * BN, (long)by * BN,
                           rbase + (shift_a ? boff : 0), rbase + (shift_a ? 0 : boff), Rz, rs);
}

// Two B operands side by side behind ONE pass over A: C[z][m][0 .. N1) = A^T . B1, C[z][m][N1 .. N1 + N2) = A^T . B2.
// Column tiles 0 .. nt1-1 are 192 wide and read B1; then either N2 / 192 tiles of 192 or (N2 == 64) one 64-wide tile read
// B2.  All column tiles of a K slice run on one XCD at about the same time, so the A rows come from HBM once and from that
// XCD's L2 for the other tiles: the LSTM weight gradients that share their gate gradients (W_ih | W_hh of a layer) cost one
// pass over them instead of two or three.
template <int BM, int BR, int STAGES>
__global__ __launch_bounds__(512, 1) void gemm_tn_dma2_kernel(const bf16_t* __restrict__ A, long lda, const bf16_t* __restrict__ B1,
                                                              long ldb1, int nt1, const bf16_t* __restrict__ B2, long ldb2, int N2,
                                                              float* __restrict__ C, long ldc, long slab_stride, long Rz,
                                                              float* __restrict__ rowsum) {
  unsigned bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  dr_xcd_remap(bx, by, bz);
  const long m0 = (long)bx * BM, rbase = (long)bz * Rz;
  float* Cz = C + (size_t)bz * slab_stride;
  float* rs = rowsum != nullptr && by == 0 ? rowsum + (size_t)bz * ((size_t)gridDim.x * BM) : nullptr;
  if ((int)by >= nt1 && N2 == 64) {
    dma_tile<BM, 64, BR, STAGES>(A, lda, B2, ldb2, Cz, ldc, m0, 0, (long)nt1 * 192, rbase, rbase, Rz, nullptr);
    return;
  }
  const bool second = (int)by >= nt1;
  const long n0 = (long)(second ? by - nt1 : by) * 192;
  dma_tile<BM, 192, BR, STAGES>(A, lda, second ? B2 : B1, second ? ldb2 : ldb1, Cz, ldc, m0, n0, (long)by * 192, rbase, rbase, Rz, rs);
}

}  // namespace

// C_slab[z] [M][N] = A[rows slice z][M]^T * B[rows slice z][N]  (bf16 operands, fp32 slabs).  M % 128 == 0,
// N % 64 == 0, R % (64*ksplit) == 0, lda/ldb multiples of 8.
// 256-row tiles are on unless NPPC_TN_BIG=0 (A/B switch for tools/bench_tn.py)
static bool tn_big_tiles() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("NPPC_TN_BIG");
    v = (e && e[0] == '0') ? 0 : 1;
  }
  return v == 1;
}

// NPPC_TN_DMA=0 switches the LDS-DMA ring kernel off (A/B switch for tools/bench_tn.py)
static bool tn_dma() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("NPPC_TN_DMA");
    v = (e && e[0] == '0') ? 0 : 1;
  }
  return v == 1;
}

// NPPC_TN_DMA192=0: 256 x 128 tiles only (A/B switch)
static bool tn_dma192() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("NPPC_TN_DMA192");
    v = (e && e[0] == '0') ? 0 : 1;
  }
  return v == 1;
}

static int launch_tn(const void* A, long lda, const void* B, long ldb, float* C, long ldc, int M, int N, long R, int ksplit,
                     int ntap, int Wp, int shift_a, void* stream, int batch = 1, long sA = 0, long sB = 0, long sC = 0,
                     float* rowsum = nullptr) {
  if (!A || !B || !C || M <= 0 || N <= 0 || R <= 0 || ksplit < 1 || (ntap != 1 && ntap != 9)) return NPPC_EBADARG;
  if (M % 128 || N % 64 || R % (64L * ksplit) || lda % 8 || ldb % 8) return NPPC_EUNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  // LDS-DMA ring kernel (plain products and the nine-tap convolution weight gradient): 256 x 192 tiles (32-row stages x 5) when N is a multiple of 192, else
  // 256 x 128 tiles (64-row stages x 3), as long as there are enough workgroups to fill the chip
  if (tn_dma() && M % DR_BM == 0 && (R / ksplit) % 64 == 0 && (ntap == 1 || batch == 1)) {
    const int bn = (N % 192 == 0 && tn_dma192()) ? 192 : (N % 128 == 0 ? 128 : 0);
    if (bn && (long)(M / DR_BM) * (N / bn) * ksplit * ntap * batch >= 256) {
      dim3 grid(M / DR_BM, N / bn, ksplit * ntap * batch);
      if (bn == 192) {
        constexpr int smem = 5 * (32 * DR_BM * 2 + 32 * 192 * 2);
        static bool attr = false;
        if (!attr) {
          if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_dma_kernel<192, 32, 5>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
            return NPPC_ELAUNCH;
          attr = true;
        }
        hipLaunchKernelGGL((gemm_tn_dma_kernel<192, 32, 5>), grid, dim3(512), smem, s, (const bf16_t*)A, lda, (const bf16_t*)B, ldb, C,
                           ldc, (long)M * ldc, R / ksplit, ksplit * ntap, sA, sB, sC, rowsum, ntap, Wp, shift_a);
      } else {
        constexpr int smem = 3 * (64 * DR_BM * 2 + 64 * 128 * 2);
        static bool attr = false;
        if (!attr) {
          if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_dma_kernel<128, 64, 3>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
            return NPPC_ELAUNCH;
          attr = true;
        }
        hipLaunchKernelGGL((gemm_tn_dma_kernel<128, 64, 3>), grid, dim3(512), smem, s, (const bf16_t*)A, lda, (const bf16_t*)B, ldb, C,
                           ldc, (long)M * ldc, R / ksplit, ksplit * ntap, sA, sB, sC, rowsum, ntap, Wp, shift_a);
      }
      NPPC_CHECK_LAUNCH();
      return NPPC_OK;
    }
  }
  if (rowsum) return NPPC_EUNSUPPORTED;            // only the LDS-DMA kernel produces row sums
  // 256-row tiles (+4...11 % at these shapes) only when they still give every CU a few workgroups
  if (N % 128 == 0 && M % 256 == 0 && tn_big_tiles() && (long)(M / 256) * (N / 128) * ksplit * ntap * batch >= 1024) {
    dim3 grid(M / 256, N / 128, ksplit * ntap * batch);
    hipLaunchKernelGGL((gemm_tn_tiled_kernel<128, 256>), grid, dim3(512), 0, s, (const bf16_t*)A, lda, (const bf16_t*)B, ldb, C,
                       ldc, (long)M * ldc, R / ksplit, ntap, Wp, shift_a, ksplit * ntap, sA, sB, sC);
  } else if (N % 128 == 0) {
    dim3 grid(M / 128, N / 128, ksplit * ntap * batch);
    hipLaunchKernelGGL(gemm_tn_tiled_kernel<128>, grid, dim3(256), 0, s, (const bf16_t*)A, lda, (const bf16_t*)B, ldb, C, ldc,
                       (long)M * ldc, R / ksplit, ntap, Wp, shift_a, ksplit * ntap, sA, sB, sC);
  } else {
    dim3 grid(M / 128, N / 64, ksplit * ntap * batch);
    hipLaunchKernelGGL(gemm_tn_tiled_kernel<64>, grid, dim3(256), 0, s, (const bf16_t*)A, lda, (const bf16_t*)B, ldb, C, ldc,
                       (long)M * ldc, R / ksplit, ntap, Wp, shift_a, ksplit * ntap, sA, sB, sC);
  }
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

extern "C" int nppc_gemm_tn_splitk(const void* A, long lda, const void* B, long ldb, float* C, long ldc, int M, int N, long R,
                                   int ksplit, void* stream) {
  return launch_tn(A, lda, B, ldb, C, ldc, M, N, R, ksplit, 1, 0, 0, stream);
}

// The same product plus rowsum[z][m] = sum over the rows of slice z of A[r][m] (fp32, [ksplit][M]): the bias gradient that
// belongs to a weight gradient, from the same pass over the gate gradients.  Needs the LDS-DMA kernel's shapes
// (M % 256 == 0, N % 128 == 0, (R / ksplit) % 64 == 0, >= 256 workgroups); NPPC_EUNSUPPORTED otherwise.
extern "C" int nppc_gemm_tn_splitk_rowsum(const void* A, long lda, const void* B, long ldb, float* C, long ldc, int M, int N, long R,
                                          int ksplit, float* rowsum, void* stream) {
  if (!rowsum) return NPPC_EBADARG;
  return launch_tn(A, lda, B, ldb, C, ldc, M, N, R, ksplit, 1, 0, 0, stream, 1, 0, 0, 0, rowsum);
}

// One pass over A for two B operands side by side (gemm_tn_dma2_kernel): C_slab[z][m][0 .. N1) = A^T . B1 and
// C_slab[z][m][N1 .. N1 + N2) = A^T . B2 over the rows of slice z; rowsum optional ([ksplit][M], as above).  bf16, M % 256 == 0,
// N1 % 192 == 0, N2 == 64 or N2 % 192 == 0, (R / ksplit) % 64 == 0, ldc >= N1 + N2; NPPC_EUNSUPPORTED otherwise (the caller
// then runs the two products separately).
extern "C" int nppc_gemm_tn_splitk2(const void* A, long lda, const void* B1, long ldb1, int N1, const void* B2, long ldb2, int N2,
                                    float* C, long ldc, int M, long R, int ksplit, float* rowsum, void* stream) {
  if (!A || !B1 || !B2 || !C || M <= 0 || N1 <= 0 || N2 <= 0 || R <= 0 || ksplit < 1) return NPPC_EBADARG;
  if (lda % 8 || ldb1 % 8 || ldb2 % 8 || ldc < N1 + N2) return NPPC_EBADARG;
  if (!tn_dma() || N1 % 192 || (N2 != 64 && N2 % 192) || R % ksplit || (R / ksplit) % 64) return NPPC_EUNSUPPORTED;
  const int nt1 = N1 / 192, nt2 = N2 == 64 ? 1 : N2 / 192;
  hipStream_t s = (hipStream_t)stream;
  // 384-row tiles (4 stages of 32 K-rows) when M allows: 128 FLOP per operand byte in flight instead of 110, and for the
  // LSTM shapes (M = 1536, 64 K slices) 4 x 4 x 64 = 1024 and 4 x 3 x 64 = 768 workgroups = whole rounds of the 256 CUs;
  // NPPC_TN_BM384=0 keeps the 256-row tiles (A/B switch)
  static const bool bm384 = [] { const char* e = getenv("NPPC_TN_BM384"); return !(e && e[0] == '0'); }();
  if (bm384 && M % 384 == 0 && (long)(M / 384) * (nt1 + nt2) * ksplit >= 256) {
    constexpr int smem = 4 * (32 * 384 * 2 + 32 * 192 * 2);
    static bool attr = false;
    if (!attr) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_dma2_kernel<384, 32, 4>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
        return NPPC_ELAUNCH;
      attr = true;
    }
    hipLaunchKernelGGL((gemm_tn_dma2_kernel<384, 32, 4>), dim3(M / 384, nt1 + nt2, ksplit), dim3(512), smem, s, (const bf16_t*)A, lda,
                       (const bf16_t*)B1, ldb1, nt1, (const bf16_t*)B2, ldb2, N2, C, ldc, (long)M * ldc, R / ksplit, rowsum);
    NPPC_CHECK_LAUNCH();
    return NPPC_OK;
  }
  if (M % DR_BM || (long)(M / DR_BM) * (nt1 + nt2) * ksplit < 256) return NPPC_EUNSUPPORTED;
  constexpr int smem = 5 * (32 * DR_BM * 2 + 32 * 192 * 2);
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_dma2_kernel<DR_BM, 32, 5>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
      return NPPC_ELAUNCH;
    attr = true;
  }
  hipLaunchKernelGGL((gemm_tn_dma2_kernel<DR_BM, 32, 5>), dim3(M / DR_BM, nt1 + nt2, ksplit), dim3(512), smem, s, (const bf16_t*)A,
                     lda, (const bf16_t*)B1, ldb1, nt1, (const bf16_t*)B2, ldb2, N2, C, ldc, (long)M * ldc, R / ksplit, rowsum);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

// `batch` independent products in one launch (operand / slab strides in elements; slabs of batch b start at C + b*sC)
extern "C" int nppc_gemm_tn_splitk_batched(const void* A, long lda, long sA, const void* B, long ldb, long sB, float* C, long ldc,
                                           long sC, int M, int N, long R, int ksplit, int batch, void* stream) {
  if (batch < 1) return NPPC_EBADARG;
  return launch_tn(A, lda, B, ldb, C, ldc, M, N, R, ksplit, 1, 0, 0, stream, batch, sA, sB, sC);
}

// Nine row-shifted TN products in one launch (3x3 convolution weight gradient, csrc/unet.hip):
//   C_slab[t * ksplit + z][m][n] = sum_{r in slice z} A[r][m] * B[r + off_t][n],  off_t = (t/3 - 1) * Wp + (t%3 - 1)
// (shift_a != 0: the shift is applied to A's rows instead: sum_r A[r + off_t][m] * B[r][n])
extern "C" int nppc_gemm_tn_splitk_taps(const void* A, long lda, const void* B, long ldb, float* C, long ldc, int M, int N,
                                        long R, int ksplit, int Wp, int shift_a, void* stream) {
  return launch_tn(A, lda, B, ldb, C, ldc, M, N, R, ksplit, 9, Wp, shift_a, stream);
}
