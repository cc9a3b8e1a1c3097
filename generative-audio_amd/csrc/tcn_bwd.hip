// Backward of the TCNBlock's normalisation / activation / depthwise stages (the 1x1-conv gradients are NT GEMMs
// in tcn.hip on transposed operands).  Reference forward: audio_zen/model/module/causal_conv.py:96-108.
// All tensors [z][B][Tp][C] time-major (C = 512 hidden channels), per-channel parameter gradients are
// accumulated with fp32 atomics (one partial per workgroup), per-sample GroupNorm sums in fp64.
#include "common.h"
#include "nppc_hip.h"

namespace {

// frames per workgroup (each workgroup ends with per-channel atomics: fewer, larger workgroups); NPPC_TCN_RPB overrides
static int tcn_rpb() {
  static const int v = [] { const char* e = getenv("NPPC_TCN_RPB"); const int r = e ? atoi(e) : 64; return r >= 4 ? r : 64; }();
  return v;
}
static int diag_noatom() {
  static const int v = [] { const char* e = getenv("NPPC_DIAG_NOATOM"); return e ? atoi(e) : 0; }();
  return v;
}

constexpr int UNR = 4;     // rows in flight per thread and trip

struct GnCtx {
  float mean, rstd;
};
__device__ __forceinline__ GnCtx gn_ctx(const double* st, int b, double cnt, float eps) {
  const double m = st[b * 2] / cnt;
  const double var = st[b * 2 + 1] / cnt - m * m;
  return {(float)m, (float)(1.0 / sqrt((var > 0 ? var : 0) + (double)eps))};
}

// pass 1 of GroupNorm backward:  S[b] = (sum dxh, sum dxh*xh), dgamma[c] += sum dA*xh, dbeta[c] += sum dA
//   xh = (y - mean) * rstd,  dxh = dA * gamma
template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_reduce_kernel(const T* __restrict__ dA, const T* __restrict__ y,
                                                            const double* __restrict__ st, const float* __restrict__ gamma,
                                                            double* __restrict__ S, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, int Cc, int Tp, int Tv, float eps,
                                                            long sAct, long sSt, long sP, int RPB, int noatom) {
  extern __shared__ float sm[];   // [2][Cc] channel partials
  const int z = blockIdx.z, b = blockIdx.y;
  dA += (size_t)z * sAct; y += (size_t)z * sAct;
  st += (size_t)z * sSt; S += (size_t)z * sSt;
  gamma += (size_t)z * sP; dgamma += (size_t)z * sP; dbeta += (size_t)z * sP;
  const GnCtx c = gn_ctx(st, b, (double)Cc * Tv, eps);
  for (int i = threadIdx.x; i < 2 * Cc; i += 256) sm[i] = 0.f;
  __syncthreads();
  const int cpr = Cc / 8, rpi = 256 / cpr;
  const int tl = threadIdx.x / cpr, c8 = (threadIdx.x % cpr) * 8;
  float s1 = 0.f, s2 = 0.f, dg[8], db[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) dg[i] = db[i] = 0.f;
  float g8[8];
  loadf8(gamma + c8, g8);
  if (tl < rpi) {
    // UNR rows per trip, all loads first: one row per trip waited out a full memory latency per 2 x 16 bytes
    const int tend = (blockIdx.x + 1) * RPB < Tv ? (blockIdx.x + 1) * RPB : Tv;
    for (int tb = blockIdx.x * RPB + tl; tb < tend; tb += UNR * rpi) {
      float dv[UNR][8], yv8[UNR][8];
#pragma unroll
      for (int j = 0; j < UNR; ++j) {
        const int t = tb + j * rpi;
        if (t < tend) {
          const size_t o = ((size_t)b * Tp + t) * Cc + c8;
          load8<T>(dA + o, dv[j]);
          load8<T>(y + o, yv8[j]);
        }
      }
#pragma unroll
      for (int j = 0; j < UNR; ++j) {
        if (tb + j * rpi < tend) {
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const float d = dv[j][i];
            const float xh = (yv8[j][i] - c.mean) * c.rstd;
            const float dxh = d * g8[i];
            s1 += dxh;
            s2 += dxh * xh;
            dg[i] += d * xh;
            db[i] += d;
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      atomicAdd(&sm[c8 + i], dg[i]);
      atomicAdd(&sm[Cc + c8 + i], db[i]);
    }
  }
  const double d1 = wave_sum((double)s1), d2 = wave_sum((double)s2);
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(S + b * 2, d1);
    atomicAdd(S + b * 2 + 1, d2);
  }
  __syncthreads();
  if (noatom) return;
  for (int i = threadIdx.x; i < Cc; i += 256) {
    atomicAdd(dgamma + i, sm[i]);
    atomicAdd(dbeta + i, sm[Cc + i]);
  }
}

// pass 2: dy = rstd * (dxh - S1/cnt - xh * S2/cnt);  PReLU backward through y = prelu(pre):
//   dpre = y > 0 ? dy : a*dy;   da += sum_{y<0} dy * y / a      (pre = y/a for y < 0; assumes slope a > 0)
template <typename T>
__global__ __launch_bounds__(256) void gn_prelu_bwd_kernel(const T* __restrict__ dA, const T* __restrict__ y,
                                                           const double* __restrict__ st, const float* __restrict__ gamma,
                                                           const double* __restrict__ S, const float* __restrict__ slope,
                                                           T* __restrict__ dpre, float* __restrict__ dslope, int Cc, int Tp,
                                                           int Tv, float eps, long sAct, long sSt, long sP, int RPB, int noatom) {
  const int z = blockIdx.z, b = blockIdx.y;
  dA += (size_t)z * sAct; y += (size_t)z * sAct; dpre += (size_t)z * sAct;
  st += (size_t)z * sSt; S += (size_t)z * sSt;
  gamma += (size_t)z * sP;
  const float a = slope[(size_t)z * sP];
  const double cnt = (double)Cc * Tv;
  const GnCtx c = gn_ctx(st, b, cnt, eps);
  const float m1 = (float)(S[b * 2] / cnt), m2 = (float)(S[b * 2 + 1] / cnt);
  const int cpr = Cc / 8, rpi = 256 / cpr;
  const int tl = threadIdx.x / cpr, c8 = (threadIdx.x % cpr) * 8;
  float da = 0.f;
  const float inva = 1.f / a;
  float g8[8];
  loadf8(gamma + c8, g8);
  constexpr int UNR = 1;      // pure streaming (2 loads + 1 store per row): occupancy hides the latency, batching rows only costs registers
  if (tl < rpi) {
    const int tend = (blockIdx.x + 1) * RPB < Tp ? (blockIdx.x + 1) * RPB : Tp;
    for (int tb = blockIdx.x * RPB + tl; tb < tend; tb += UNR * rpi) {
      float dv[UNR][8], yv8[UNR][8];
#pragma unroll
      for (int j = 0; j < UNR; ++j) {
        const int t = tb + j * rpi;
        if (t < tend && t < Tv) {
          const size_t o = ((size_t)b * Tp + t) * Cc + c8;
          load8<T>(dA + o, dv[j]);
          load8<T>(y + o, yv8[j]);
        }
      }
#pragma unroll
      for (int j = 0; j < UNR; ++j) {
        const int t = tb + j * rpi;
        if (t < tend) {
          float out[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) out[i] = 0.f;
          if (t < Tv) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              const float yv = yv8[j][i];
              const float xh = (yv - c.mean) * c.rstd;
              const float dxh = dv[j][i] * g8[i];
              const float dy = c.rstd * (dxh - m1 - xh * m2);
              if (yv > 0.f) out[i] = dy;
              else { out[i] = a * dy; da += dy * yv * inva; }
            }
          }
          store8<T>(dpre + ((size_t)b * Tp + t) * Cc + c8, out);
        }
      }
    }
  }
  const float ds = wave_sum(da);
  if ((threadIdx.x & 63) == 0) atomicAdd(dslope + (size_t)z * sP, ds);
}

// depthwise conv backward.  forward: u[t] = bd + sum_k wd[k] * zz[t + (k-1)d],  zz = GN1(y1) on valid frames else 0
//   dz[t]   = sum_k wd[k] * du[t - (k-1)d]          (valid t, output frame inside [0, Tv))
//   dwd[k] += sum_t du[t - (k-1)d] * zz[t]          (the same sum as sum_t' du[t'] * zz[t' + (k-1)d], indexed by the SOURCE
//             frame t, so that one row needs zz[t] and the three du rows dz[t] reads anyway: 4 loads per row, not 7)
//   dbd    += sum_t du[t]
template <typename T>
__global__ __launch_bounds__(256) void dwconv_bwd_kernel(const T* __restrict__ du, const T* __restrict__ y1,
                                                         const double* __restrict__ st1, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, const float* __restrict__ wd,
                                                         T* __restrict__ dz, float* __restrict__ dwd, float* __restrict__ dbd,
                                                         int Cc, int Tp, int Tv, int dil, float eps, long sAct, long sSt,
                                                         long sP, int RPB, int noatom) {
  extern __shared__ float sm[];   // [4][Cc]: dwd k=0..2, dbd
  const int z = blockIdx.z, b = blockIdx.y;
  du += (size_t)z * sAct; y1 += (size_t)z * sAct; dz += (size_t)z * sAct;
  st1 += (size_t)z * sSt;
  gamma += (size_t)z * sP; beta += (size_t)z * sP; wd += (size_t)z * sP; dwd += (size_t)z * sP; dbd += (size_t)z * sP;
  const GnCtx c = gn_ctx(st1, b, (double)Cc * Tv, eps);
  for (int i = threadIdx.x; i < 4 * Cc; i += 256) sm[i] = 0.f;
  __syncthreads();
  const int cpr = Cc / 8, rpi = 256 / cpr;
  const int tl = threadIdx.x / cpr, c8 = (threadIdx.x % cpr) * 8;
  float aw[3][8], ab[8], g8[8], be8[8], w8[3][8];
  {
    float ga[8], bt[8], w24[24];
    loadf8(gamma + c8, ga);
    loadf8(beta + c8, bt);
#pragma unroll
    for (int v = 0; v < 3; ++v) loadf8(wd + c8 * 3 + 8 * v, *reinterpret_cast<float(*)[8]>(&w24[8 * v]));
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      ab[i] = 0.f;
      g8[i] = ga[i] * c.rstd;
      be8[i] = bt[i] - c.mean * g8[i];
#pragma unroll
      for (int k = 0; k < 3; ++k) { aw[k][i] = 0.f; w8[k][i] = w24[i * 3 + k]; }
    }
  }
  if (tl < rpi) {
    constexpr int UD = 2;
    const int tend = (blockIdx.x + 1) * RPB < Tp ? (blockIdx.x + 1) * RPB : Tp;
    for (int tb = blockIdx.x * RPB + tl; tb < tend; tb += UD * rpi) {
      float yv8[UD][8], du8[UD][3][8];
#pragma unroll
      for (int j = 0; j < UD; ++j) {
        const int t = tb + j * rpi;
        if (t < tend && t < Tv) {
          load8<T>(y1 + ((size_t)b * Tp + t) * Cc + c8, yv8[j]);
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const int tu = t - (k - 1) * dil;       // output frame whose tap k reads (source) frame t
            if (tu >= 0 && tu < Tv) load8<T>(du + ((size_t)b * Tp + tu) * Cc + c8, du8[j][k]);
          }
        }
      }
#pragma unroll
      for (int j = 0; j < UD; ++j) {
        const int t = tb + j * rpi;
        if (t < tend) {
          float out[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) out[i] = 0.f;
          if (t < Tv) {
#pragma unroll
            for (int i = 0; i < 8; ++i) ab[i] += du8[j][1][i];           // tap 1: tu == t, always valid here
#pragma unroll
            for (int k = 0; k < 3; ++k) {
              const int tu = t - (k - 1) * dil;
              if (tu >= 0 && tu < Tv) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                  out[i] += w8[k][i] * du8[j][k][i];
                  aw[k][i] += du8[j][k][i] * (yv8[j][i] * g8[i] + be8[i]);
                }
              }
            }
          }
          store8<T>(dz + ((size_t)b * Tp + t) * Cc + c8, out);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int k = 0; k < 3; ++k) atomicAdd(&sm[k * Cc + c8 + i], aw[k][i]);
      atomicAdd(&sm[3 * Cc + c8 + i], ab[i]);
    }
  }
  __syncthreads();
  if (noatom) return;
  for (int i = threadIdx.x; i < Cc; i += 256) {
#pragma unroll
    for (int k = 0; k < 3; ++k) atomicAdd(dwd + i * 3 + k, sm[k * Cc + i]);
    atomicAdd(dbd + i, sm[3 * Cc + i]);
  }
}


// =====================================================================================================
// Fused backward of everything between the two 1x1 convolutions of a TCNBlock:
//   dA (gradient of the normalised depthwise output a2 = GN2(y2))  ->  dpre1 (gradient of the conv1x1 pre-activation)
// through GroupNorm-2, PReLU-2, the depthwise dilated convolution, GroupNorm-1 and PReLU-1, in ONE reduce pass and ONE
// apply pass over the saved activations (y1 = PReLU1 output, y2 = PReLU2 output), instead of five passes with three
// intermediate gradient tensors.
//
// Notation per sample (n = C*Tv valid elements): xh2 = (y2 - mean2) rstd2, p = PReLU2'(y2), xh1 = (y1 - mean1) rstd1,
//   d xh2 = dA gamma2,  S1 = sum d xh2,  S2 = sum d xh2 xh2,
//   du    = p rstd2 (d xh2 - S1/n - xh2 S2/n)                       (gradient of the depthwise output before PReLU-2)
//   dz1[t]= sum_k wd[k] du[t - (k-1) dil]                            (gradient of z1 = GN1(y1))
//   R1    = sum gamma1 dz1,  R2 = sum gamma1 dz1 xh1,   dy1 = rstd1 (gamma1 dz1 - R1/n - xh1 R2/n),  dpre1 = PReLU1'(y1) dy1
// GroupNorm-1's sums R1, R2 depend on du, i.e. on S1, S2 -- but LINEARLY: du = f0 - (S1/n) f1 - (S2/n) f2 with
// f0 = p rstd2 d xh2, f1 = p rstd2, f2 = p rstd2 xh2, and R1 = sum du m1, R2 = sum du m2 with the fields
//   m1[t] = gamma1 sum_k wd[k] [0 <= t + (k-1) dil < Tv],   m2[t] = gamma1 sum_k wd[k] xh1[t + (k-1) dil]
// (the depthwise convolution moved onto the other factor).  So ONE pass yields S1, S2 and the six sums
// A_j = sum f_j m1, B_j = sum f_j m2, and R1 = A0 - (S1/n) A1 - (S2/n) A2, R2 = B0 - (S1/n) B1 - (S2/n) B2 follow.
// Per-sample sums (S1, S2, A0, A1, A2, B0, B1, B2): every channel-group workgroup stores its share in S[z][b][cg][8] (fp64) and
// the apply pass adds the shares in index order; per-channel parameter gradients: a fixed shuffle / LDS tree per workgroup,
// one partial row per (branch, sample), summed over the samples by the finishing launch.  No atomics (round 4).
constexpr int MB_SUMS = 8;
constexpr int MB_CG = 64;          // channels per workgroup
constexpr int MB_FL = 32;          // frame lanes per workgroup (256 threads = 32 frame lanes x 8 threads of 8 channels)
constexpr int MB_PART_ROWS = 9;    // dgamma2, dbeta2 | dgamma1, dbeta1, dbias1, dbd, dwd k = 0..2

// Work split: one workgroup = one sample x 64 channels x ALL frames.  The depthwise convolution couples frames, so with
// the whole time axis in one workgroup the shifted rows (t +- dil) are the workgroup's own rows a few iterations away
// (cache hits, no halo re-reads: frame-chunked workgroups re-read 2 dil of every 64 frames and each ended with 3.6 K
// per-channel partials), every per-channel sum is complete inside the workgroup, and a frame row of a channel group is one
// full 128-byte line (bf16).  Per-channel parameter-gradient partials go to part[z][b][9 C + 2 C/64] (rows as
// MB_PART_ROWS, then the two PReLU-slope partials of each channel group); mid_bwd_finish_kernel sums them over the samples
// into the gradients -- global atomics from every workgroup onto the same addresses ran at the contended rate.
struct MidBwdArgs {
  const void* dA; const void* y2; const void* y1;     // [z][B][Tp][C]
  const double* st1; const double* st2;               // GroupNorm (sum, sumsq) per sample: [z][B][2]
  double* S;                                          // [z][B][Cc/64][8]: per channel group, summed in order by the apply pass
  const float* gamma1; const float* beta1; const float* gamma2; const float* beta2; const float* wd;
  const float* slope1; const float* slope2;
  void* a2;                                           // reduce pass: GN2(y2) for the sconv weight gradient (may be null)
  void* dpre1;                                        // apply pass output
  float* dgamma2; float* dbeta2;                      // reduce pass
  float* dgamma1; float* dbeta1; float* dwd; float* dbd; float* dslope1; float* dslope2; float* dbias1;   // apply pass
  int Cc, Tp, Tv, dil; float eps;
  long sAct, sSt, sP;
  float* part;
  // optional: tile column sums of the block's upstream gradient (nppc_gemm_nt_colsum: [z][cp_tiles][cp_ld]) -> the sconv bias
  // gradient dbias2[z*sP + c], c < cp_cols, added up by the finishing launch
  const float* colpart; int cp_tiles, cp_ld, cp_cols; float* dbias2;
  // finishing launch over several TCN blocks at once (blockIdx.z = block): element strides between consecutive blocks of the
  // partial rows, the tile column sums and the parameter gradients (0 for the one-block launch)
  long partL, colpartL, sL;
};
__host__ __device__ __forceinline__ size_t mb_part_stride(int Cc) { return (size_t)MB_PART_ROWS * Cc + 2 * (Cc / MB_CG) + 6; }

// per-channel partials of a workgroup: 256 threads -> 64 channels.  acc[r][i] of thread (fl, ct) belongs to channel
// ct*8 + i of row r.  No atomics (round 4: LDS float atomics from four waves arrive in any order): the 8 frame lanes of a
// wave that share a channel are folded with three xor-shuffles (a fixed tree), lanes 0-7 leave the wave's sums in LDS and
// the four waves are added in index order, one plain store per channel.  sm: [4 waves][ROWS][64] floats.
template <int ROWS>
__device__ __forceinline__ void mb_channel_sums(float (&acc)[ROWS][8], float* sm /* [4][ROWS][64] */, float* dst /* row 0 */, int Cc,
                                                int c0, int ct) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();                                     // (sm may still be read by an earlier call)
#pragma unroll
  for (int r = 0; r < ROWS; ++r)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float v = acc[r][i];
      v += __shfl_xor(v, 8, 64);
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (lane < 8) sm[(wave * ROWS + r) * MB_CG + ct * 8 + i] = v;
    }
  __syncthreads();
  for (int i = threadIdx.x; i < ROWS * MB_CG; i += 256)
    dst[(size_t)(i / MB_CG) * Cc + c0 + (i % MB_CG)] =
        ((sm[i] + sm[ROWS * MB_CG + i]) + sm[2 * ROWS * MB_CG + i]) + sm[3 * ROWS * MB_CG + i];
}

template <typename T>
__global__ __launch_bounds__(256) void mid_bwd_reduce_kernel(MidBwdArgs g) {
  __shared__ float sm[4 * 2 * MB_CG];
  __shared__ double red[MB_SUMS][4];
  const int z = blockIdx.z, b = blockIdx.y, Cc = g.Cc, Tp = g.Tp, Tv = g.Tv, dil = g.dil;
  const int ct = threadIdx.x & 7, fl = threadIdx.x >> 3, c0 = blockIdx.x * MB_CG, c8 = c0 + ct * 8;
  const T* dA = reinterpret_cast<const T*>(g.dA) + (size_t)z * g.sAct + (size_t)b * Tp * Cc + c8;
  const T* y2 = reinterpret_cast<const T*>(g.y2) + (size_t)z * g.sAct + (size_t)b * Tp * Cc + c8;
  const T* y1 = reinterpret_cast<const T*>(g.y1) + (size_t)z * g.sAct + (size_t)b * Tp * Cc + c8;
  T* a2 = g.a2 ? reinterpret_cast<T*>(g.a2) + (size_t)z * g.sAct + (size_t)b * Tp * Cc + c8 : nullptr;
  const GnCtx c1 = gn_ctx(g.st1 + (size_t)z * g.sSt, b, (double)Cc * Tv, g.eps);
  const GnCtx c2 = gn_ctx(g.st2 + (size_t)z * g.sSt, b, (double)Cc * Tv, g.eps);
  const float sl2 = g.slope2[(size_t)z * g.sP];
  float acc[MB_SUMS];
#pragma unroll
  for (int i = 0; i < MB_SUMS; ++i) acc[i] = 0.f;
  float g1[8], g2[8], b2[8], w8[3][8], ch[2][8];
  {
    float w24[24];
    loadf8(g.gamma1 + (size_t)z * g.sP + c8, g1);
    loadf8(g.gamma2 + (size_t)z * g.sP + c8, g2);
    loadf8(g.beta2 + (size_t)z * g.sP + c8, b2);
#pragma unroll
    for (int v = 0; v < 3; ++v) loadf8(g.wd + (size_t)z * g.sP + (size_t)c8 * 3 + 8 * v, *reinterpret_cast<float(*)[8]>(&w24[8 * v]));
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      ch[0][i] = ch[1][i] = 0.f;
#pragma unroll
      for (int k = 0; k < 3; ++k) w8[k][i] = w24[i * 3 + k];
    }
  }
  for (int t = fl; t < Tv; t += MB_FL) {
    const size_t o = (size_t)t * Cc;
    float dv[8], yv[8], x1[3][8];
    load8<T>(dA + o, dv);
    load8<T>(y2 + o, yv);
    bool ok[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int ts = t + (k - 1) * dil;
      ok[k] = ts >= 0 && ts < Tv;
      if (ok[k]) load8<T>(y1 + (size_t)ts * Cc, x1[k]);
    }
    float av[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float xh2 = (yv[i] - c2.mean) * c2.rstd;
      const float dxh = dv[i] * g2[i];
      ch[0][i] += dv[i] * xh2;
      ch[1][i] += dv[i];
      av[i] = xh2 * g2[i] + b2[i];
      const float pr = (yv[i] > 0.f ? 1.f : sl2) * c2.rstd;
      const float f0 = pr * dxh, f1 = pr, f2 = pr * xh2;
      float m1 = 0.f, m2 = 0.f;
#pragma unroll
      for (int k = 0; k < 3; ++k)
        if (ok[k]) {
          m1 += w8[k][i];
          m2 += w8[k][i] * ((x1[k][i] - c1.mean) * c1.rstd);
        }
      m1 *= g1[i];
      m2 *= g1[i];
      acc[0] += dxh;
      acc[1] += dxh * xh2;
      acc[2] += f0 * m1; acc[3] += f1 * m1; acc[4] += f2 * m1;
      acc[5] += f0 * m2; acc[6] += f1 * m2; acc[7] += f2 * m2;
    }
    if (a2) store8<T>(a2 + o, av);
  }
#pragma unroll
  for (int i = 0; i < MB_SUMS; ++i) {
    const double d = wave_sum((double)acc[i]);
    if ((threadIdx.x & 63) == 0) red[i][threadIdx.x >> 6] = d;
  }
  __syncthreads();
  if (threadIdx.x < MB_SUMS)     // this channel group's share of the per-sample sums: one plain store, added up in order by the apply pass
    g.S[(((size_t)z * gridDim.y + b) * gridDim.x + blockIdx.x) * MB_SUMS + threadIdx.x] =
        ((red[threadIdx.x][0] + red[threadIdx.x][1]) + red[threadIdx.x][2]) + red[threadIdx.x][3];
  float* part = g.part + ((size_t)z * gridDim.y + b) * mb_part_stride(Cc);
  mb_channel_sums<2>(ch, sm, part, Cc, c0, ct);
}

// Apply pass.  Round 4: the gradient of the depthwise OUTPUT, du[t] = PReLU2'(y2[t]) rstd2 (dA[t] gamma2 - S1/n - xh2[t] S2/n), is
// computed ONCE per element into LDS (phase A) and read back at the three taps (phase B); round 3 recomputed it per tap --
// two extra loads of (dA, y2) and ~20 of the ~50 vector instructions per element of a pass that is VALU-bound.  The time axis is
// walked in chunks of `chunk` frames (+ dil frames of halo on either side) so that the staging rows fit two workgroups per CU
// at any clip length: LDS = (chunk + 2 dil) x (64 + 4) floats + the reduction scratch.  Same arithmetic per element as before.
constexpr int MB_DU_LD = MB_CG + 4;      // row stride (floats) of the du staging: the 8 frame lanes of a wave hit disjoint banks
__host__ __device__ __forceinline__ int mb_apply_chunk(int Tv) { const int n = (Tv + 223) / 224; return (Tv + n - 1) / n; }
__host__ __device__ __forceinline__ size_t mb_apply_smem(int Tv, int dil) {
  return ((size_t)(mb_apply_chunk(Tv) + 2 * dil) * MB_DU_LD + 4 * 7 * MB_CG + 8) * sizeof(float);
}

template <typename T>
__global__ __launch_bounds__(256) void mid_bwd_apply_kernel(MidBwdArgs g) {
  extern __shared__ __attribute__((aligned(16))) float mb_dyn[];
  const int z = blockIdx.z, b = blockIdx.y, Cc = g.Cc, Tp = g.Tp, Tv = g.Tv, dil = g.dil;
  const int chunk = mb_apply_chunk(Tv);
  float* du_s = mb_dyn;                                        // [(chunk + 2 dil)][MB_DU_LD]
  float* sm = du_s + (size_t)(chunk + 2 * dil) * MB_DU_LD;     // [4][7][64]
  float (*sl)[4] = reinterpret_cast<float(*)[4]>(sm + 4 * 7 * MB_CG);
  const int ct = threadIdx.x & 7, fl = threadIdx.x >> 3, c0 = blockIdx.x * MB_CG, c8 = c0 + ct * 8;
  const T* dA = reinterpret_cast<const T*>(g.dA) + (size_t)z * g.sAct + (size_t)b * Tp * Cc + c8;
  const T* y2 = reinterpret_cast<const T*>(g.y2) + (size_t)z * g.sAct + (size_t)b * Tp * Cc + c8;
  const T* y1 = reinterpret_cast<const T*>(g.y1) + (size_t)z * g.sAct + (size_t)b * Tp * Cc + c8;
  T* dpre1 = reinterpret_cast<T*>(g.dpre1) + (size_t)z * g.sAct + (size_t)b * Tp * Cc + c8;
  const double cnt = (double)Cc * Tv;
  const GnCtx c1 = gn_ctx(g.st1 + (size_t)z * g.sSt, b, cnt, g.eps);
  const GnCtx c2 = gn_ctx(g.st2 + (size_t)z * g.sSt, b, cnt, g.eps);
  double S[MB_SUMS];             // per-sample sums: the channel groups' shares in index order (no atomics: bit-reproducible)
  {
    const double* Sp = g.S + ((size_t)z * gridDim.y + b) * gridDim.x * MB_SUMS;
#pragma unroll
    for (int i = 0; i < MB_SUMS; ++i) S[i] = Sp[i];
    for (int cg = 1; cg < (int)gridDim.x; ++cg)
#pragma unroll
      for (int i = 0; i < MB_SUMS; ++i) S[i] += Sp[cg * MB_SUMS + i];
  }
  const double s1d = S[0] / cnt, s2d = S[1] / cnt;
  const float s1 = (float)s1d, s2 = (float)s2d;
  const float r1 = (float)((S[2] - s1d * S[3] - s2d * S[4]) / cnt), r2 = (float)((S[5] - s1d * S[6] - s2d * S[7]) / cnt);
  const float sl1 = g.slope1[(size_t)z * g.sP], sl2 = g.slope2[(size_t)z * g.sP];
  const float isl1 = 1.f / sl1, isl2 = 1.f / sl2;
  float ds1 = 0.f, ds2 = 0.f;
  float g1[8], be1[8], g2[8], w8[3][8];
  float ch[7][8];          // dgamma1, dbeta1, dbias1, dbd, dwd k = 0..2
  {
    float w24[24];
    loadf8(g.gamma1 + (size_t)z * g.sP + c8, g1);
    loadf8(g.beta1 + (size_t)z * g.sP + c8, be1);
    loadf8(g.gamma2 + (size_t)z * g.sP + c8, g2);
#pragma unroll
    for (int v = 0; v < 3; ++v) loadf8(g.wd + (size_t)z * g.sP + (size_t)c8 * 3 + 8 * v, *reinterpret_cast<float(*)[8]>(&w24[8 * v]));
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int r = 0; r < 7; ++r) ch[r][i] = 0.f;
#pragma unroll
      for (int k = 0; k < 3; ++k) w8[k][i] = w24[i * 3 + k];
    }
  }
  for (int f0 = 0; f0 < Tv; f0 += chunk) {
    const int f1 = f0 + chunk < Tv ? f0 + chunk : Tv;          // the chunk's own frames [f0, f1)
    const int lo = f0 - dil > 0 ? f0 - dil : 0, hi = f1 + dil < Tv ? f1 + dil : Tv;   // ... and the du rows its taps read
    __syncthreads();                                           // everyone is done reading the previous chunk's rows
    // ---- phase A: du of frames [lo, hi) -> LDS; the sums indexed by the depthwise OUTPUT frame for the chunk's own frames
    for (int u = lo + fl; u < hi; u += MB_FL) {
      float dv[8], yv[8], du8[8];
      load8<T>(dA + (size_t)u * Cc, dv);
      load8<T>(y2 + (size_t)u * Cc, yv);
      const bool own = u >= f0 && u < f1;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float y = yv[i];
        const float xh2 = (y - c2.mean) * c2.rstd;
        const float dy2 = c2.rstd * (dv[i] * g2[i] - s1 - xh2 * s2);
        const float du = y > 0.f ? dy2 : sl2 * dy2;
        du8[i] = du;
        if (own) {
          ch[3][i] += du;
          if (!(y > 0.f)) ds2 += dy2 * y * isl2;
        }
      }
      float* dp = du_s + (size_t)(u - lo) * MB_DU_LD + ct * 8;
      *reinterpret_cast<float4*>(dp) = make_float4(du8[0], du8[1], du8[2], du8[3]);
      *reinterpret_cast<float4*>(dp + 4) = make_float4(du8[4], du8[5], du8[6], du8[7]);
    }
    __syncthreads();
    // ---- phase B: the chunk's own frames
    for (int t = f0 + fl; t < f1; t += MB_FL) {
      float out[8], yv1[8], du3[3][8];
      const size_t o = (size_t)t * Cc;
      load8<T>(y1 + o, yv1);
      bool ok[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int tu = t - (k - 1) * dil;           // depthwise output frame whose tap k reads frame t
        ok[k] = tu >= 0 && tu < Tv;
        if (ok[k]) {
          const float* dp = du_s + (size_t)(tu - lo) * MB_DU_LD + ct * 8;
          const float4 a = *reinterpret_cast<const float4*>(dp), c = *reinterpret_cast<const float4*>(dp + 4);
          du3[k][0] = a.x; du3[k][1] = a.y; du3[k][2] = a.z; du3[k][3] = a.w;
          du3[k][4] = c.x; du3[k][5] = c.y; du3[k][6] = c.z; du3[k][7] = c.w;
        }
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float xh1 = (yv1[i] - c1.mean) * c1.rstd;
        const float z1 = xh1 * g1[i] + be1[i];
        float dz = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k)
          if (ok[k]) {
            const float du = du3[k][i];
            dz += w8[k][i] * du;
            ch[4 + k][i] += du * z1;
          }
        ch[0][i] += dz * xh1;
        ch[1][i] += dz;
        const float dy1 = c1.rstd * (dz * g1[i] - r1 - xh1 * r2);
        float dp;
        if (yv1[i] > 0.f) dp = dy1;
        else { dp = sl1 * dy1; ds1 += dy1 * yv1[i] * isl1; }
        out[i] = dp;
        ch[2][i] += to_f32<T>(from_f32<T>(dp));       // the bias gradient of the STORED (rounded) tensor, as a column sum of it was
      }
      store8<T>(dpre1 + o, out);
    }
  }
  {                                                            // padded frames [Tv, Tp): zero
    float zero8[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) zero8[i] = 0.f;
    for (int t = Tv + fl; t < Tp; t += MB_FL) store8<T>(dpre1 + (size_t)t * Cc, zero8);
  }
  const float w1 = wave_sum(ds1), w2 = wave_sum(ds2);
  __syncthreads();                                             // (the staging rows are dead: sl / sm follow)
  if ((threadIdx.x & 63) == 0) { sl[0][threadIdx.x >> 6] = w1; sl[1][threadIdx.x >> 6] = w2; }
  float* part = g.part + ((size_t)z * gridDim.y + b) * mb_part_stride(Cc);
  mb_channel_sums<7>(ch, sm, part + 2 * (size_t)Cc, Cc, c0, ct);      // (its barriers also publish sl)
  if (threadIdx.x < 2)
    part[(size_t)MB_PART_ROWS * Cc + 2 * blockIdx.x + threadIdx.x] = sl[threadIdx.x][0] + sl[threadIdx.x][1] + sl[threadIdx.x][2] + sl[threadIdx.x][3];
}

// sums the per-workgroup partials over the samples and ADDS them to the parameter gradients of branch blockIdx.y.
// One workgroup = 64 consecutive partial columns x 16 slices of the sample list (256 contiguous bytes per load instruction);
// the two PReLU-slope gradients are columns 9 Cc and 9 Cc + 1, each the sum over (sample, channel group).  Every gradient
// element has ONE writer and a fixed summation order: no atomics.
__global__ __launch_bounds__(1024) void mid_bwd_finish_kernel(MidBwdArgs g, int B, int nb_main) {
  __shared__ float red[16][64];
  const int z = blockIdx.y, Cc = g.Cc;
  {
    const long blk = blockIdx.z;
    g.part += blk * g.partL;
    if (g.colpart) { g.colpart += blk * g.colpartL; g.dbias2 += blk * g.sL; }
    g.dgamma2 += blk * g.sL; g.dbeta2 += blk * g.sL; g.dgamma1 += blk * g.sL; g.dbeta1 += blk * g.sL; g.dbias1 += blk * g.sL;
    g.dbd += blk * g.sL; g.dwd += blk * g.sL; g.dslope1 += blk * g.sL; g.dslope2 += blk * g.sL;
  }
  const int col = threadIdx.x & 63, slice = threadIdx.x >> 6;
  if ((int)blockIdx.x >= nb_main) {                   // workgroups behind the main ones: the sconv bias gradient from tile sums
    const int c = ((int)blockIdx.x - nb_main) * 64 + col;
    float s = 0.f;
    if (c < g.cp_cols) {
      const float* p = g.colpart + (size_t)z * g.cp_tiles * g.cp_ld + c;
      for (int t = slice; t < g.cp_tiles; t += 16) s += p[(size_t)t * g.cp_ld];
    }
    red[slice][col] = s;
    __syncthreads();
    if (slice != 0 || c >= g.cp_cols) return;
#pragma unroll
    for (int k = 1; k < 16; ++k) s += red[k][col];
    g.dbias2[(size_t)z * g.sP + c] = s;
    return;
  }
  const int i = blockIdx.x * 64 + col;
  const int ncg = Cc / MB_CG, ncols = MB_PART_ROWS * Cc + 2;
  const size_t ps = mb_part_stride(Cc);
  float s = 0.f;
  if (i < MB_PART_ROWS * Cc) {
    const float* p = g.part + (size_t)z * B * ps + i;
    for (int b = slice; b < B; b += 16) s += p[(size_t)b * ps];
  } else if (i < ncols) {                               // slope k = i - 9 Cc: entries (b, cg) at part[b][9 Cc + 2 cg + k]
    const float* p = g.part + (size_t)z * B * ps + (size_t)MB_PART_ROWS * Cc + (i - MB_PART_ROWS * Cc);
    for (int e = slice; e < B * ncg; e += 16) s += p[(size_t)(e / ncg) * ps + 2 * (e % ncg)];
  }
  red[slice][col] = s;
  __syncthreads();
  if (slice != 0 || i >= ncols) return;
#pragma unroll
  for (int k = 1; k < 16; ++k) s += red[k][col];
  const size_t zo = (size_t)z * g.sP;
  if (i >= MB_PART_ROWS * Cc) {
    float* d = ((i - MB_PART_ROWS * Cc) & 1 ? g.dslope2 : g.dslope1) + zo;
    *d += s;
    return;
  }
  const int row = i / Cc, c = i % Cc;
  switch (row) {
    case 0: g.dgamma2[zo + c] += s; break;
    case 1: g.dbeta2[zo + c] += s; break;
    case 2: g.dgamma1[zo + c] += s; break;
    case 3: g.dbeta1[zo + c] += s; break;
    case 4: g.dbias1[zo + c] += s; break;
    case 5: g.dbd[zo + c] += s; break;
    default: g.dwd[zo + c * 3 + (row - 6)] += s; break;
  }
}

}  // namespace

extern "C" {

int nppc_tcn_gn_bwd(int prec, const void* dA, const void* y, const double* st, const float* gamma, const float* slope,
                    double* S, void* dpre, float* dgamma, float* dbeta, float* dslope, int B, int Cc, int Tp, int Tv, float eps,
                    long sAct, long sSt, long sP, int batch, void* stream) {
  if (!dA || !y || !st || !gamma || !slope || !S || !dpre || !dgamma || !dbeta || !dslope || Cc % 8 || Cc / 8 > 256)
    return NPPC_EBADARG;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(S, 0, sizeof(double) * 2 * B * batch, s) != hipSuccess) return NPPC_ELAUNCH;   // S laid out [batch][B][2], sSt = 2B
  const int RPB = tcn_rpb(), noatom = diag_noatom();
  dim3 g1(ceil_div(Tv, RPB), B, batch), g2(ceil_div(Tp, RPB), B, batch);
  const size_t smem = (size_t)2 * Cc * sizeof(float);
  if (prec == NPPC_PREC_BF16) {
    hipLaunchKernelGGL(gn_bwd_reduce_kernel<bf16_t>, g1, dim3(256), smem, s, (const bf16_t*)dA, (const bf16_t*)y, st, gamma, S,
                       dgamma, dbeta, Cc, Tp, Tv, eps, sAct, sSt, sP, RPB, noatom);
    hipLaunchKernelGGL(gn_prelu_bwd_kernel<bf16_t>, g2, dim3(256), 0, s, (const bf16_t*)dA, (const bf16_t*)y, st, gamma, S,
                       slope, (bf16_t*)dpre, dslope, Cc, Tp, Tv, eps, sAct, sSt, sP, RPB, noatom);
  } else {
    hipLaunchKernelGGL(gn_bwd_reduce_kernel<float>, g1, dim3(256), smem, s, (const float*)dA, (const float*)y, st, gamma, S,
                       dgamma, dbeta, Cc, Tp, Tv, eps, sAct, sSt, sP, RPB, noatom);
    hipLaunchKernelGGL(gn_prelu_bwd_kernel<float>, g2, dim3(256), 0, s, (const float*)dA, (const float*)y, st, gamma, S, slope,
                       (float*)dpre, dslope, Cc, Tp, Tv, eps, sAct, sSt, sP, RPB, noatom);
  }
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

int nppc_tcn_dwconv_bwd(int prec, const void* du, const void* y1, const double* st1, const float* gamma, const float* beta,
                        const float* wd, void* dz, float* dwd, float* dbd, int B, int Cc, int Tp, int Tv, int dil, float eps,
                        long sAct, long sSt, long sP, int batch, void* stream) {
  if (!du || !y1 || !st1 || !gamma || !beta || !wd || !dz || !dwd || !dbd || Cc % 8 || Cc / 8 > 256) return NPPC_EBADARG;
  hipStream_t s = (hipStream_t)stream;
  const int RPB = tcn_rpb(), noatom = diag_noatom();
  dim3 g(ceil_div(Tp, RPB), B, batch);
  const size_t smem = (size_t)4 * Cc * sizeof(float);
  if (prec == NPPC_PREC_BF16)
    hipLaunchKernelGGL(dwconv_bwd_kernel<bf16_t>, g, dim3(256), smem, s, (const bf16_t*)du, (const bf16_t*)y1, st1, gamma, beta,
                       wd, (bf16_t*)dz, dwd, dbd, Cc, Tp, Tv, dil, eps, sAct, sSt, sP, RPB, noatom);
  else
    hipLaunchKernelGGL(dwconv_bwd_kernel<float>, g, dim3(256), smem, s, (const float*)du, (const float*)y1, st1, gamma, beta, wd,
                       (float*)dz, dwd, dbd, Cc, Tp, Tv, dil, eps, sAct, sSt, sP, RPB, noatom);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}


/* fused backward of a TCNBlock's middle (GroupNorm-2, PReLU-2, depthwise conv, GroupNorm-1, PReLU-1): dA -> dpre1 and
 * every parameter gradient of those stages plus the conv1x1 bias gradient, one reduce + one apply launch
 * (S: [batch][B][Cc/64][8] fp64 workspace, no initial state; part: nppc_tcn_mid_bwd_part_elems(...) floats of workspace for the
 * per-workgroup partial sums; a2 (nullable): GN2(y2) written for the sconv weight gradient) */
int nppc_tcn_mid_bwd_part_elems(int B, int Cc, int Tp, int batch, long* n) {
  if (!n || B <= 0 || Cc <= 0 || Tp <= 0 || batch <= 0) return NPPC_EBADARG;
  *n = (long)batch * B * (long)mb_part_stride(Cc);
  return NPPC_OK;
}

int nppc_tcn_mid_bwd(int prec, const void* dA, const void* y2, const void* y1, const double* st1, const double* st2, double* S,
                     float* part,
                     const float* gamma1, const float* beta1, const float* gamma2, const float* beta2, const float* wd,
                     const float* slope1, const float* slope2, void* a2, void* dpre1, float* dgamma2, float* dbeta2,
                     float* dgamma1, float* dbeta1, float* dwd, float* dbd, float* dslope1, float* dslope2, float* dbias1,
                     const float* colpart, int cp_tiles, int cp_ld, int cp_cols, float* dbias2, int B,
                     int Cc, int Tp, int Tv, int dil, float eps, long sAct, long sSt, long sP, int batch, int finish_now,
                     void* stream) {
  if (!dA || !y2 || !y1 || !st1 || !st2 || !S || !part || !gamma1 || !beta1 || !gamma2 || !beta2 || !wd || !slope1 || !slope2 || !dpre1 ||
      !dgamma2 || !dbeta2 || !dgamma1 || !dbeta1 || !dwd || !dbd || !dslope1 || !dslope2 || !dbias1 || dil < 1)
    return NPPC_EBADARG;
  if (colpart && (!dbias2 || cp_tiles < 1 || cp_cols < 1 || cp_ld < cp_cols)) return NPPC_EBADARG;
  if (Cc % MB_CG) return NPPC_EUNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  MidBwdArgs g{dA, y2, y1, st1, st2, S, gamma1, beta1, gamma2, beta2, wd, slope1, slope2, a2, dpre1, dgamma2, dbeta2,
               dgamma1, dbeta1, dwd, dbd, dslope1, dslope2, dbias1, Cc, Tp, Tv, dil, eps, sAct, sSt, sP, part,
               colpart, cp_tiles, cp_ld, cp_cols, dbias2, 0, 0, 0};
  dim3 grid(Cc / MB_CG, B, batch);
  const size_t smem = mb_apply_smem(Tv, dil);
  if (smem > 80 * 1024) return NPPC_EUNSUPPORTED;          // (dil far beyond the TCN's 1..9: the staging rows would not fit)
  if (prec == NPPC_PREC_BF16) {
    static bool attr = false;
    if (!attr) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(mid_bwd_apply_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              80 * 1024) != hipSuccess)
        return NPPC_ELAUNCH;
      attr = true;
    }
    hipLaunchKernelGGL(mid_bwd_reduce_kernel<bf16_t>, grid, dim3(256), 0, s, g);
    hipLaunchKernelGGL(mid_bwd_apply_kernel<bf16_t>, grid, dim3(256), smem, s, g);
  } else {
    static bool attr = false;
    if (!attr) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(mid_bwd_apply_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              80 * 1024) != hipSuccess)
        return NPPC_ELAUNCH;
      attr = true;
    }
    hipLaunchKernelGGL(mid_bwd_reduce_kernel<float>, grid, dim3(256), 0, s, g);
    hipLaunchKernelGGL(mid_bwd_apply_kernel<float>, grid, dim3(256), smem, s, g);
  }
  if (finish_now) {
    const int nb_main = (int)ceil_div((long)MB_PART_ROWS * Cc + 2, 64);
    hipLaunchKernelGGL(mid_bwd_finish_kernel, dim3(nb_main + (colpart ? (int)ceil_div(cp_cols, 64) : 0), batch), dim3(1024), 0, s, g,
                       B, nb_main);
  }
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

/* the finishing launch of nppc_tcn_mid_bwd(..., finish_now = 0) for `nblk` TCN blocks at once: block k's partial rows at
 * part + k * partL (each nppc_tcn_mid_bwd_part_elems floats), its tile column sums at colpart + k * colpartL (nullable), its
 * parameter gradients k * sL elements behind the pointers given (the blocks of a TCN stack sit at a constant stride in the
 * flat parameter buffer).  Fixed summation order, no atomics; gradients are ACCUMULATED. */
int nppc_tcn_mid_bwd_finish(const float* part, long partL, const float* colpart, long colpartL, int cp_tiles, int cp_ld,
                            int cp_cols, float* dgamma2, float* dbeta2, float* dgamma1, float* dbeta1, float* dwd, float* dbd,
                            float* dslope1, float* dslope2, float* dbias1, float* dbias2, int B, int Cc, long sP, long sL,
                            int batch, int nblk, void* stream) {
  if (!part || !dgamma2 || !dbeta2 || !dgamma1 || !dbeta1 || !dwd || !dbd || !dslope1 || !dslope2 || !dbias1 || B <= 0 ||
      batch <= 0 || nblk <= 0)
    return NPPC_EBADARG;
  if (colpart && (!dbias2 || cp_tiles < 1 || cp_cols < 1 || cp_ld < cp_cols)) return NPPC_EBADARG;
  if (Cc % MB_CG) return NPPC_EUNSUPPORTED;
  MidBwdArgs g{};
  g.part = const_cast<float*>(part);
  g.colpart = colpart; g.cp_tiles = cp_tiles; g.cp_ld = cp_ld; g.cp_cols = cp_cols; g.dbias2 = dbias2;
  g.dgamma2 = dgamma2; g.dbeta2 = dbeta2; g.dgamma1 = dgamma1; g.dbeta1 = dbeta1; g.dwd = dwd; g.dbd = dbd;
  g.dslope1 = dslope1; g.dslope2 = dslope2; g.dbias1 = dbias1;
  g.Cc = Cc; g.sP = sP; g.partL = partL; g.colpartL = colpartL; g.sL = sL;
  const int nb_main = (int)ceil_div((long)MB_PART_ROWS * Cc + 2, 64);
  hipLaunchKernelGGL(mid_bwd_finish_kernel, dim3(nb_main + (colpart ? (int)ceil_div(cp_cols, 64) : 0), batch, nblk), dim3(1024), 0,
                     (hipStream_t)stream, g, B, nb_main);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

}  // extern "C"
