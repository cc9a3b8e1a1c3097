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

}  // extern "C"
