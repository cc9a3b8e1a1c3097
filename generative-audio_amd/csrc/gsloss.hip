// Gram-Schmidt on the K complex mask directions + the NPPC residual-projection loss.
// Reference: nppc_audio/pc_wrapper.py:8-44 (gram_schmidt_to_crm: coefficient sum(conj(w) * w_hat), w_hat detached,
// no epsilon), nppc_audio/trainer.py:259-317 (base_step loss), :337-342 (lambda ramp, done on the host).
//
// Everything is expressed through per-sample Gram matrices: one streaming pass builds all inner products
// <a_i, b_n> = sum_t conj(a_i[t]) b_n[t] (fp32 partials, fp64 reduction), a one-thread-per-sample fp64 solve turns
// them into combination coefficients, and one streaming pass writes the combined vectors.  HBM-bound.
#include <stdlib.h>
#include "common.h"
#include "nppc_hip.h"

namespace {

// vector set: K vectors stored [B][K][2][N] (+ optional extra vector e = gt - pred as index K)
struct VecSet {
  const float* v;     // [B][K][2][N]
  const float* gt;    // [B][2][N] or null
  const float* pred;  // [B][2][N] or null
};

template <int KV>
__device__ __forceinline__ void load_set(const VecSet& s, int K, int b, long N, long t, float (&re)[KV], float (&im)[KV]) {
#pragma unroll
  for (int i = 0; i < KV; ++i) {
    if (i < K) {
      const float* p = s.v + ((size_t)(b * K + i) * 2) * N + t;
      re[i] = p[0];
      im[i] = p[N];
    } else if (s.gt) {
      const float* g = s.gt + (size_t)b * 2 * N + t;
      const float* q = s.pred + (size_t)b * 2 * N + t;
      re[i] = g[0] - q[0];
      im[i] = g[N] - q[N];
    } else {
      re[i] = im[i] = 0.f;
    }
  }
}

// out[b][i][n] (double2) += sum_t conj(a_i[t]) * b_n[t],   i,n < KV   (KV = K or K+1)
// fp64 products and sums: fp32 x fp32 is exact in fp64, so nearly collinear directions (where Gram-Schmidt
// subtracts almost equal vectors) lose nothing in the Gram matrix.  SAME: a == b, only n >= i is accumulated.
template <int KV, bool SAME>
__global__ __launch_bounds__(256) void gram_kernel(VecSet A, VecSet Bs, double* __restrict__ out, int K, long N, long chunk) {
  constexpr int NP = SAME ? KV * (KV + 1) / 2 : KV * KV;
  __shared__ double red[4][NP * 2];
  const int b = blockIdx.y;
  const long t0 = (long)blockIdx.x * chunk;
  const long t1 = t0 + chunk < N ? t0 + chunk : N;
  double ar[NP], ai[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) ar[i] = ai[i] = 0.0;
  for (long t = t0 + threadIdx.x; t < t1; t += 256) {
    float xr[KV], xi[KV], yr[KV], yi[KV];
    load_set<KV>(A, K, b, N, t, xr, xi);
    if (SAME) {
#pragma unroll
      for (int i = 0; i < KV; ++i) { yr[i] = xr[i]; yi[i] = xi[i]; }
    } else {
      load_set<KV>(Bs, K, b, N, t, yr, yi);
    }
    int p = 0;
#pragma unroll
    for (int i = 0; i < KV; ++i)
#pragma unroll
      for (int n = SAME ? i : 0; n < KV; ++n) {
        ar[p] += (double)xr[i] * (double)yr[n] + (double)xi[i] * (double)yi[n];
        ai[p] += (double)xr[i] * (double)yi[n] - (double)xi[i] * (double)yr[n];
        ++p;
      }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const double r = wave_sum(ar[i]), m = wave_sum(ai[i]);
    if (lane == 0) { red[wave][2 * i] = r; red[wave][2 * i + 1] = m; }
  }
  __syncthreads();
  if (threadIdx.x < NP) {
    const int p = threadIdx.x;
    const double r = red[0][2 * p] + red[1][2 * p] + red[2][2 * p] + red[3][2 * p];
    const double m = red[0][2 * p + 1] + red[1][2 * p + 1] + red[2][2 * p + 1] + red[3][2 * p + 1];
    int i = 0, n = p;
    if (SAME) {
      int rem = p;
      for (i = 0; i < KV; ++i) {
        if (rem < KV - i) break;
        rem -= KV - i;
      }
      n = i + rem;
    } else {
      i = p / KV;
      n = p % KV;
    }
    double* o = out + (size_t)b * KV * KV * 2;
    atomicAdd(o + (i * KV + n) * 2, r);
    atomicAdd(o + (i * KV + n) * 2 + 1, m);
    if (SAME && n != i) {
      atomicAdd(o + (n * KV + i) * 2, r);
      atomicAdd(o + (n * KV + i) * 2 + 1, -m);
    }
  }
}

// four consecutive t per lane (16-byte loads; N % 4 == 0, 16-byte aligned vectors)
template <int KV>
__device__ __forceinline__ void load_set4(const VecSet& s, int K, int b, long N, long t, float4 (&re)[KV], float4 (&im)[KV]) {
#pragma unroll
  for (int i = 0; i < KV; ++i) {
    if (i < K) {
      const float* p = s.v + ((size_t)(b * K + i) * 2) * N + t;
      re[i] = *reinterpret_cast<const float4*>(p);
      im[i] = *reinterpret_cast<const float4*>(p + N);
    } else if (s.gt) {
      const float* g = s.gt + (size_t)b * 2 * N + t;
      const float* q = s.pred + (size_t)b * 2 * N + t;
      const float4 g0 = *reinterpret_cast<const float4*>(g), q0 = *reinterpret_cast<const float4*>(q);
      const float4 g1 = *reinterpret_cast<const float4*>(g + N), q1 = *reinterpret_cast<const float4*>(q + N);
      re[i] = make_float4(g0.x - q0.x, g0.y - q0.y, g0.z - q0.z, g0.w - q0.w);
      im[i] = make_float4(g1.x - q1.x, g1.y - q1.y, g1.z - q1.z, g1.w - q1.w);
    } else {
      re[i] = im[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
}
__device__ __forceinline__ float f4(const float4& v, int e) { return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w; }

// the same Gram matrix with 16-byte loads: every lane owns four consecutive t per iteration and keeps 2 x KV (SAME) or 4 x KV
// such loads in flight -- the scalar kernel had a dozen 4-byte loads per lane in flight on 256 workgroups and ran at 1 TB/s.
// Products and sums stay fp64 (fp32 x fp32 is exact in fp64): 84 fp64 FMAs per t at KV = 6 are a few microseconds chip-wide,
// the kernel is bound by its memory-level parallelism, not by the arithmetic.
template <int KV, bool SAME>
__global__ __launch_bounds__(256) void gram4_kernel(VecSet A, VecSet Bs, double* __restrict__ out, int K, long N, long chunk) {
  constexpr int NP = SAME ? KV * (KV + 1) / 2 : KV * KV;
  __shared__ double red[4][NP * 2];
  const int b = blockIdx.y;
  const long t0 = (long)blockIdx.x * chunk;
  const long t1 = t0 + chunk < N ? t0 + chunk : N;
  double ar[NP], ai[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) ar[i] = ai[i] = 0.0;
  for (long t = t0 + 4 * threadIdx.x; t < t1; t += 1024) {
    float4 xr[KV], xi[KV], yr[KV], yi[KV];
    load_set4<KV>(A, K, b, N, t, xr, xi);
    if (!SAME) load_set4<KV>(Bs, K, b, N, t, yr, yi);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int p = 0;
#pragma unroll
      for (int i = 0; i < KV; ++i)
#pragma unroll
        for (int n = SAME ? i : 0; n < KV; ++n) {
          const double a_r = f4(xr[i], e), a_i = f4(xi[i], e);
          const double b_r = SAME ? f4(xr[n], e) : f4(yr[n], e), b_i = SAME ? f4(xi[n], e) : f4(yi[n], e);
          ar[p] += a_r * b_r + a_i * b_i;
          ai[p] += a_r * b_i - a_i * b_r;
          ++p;
        }
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const double r = wave_sum(ar[i]), m = wave_sum(ai[i]);
    if (lane == 0) { red[wave][2 * i] = r; red[wave][2 * i + 1] = m; }
  }
  __syncthreads();
  if (threadIdx.x < NP) {
    const int p = threadIdx.x;
    const double r = red[0][2 * p] + red[1][2 * p] + red[2][2 * p] + red[3][2 * p];
    const double m = red[0][2 * p + 1] + red[1][2 * p + 1] + red[2][2 * p + 1] + red[3][2 * p + 1];
    int i = 0, n = p;
    if (SAME) {
      int rem = p;
      for (i = 0; i < KV; ++i) {
        if (rem < KV - i) break;
        rem -= KV - i;
      }
      n = i + rem;
    } else {
      i = p / KV;
      n = p % KV;
    }
    double* o = out + (size_t)b * KV * KV * 2;
    atomicAdd(o + (i * KV + n) * 2, r);
    atomicAdd(o + (i * KV + n) * 2 + 1, m);
    if (SAME && n != i) {
      atomicAdd(o + (n * KV + i) * 2, r);
      atomicAdd(o + (n * KV + i) * 2 + 1, -m);
    }
  }
}

// Two DIFFERENT sets at KV >= 8 (BASELINE config 5: K = 8 directions): 64 complex fp64 accumulators per lane do not fit, so the
// rows i of the Gram matrix are split over blockIdx.z in NH groups of KV / NH (each group re-reads the second set: still 16-byte
// loads, against 465 us on the scalar kernel for one 8 x 8 pass at C5)
template <int KV, int NH>
__global__ __launch_bounds__(256) void gram4_rows_kernel(VecSet A, VecSet Bs, double* __restrict__ out, int K, long N, long chunk) {
  constexpr int NI = KV / NH, NP = NI * KV;
  static_assert(KV % NH == 0, "row groups");
  __shared__ double red[4][NP * 2];
  const int b = blockIdx.y, i0 = blockIdx.z * NI;
  const long t0 = (long)blockIdx.x * chunk;
  const long t1 = t0 + chunk < N ? t0 + chunk : N;
  double ar[NP], ai[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) ar[i] = ai[i] = 0.0;
  for (long t = t0 + 4 * threadIdx.x; t < t1; t += 1024) {
    float4 xr[NI], xi[NI], yr[KV], yi[KV];
#pragma unroll
    for (int i = 0; i < NI; ++i) {                         // rows i0 .. i0 + NI - 1 of set A (plain vectors: no gt / pred row here)
      const float* p = A.v + ((size_t)(b * K + i0 + i) * 2) * N + t;
      xr[i] = (i0 + i < K) ? *reinterpret_cast<const float4*>(p) : make_float4(0.f, 0.f, 0.f, 0.f);
      xi[i] = (i0 + i < K) ? *reinterpret_cast<const float4*>(p + N) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    load_set4<KV>(Bs, K, b, N, t, yr, yi);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
      for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int n = 0; n < KV; ++n) {
          const double a_r = f4(xr[i], e), a_i = f4(xi[i], e), b_r = f4(yr[n], e), b_i = f4(yi[n], e);
          ar[i * KV + n] += a_r * b_r + a_i * b_i;
          ai[i * KV + n] += a_r * b_i - a_i * b_r;
        }
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const double r = wave_sum(ar[i]), m = wave_sum(ai[i]);
    if (lane == 0) { red[wave][2 * i] = r; red[wave][2 * i + 1] = m; }
  }
  __syncthreads();
  if (threadIdx.x < NP) {
    const int p = threadIdx.x, i = i0 + p / KV, n = p % KV;
    double* o = out + (size_t)b * KV * KV * 2;
    atomicAdd(o + (i * KV + n) * 2, ((red[0][2 * p] + red[1][2 * p]) + red[2][2 * p]) + red[3][2 * p]);
    atomicAdd(o + (i * KV + n) * 2 + 1, ((red[0][2 * p + 1] + red[1][2 * p + 1]) + red[2][2 * p + 1]) + red[3][2 * p + 1]);
  }
}

// out_i[t] = sum_m M1[i][m] * A_m[t] + sum_m M2[i][m] * B_m[t]      (complex coefficients, i,m < K; set index K of
// A = gt - pred when present).  M1/M2: [B][KV][KV] double2, null = skip.  out [B][K][2][N].  fp64 combination.
template <int KV>
__global__ __launch_bounds__(256) void combine_kernel(VecSet A, const double* __restrict__ M1, VecSet Bs,
                                                      const double* __restrict__ M2, float* __restrict__ out, int K, long N) {
  __shared__ double2 c1[KV * KV], c2[KV * KV];
  const int b = blockIdx.y;
  for (int i = threadIdx.x; i < KV * KV; i += 256) {
    c1[i] = M1 ? make_double2(M1[((size_t)b * KV * KV + i) * 2], M1[((size_t)b * KV * KV + i) * 2 + 1]) : make_double2(0, 0);
    c2[i] = M2 ? make_double2(M2[((size_t)b * KV * KV + i) * 2], M2[((size_t)b * KV * KV + i) * 2 + 1]) : make_double2(0, 0);
  }
  __syncthreads();
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < N; t += (long)gridDim.x * 256) {
    float xr[KV], xi[KV], yr[KV], yi[KV];
    load_set<KV>(A, K, b, N, t, xr, xi);
    if (M2) load_set<KV>(Bs, K, b, N, t, yr, yi);
#pragma unroll
    for (int i = 0; i < KV; ++i) {
      if (i >= K) break;
      double orr = 0.0, oi = 0.0;
#pragma unroll
      for (int m = 0; m < KV; ++m) {
        const double2 c = c1[i * KV + m];
        orr += c.x * xr[m] - c.y * xi[m];
        oi += c.x * xi[m] + c.y * xr[m];
      }
      if (M2) {
#pragma unroll
        for (int m = 0; m < KV; ++m) {
          const double2 c = c2[i * KV + m];
          orr += c.x * yr[m] - c.y * yi[m];
          oi += c.x * yi[m] + c.y * yr[m];
        }
      }
      float* p = out + ((size_t)(b * K + i) * 2) * N + t;
      p[0] = (float)orr;
      p[N] = (float)oi;
    }
  }
}

// the same combination, four consecutive t per lane (16-byte loads and stores)
template <int KV>
__global__ __launch_bounds__(256) void combine4_kernel(VecSet A, const double* __restrict__ M1, VecSet Bs,
                                                       const double* __restrict__ M2, float* __restrict__ out, int K, long N) {
  __shared__ double2 c1[KV * KV], c2[KV * KV];
  const int b = blockIdx.y;
  for (int i = threadIdx.x; i < KV * KV; i += 256) {
    c1[i] = M1 ? make_double2(M1[((size_t)b * KV * KV + i) * 2], M1[((size_t)b * KV * KV + i) * 2 + 1]) : make_double2(0, 0);
    c2[i] = M2 ? make_double2(M2[((size_t)b * KV * KV + i) * 2], M2[((size_t)b * KV * KV + i) * 2 + 1]) : make_double2(0, 0);
  }
  __syncthreads();
  for (long t = ((long)blockIdx.x * 256 + threadIdx.x) * 4; t < N; t += (long)gridDim.x * 1024) {
    float4 xr[KV], xi[KV], yr[KV], yi[KV];
    load_set4<KV>(A, K, b, N, t, xr, xi);
    if (M2) load_set4<KV>(Bs, K, b, N, t, yr, yi);
#pragma unroll
    for (int i = 0; i < KV; ++i) {
      if (i >= K) break;
      float o_r[4], o_i[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        double orr = 0.0, oi = 0.0;
#pragma unroll
        for (int m = 0; m < KV; ++m) {
          const double2 c = c1[i * KV + m];
          orr += c.x * f4(xr[m], e) - c.y * f4(xi[m], e);
          oi += c.x * f4(xi[m], e) + c.y * f4(xr[m], e);
        }
        if (M2) {
#pragma unroll
          for (int m = 0; m < KV; ++m) {
            const double2 c = c2[i * KV + m];
            orr += c.x * f4(yr[m], e) - c.y * f4(yi[m], e);
            oi += c.x * f4(yi[m], e) + c.y * f4(yr[m], e);
          }
        }
        o_r[e] = (float)orr;
        o_i[e] = (float)oi;
      }
      float* p = out + ((size_t)(b * K + i) * 2) * N + t;
      *reinterpret_cast<float4*>(p) = make_float4(o_r[0], o_r[1], o_r[2], o_r[3]);
      *reinterpret_cast<float4*>(p + N) = make_float4(o_i[0], o_i[1], o_i[2], o_i[3]);
    }
  }
}

struct cd { double r, i; };
__device__ __forceinline__ cd cmul(cd a, cd b) { return {a.r * b.r - a.i * b.i, a.r * b.i + a.i * b.r}; }
__device__ __forceinline__ cd cconj(cd a) { return {a.r, -a.i}; }
__device__ __forceinline__ cd cadd(cd a, cd b) { return {a.r + b.r, a.i + b.i}; }
__device__ __forceinline__ cd csub(cd a, cd b) { return {a.r - b.r, a.i - b.i}; }
constexpr int KMAX = 9;

// <sum_m a_m x_m, sum_n b_n x_n> = sum_{m,n} conj(a_m) b_n G[m][n]
__device__ cd gdot(const cd* a, const cd* bvec, const double* G, int KV, int K) {
  cd s{0, 0};
  for (int m = 0; m < K; ++m)
    for (int n = 0; n < K; ++n) {
      const cd g{G[(m * KV + n) * 2], G[(m * KV + n) * 2 + 1]};
      s = cadd(s, cmul(cmul(cconj(a[m]), bvec[n]), g));
    }
  return s;
}

// Gram-Schmidt in coefficient space.  G [B][KV][KV] (x-Gram).  Outputs (all [B][KV][KV] double2):
//   C  : w_i   = sum_m C[i][m] x_m        Ch : w_hat_i = sum_m Ch[i][m] x_m
__global__ void gs_solve_kernel(const double* __restrict__ Gall, double* __restrict__ Call, double* __restrict__ Chall,
                                int B, int K, int KV) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const double* G = Gall + (size_t)b * KV * KV * 2;
  cd C[KMAX][KMAX], Ch[KMAX][KMAX];
  for (int i = 0; i < K; ++i) {
    for (int m = 0; m < K; ++m) C[i][m] = {m == i ? 1.0 : 0.0, 0.0};
    for (int j = 0; j < i; ++j) {
      const cd s = gdot(C[i], Ch[j], G, KV, K);          // sum conj(w) * w_hat_j
      for (int m = 0; m < K; ++m) C[i][m] = csub(C[i][m], cmul(s, Ch[j][m]));
    }
    const double nrm = sqrt(gdot(C[i], C[i], G, KV, K).r);
    for (int m = 0; m < K; ++m) Ch[i][m] = {C[i][m].r / nrm, C[i][m].i / nrm};
  }
  double* Co = Call + (size_t)b * KV * KV * 2;
  double* Cho = Chall + (size_t)b * KV * KV * 2;
  for (int i = 0; i < KV; ++i)
    for (int m = 0; m < KV; ++m) {
      const bool in = i < K && m < K;
      Co[(i * KV + m) * 2] = in ? C[i][m].r : 0.0;
      Co[(i * KV + m) * 2 + 1] = in ? C[i][m].i : 0.0;
      Cho[(i * KV + m) * 2] = in ? Ch[i][m].r : 0.0;
      Cho[(i * KV + m) * 2 + 1] = in ? Ch[i][m].i : 0.0;
    }
}

// Backward of Gram-Schmidt (w_hat detached): dx_i = A_0^T ... A_{i-1}^T g_i with A_j^T(u) = u - w_hat_j <u, w_hat_j>,
// <u, v> = sum conj(u) v.  With u = g_i + sum_m d_m x_m:
//   <u, w_hat_j> = sum_n Ch[j][n] P[i][n] + sum_{m,n} conj(d_m) Ch[j][n] G[m][n],   P[i][n] = <g_i, x_n>.
// Output D [B][KV][KV]: dx_i = g_i + sum_m D[i][m] x_m.
__global__ void gs_bwd_solve_kernel(const double* __restrict__ Gall, const double* __restrict__ Pall,
                                    const double* __restrict__ Chall, double* __restrict__ Dall, int B, int K, int KV) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const double* G = Gall + (size_t)b * KV * KV * 2;
  const double* P = Pall + (size_t)b * KV * KV * 2;
  const double* Chp = Chall + (size_t)b * KV * KV * 2;
  double* Do = Dall + (size_t)b * KV * KV * 2;
  for (int i = 0; i < KV * KV * 2; ++i) Do[i] = 0.0;
  for (int i = 0; i < K; ++i) {
    cd d[KMAX];
    for (int m = 0; m < K; ++m) d[m] = {0, 0};
    for (int j = i - 1; j >= 0; --j) {
      cd chj[KMAX];
      for (int n = 0; n < K; ++n) chj[n] = {Chp[(j * KV + n) * 2], Chp[(j * KV + n) * 2 + 1]};
      cd s{0, 0};
      for (int n = 0; n < K; ++n) s = cadd(s, cmul(chj[n], cd{P[(i * KV + n) * 2], P[(i * KV + n) * 2 + 1]}));
      s = cadd(s, gdot(d, chj, G, KV, K));
      for (int m = 0; m < K; ++m) d[m] = csub(d[m], cmul(s, chj[m]));
    }
    for (int m = 0; m < K; ++m) { Do[(i * KV + m) * 2] = d[m].r; Do[(i * KV + m) * 2 + 1] = d[m].i; }
  }
}

// The same two solves with K a compile-time constant (KV may exceed K): every loop unrolls, the K x K coefficient
// matrices live in registers -- the generic kernels index them dynamically, i.e. through scratch memory (98 / 73 us per
// call for 32 5 x 5 problems; the arithmetic is microseconds).  Same operation order as the generic kernels.
template <int KT>
__device__ __forceinline__ cd gdot_t(const cd (&a)[KT], const cd (&bvec)[KT], const cd (&G)[KT][KT]) {
  cd s{0, 0};
#pragma unroll
  for (int m = 0; m < KT; ++m)
#pragma unroll
    for (int n = 0; n < KT; ++n) s = cadd(s, cmul(cmul(cconj(a[m]), bvec[n]), G[m][n]));
  return s;
}

template <int KT>
__global__ __launch_bounds__(64) void gs_solve_t_kernel(const double* __restrict__ Gall, double* __restrict__ Call, double* __restrict__ Chall, int B,
                                  int KV) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const double* Gp = Gall + (size_t)b * KV * KV * 2;
  cd G[KT][KT], C[KT][KT], Ch[KT][KT];
#pragma unroll
  for (int m = 0; m < KT; ++m)
#pragma unroll
    for (int n = 0; n < KT; ++n) G[m][n] = {Gp[(m * KV + n) * 2], Gp[(m * KV + n) * 2 + 1]};
#pragma unroll
  for (int i = 0; i < KT; ++i) {
#pragma unroll
    for (int m = 0; m < KT; ++m) C[i][m] = {m == i ? 1.0 : 0.0, 0.0};
#pragma unroll
    for (int j = 0; j < i; ++j) {
      const cd sj = gdot_t<KT>(C[i], Ch[j], G);
#pragma unroll
      for (int m = 0; m < KT; ++m) C[i][m] = csub(C[i][m], cmul(sj, Ch[j][m]));
    }
    const double nrm = sqrt(gdot_t<KT>(C[i], C[i], G).r);
#pragma unroll
    for (int m = 0; m < KT; ++m) Ch[i][m] = {C[i][m].r / nrm, C[i][m].i / nrm};
  }
  double* Co = Call + (size_t)b * KV * KV * 2;
  double* Cho = Chall + (size_t)b * KV * KV * 2;
  for (int e = 0; e < KV * KV * 2; ++e) { Co[e] = 0.0; Cho[e] = 0.0; }
#pragma unroll
  for (int i = 0; i < KT; ++i)
#pragma unroll
    for (int m = 0; m < KT; ++m) {
      Co[(i * KV + m) * 2] = C[i][m].r;
      Co[(i * KV + m) * 2 + 1] = C[i][m].i;
      Cho[(i * KV + m) * 2] = Ch[i][m].r;
      Cho[(i * KV + m) * 2 + 1] = Ch[i][m].i;
    }
}

template <int KT>
__global__ __launch_bounds__(64) void gs_bwd_solve_t_kernel(const double* __restrict__ Gall, const double* __restrict__ Pall,
                                      const double* __restrict__ Chall, double* __restrict__ Dall, int B, int KV) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const double* Gp = Gall + (size_t)b * KV * KV * 2;
  const double* P = Pall + (size_t)b * KV * KV * 2;
  const double* Chp = Chall + (size_t)b * KV * KV * 2;
  double* Do = Dall + (size_t)b * KV * KV * 2;
  cd G[KT][KT], Ch[KT][KT];
#pragma unroll
  for (int m = 0; m < KT; ++m)
#pragma unroll
    for (int n = 0; n < KT; ++n) {
      G[m][n] = {Gp[(m * KV + n) * 2], Gp[(m * KV + n) * 2 + 1]};
      Ch[m][n] = {Chp[(m * KV + n) * 2], Chp[(m * KV + n) * 2 + 1]};
    }
  for (int e = 0; e < KV * KV * 2; ++e) Do[e] = 0.0;
#pragma unroll
  for (int i = 0; i < KT; ++i) {
    cd d[KT], Pi[KT];
#pragma unroll
    for (int m = 0; m < KT; ++m) {
      d[m] = {0, 0};
      Pi[m] = {P[(i * KV + m) * 2], P[(i * KV + m) * 2 + 1]};
    }
#pragma unroll
    for (int j = i - 1; j >= 0; --j) {
      cd sj{0, 0};
#pragma unroll
      for (int n = 0; n < KT; ++n) sj = cadd(sj, cmul(Ch[j][n], Pi[n]));
      sj = cadd(sj, gdot_t<KT>(d, Ch[j], G));
#pragma unroll
      for (int m = 0; m < KT; ++m) d[m] = csub(d[m], cmul(sj, Ch[j][m]));
    }
#pragma unroll
    for (int m = 0; m < KT; ++m) { Do[(i * KV + m) * 2] = d[m].r; Do[(i * KV + m) * 2 + 1] = d[m].i; }
  }
}

// ---- the same two solves, ONE WAVE PER SAMPLE (round 4): lane l < K^2 owns Gram entry (m, n) = (l / K, l % K) and forms its term
// conj(a_m) b_n G_mn of every inner product <sum a_m x_m, sum b_n x_n>; a fixed xor-shuffle tree adds the terms (fp64, same on every
// run).  The one-thread-per-sample kernels above walk ~K^2 (K+1) / 2 dependent complex products one after the other (16.7 / 18.7 us
// for 32 5 x 5 problems; 320 / 271 us for 8 x 8 ones, whose register version does not exist: 3 x 64 complex doubles per thread).
// Vectors live in LDS as [K] complex doubles; K <= 8 (K^2 <= 64 lanes).
__device__ __forceinline__ cd wave_csum(cd v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { v.r += __shfl_xor(v.r, o, 64); v.i += __shfl_xor(v.i, o, 64); }
  return v;
}
__global__ __launch_bounds__(64) void gs_solve_wave_kernel(const double* __restrict__ Gall, double* __restrict__ Call,
                                                           double* __restrict__ Chall, int K, int KV) {
  __shared__ cd Cs[8][8], Chs[8][8];
  const int b = blockIdx.x, lane = threadIdx.x, m = lane / K, n = lane % K;
  const bool on = lane < K * K;
  const double* Gp = Gall + (size_t)b * KV * KV * 2;
  const cd g = on ? cd{Gp[(m * KV + n) * 2], Gp[(m * KV + n) * 2 + 1]} : cd{0, 0};
  for (int i = 0; i < K; ++i) {
    if (lane < K) Cs[i][lane] = {lane == i ? 1.0 : 0.0, 0.0};
    __syncthreads();
    for (int j = 0; j < i; ++j) {
      const cd t = on ? cmul(cmul(cconj(Cs[i][m]), Chs[j][n]), g) : cd{0, 0};
      const cd sj = wave_csum(t);                          // sum conj(w) * w_hat_j
      __syncthreads();
      if (lane < K) Cs[i][lane] = csub(Cs[i][lane], cmul(sj, Chs[j][lane]));
      __syncthreads();
    }
    const cd t = on ? cmul(cmul(cconj(Cs[i][m]), Cs[i][n]), g) : cd{0, 0};
    const double nrm = sqrt(wave_csum(t).r);
    if (lane < K) Chs[i][lane] = {Cs[i][lane].r / nrm, Cs[i][lane].i / nrm};
    __syncthreads();
  }
  double* Co = Call + (size_t)b * KV * KV * 2;
  double* Cho = Chall + (size_t)b * KV * KV * 2;
  for (int e = lane; e < KV * KV; e += 64) {
    const int i = e / KV, mm = e % KV;
    const bool in = i < K && mm < K;
    Co[e * 2] = in ? Cs[i][mm].r : 0.0;
    Co[e * 2 + 1] = in ? Cs[i][mm].i : 0.0;
    Cho[e * 2] = in ? Chs[i][mm].r : 0.0;
    Cho[e * 2 + 1] = in ? Chs[i][mm].i : 0.0;
  }
}

__global__ __launch_bounds__(64) void gs_bwd_solve_wave_kernel(const double* __restrict__ Gall, const double* __restrict__ Pall,
                                                               const double* __restrict__ Chall, double* __restrict__ Dall, int K,
                                                               int KV) {
  __shared__ cd Chs[8][8], ds[8];
  const int b = blockIdx.x, lane = threadIdx.x, m = lane / K, n = lane % K;
  const bool on = lane < K * K;
  const double* Gp = Gall + (size_t)b * KV * KV * 2;
  const double* P = Pall + (size_t)b * KV * KV * 2;
  const double* Chp = Chall + (size_t)b * KV * KV * 2;
  double* Do = Dall + (size_t)b * KV * KV * 2;
  const cd g = on ? cd{Gp[(m * KV + n) * 2], Gp[(m * KV + n) * 2 + 1]} : cd{0, 0};
  if (on) Chs[m][n] = {Chp[(m * KV + n) * 2], Chp[(m * KV + n) * 2 + 1]};
  for (int e = lane; e < KV * KV * 2; e += 64) Do[e] = 0.0;
  __syncthreads();
  for (int i = 0; i < K; ++i) {
    if (lane < K) ds[lane] = {0, 0};
    const cd pin = lane < K ? cd{P[(i * KV + lane) * 2], P[(i * KV + lane) * 2 + 1]} : cd{0, 0};   // P[i][n] on lane n
    __syncthreads();
    for (int j = i - 1; j >= 0; --j) {
      cd t = on ? cmul(cmul(cconj(ds[m]), Chs[j][n]), g) : cd{0, 0};
      if (lane < K) t = cadd(t, cmul(Chs[j][lane], pin));
      const cd sj = wave_csum(t);
      __syncthreads();
      if (lane < K) ds[lane] = csub(ds[lane], cmul(sj, Chs[j][lane]));
      __syncthreads();
    }
    if (lane < K) { Do[(i * KV + lane) * 2] = ds[lane].r; Do[(i * KV + lane) * 2 + 1] = ds[lane].i; }
    __syncthreads();
  }
}

// Loss scalars from the Gram of [w_0..w_{K-1}, e] (KV = K+1), trainer.py:269-298.
// Also the backward coefficients: dL/dw_i = a_i w_i + b_i e  with upstream weights
//   dL/d(reconst_b) = gr[b] (+ gobj/B folded in by the host), dL/d(sm_bi) = gs (= gobj*lambda/(B*K)).
__global__ void loss_solve_kernel(const double* __restrict__ Gall, float* __restrict__ err_norm, float* __restrict__ proj_re,
                                  float* __restrict__ proj_im, float* __restrict__ proj_mag, float* __restrict__ w_norms,
                                  float* __restrict__ reconst, float* __restrict__ sm, double* __restrict__ coefA,
                                  double* __restrict__ coefE, int B, int K, double eps, int eps_in_norms, float lam,
                                  float* __restrict__ objective) {
  // eps_in_norms = 0: enhancement trainer (trainer.py:269-298): eps only in the divisors, reported norms are plain.
  // eps_in_norms = 1: inpainting trainer (inpainting/trainer/nppc_trainer.py:352-372): w_norms and err_norm carry the
  //                   eps themselves (w_norms = (|w| + eps) / (|e| + eps), err_norm = |e| + eps).
  // objective != null (ONE workgroup, B <= blockDim.x): also objective = mean_b reconst + lam * mean_{b,i} sm
  // (trainer.py:300-304 `_calculate_final_objective`), the samples added in index order -- four ATen launches less per step
  __shared__ double osum[2][16];
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  double rec_b = 0.0, sm_b = 0.0;
  if (b < B) {
  const int KV = K + 1;
  const double* G = Gall + (size_t)b * KV * KV * 2;
  const double en = sqrt(G[(K * KV + K) * 2]);
  const double de = en + eps;
  err_norm[b] = (float)(eps_in_norms ? de : en);
  double rec = 1.0;
  for (int i = 0; i < K; ++i) {
    const double wn = sqrt(G[(i * KV + i) * 2]);
    const double dw = wn + eps;
    const double qr = G[(i * KV + K) * 2], qi = G[(i * KV + K) * 2 + 1];   // q = <w_i, e> = sum conj(w_i) e
    const double pr = qr / (dw * de), pi = qi / (dw * de);
    const double pm2 = pr * pr + pi * pi;
    const double wno = (eps_in_norms ? dw : wn) / de;
    proj_re[b * K + i] = (float)pr;
    proj_im[b * K + i] = (float)pi;
    proj_mag[b * K + i] = (float)sqrt(pm2);
    w_norms[b * K + i] = (float)wno;
    const double dsm = wno * wno - pm2;
    sm[b * K + i] = (float)(dsm * dsm);
    sm_b += (double)(float)(dsm * dsm);
    rec -= pm2;
    // d(pm2)/dw = 2 conj(q) e / (dw^2 de^2) - 2 |q|^2 / (dw^3 de^2) * w / wn ;  d(wno^2)/dw = 2 wn/de^2 * w / wn
    // store unit-weight pieces; the host-provided upstream weights are applied in loss_bwd_coef_kernel
    coefA[(b * K + i) * 4 + 0] = -2.0 * (qr * qr + qi * qi) / (dw * dw * dw * de * de) / (wn > 0 ? wn : 1.0);  // d pm2 / dw : w part
    // d sm / dw : w part (x w):  d(wno^2)/dw = 2 wno / de * w / wn
    coefA[(b * K + i) * 4 + 1] = 2.0 * dsm * 2.0 * wno / (de * (wn > 0 ? wn : 1.0));
    coefE[(b * K + i) * 2 + 0] = 2.0 * qr / (dw * dw * de * de);     // d pm2 / dw : e part = 2 conj(q) e / (..), conj(q) = (qr, -qi)
    coefE[(b * K + i) * 2 + 1] = -2.0 * qi / (dw * dw * de * de);
  }
  reconst[b] = (float)rec;
  rec_b = (double)(float)rec;
  }
  if (objective) {
    const double r = wave_sum(rec_b), m = wave_sum(sm_b);
    if ((threadIdx.x & 63) == 0) { osum[0][threadIdx.x >> 6] = r; osum[1][threadIdx.x >> 6] = m; }
    __syncthreads();
    if (threadIdx.x == 0) {
      double R = 0.0, M = 0.0;
      for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { R += osum[0][w]; M += osum[1][w]; }
      *objective = (float)(R / B) + lam * (float)(M / ((double)B * K));
    }
  }
}

// M1[b][i][i] = -gr_b * A0 + gs * A1 ; M1[b][i][K] = -gr_b * E   (gr_b = dL/d reconst_b, gs = dL/d sm)
__global__ void loss_bwd_coef_kernel(const double* __restrict__ coefA, const double* __restrict__ coefE,
                                     const float* __restrict__ grec, float gobj_over_B, float gsm, double* __restrict__ M1,
                                     int B, int K, const float* __restrict__ gobj_dev) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  if (gobj_dev) {            // upstream d L / d objective stays on the device: no host read-back inside backward()
    gobj_over_B *= *gobj_dev;
    gsm *= *gobj_dev;
  }
  const int KV = K + 1;
  double* M = M1 + (size_t)b * KV * KV * 2;
  for (int i = 0; i < KV * KV * 2; ++i) M[i] = 0.0;
  const double gr = (double)gobj_over_B + (grec ? (double)grec[b] : 0.0);
  for (int i = 0; i < K; ++i) {
    M[(i * KV + i) * 2] = -gr * coefA[(b * K + i) * 4 + 0] + (double)gsm * coefA[(b * K + i) * 4 + 1];
    M[(i * KV + K) * 2] = -gr * coefE[(b * K + i) * 2 + 0];
    M[(i * KV + K) * 2 + 1] = -gr * coefE[(b * K + i) * 2 + 1];
  }
}

template <int KV>
static void launch_gram(const VecSet& A, const VecSet& Bs, int same, double* out, int B, int K, long N, hipStream_t s) {
  auto al16 = [](const void* p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  if constexpr (KV == 8) {
    if (!same && N % 4 == 0 && al16(A.v) && al16(Bs.v) && !A.gt && !Bs.gt) {     // two sets at K = 8: row groups (registers)
      const long chunk4 = 1024L * 2;
      hipLaunchKernelGGL((gram4_rows_kernel<KV, 2>), dim3(ceil_div(N, chunk4), B, 2), dim3(256), 0, s, A, Bs, out, K, N, chunk4);
      return;
    }
  }
  if (N % 4 == 0 && al16(A.v) && al16(A.gt) && al16(A.pred) && al16(Bs.v) && (same || KV <= 7)) {   // (KV > 7, two sets: registers)
    const long chunk4 = 1024L * 2;                     // two iterations of 4 t per lane: 512 workgroups at C2
    dim3 grid4(ceil_div(N, chunk4), B);
    if (same)
      hipLaunchKernelGGL((gram4_kernel<KV, true>), grid4, dim3(256), 0, s, A, Bs, out, K, N, chunk4);
    else
      hipLaunchKernelGGL((gram4_kernel<KV, false>), grid4, dim3(256), 0, s, A, Bs, out, K, N, chunk4);
    return;
  }
  const long chunk = 256L * 16;
  dim3 grid(ceil_div(N, chunk), B);
  if (same)
    hipLaunchKernelGGL((gram_kernel<KV, true>), grid, dim3(256), 0, s, A, Bs, out, K, N, chunk);
  else
    hipLaunchKernelGGL((gram_kernel<KV, false>), grid, dim3(256), 0, s, A, Bs, out, K, N, chunk);
}
template <int KV>
static void launch_combine(const VecSet& A, const double* M1, const VecSet& Bs, const double* M2, float* out, int B, int K,
                           long N, hipStream_t s) {
  auto al16 = [](const void* p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  if (N % 4 == 0 && al16(A.v) && al16(A.gt) && al16(A.pred) && al16(Bs.v) && al16(out)) {
    long g4 = (N / 4 + 255) / 256;
    if (g4 > 64) g4 = 64;
    hipLaunchKernelGGL(combine4_kernel<KV>, dim3((int)g4, B), dim3(256), 0, s, A, M1, Bs, M2, out, K, N);
    return;
  }
  long g = (N + 255) / 256;
  if (g > 256) g = 256;
  hipLaunchKernelGGL(combine_kernel<KV>, dim3((int)g, B), dim3(256), 0, s, A, M1, Bs, M2, out, K, N);
}

#define KV_SWITCH(KVv, CALL)                     \
  switch (KVv) {                                 \
    case 1: { constexpr int KVc = 1; CALL; } break; \
    case 2: { constexpr int KVc = 2; CALL; } break; \
    case 3: { constexpr int KVc = 3; CALL; } break; \
    case 4: { constexpr int KVc = 4; CALL; } break; \
    case 5: { constexpr int KVc = 5; CALL; } break; \
    case 6: { constexpr int KVc = 6; CALL; } break; \
    case 7: { constexpr int KVc = 7; CALL; } break; \
    case 8: { constexpr int KVc = 8; CALL; } break; \
    case 9: { constexpr int KVc = 9; CALL; } break; \
    default: return NPPC_EUNSUPPORTED;           \
  }

}  // namespace

extern "C" {

// out[b][i][n] = <a_i, b_n>;  vector sets of K vectors [B][K][2][N]; when gt/pred are given the set has the extra
// vector e = gt - pred at index K (KV = K+1).  `out` must be zeroed by the caller.
int nppc_gram(const float* a, const float* b_or_null, const float* gt, const float* pred, double* out, int B, int K,
              long N, void* stream) {
  if (!a || !out || B <= 0 || K < 1 || N <= 0) return NPPC_EBADARG;
  const int KV = K + (gt ? 1 : 0);
  VecSet A{a, gt, pred}, Bs{b_or_null ? b_or_null : a, gt, pred};
  hipStream_t s = (hipStream_t)stream;
  KV_SWITCH(KV, launch_gram<KVc>(A, Bs, b_or_null ? 0 : 1, out, B, K, N, s));
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

// out_i = sum_m M1[i][m] A_m + sum_m M2[i][m] B_m   (M*: [B][KV][KV] complex double; KV = K + (gt ? 1 : 0))
int nppc_combine(const float* a, const double* M1, const float* b, const double* M2, const float* gt, const float* pred,
                 float* out, int B, int K, long N, void* stream) {
  if (!a || !M1 || !out || B <= 0 || K < 1) return NPPC_EBADARG;
  const int KV = K + (gt ? 1 : 0);
  VecSet A{a, gt, pred}, Bs{b, nullptr, nullptr};
  hipStream_t s = (hipStream_t)stream;
  KV_SWITCH(KV, launch_combine<KVc>(A, M1, Bs, (b && M2) ? M2 : nullptr, out, B, K, N, s));
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

int nppc_gs_solve(const double* G, double* C, double* Ch, int B, int K, int KV, void* stream) {
  if (!G || !C || !Ch || K > KMAX || KV < K) return NPPC_EBADARG;
  const dim3 grid(ceil_div(B, 64));
  hipStream_t s = (hipStream_t)stream;
  static const bool wave_solve = [] { const char* e = getenv("NPPC_GS_WAVE_SOLVE"); return !(e && e[0] == '0'); }();   // A/B switch
  if (wave_solve && K <= 8) {        // one wave per sample, one lane per Gram entry
    hipLaunchKernelGGL(gs_solve_wave_kernel, dim3(B), dim3(64), 0, s, G, C, Ch, K, KV);
    NPPC_CHECK_LAUNCH();
    return NPPC_OK;
  }
  switch (K) {      // one thread per sample: register-resident specialisations for K = 2..6, generic beyond
    case 2: hipLaunchKernelGGL(gs_solve_t_kernel<2>, grid, dim3(64), 0, s, G, C, Ch, B, KV); break;
    case 3: hipLaunchKernelGGL(gs_solve_t_kernel<3>, grid, dim3(64), 0, s, G, C, Ch, B, KV); break;
    case 4: hipLaunchKernelGGL(gs_solve_t_kernel<4>, grid, dim3(64), 0, s, G, C, Ch, B, KV); break;
    case 5: hipLaunchKernelGGL(gs_solve_t_kernel<5>, grid, dim3(64), 0, s, G, C, Ch, B, KV); break;
    case 6: hipLaunchKernelGGL(gs_solve_t_kernel<6>, grid, dim3(64), 0, s, G, C, Ch, B, KV); break;
    default: hipLaunchKernelGGL(gs_solve_kernel, grid, dim3(64), 0, s, G, C, Ch, B, K, KV);
  }
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

int nppc_gs_bwd_solve(const double* G, const double* P, const double* Ch, double* D, int B, int K, int KV, void* stream) {
  if (!G || !P || !Ch || !D || K > KMAX || KV < K) return NPPC_EBADARG;
  const dim3 grid(ceil_div(B, 64));
  hipStream_t s = (hipStream_t)stream;
  static const bool wave_solve = [] { const char* e = getenv("NPPC_GS_WAVE_SOLVE"); return !(e && e[0] == '0'); }();
  if (wave_solve && K <= 8) {
    hipLaunchKernelGGL(gs_bwd_solve_wave_kernel, dim3(B), dim3(64), 0, s, G, P, Ch, D, K, KV);
    NPPC_CHECK_LAUNCH();
    return NPPC_OK;
  }
  switch (K) {
    case 2: hipLaunchKernelGGL(gs_bwd_solve_t_kernel<2>, grid, dim3(64), 0, s, G, P, Ch, D, B, KV); break;
    case 3: hipLaunchKernelGGL(gs_bwd_solve_t_kernel<3>, grid, dim3(64), 0, s, G, P, Ch, D, B, KV); break;
    case 4: hipLaunchKernelGGL(gs_bwd_solve_t_kernel<4>, grid, dim3(64), 0, s, G, P, Ch, D, B, KV); break;
    case 5: hipLaunchKernelGGL(gs_bwd_solve_t_kernel<5>, grid, dim3(64), 0, s, G, P, Ch, D, B, KV); break;
    case 6: hipLaunchKernelGGL(gs_bwd_solve_t_kernel<6>, grid, dim3(64), 0, s, G, P, Ch, D, B, KV); break;
    default: hipLaunchKernelGGL(gs_bwd_solve_kernel, grid, dim3(64), 0, s, G, P, Ch, D, B, K, KV);
  }
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

int nppc_loss_solve(const double* G, float* err_norm, float* proj_re, float* proj_im, float* proj_mag, float* w_norms,
                    float* reconst, float* sm, double* coefA, double* coefE, int B, int K, void* stream) {
  if (!G || !err_norm || !coefA || !coefE || K + 1 > KMAX) return NPPC_EBADARG;
  hipLaunchKernelGGL(loss_solve_kernel, dim3(ceil_div(B, 64)), dim3(64), 0, (hipStream_t)stream, G, err_norm, proj_re,
                     proj_im, proj_mag, w_norms, reconst, sm, coefA, coefE, B, K, 1e-8, 0, 0.f, (float*)nullptr);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

// Same with the inpainting trainer's epsilon conventions (inpainting/trainer/nppc_trainer.py:352-372): real vectors
// are passed with a zero imaginary plane; eps = 1e-6 sits inside w_norms and err_norm.
int nppc_loss_solve_eps(const double* G, float* err_norm, float* proj_re, float* proj_im, float* proj_mag, float* w_norms,
                        float* reconst, float* sm, double* coefA, double* coefE, int B, int K, double eps, int eps_in_norms,
                        void* stream) {
  if (!G || !err_norm || !coefA || !coefE || K + 1 > KMAX || !(eps >= 0)) return NPPC_EBADARG;
  hipLaunchKernelGGL(loss_solve_kernel, dim3(ceil_div(B, 64)), dim3(64), 0, (hipStream_t)stream, G, err_norm, proj_re,
                     proj_im, proj_mag, w_norms, reconst, sm, coefA, coefE, B, K, eps, eps_in_norms, 0.f, (float*)nullptr);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

// ... and the step's objective = mean(reconst) + lam * mean(sm) (trainer.py:300-304) from the same launch: one workgroup,
// B <= 1024 (NPPC_EUNSUPPORTED beyond: the caller adds the means up itself)
int nppc_loss_solve_obj(const double* G, float* err_norm, float* proj_re, float* proj_im, float* proj_mag, float* w_norms,
                        float* reconst, float* sm, double* coefA, double* coefE, int B, int K, double eps, int eps_in_norms,
                        float lam, float* objective, void* stream) {
  if (!G || !err_norm || !coefA || !coefE || !objective || K + 1 > KMAX || !(eps >= 0)) return NPPC_EBADARG;
  if (B > 1024) return NPPC_EUNSUPPORTED;
  hipLaunchKernelGGL(loss_solve_kernel, dim3(1), dim3(round_up(B, 64)), 0, (hipStream_t)stream, G, err_norm, proj_re, proj_im,
                     proj_mag, w_norms, reconst, sm, coefA, coefE, B, K, eps, eps_in_norms, lam, objective);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

int nppc_loss_bwd_coef(const double* coefA, const double* coefE, const float* grec, float gobj_over_B, float gsm, double* M1,
                       int B, int K, void* stream) {
  if (!coefA || !coefE || !M1) return NPPC_EBADARG;
  hipLaunchKernelGGL(loss_bwd_coef_kernel, dim3(ceil_div(B, 64)), dim3(64), 0, (hipStream_t)stream, coefA, coefE, grec,
                     gobj_over_B, gsm, M1, B, K, (const float*)nullptr);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

int nppc_loss_bwd_coef_dev(const double* coefA, const double* coefE, const float* grec, const float* gobj, float inv_B,
                           float sm_weight, double* M1, int B, int K, void* stream) {
  if (!coefA || !coefE || !M1 || !gobj) return NPPC_EBADARG;
  hipLaunchKernelGGL(loss_bwd_coef_kernel, dim3(ceil_div(B, 64)), dim3(64), 0, (hipStream_t)stream, coefA, coefE, grec, inv_B,
                     sm_weight, M1, B, K, gobj);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

}  // extern "C"
