// MC-dropout + per-sample PCA baseline of the inpainting path for gfx950.
//   * nn.Dropout on a haloed NHWC activation (tmp_utils.py:28-29: last entry of double_conv in down3/down4/up1/up2),
//     counter-based Philox4x32-10 so a (seed, stream, element) triple always gives the same keep bit;
//   * the PCA the reference runs per batch item on the CPU with scikit-learn / torch.linalg.svd
//     (utils.py:393-545 compute_pca_sklearn_batch, compute_pca_and_importance_weights) for ALL items in four launches:
//     mean -> centred K x K Gram (fp64) -> cyclic Jacobi eigen-solver in LDS (fp64) -> components = U^T Xc / s with
//     scikit-learn's sign rule (svd_flip on V: the largest-magnitude entry of every component is positive).
// HBM-bound: every MC sample is read twice (Gram, components); the K x K eigenproblem is on-chip.
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------ Philox4x32-10
struct u32x4 {
  unsigned x, y, z, w;
};

__device__ __forceinline__ u32x4 philox4x32(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
    c1 = (unsigned)p1;
    c3 = (unsigned)p0;
    c0 = n0;
    c2 = n2;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return u32x4{c0, c1, c2, c3};
}

// in place on X [rows][ld], channels [0, C): x <- keep ? x / (1 - p) : 0, keep = (u32 >= p * 2^32); four channels per
// thread share one Philox block (counter = row, channel / 4, stream).  keep_out (nullable) [rows][C] u8 for the tests.
template <typename T>
__global__ __launch_bounds__(256) void dropout_kernel(T* __restrict__ X, long ld, long rows, int C, unsigned thresh,
                                                      float scale, unsigned seed_lo, unsigned seed_hi, unsigned stream_id,
                                                      unsigned char* __restrict__ keep_out) {
  const int c4 = C / 4;
  const long n = rows * c4;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
    const long r = e / c4;
    const int c = (int)(e % c4) * 4;
    const u32x4 u = philox4x32((unsigned)r, (unsigned)((unsigned long long)r >> 32), (unsigned)(c / 4), stream_id, seed_lo,
                               seed_hi);
    const unsigned uu[4] = {u.x, u.y, u.z, u.w};
    T* px = X + r * ld + c;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool keep = uu[i] >= thresh;
      px[i] = from_f32<T>(keep ? to_f32<T>(px[i]) * scale : 0.f);
      if (keep_out) keep_out[r * C + c + i] = keep ? 1 : 0;
    }
  }
}

// ------------------------------------------------------------------------------------------------ PCA
// X [K][B][D] (K Monte-Carlo samples); mean [B][D]
__global__ __launch_bounds__(256) void pca_mean_kernel(const float* __restrict__ X, float* __restrict__ mean, int K, long BD) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= BD) return;
  double s = 0.0;
  for (int k = 0; k < K; ++k) s += (double)X[(size_t)k * BD + i];
  mean[i] = (float)(s / K);
}

// G[b][i][j] += sum_{d in chunk} xc[i][d] * xc[j][d];  xc = float(X - mean) as scikit-learn centres in the input dtype
constexpr int PCA_CH = 64, PCA_KMAX = 60;
__global__ __launch_bounds__(256) void pca_gram_kernel(const float* __restrict__ X, const float* __restrict__ mean,
                                                       double* __restrict__ G, int K, int B, int D) {
  __shared__ float xc[PCA_KMAX][PCA_CH + 1];
  const int b = blockIdx.y, d0 = blockIdx.x * PCA_CH, tid = threadIdx.x;
  for (int e = tid; e < K * PCA_CH; e += 256) {
    const int k = e / PCA_CH, j = e % PCA_CH, d = d0 + j;
    xc[k][j] = d < D ? X[((size_t)k * B + b) * D + d] - mean[(size_t)b * D + d] : 0.f;
  }
  __syncthreads();
  for (int e = tid; e < K * K; e += 256) {
    const int i = e / K, j = e % K;
    if (j < i) continue;
    double s = 0.0;
#pragma unroll 8
    for (int d = 0; d < PCA_CH; ++d) s += (double)xc[i][d] * (double)xc[j][d];
    atomicAdd(&G[((size_t)b * K + i) * K + j], s);
    if (j != i) atomicAdd(&G[((size_t)b * K + j) * K + i], s);
  }
}

// cyclic Jacobi on the symmetric K x K Gram of one item (fp64, LDS); round-robin pairing -> K/2 disjoint rotations per
// round.  Writes the n largest eigenvalues (descending) and their eigenvectors evec[b][i][k].
__global__ __launch_bounds__(256) void pca_eigh_kernel(const double* __restrict__ G, double* __restrict__ eval,
                                                       double* __restrict__ evec, int K, int n) {
  __shared__ double A[PCA_KMAX][PCA_KMAX];
  __shared__ double V[PCA_KMAX][PCA_KMAX];
  __shared__ double cs[PCA_KMAX / 2][2];
  __shared__ int pq[PCA_KMAX / 2][2];
  __shared__ double red[4], red2[4];
  __shared__ int order[PCA_KMAX];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Kp = (K + 1) & ~1, half = Kp / 2;                  // odd K: one padding row/column of zeros (eigenvalue 0)
  for (int e = tid; e < Kp * Kp; e += 256) {
    const int i = e / Kp, j = e % Kp;
    A[i][j] = (i < K && j < K) ? G[((size_t)b * K + i) * K + j] : 0.0;
    V[i][j] = i == j ? 1.0 : 0.0;
  }
  __syncthreads();
  for (int sweep = 0; sweep < 30; ++sweep) {
    double off = 0.0, dg = 0.0;
    for (int e = tid; e < Kp * Kp; e += 256) {
      const int i = e / Kp, j = e % Kp;
      const double a = A[i][j];
      if (i == j) dg += a * a; else off += a * a;
    }
    off = wave_sum(off);
    dg = wave_sum(dg);
    __syncthreads();
    if (lane == 0) { red[wave] = off; red2[wave] = dg; }
    __syncthreads();
    const double offt = red[0] + red[1] + red[2] + red[3], dgt = red2[0] + red2[1] + red2[2] + red2[3];
    __syncthreads();
    if (offt <= 1e-30 * dgt || offt == 0.0) break;
    for (int r = 0; r < Kp - 1; ++r) {
      if (tid < half) {
        int p, q;
        if (tid == 0) { p = Kp - 1; q = r; }
        else { p = (r + tid) % (Kp - 1); q = (r - tid + Kp - 1) % (Kp - 1); }
        if (p > q) { const int t = p; p = q; q = t; }
        const double apq = A[p][q], app = A[p][p], aqq = A[q][q];
        double c = 1.0, s = 0.0;
        if (fabs(apq) > 1e-300 && fabs(apq) > 1e-17 * sqrt(fabs(app * aqq)) ) {
          const double theta = (aqq - app) / (2.0 * apq);
          const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
          c = 1.0 / sqrt(t * t + 1.0);
          s = t * c;
        }
        pq[tid][0] = p; pq[tid][1] = q;
        cs[tid][0] = c; cs[tid][1] = s;
      }
      __syncthreads();
      for (int e = tid; e < half * Kp; e += 256) {               // columns p, q of A and V
        const int m = e / Kp, k = e % Kp, p = pq[m][0], q = pq[m][1];
        const double c = cs[m][0], s = cs[m][1];
        const double ap = A[k][p], aq = A[k][q];
        A[k][p] = c * ap - s * aq;
        A[k][q] = s * ap + c * aq;
        const double vp = V[k][p], vq = V[k][q];
        V[k][p] = c * vp - s * vq;
        V[k][q] = s * vp + c * vq;
      }
      __syncthreads();
      for (int e = tid; e < half * Kp; e += 256) {               // rows p, q of A
        const int m = e / Kp, k = e % Kp, p = pq[m][0], q = pq[m][1];
        const double c = cs[m][0], s = cs[m][1];
        const double ap = A[p][k], aq = A[q][k];
        A[p][k] = c * ap - s * aq;
        A[q][k] = s * ap + c * aq;
      }
      __syncthreads();
    }
  }
  if (tid == 0) {                                               // selection of the n largest (K <= 60, n <= 8)
    for (int i = 0; i < Kp; ++i) order[i] = i;
    for (int i = 0; i < n; ++i) {
      int best = i;
      for (int j = i + 1; j < Kp; ++j)
        if (A[order[j]][order[j]] > A[order[best]][order[best]]) best = j;
      const int t = order[i]; order[i] = order[best]; order[best] = t;
    }
  }
  __syncthreads();
  for (int e = tid; e < n * K; e += 256) {
    const int i = e / K, k = e % K;
    evec[((size_t)b * n + i) * K + k] = V[k][order[i]];
  }
  if (tid < n) eval[(size_t)b * n + tid] = A[order[tid]][order[tid]];
}

// component i of item b: v = sum_k U[k][i] * xc[k] / s_i  (s_i = sqrt(lambda_i)); sign so that the entry of largest
// magnitude is positive (sklearn.utils.extmath.svd_flip(u_based_decision=False), first index on ties like argmax);
// scaled = v * s_i (utils.py:449), weights = s / sum(s over the n kept) (utils.py:443).
__global__ __launch_bounds__(256) void pca_components_kernel(const float* __restrict__ X, const float* __restrict__ mean,
                                                             const double* __restrict__ eval, const double* __restrict__ evec,
                                                             float* __restrict__ comps, float* __restrict__ scaled,
                                                             float* __restrict__ svals, float* __restrict__ weights, int K,
                                                             int B, int D, int n) {
  __shared__ double u[PCA_KMAX];
  __shared__ float bestv[4];
  __shared__ int besti[4];
  __shared__ float sgn;
  const int i = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const double lam = eval[(size_t)b * n + i];
  const double s = lam > 0.0 ? sqrt(lam) : 0.0;
  for (int k = tid; k < K; k += 256) u[k] = evec[((size_t)b * n + i) * K + k];
  __syncthreads();
  const double inv = s > 0.0 ? 1.0 / s : 0.0;
  float* cv = comps + ((size_t)b * n + i) * D;
  float bv = -1.f;
  int bi = 0x7fffffff;
  for (int d = tid; d < D; d += 256) {
    double a = 0.0;
    const float m = mean[(size_t)b * D + d];
    for (int k = 0; k < K; ++k) a += u[k] * (double)(X[((size_t)k * B + b) * D + d] - m);
    const float v = (float)(a * inv);
    cv[d] = v;
    if (fabsf(v) > bv) { bv = fabsf(v); bi = d; }              // ascending d per thread: first index kept on ties
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(bv, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  if (lane == 0) { bestv[wave] = bv; besti[wave] = bi; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 4; ++w)
      if (bestv[w] > bv || (bestv[w] == bv && besti[w] < bi)) { bv = bestv[w]; bi = besti[w]; }
    sgn = (bi < D && cv[bi] < 0.f) ? -1.f : 1.f;
    double tot = 0.0;
    for (int j = 0; j < n; ++j) { const double l = eval[(size_t)b * n + j]; tot += l > 0.0 ? sqrt(l) : 0.0; }
    svals[(size_t)b * n + i] = (float)s;
    weights[(size_t)b * n + i] = (float)(s / tot);
  }
  __syncthreads();
  const float sg = sgn, sf = (float)s;
  float* sc = scaled + ((size_t)b * n + i) * D;
  for (int d = tid; d < D; d += 256) {
    const float v = cv[d] * sg;
    cv[d] = v;
    sc[d] = v * sf;
  }
}

// ------------------------------------------------------------------------------------------------ base_step2 loss
// inpainting/trainer/nppc_trainer.py:285-323: per (item, direction) row of N elements, NPPC direction w against the scaled
// MC-dropout component m: proj = <w / (|w| + eps), m / (|m| + eps)>, reconst = mean_k (1 - proj^2),
// second moment = mean_k ((|w| + eps)^2 - sv^2)^2.  HBM-bound: w and m are read once forward, once backward.
__global__ __launch_bounds__(256) void pair_sums_kernel(const float* __restrict__ w, const float* __restrict__ m,
                                                        double* __restrict__ sums, long N) {
  __shared__ double red[3][4];
  const int row = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* wr = w + (size_t)row * N;
  const float* mr = m + (size_t)row * N;
  double a = 0.0, b = 0.0, c = 0.0;
  for (long i = (long)blockIdx.x * 256 + tid; i < N; i += (long)gridDim.x * 256) {
    const double x = wr[i], y = mr[i];
    a += x * y;
    b += x * x;
    c += y * y;
  }
  a = wave_sum(a); b = wave_sum(b); c = wave_sum(c);
  if (lane == 0) { red[0][wave] = a; red[1][wave] = b; red[2][wave] = c; }
  __syncthreads();
  if (tid < 3) atomicAdd(&sums[(size_t)row * 3 + tid], red[tid][0] + red[tid][1] + red[tid][2] + red[tid][3]);
}

__global__ __launch_bounds__(64) void pair_loss_kernel(const double* __restrict__ sums, const float* __restrict__ sv,
                                                       float* __restrict__ proj, float* __restrict__ w_norms,
                                                       float* __restrict__ reconst, float* __restrict__ second,
                                                       double* __restrict__ coef, int B, int K, double eps) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  double rec = 0.0, sm = 0.0;
  for (int k = 0; k < K; ++k) {
    const size_t r = (size_t)b * K + k;
    const double wm = sums[3 * r], nw = sqrt(sums[3 * r + 1]), a = nw + eps, c = sqrt(sums[3 * r + 2]) + eps;
    const double p = wm / (a * c), s2 = (double)sv[r] * (double)sv[r], d = a * a - s2;
    rec += 1.0 - p * p;
    sm += d * d;
    proj[r] = (float)p;
    w_norms[r] = (float)a;
    const double inw = nw > 0.0 ? 1.0 / nw : 0.0;
    coef[3 * r + 0] = -2.0 * p / (a * c) / K;                 // d reconst_b / d w_k = c0 * m + c1 * w
    coef[3 * r + 1] = 2.0 * p * wm / (a * a * c) * inw / K;
    coef[3 * r + 2] = 4.0 * a * d * inw / K;                  // d second_b / d w_k = c2 * w
  }
  reconst[b] = (float)(rec / K);
  second[b] = (float)(sm / K);
}

__global__ __launch_bounds__(256) void pair_bwd_kernel(const float* __restrict__ w, const float* __restrict__ m,
                                                       const double* __restrict__ coef, const float* __restrict__ grec,
                                                       float g_rec_all, float g_sm_all, float* __restrict__ dw, int K, long N) {
  const int row = blockIdx.y, b = row / K;
  const float gr = (grec ? grec[b] : 0.f) + g_rec_all;
  const float alpha = (float)(coef[3 * (size_t)row] * gr);
  const float beta = (float)(coef[3 * (size_t)row + 1] * gr + coef[3 * (size_t)row + 2] * g_sm_all);
  const float* wr = w + (size_t)row * N;
  const float* mr = m + (size_t)row * N;
  float* dr = dw + (size_t)row * N;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < N; i += (long)gridDim.x * 256) dr[i] = alpha * mr[i] + beta * wr[i];
}

// ------------------------------------------------------------------------------------------------ compute_metrics
// validator_nppc_model.py:742-828: every number it reports is a function of inner products between the direction rows
// and three error rows.  metric_rows builds the error rows, rows_gram the inner products (fp64); the 5 x 5 algebra
// (norms, residual, principal angles) is host-side.
__global__ __launch_bounds__(256) void metric_rows_kernel(const float* __restrict__ pred, const float* __restrict__ clean,
                                                          const float* __restrict__ mean, const float* __restrict__ mask,
                                                          float* __restrict__ out, long N) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const float hole = mask[i] == 0.f ? 1.f : 0.f;
  const float e = pred[i] - clean[i];
  out[i] = e;                                   // error (:807), used unmasked by compute_residual_error_magnitude
  out[N + i] = e * hole;                        // compute_rmse(pred, clean, mask)  (:758-762)
  out[2 * N + i] = (mean[i] - clean[i]) * hole; // compute_rmse(mean_prediction, clean, mask)
}

__global__ __launch_bounds__(256) void rows_gram_kernel(const float* __restrict__ A, const float* __restrict__ Bm, int Rb,
                                                        long N, double* __restrict__ G) {
  __shared__ double red[4];
  const int pair = blockIdx.y, i = pair / Rb, j = pair % Rb, tid = threadIdx.x;
  const float* a = A + (size_t)i * N;
  const float* b = Bm + (size_t)j * N;
  double s = 0.0;
  for (long k = (long)blockIdx.x * 256 + tid; k < N; k += (long)gridDim.x * 256) s += (double)a[k] * (double)b[k];
  s = wave_sum(s);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  if (tid == 0) atomicAdd(&G[pair], red[0] + red[1] + red[2] + red[3]);
}

}  // namespace

extern "C" {

int nppc_dropout(int prec, void* X, long ld, long rows, int C, float p, long seed, int stream_id,
                 unsigned char* keep_out, void* stream) {
  if (!X || rows <= 0 || C <= 0 || C % 4 || ld < C || p < 0.f || p >= 1.f) return NPPC_EBADARG;
  const double t = (double)p * 4294967296.0;
  const unsigned thresh = t >= 4294967295.0 ? 4294967295u : (unsigned)t;
  const float scale = 1.f / (1.f - p);
  const long n = rows * (C / 4);
  const dim3 grid((unsigned)(n / 256 + 1 > 65536 ? 65536 : n / 256 + 1));
  hipStream_t s = (hipStream_t)stream;
  if (prec == NPPC_PREC_BF16)
    hipLaunchKernelGGL(dropout_kernel<bf16_t>, grid, dim3(256), 0, s, (bf16_t*)X, ld, rows, C, thresh, scale, (unsigned)seed,
                       (unsigned)((unsigned long long)seed >> 32), (unsigned)stream_id, keep_out);
  else if (prec == NPPC_PREC_F32)
    hipLaunchKernelGGL(dropout_kernel<float>, grid, dim3(256), 0, s, (float*)X, ld, rows, C, thresh, scale, (unsigned)seed,
                       (unsigned)((unsigned long long)seed >> 32), (unsigned)stream_id, keep_out);
  else
    return NPPC_EBADARG;
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

int nppc_pca_work_elems(int K, int B, int n, long* elems) {
  if (!elems || K <= 0 || B <= 0 || n <= 0) return NPPC_EBADARG;
  *elems = (long)B * ((long)K * K + n + (long)n * K);
  return NPPC_OK;
}

int nppc_pca_batch(const float* X, int K, int B, int D, int n, float* mean, float* comps, float* scaled, float* svals,
                   float* weights, double* work, void* stream) {
  if (!X || !mean || !comps || !scaled || !svals || !weights || !work || B <= 0 || D <= 0 || n <= 0) return NPPC_EBADARG;
  if (K < 2 || K > PCA_KMAX || n > K || n > 8) return NPPC_EUNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  double* G = work;
  double* eval = G + (size_t)B * K * K;
  double* evec = eval + (size_t)B * n;
  if (hipMemsetAsync(G, 0, sizeof(double) * (size_t)B * K * K, s) != hipSuccess) return NPPC_ELAUNCH;
  const long BD = (long)B * D;
  hipLaunchKernelGGL(pca_mean_kernel, dim3(ceil_div(BD, 256L)), dim3(256), 0, s, X, mean, K, BD);
  hipLaunchKernelGGL(pca_gram_kernel, dim3(ceil_div(D, PCA_CH), B), dim3(256), 0, s, X, mean, G, K, B, D);
  hipLaunchKernelGGL(pca_eigh_kernel, dim3(B), dim3(256), 0, s, G, eval, evec, K, n);
  hipLaunchKernelGGL(pca_components_kernel, dim3(n, B), dim3(256), 0, s, X, mean, eval, evec, comps, scaled, svals, weights, K,
                     B, D, n);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

int nppc_pair_loss(const float* w, const float* m, const float* sv, double* sums, float* proj, float* w_norms, float* reconst,
                   float* second, double* coef, int B, int K, long N, double eps, void* stream) {
  if (!w || !m || !sv || !sums || !proj || !w_norms || !reconst || !second || !coef || B <= 0 || K <= 0 || N <= 0)
    return NPPC_EBADARG;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(sums, 0, sizeof(double) * 3 * (size_t)B * K, s) != hipSuccess) return NPPC_ELAUNCH;
  const int chunks = (int)(N / 8192 + 1 > 32 ? 32 : N / 8192 + 1);
  hipLaunchKernelGGL(pair_sums_kernel, dim3(chunks, B * K), dim3(256), 0, s, w, m, sums, N);
  hipLaunchKernelGGL(pair_loss_kernel, dim3(ceil_div(B, 64)), dim3(64), 0, s, sums, sv, proj, w_norms, reconst, second, coef, B,
                     K, eps);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

int nppc_pair_loss_bwd(const float* w, const float* m, const double* coef, const float* grec, float g_rec_all, float g_sm_all,
                       float* dw, int B, int K, long N, void* stream) {
  if (!w || !m || !coef || !dw || B <= 0 || K <= 0 || N <= 0) return NPPC_EBADARG;
  const int chunks = (int)(N / 2048 + 1 > 64 ? 64 : N / 2048 + 1);
  hipLaunchKernelGGL(pair_bwd_kernel, dim3(chunks, B * K), dim3(256), 0, (hipStream_t)stream, w, m, coef, grec, g_rec_all,
                     g_sm_all, dw, K, N);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

int nppc_metric_rows(const float* pred, const float* clean, const float* mean, const float* mask, float* out, long N,
                     void* stream) {
  if (!pred || !clean || !mean || !mask || !out || N <= 0) return NPPC_EBADARG;
  hipLaunchKernelGGL(metric_rows_kernel, dim3(ceil_div(N, 256L)), dim3(256), 0, (hipStream_t)stream, pred, clean, mean, mask, out,
                     N);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

int nppc_rows_gram(const float* A, int Ra, const float* Bm, int Rb, long N, double* G, void* stream) {
  if (!A || !Bm || !G || Ra <= 0 || Rb <= 0 || N <= 0 || (long)Ra * Rb > 65535) return NPPC_EBADARG;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(G, 0, sizeof(double) * (size_t)Ra * Rb, s) != hipSuccess) return NPPC_ELAUNCH;
  const int chunks = (int)(N / 8192 + 1 > 32 ? 32 : N / 8192 + 1);
  hipLaunchKernelGGL(rows_gram_kernel, dim3(chunks, Ra * Rb), dim3(256), 0, s, A, Bm, Rb, N, G);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

}  // extern "C"
