// Full-band front: offline laplace norm (per-sample mean), TSSE channel attention, and the
// scale + transpose into the time-major activation layout the TCN GEMMs consume.
// Reference: audio_zen/model/base_model.py:210-224 (offline_laplace_norm),
//            audio_zen/model/module/attention_model.py:43-98 (ChannelTimeSenseSELayer),
//            fullsubnet_plus.py:158-185 / nppc_audio/networks.py:80-112 (pad look_ahead, norm, attention, concat).
#include "common.h"
#include "nppc_hip.h"

namespace {

// one wave per row: sums[r] = sum_t x[r][t]   (fp64 accumulate: the real/imag maps have |mean| << |x|)
// blockIdx.y = map j of a set of equally shaped maps (the 3 or 6 input maps of a net: one launch for all of them)
struct MapSet { const float* x[6]; };

__global__ __launch_bounds__(256) void rowsum_kernel(MapSet ms, double* __restrict__ sums, long R, int T) {
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  const int lane = threadIdx.x & 63;
  const float* x = ms.x[blockIdx.y];
  sums += (size_t)blockIdx.y * R;
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
// blockDim = C rounded up to 64 (<= 1024): with 256 threads a C = 257 map took two trips through every per-channel loop,
// the second for ONE channel
// blockIdx.y = map j = m*3 + z (z: mag / real / imag branch, whose attention weights sit z * sW elements further; m: noisy /
// enhanced call sharing those weights): rowsum [nmaps][B][C] by j; scale and the saved tensors are laid out [3][nm][B][..]
__device__ __forceinline__ TsseW tsse_branch(TsseW w, long off) {
  for (int i = 0; i < 3; ++i) { w.cw[i] += off; if (w.cb[i]) w.cb[i] += off; }
  w.fcw += off; if (w.fcb) w.fcb += off;
  w.w1 += off; if (w.b1) w.b1 += off;
  w.w2 += off; if (w.b2) w.b2 += off;
  return w;
}

__global__ __launch_bounds__(1024) void tsse_fwd_kernel(MapSet ms, const double* __restrict__ rowsum,
                                                       TsseW w, long sW, int nm, float* __restrict__ scale, float* __restrict__ ns_out,
                                                       float* __restrict__ pre_out, float* __restrict__ sq_out,
                                                       float* __restrict__ h1_out, float* __restrict__ sg_out, int C, int C2,
                                                       int T, int la) {
  __shared__ double red[16];
  __shared__ float sq[TSSE_MAXC];
  __shared__ float h1[TSSE_MAXC / 2];
  __shared__ float ns_s;
  const int b = blockIdx.x, tid = threadIdx.x;
  {
    const int j = blockIdx.y, z = j % 3, m = j / 3, B = gridDim.x;
    const size_t slot = (size_t)z * nm + m;
    w = tsse_branch(w, (long)z * sW);
    rowsum += (size_t)j * B * C;
    scale += slot * B * C;
    if (ns_out) ns_out += slot * B;
    if (pre_out) pre_out += slot * B * C * 3;
    if (sq_out) sq_out += slot * B * C;
    if (h1_out) h1_out += slot * B * C2;
    if (sg_out) sg_out += slot * B * C;
  }
  const float* __restrict__ x = ms.x[blockIdx.y];
  const int Tp = T + la;
  double part = 0.0;
  for (int c = tid; c < C; c += blockDim.x) part += rowsum[(size_t)b * C + c];
  part = wave_sum(part);
  if ((tid & 63) == 0) red[tid >> 6] = part;
  __syncthreads();
  if (tid == 0) {
    double tot_all = 0.0;
    for (int wv = 0; wv < (int)(blockDim.x >> 6); ++wv) tot_all += red[wv];
    const float mu = (float)(tot_all / ((double)C * Tp));
    ns_s = 1.0f / (mu + 1e-5f);
    if (ns_out) ns_out[b] = ns_s;
  }
  __syncthreads();
  const float ns = ns_s;
  for (int c = tid; c < C; c += blockDim.x) {
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
  for (int j = tid; j < C2; j += blockDim.x) {
    float a = w.b1[j];
    const float* wr = w.w1 + (size_t)j * C;
    for (int c = 0; c < C; ++c) a += wr[c] * sq[c];
    a = fmaxf(a, 0.f);
    h1[j] = a;
    if (h1_out) h1_out[(size_t)b * C2 + j] = a;
  }
  __syncthreads();
  for (int c = tid; c < C; c += blockDim.x) {
    float a = w.b2[c];
    const float* wr = w.w2 + (size_t)c * C2;
    for (int j = 0; j < C2; ++j) a += wr[j] * h1[j];
    const float sg = 1.0f / (1.0f + expf(-a));
    if (sg_out) sg_out[(size_t)b * C + c] = sg;
    scale[(size_t)b * C + c] = ns * sg;
  }
}

// y[b][t][coff + c] = x[b][c][t] * scale[b][c]   for t < T  (rows T..Tp-1 and pad columns stay as the caller left them: zero)
// batched over the maps of a net: blockIdx.z = b + B * j, j = m*3 + z: source map j, scale [3][nm][B][C], destination
// branch z (stride sY elements) at column offset m * C
template <typename T>
__global__ __launch_bounds__(256) void scale_transpose_kernel(MapSet ms, const float* __restrict__ scale,
                                                              T* __restrict__ y, int B, int nm, long sY, int C, int Tn, int Tp,
                                                              int ld, int coff) {
  __shared__ float tile[32][33];
  const int j = blockIdx.z / B, b = blockIdx.z % B, c0 = blockIdx.y * 32, t0 = blockIdx.x * 32;
  const float* __restrict__ x = ms.x[j];
  {
    const int z = j % 3, m = j / 3;
    if (scale) scale += ((size_t)z * nm + m) * B * C;
    y += (size_t)z * sY;
    coff += m * C;
  }
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

// ---------------------------------------------------------------- TSSE backward (direction net: attention weights train)
// Round 4: NO atomics on this path -- every sum below has ONE writer and a fixed order (samples in index order, maps m in
// index order), so two runs of the same step give bit-identical attention gradients.
// Workspace of map j (stride sWs = tsse_ws_stride(B, C, KS) floats, KS = ks0 + ks1 + ks2):
//   dsg [B][C] | da2 [B][C] | da1 [B][C/2] | pcb [B][3][C] | pcw [B][C*KS] (conv i at column C * (ks_0 + .. + ks_{i-1})) | pf [B][4]
// pcb / pcw / pf are the per-sample contributions to the conv bias / conv weight / feature_concate_fc (weight x3, bias)
// gradients; tsse_bwd_outer_kernel adds them up over the samples.
__host__ __device__ __forceinline__ long tsse_ws_stride(int B, int C, int KS) { return (long)B * (5L * C + C / 2 + (long)C * KS + 4); }

// X0[b][t][coff+c] = x[b][c][t] * ns_b * sg[b][c]  ->  dsg[b][c] = ns_b * sum_t dX0[b][t][coff+c] * x[b][c][t]
// one workgroup = 64 channels x ALL frames of one sample (64-frame chunks through an LDS tile: x is read along t, dX0 along
// c, both coalesced); thread (cl, tl) sums frames tl + 4 k of every chunk, the four tl partials are added in index order
// batched: blockIdx.z = b + B * j, j = m*3 + z; dX0 of branch z (stride sY) at column offset m*C; ns [3][nm][B];
// dsg workspace of map j at j * sWs
template <typename T>
__global__ __launch_bounds__(256) void tsse_bwd_ds_kernel(const T* __restrict__ dX0, MapSet ms,
                                                          const float* __restrict__ ns, float* __restrict__ dsg, int B, int nm,
                                                          long sY, long sWs, int C, int Tn, int Tp, int ld, int coff) {
  __shared__ float xt[64][65];
  __shared__ float part[4][64];
  const int j = blockIdx.z / B, b = blockIdx.z % B, c0 = blockIdx.x * 64, tid = threadIdx.x;
  const float* __restrict__ x = ms.x[j];
  {
    const int z = j % 3, m = j / 3;
    dX0 += (size_t)z * sY;
    coff += m * C;
    ns += ((size_t)z * nm + m) * B;
    dsg += (size_t)j * sWs;
  }
  const int cl = tid & 63, tl = tid >> 6;
  float s = 0.f;
  for (int t0 = 0; t0 < Tn; t0 += 64) {
    __syncthreads();                                   // the previous chunk's tile has been consumed
    for (int e = tid; e < 64 * 64; e += 256) {
      const int cc = e / 64, tt = e % 64;
      xt[cc][tt] = (c0 + cc < C && t0 + tt < Tn) ? x[((size_t)b * C + c0 + cc) * Tn + t0 + tt] : 0.f;
    }
    __syncthreads();
    if (c0 + cl < C) {
      const T* dp = dX0 + (size_t)b * Tp * ld + coff + c0 + cl;
#pragma unroll 4
      for (int k = 0; k < 16; ++k) {
        const int tt = tl + 4 * k;
        if (t0 + tt < Tn) s += to_f32<T>(dp[(size_t)(t0 + tt) * ld]) * xt[cl][tt];
      }
    }
  }
  part[tl][cl] = s;
  __syncthreads();
  if (tid < 64 && c0 + tid < C)
    dsg[(size_t)b * C + c0 + tid] = ns[b] * (((part[0][tid] + part[1][tid]) + part[2][tid]) + part[3][tid]);
}

struct TsseG {
  float* cw[3];
  float* cb[3];
  float* fcw; float* fcb; float* w1; float* b1; float* w2; float* b2;
};

__device__ __forceinline__ TsseG tsse_branch_g(TsseG g, long off) {
  for (int i = 0; i < 3; ++i) { g.cw[i] += off; g.cb[i] += off; }
  g.fcw += off; g.fcb += off; g.w1 += off; g.b1 += off; g.w2 += off; g.b2 += off;
  return g;
}

// one workgroup per (sample, map): backprop dsg through sigmoid/fc2/relu/fc1/feature_concate_fc/relu/conv-means; leaves
// da2 / da1 (operands of the fc weight gradients) and this sample's contributions pcb / pcw / pf in the workspace
// batched: blockIdx.y = map j = m*3 + z
__global__ __launch_bounds__(1024) void tsse_bwd_mlp_kernel(MapSet ms, const double* __restrict__ rowsum,
                                                           TsseW w, long sW, int nm, const float* __restrict__ ns_in,
                                                           const float* __restrict__ pre, const float* __restrict__ sq,
                                                           const float* __restrict__ h1, const float* __restrict__ sg,
                                                           float* __restrict__ ws, long sWs, int C, int C2, int T, int la) {
  __shared__ float da2[TSSE_MAXC];
  __shared__ float da1[TSSE_MAXC / 2];
  __shared__ float wsum[4][16];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int B = gridDim.x;
  const int KS = w.ks[0] + w.ks[1] + w.ks[2];
  const float* __restrict__ x = ms.x[blockIdx.y];
  float* __restrict__ wj = ws + (size_t)blockIdx.y * sWs;
  const float* __restrict__ dsg = wj;
  float* __restrict__ da2_ws = wj + (size_t)B * C;
  float* __restrict__ da1_ws = da2_ws + (size_t)B * C;
  float* __restrict__ pcb = da1_ws + (size_t)B * C2 + (size_t)b * 3 * C;
  float* __restrict__ pcw = da1_ws + (size_t)B * C2 + (size_t)B * 3 * C + (size_t)b * C * KS;
  float* __restrict__ pf = da1_ws + (size_t)B * C2 + (size_t)B * 3 * C + (size_t)B * C * KS + (size_t)b * 4;
  {
    const int j = blockIdx.y, z = j % 3, m = j / 3;
    const size_t slot = (size_t)z * nm + m;
    w = tsse_branch(w, (long)z * sW);
    rowsum += (size_t)j * B * C;
    ns_in += slot * B;
    pre += slot * B * C * 3;
    sq += slot * B * C;
    h1 += slot * B * C2;
    sg += slot * B * C;
  }
  const int Tp = T + la;
  const float ns = ns_in[b];
  for (int c = tid; c < C; c += blockDim.x) {
    const float s = sg[(size_t)b * C + c];
    const float d = dsg[(size_t)b * C + c] * s * (1.f - s);
    da2[c] = d;
    da2_ws[(size_t)b * C + c] = d;          // fc2 / fc1 weight gradients: tsse_bwd_outer_kernel (sum over samples)
  }
  __syncthreads();
  for (int j = tid; j < C2; j += blockDim.x) {
    float a = 0.f;
    for (int c = 0; c < C; ++c) a += w.w2[(size_t)c * C2 + j] * da2[c];
    a = h1[(size_t)b * C2 + j] > 0.f ? a : 0.f;
    da1[j] = a;
    da1_ws[(size_t)b * C2 + j] = a;
  }
  __syncthreads();
  float fsum[3] = {0.f, 0.f, 0.f}, bsum = 0.f;
  for (int c = tid; c < C; c += blockDim.x) {
    float a = 0.f;
    for (int j = 0; j < C2; ++j) a += w.w1[(size_t)j * C + c] * da1[j];
    bsum += a;
    const float* xr = x + ((size_t)b * C + c) * T;
    const double tot = rowsum[(size_t)b * C + c];
    int koff = 0;
    for (int i = 0; i < 3; ++i) {
      const float pv = pre[((size_t)b * C + c) * 3 + i];
      fsum[i] += a * fmaxf(pv, 0.f);
      const float dp = pv > 0.f ? a * w.fcw[i] : 0.f;
      pcb[i * C + c] = dp;
      const int ks = w.ks[i];
      const int Lout = Tp - ks + 1;
      for (int k = 0; k < ks; ++k) {
        float v = 0.f;
        if (dp != 0.f) {
          double prf = 0.0, suf = 0.0;
          for (int t = 0; t < k; ++t) prf += (t < T) ? (double)xr[t] : 0.0;
          for (int m = 0; m < ks - 1 - k; ++m) {
            const int t = Tp - 1 - m;
            suf += (t < T && t >= 0) ? (double)xr[t] : 0.0;
          }
          v = dp * ns * (float)((tot - prf - suf) / Lout);
        }
        pcw[(size_t)C * koff + c * ks + k] = v;
      }
      koff += ks;
    }
  }
  // feature_concate_fc: the workgroup's sums over the channels, waves added in index order
  for (int i = 0; i < 4; ++i) {
    const float v = wave_sum(i < 3 ? fsum[i] : bsum);
    if ((tid & 63) == 0) wsum[i][tid >> 6] = v;
  }
  __syncthreads();
  if (tid < 4) {
    float v = 0.f;
    for (int wv = 0; wv < (int)(blockDim.x >> 6); ++wv) v += wsum[tid][wv];
    pf[tid] = v;
  }
}

// every attention gradient of branch blockIdx.y = z as sums over the samples, ONE thread per element, samples in index
// order, the nm maps (noisy / enhanced call) of the branch one after the other: G += (sum_b of map 0), G += (sum_b of map 1)
//   w2[c][j]: sum_b da2[b][c] h1[b][j];  b2[c]: sum_b da2[b][c];  w1[j][c]: sum_b da1[b][j] sq[b][c];  b1[j]: sum_b da1[b][j]
//   cb_i[c] / cw_i[c][k] / fcw[i] / fcb: sum_b of the per-sample contributions left by tsse_bwd_mlp_kernel
__global__ __launch_bounds__(256) void tsse_bwd_outer_kernel(const float* __restrict__ ws, long sWs,
                                                             const float* __restrict__ h1, const float* __restrict__ sq, TsseG g,
                                                             long sW, int nm, int B, int C, int C2, int ks0, int ks1, int ks2,
                                                             int map0 /* per-map entry point: the map's slot is 0 */) {
  const int z = blockIdx.y;
  g = tsse_branch_g(g, (long)z * sW);
  const int KS = ks0 + ks1 + ks2;
  const int n2 = C * C2;
  const long nW = (long)C * KS;
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = 2L * n2 + C + C2 + 3L * C + nW + 4;
  if (e >= total) return;
  float* dst;
  {
    long k = e;
    if (k < n2) dst = g.w2 + k;
    else if ((k -= n2) < n2) dst = g.w1 + k;
    else if ((k -= n2) < C) dst = g.b2 + k;
    else if ((k -= C) < C2) dst = g.b1 + k;
    else if ((k -= C2) < 3L * C) dst = g.cb[k / C] + k % C;
    else if ((k -= 3L * C) < nW) {
      if (k < (long)C * ks0) dst = g.cw[0] + k;
      else if (k < (long)C * (ks0 + ks1)) dst = g.cw[1] + (k - (long)C * ks0);
      else dst = g.cw[2] + (k - (long)C * (ks0 + ks1));
    } else { k -= nW; dst = k < 3 ? g.fcw + k : g.fcb; }
  }
  float acc = *dst;
  for (int m = 0; m < nm; ++m) {
    const int j = map0 >= 0 ? map0 : m * 3 + z;
    const size_t slot = map0 >= 0 ? 0 : (size_t)z * nm + m;
    const float* __restrict__ da2 = ws + (size_t)j * sWs + (size_t)B * C;
    const float* __restrict__ da1 = da2 + (size_t)B * C;
    const float* __restrict__ pcb = da1 + (size_t)B * C2;
    const float* __restrict__ pcw = pcb + (size_t)B * 3 * C;
    const float* __restrict__ pf = pcw + (size_t)B * nW;
    const float* __restrict__ h1m = h1 + slot * B * C2;
    const float* __restrict__ sqm = sq + slot * B * C;
    float a = 0.f;
    long k = e;
    if (k < n2) {
      const int c = (int)(k / C2), jj = (int)(k % C2);
      for (int b = 0; b < B; ++b) a += da2[(size_t)b * C + c] * h1m[(size_t)b * C2 + jj];
    } else if ((k -= n2) < n2) {
      const int jj = (int)(k / C), c = (int)(k % C);
      for (int b = 0; b < B; ++b) a += da1[(size_t)b * C2 + jj] * sqm[(size_t)b * C + c];
    } else if ((k -= n2) < C) {
      for (int b = 0; b < B; ++b) a += da2[(size_t)b * C + k];
    } else if ((k -= C) < C2) {
      for (int b = 0; b < B; ++b) a += da1[(size_t)b * C2 + k];
    } else if ((k -= C2) < 3L * C) {
      for (int b = 0; b < B; ++b) a += pcb[(size_t)b * 3 * C + k];
    } else if ((k -= 3L * C) < nW) {
      for (int b = 0; b < B; ++b) a += pcw[(size_t)b * nW + k];
    } else {
      k -= nW;
      for (int b = 0; b < B; ++b) a += pf[(size_t)b * 4 + k];
    }
    acc += a;
  }
  *dst = acc;
}

}  // namespace

extern "C" {

/* TSSE backward for ALL input maps of a net in four launches (nmaps = 3 * nm maps, j = m*3 + z): dX0 [3][B][Tp][ld] is the
 * gradient of the TCN input (branch stride sY), the saved forward tensors are [3][nm][B][..], parameters / gradients of
 * branch z sit z * sW elements behind the pointers given, ws holds nmaps * B * (2 C + C/2) floats */
int nppc_tsse_bwd_maps(int prec, const void* dX0, long sY, const float* const* maps, int nmaps, const double* rowsum,
                       const float* cw0, const float* cw1, const float* cw2, int ks0, int ks1, int ks2, const float* fcw,
                       const float* w1, const float* w2, long sW, const float* ns, const float* pre, const float* sq,
                       const float* h1, const float* sg, float* ws, float* g_cw0, float* g_cb0, float* g_cw1, float* g_cb1,
                       float* g_cw2, float* g_cb2, float* g_fcw, float* g_fcb, float* g_w1, float* g_b1, float* g_w2, float* g_b2,
                       int B, int C, int T, int look_ahead, int Tp, int ld, void* stream) {
  if (!dX0 || !maps || !rowsum || !ws || B <= 0 || C > TSSE_MAXC || (nmaps != 3 && nmaps != 6)) return NPPC_EBADARG;
  hipStream_t s = (hipStream_t)stream;
  MapSet ms{};
  for (int j = 0; j < nmaps; ++j) { if (!maps[j]) return NPPC_EBADARG; ms.x[j] = maps[j]; }
  const int nm = nmaps / 3, C2 = C / 2, KS = ks0 + ks1 + ks2;
  const long sWs = tsse_ws_stride(B, C, KS);
  dim3 g1(ceil_div(C, 64), 1, B * nmaps);
  if (prec == NPPC_PREC_BF16)
    hipLaunchKernelGGL(tsse_bwd_ds_kernel<bf16_t>, g1, dim3(256), 0, s, (const bf16_t*)dX0, ms, ns, ws, B, nm, sY, sWs, C, T, Tp, ld, 0);
  else
    hipLaunchKernelGGL(tsse_bwd_ds_kernel<float>, g1, dim3(256), 0, s, (const float*)dX0, ms, ns, ws, B, nm, sY, sWs, C, T, Tp, ld, 0);
  TsseW w{{cw0, cw1, cw2}, {nullptr, nullptr, nullptr}, {ks0, ks1, ks2}, fcw, nullptr, w1, nullptr, w2, nullptr};
  TsseG g{{g_cw0, g_cw1, g_cw2}, {g_cb0, g_cb1, g_cb2}, g_fcw, g_fcb, g_w1, g_b1, g_w2, g_b2};
  const int nt = round_up(C, 64) > 1024 ? 1024 : round_up(C, 64);
  hipLaunchKernelGGL(tsse_bwd_mlp_kernel, dim3(B, nmaps), dim3(nt), 0, s, ms, rowsum, w, sW, nm, ns, pre, sq, h1, sg, ws, sWs, C,
                     C2, T, look_ahead);
  const long total = 2L * C * C2 + C + C2 + 3L * C + (long)C * KS + 4;
  hipLaunchKernelGGL(tsse_bwd_outer_kernel, dim3(ceil_div(total, 256), 3), dim3(256), 0, s, ws, sWs, h1, sq, g, sW, nm, B, C, C2,
                     ks0, ks1, ks2, -1);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

/* laplace-norm row sums, TSSE attention and scale + transpose into the TCN input for ALL maps of a net, three launches:
 * rowsum [nmaps][B][C] (fp64), scale / saved tensors [3][nm][B][..], X0 [3][B][Tp][ld] (branch stride sY), map j = m*3 + z
 * lands in branch z at columns [m*C, m*C + C) */
int nppc_tsse_fwd_maps(int prec, const float* const* maps, int nmaps, double* rowsum, const float* cw0, const float* cb0,
                       const float* cw1, const float* cb1, const float* cw2, const float* cb2, int ks0, int ks1, int ks2,
                       const float* fcw, const float* fcb, const float* w1, const float* b1, const float* w2, const float* b2, long sW,
                       float* scale, float* ns, float* pre, float* sq, float* h1, float* sg, void* X0, long sY, int B, int C, int T,
                       int look_ahead, int Tp, int ld, void* stream) {
  if (!maps || !rowsum || !scale || !X0 || B <= 0 || C <= 0 || C > TSSE_MAXC || (nmaps != 3 && nmaps != 6)) return NPPC_EBADARG;
  if (ks0 > T || ks1 > T || ks2 > T) return NPPC_EUNSUPPORTED;
  const int nm = nmaps / 3;
  if (nm * C > ld || T > Tp) return NPPC_EBADARG;
  hipStream_t s = (hipStream_t)stream;
  MapSet ms{};
  for (int j = 0; j < nmaps; ++j) { if (!maps[j]) return NPPC_EBADARG; ms.x[j] = maps[j]; }
  hipLaunchKernelGGL(rowsum_kernel, dim3(ceil_div((long)B * C, 4), nmaps), dim3(256), 0, s, ms, rowsum, (long)B * C, T);
  TsseW w{{cw0, cw1, cw2}, {cb0, cb1, cb2}, {ks0, ks1, ks2}, fcw, fcb, w1, b1, w2, b2};
  const int nt = round_up(C, 64) > 1024 ? 1024 : round_up(C, 64);
  hipLaunchKernelGGL(tsse_fwd_kernel, dim3(B, nmaps), dim3(nt), 0, s, ms, rowsum, w, sW, nm, scale, ns, pre, sq, h1, sg, C, C / 2, T,
                     look_ahead);
  dim3 grid(ceil_div(T, 32), ceil_div(C, 32), B * nmaps);
  if (prec == NPPC_PREC_BF16)
    hipLaunchKernelGGL(scale_transpose_kernel<bf16_t>, grid, dim3(256), 0, s, ms, scale, (bf16_t*)X0, B, nm, sY, C, T, Tp, ld, 0);
  else if (prec == NPPC_PREC_F32)
    hipLaunchKernelGGL(scale_transpose_kernel<float>, grid, dim3(256), 0, s, ms, scale, (float*)X0, B, nm, sY, C, T, Tp, ld, 0);
  else
    return NPPC_EBADARG;
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}


int nppc_tsse_bwd(int prec, const void* dX0, const float* x, const double* rowsum, const float* cw0, const float* cw1,
                  const float* cw2, int ks0, int ks1, int ks2, const float* fcw, const float* w1, const float* w2,
                  const float* ns, const float* pre, const float* sq, const float* h1, const float* sg, float* dsg_ws,
                  float* g_cw0, float* g_cb0, float* g_cw1, float* g_cb1, float* g_cw2, float* g_cb2, float* g_fcw,
                  float* g_fcb, float* g_w1, float* g_b1, float* g_w2, float* g_b2, int B, int C, int T, int look_ahead, int Tp,
                  int ld, int coff, void* stream) {
  if (!dX0 || !x || !rowsum || !dsg_ws || B <= 0 || C > TSSE_MAXC) return NPPC_EBADARG;
  hipStream_t s = (hipStream_t)stream;
  dim3 g1(ceil_div(C, 64), 1, B);
  MapSet ms{};
  ms.x[0] = x;
  const int C2 = C / 2, KS = ks0 + ks1 + ks2;
  const long sWs1 = tsse_ws_stride(B, C, KS);        // dsg_ws: one map's workspace (nppc_tsse_bwd_ws_elems(1, ..))
  if (prec == NPPC_PREC_BF16)
    hipLaunchKernelGGL(tsse_bwd_ds_kernel<bf16_t>, g1, dim3(256), 0, s, (const bf16_t*)dX0, ms, ns, dsg_ws, B, 1, 0L, sWs1, C, T, Tp, ld, coff);
  else
    hipLaunchKernelGGL(tsse_bwd_ds_kernel<float>, g1, dim3(256), 0, s, (const float*)dX0, ms, ns, dsg_ws, B, 1, 0L, sWs1, C, T, Tp, ld, coff);
  TsseW w{{cw0, cw1, cw2}, {nullptr, nullptr, nullptr}, {ks0, ks1, ks2}, fcw, nullptr, w1, nullptr, w2, nullptr};
  TsseG g{{g_cw0, g_cw1, g_cw2}, {g_cb0, g_cb1, g_cb2}, g_fcw, g_fcb, g_w1, g_b1, g_w2, g_b2};
  const int nt = round_up(C, 64) > 1024 ? 1024 : round_up(C, 64);
  hipLaunchKernelGGL(tsse_bwd_mlp_kernel, dim3(B, 1), dim3(nt), 0, s, ms, rowsum, w, 0L, 1, ns, pre, sq, h1, sg, dsg_ws, sWs1,
                     C, C2, T, look_ahead);
  const long total = 2L * C * C2 + C + C2 + 3L * C + (long)C * KS + 4;
  hipLaunchKernelGGL(tsse_bwd_outer_kernel, dim3(ceil_div(total, 256), 1), dim3(256), 0, s, dsg_ws, sWs1, h1, sq, g, 0L, 1, B, C, C2,
                     ks0, ks1, ks2, 0);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

/* floats of workspace nppc_tsse_bwd_maps (nmaps = 3 or 6) / nppc_tsse_bwd (nmaps = 1) need */
int nppc_tsse_bwd_ws_elems(int nmaps, int B, int C, int ks0, int ks1, int ks2, long* elems) {
  if (!elems || nmaps < 1 || B < 1 || C < 1) return NPPC_EBADARG;
  *elems = (long)nmaps * tsse_ws_stride(B, C, ks0 + ks1 + ks2);
  return NPPC_OK;
}

int nppc_rowsum(const float* x, double* sums, long R, int T, void* stream) {
  if (!x || !sums || R <= 0 || T <= 0) return NPPC_EBADARG;
  MapSet ms{};
  ms.x[0] = x;
  hipLaunchKernelGGL(rowsum_kernel, dim3(ceil_div(R, 4), 1), dim3(256), 0, (hipStream_t)stream, ms, sums, R, T);
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
  const int nt = round_up(C, 64) > 1024 ? 1024 : round_up(C, 64);
  MapSet ms{};
  ms.x[0] = x;
  hipLaunchKernelGGL(tsse_fwd_kernel, dim3(B, 1), dim3(nt), 0, (hipStream_t)stream, ms, rowsum, w, 0L, 1, scale, ns, pre, sq, h1, sg,
                     C, C / 2, T, look_ahead);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

int nppc_scale_transpose(int prec, const float* x, const float* scale, void* y, int B, int C, int T, int Tp, int ld,
                         int coff, void* stream) {
  if (!x || !y || B <= 0 || coff + C > ld || T > Tp) return NPPC_EBADARG;
  dim3 grid(ceil_div(T, 32), ceil_div(C, 32), B);
  MapSet ms{};
  ms.x[0] = x;
  if (prec == NPPC_PREC_BF16)
    hipLaunchKernelGGL(scale_transpose_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, ms, scale, (bf16_t*)y, B, 1, 0L, C, T,
                       Tp, ld, coff);
  else if (prec == NPPC_PREC_F32)
    hipLaunchKernelGGL(scale_transpose_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, ms, scale, (float*)y, B, 1, 0L, C, T, Tp,
                       ld, coff);
  else
    return NPPC_EBADARG;
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

}  // extern "C"
