// Training-only helpers of the direction net: transposes / column sums feeding the weight-gradient GEMMs,
// head backward, sub-band staging backward, split-K slab reduction into the parameter gradients, Adam.
// Reference: the autograd of nppc_audio/networks.py:63-163 + torch.optim.Adam (nppc_audio/trainer.py:64-69,102-104).
#include "common.h"
#include "nppc_hip.h"

namespace {

// out[c][r] = in[r][c] (optionally relu'd), batched over blockIdx.z
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T* __restrict__ in, T* __restrict__ out, int rows, int cols,
                                                        long ld_in, long ld_out, long sIn, long sOut, int relu) {
  __shared__ float tile[32][33];
  in += (size_t)blockIdx.z * sIn;
  out += (size_t)blockIdx.z * sOut;
  const int r0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + tx;
    float v = 0.f;
    if (r < rows && c < cols) v = to_f32<T>(in[(size_t)r * ld_in + c]);
    if (relu) v = fmaxf(v, 0.f);
    tile[i][tx] = v;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + tx;
    if (c < cols && r < rows) out[(size_t)c * ld_out + r] = from_f32<T>(tile[tx][i]);
  }
}

// out[c] += sum_r M[r][c], batched over blockIdx.z, in two launches and WITHOUT atomics (round 4: float atomics add up in
// arrival order, so two runs of one step differed in the last bits of these bias gradients): every row block leaves its
// column sums in part[z][row block][cols] (plain stores), colsum_finish_kernel adds the row blocks in a fixed order.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ M, float* __restrict__ part, int rows, int cols,
                                                     long ld, long sM, int rows_per_block) {
  M += (size_t)blockIdx.z * sM;
  part += ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * cols;
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int r0 = blockIdx.y * rows_per_block;
  const int r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  if (c >= cols) return;
  float s = 0.f;
  for (int r = r0; r < r1; ++r) s += to_f32<T>(M[(size_t)r * ld + c]);
  part[c] = s;
}

// one wave per (column, batch entry): lane l adds row blocks l, l + 64, ... in order, then a fixed xor-shuffle tree
__global__ __launch_bounds__(256) void colsum_finish_kernel(const float* __restrict__ part, float* __restrict__ out, int nblk, int cols,
                                                            long sOut) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63, z = blockIdx.y;
  if (c >= cols) return;
  const float* p = part + (size_t)z * nblk * cols + c;
  float s = 0.f;
  for (int k = lane; k < nblk; k += 64) s += p[(size_t)k * cols];
  s = wave_sum(s);
  if (lane == 0) out[(size_t)z * sOut + c] += s;
}

// the same for 16-byte aligned rows padded to a multiple of 8 columns (pad columns are read, never written): a workgroup covers 256 columns (32 lanes x 8 channels, one 16-byte
// load per lane and row) x CS_ROWS rows (8 row lanes, CS_ROWS / 8 independent loads in flight per thread)
constexpr int CS_ROWS = 64;
template <typename T>
__global__ __launch_bounds__(256) void colsum8_kernel(const T* __restrict__ M, float* __restrict__ pout, int rows, int cols,
                                                      long ld, long sM, int rows_per_block) {
  __shared__ float part[8][256 + 8];
  M += (size_t)blockIdx.z * sM;
  pout += ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * cols;
  const int cl = (threadIdx.x & 31) * 8, c = blockIdx.x * 256 + cl, rl = threadIdx.x >> 5;
  float s[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) s[i] = 0.f;
  if (c < cols) {
    // rows_per_block (a multiple of CS_ROWS): tall matrices use few large workgroups (one partial row per workgroup)
    for (int rb = blockIdx.y * rows_per_block; rb < (blockIdx.y + 1) * rows_per_block && rb < rows; rb += CS_ROWS) {
#pragma unroll
      for (int j = 0; j < CS_ROWS / 8; ++j) {
        const int r = rb + rl + 8 * j;
        if (r < rows) {
          float v[8];
          load8<T>(M + (size_t)r * ld + c, v);
#pragma unroll
          for (int i = 0; i < 8; ++i) s[i] += v[i];
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) part[rl][cl + i] = s[i];
  __syncthreads();
  const int cc = blockIdx.x * 256 + threadIdx.x;
  if (cc < cols) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += part[k][threadIdx.x];
    pout[cc] = t;
  }
}

// ---------------------------------------------------------------- head backward
// dh2[t][n][u] = sum_o dY[t][n][o] * Wh[o][u],   dY[t][n][o] = dout[bo][o][fo][t-la] (0 for t < la)
// one wave = 16 rows; A fragment gathered from dout (K = O <= 32), B = WhT packed [Hd][32].
template <typename T>
__global__ __launch_bounds__(256) void head_bwd_dh_kernel(const float* __restrict__ dout, const T* __restrict__ whT,
                                                          T* __restrict__ dh2, long Nseq, int Tn, int la, int Hd, int O,
                                                          int Fo) {
  typedef typename Frag<T>::type frag;
  const int lane = threadIdx.x & 63, n = lane & 15, q = lane >> 4;
  const long tiles_per_t = (Nseq + 15) / 16;
  const long tile = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tile >= tiles_per_t * Tn) return;
  const int t = (int)(tile / tiles_per_t);
  const long n0 = (tile % tiles_per_t) * 16;
  const int To = Tn - la;
  frag a;
  {
    const long nn = n0 + n;
    const bool ok = nn < Nseq && t >= la;
    const long bo = ok ? nn / Fo : 0, fo = ok ? nn % Fo : 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int o = 8 * q + j;
      float v = 0.f;
      if (ok && o < O) v = dout[((bo * O + o) * Fo + fo) * To + (t - la)];
      if constexpr (sizeof(T) == 2) a[j] = (__bf16)v; else a[j] = v;
    }
  }
  for (int c0 = 0; c0 < Hd; c0 += 16) {
    const int col = c0 + n < Hd ? c0 + n : Hd - 1;
    const frag b = load_frag<T>(whT + (size_t)col * 32 + 8 * q);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = mma16(a, b, acc);
    if (c0 + n < Hd) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const long nn = n0 + 4 * q + j;
        if (nn < Nseq) dh2[((size_t)t * Nseq + nn) * Hd + c0 + n] = from_f32<T>(acc[j]);
      }
    }
  }
}

// dWh[o][u] += sum_rows dY[row][o] * h2[row][u];  dbh[o] += sum_rows dY[row][o]
// block = (t, chunk of 64 rows); thread u (blockDim = Hd rounded up to 64)
constexpr int HB_ROWS = 64;
template <typename T, int OP>
__global__ void head_bwd_w_kernel(const float* __restrict__ dout, const T* __restrict__ h2, float* __restrict__ dWh,
                                  float* __restrict__ dbh, long Nseq, int Tn, int la, int Hd, int O, int Fo) {
  // one workgroup = frame t and every gridDim.x-th chunk of 64 sequences: the O*Hd partial sums stay in registers over
  // all its chunks and reach memory with ONE atomic each (one workgroup per chunk meant 16k workgroups x 3840 atomics on
  // the same 3840 addresses).  OP = O rounded up to 8/16/32: the dY row is read from LDS as broadcast float4s, so a row
  // costs OP/4 LDS instructions + OP FMAs per thread (with OP fixed at 32 the LDS issue rate bounded the kernel).
  // (Gathering 8 frames per staging round -- 32-byte runs of dout instead of single floats -- needed 32 KB of LDS and
  // halved the workgroups: 0.62 -> 1.1 ms, the 2-byte h2 loads lost their latency cover.)
  __shared__ __attribute__((aligned(16))) float dy[HB_ROWS][OP];
  const int t = la + blockIdx.y;
  const int To = Tn - la;
  const int u = threadIdx.x;
  float acc[OP];
#pragma unroll
  for (int o = 0; o < OP; ++o) acc[o] = 0.f;
  float bsum = 0.f;
  for (long n0 = (long)blockIdx.x * HB_ROWS; n0 < Nseq; n0 += (long)gridDim.x * HB_ROWS) {
    __syncthreads();
    for (int e = threadIdx.x; e < HB_ROWS * OP; e += blockDim.x) {
      const int r = e / OP, o = e % OP;
      const long nn = n0 + r;
      float v = 0.f;
      if (nn < Nseq && o < O) {
        const long bo = nn / Fo, fo = nn % Fo;
        v = dout[((bo * O + o) * Fo + fo) * To + (t - la)];
      }
      dy[r][o] = v;
    }
    __syncthreads();
    if (u < Hd) {
      const int nr = Nseq - n0 < HB_ROWS ? (int)(Nseq - n0) : HB_ROWS;
      const T* hp = h2 + ((size_t)t * Nseq + n0) * Hd + u;
#pragma unroll 4
      for (int r = 0; r < nr; ++r) {
        const float hv = to_f32<T>(hp[(size_t)r * Hd]);
#pragma unroll
        for (int o4 = 0; o4 < OP / 4; ++o4) {
          const float4 d = *reinterpret_cast<const float4*>(&dy[r][4 * o4]);
          acc[4 * o4] += d.x * hv;
          acc[4 * o4 + 1] += d.y * hv;
          acc[4 * o4 + 2] += d.z * hv;
          acc[4 * o4 + 3] += d.w * hv;
        }
      }
    }
    if (threadIdx.x < OP && threadIdx.x < O)
      for (int r = 0; r < HB_ROWS; ++r) bsum += dy[r][threadIdx.x];
  }
  if (u < Hd) {
#pragma unroll
    for (int o = 0; o < OP; ++o)
      if (o < O) atomicAdd(dWh + (size_t)o * Hd + u, acc[o]);
  }
  if (threadIdx.x < OP && threadIdx.x < O) atomicAdd(dbh + threadIdx.x, bsum);
}

template <typename T>
static void launch_head_bwd_w(dim3 grid, int bw, hipStream_t s, const float* dout, const T* h2, float* dWh, float* dbh, long Nseq,
                              int Tn, int la, int Hd, int O, int Fo) {
  if (O <= 8) hipLaunchKernelGGL((head_bwd_w_kernel<T, 8>), grid, dim3(bw), 0, s, dout, h2, dWh, dbh, Nseq, Tn, la, Hd, O, Fo);
  else if (O <= 16) hipLaunchKernelGGL((head_bwd_w_kernel<T, 16>), grid, dim3(bw), 0, s, dout, h2, dWh, dbh, Nseq, Tn, la, Hd, O, Fo);
  else hipLaunchKernelGGL((head_bwd_w_kernel<T, 32>), grid, dim3(bw), 0, s, dout, h2, dWh, dbh, Nseq, Tn, la, Hd, O, Fo);
}

// dyt[t][n][0..15] (bf16) = dout[bo][o][fo][t - la] for o < O, t >= la, else 0: the dY rows in the order the fused head
// backward of lstm2_coop_bwd2_kernel<true> takes them (32 bytes per (t, n)); threads run along t, the contiguous axis of dout
__global__ __launch_bounds__(256) void head_dy_gather_kernel(const float* __restrict__ dout, bf16_t* __restrict__ dyt, long Nseq,
                                                             int Tn, int la, int O, int Fo) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= Nseq * Tn) return;
  const long nn = i / Tn;
  const int t = (int)(i % Tn), To = Tn - la;
  const long bo = nn / Fo, fo = nn % Fo;
  unsigned w[8];
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    float v0 = 0.f, v1 = 0.f;
    if (t >= la) {
      if (2 * p < O) v0 = dout[((bo * O + 2 * p) * Fo + fo) * To + (t - la)];
      if (2 * p + 1 < O) v1 = dout[((bo * O + 2 * p + 1) * Fo + fo) * To + (t - la)];
    }
    w[p] = (unsigned)f2bf(v0) | ((unsigned)f2bf(v1) << 16);
  }
  uint4* dst = reinterpret_cast<uint4*>(dyt + ((size_t)t * Nseq + nn) * 16);
  dst[0] = uint4{w[0], w[1], w[2], w[3]};
  dst[1] = uint4{w[4], w[5], w[6], w[7]};
}

// ---------------------------------------------------------------- sub-band staging backward
// D[bo] = sum_{t, fo, j < nfeat} dx[t][n][j] * x[t][n][j]      (x = normalised LSTM input as staged)
// one workgroup per (frame, sample): rows of KX elements, of which the first nfeat count; 8 elements (one 16-byte load of
// bf16) per thread and trip, only the ceil(nfeat / 8) chunks of a row that hold features are touched (scalar 2-byte loads
// with a modulo per element ran at 0.65 TB/s: 417 us on the step's critical path right behind the LSTM backward).
// The same pass drops the three values of a row that the scatter pass needs -- dx[t][n][W + m], the gradients of the three
// full-band features -- at their final place dpre[m][b][t][f] (raw): the scatter pass, which used to gather them with 2-byte
// loads that touched every 128-byte row of dx a second time (133 MB for 6 MB), became an in-place pass over dpre.
template <typename T>
__global__ __launch_bounds__(256) void sb_bwd_reduce_kernel(const T* __restrict__ dx, const T* __restrict__ x,
                                                            double* __restrict__ D, T* __restrict__ dpre, int B, int Tp, int ldF,
                                                            long strideFb, int W, int G, int Fo, int KX, int nfeat, long Nseq) {
  __shared__ double red[4];
  const int t = blockIdx.x, bo = blockIdx.y;
  // sample b and drop-band group g of sequence block bo (the inverse of the scatter pass's b -> bo)
  int g = 0, b = bo;
  if (G > 1) {
    int start = 0;
    for (g = 0; g < G; ++g) {
      const int cnt = (B - g + G - 1) / G;
      if (bo < start + cnt) break;
      start += cnt;
    }
    b = (bo - start) * G + g;
  }
  const size_t base = ((size_t)t * Nseq + (size_t)bo * Fo) * KX;
  const size_t orow = ((size_t)b * Tp + t) * ldF;
  const int cpr = (nfeat + 7) / 8;                         // chunks per row that hold features (KX % 8 == 0)
  float s = 0.f;
  for (int e = threadIdx.x; e < Fo * cpr; e += 256) {
    const int r = e / cpr, c = e % cpr;
    float a[8], bb[8];
    load8<T>(dx + base + (size_t)r * KX + 8 * c, a);
    load8<T>(x + base + (size_t)r * KX + 8 * c, bb);
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (8 * c + i < nfeat) s += a[i] * bb[i];
#pragma unroll
    for (int m = 0; m < 3; ++m)
      if (c == (W + m) / 8) {
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (i == (W + m) % 8) v = a[i];
        dpre[(size_t)m * strideFb + orow + (size_t)r * G + g] = from_f32<T>(v);
      }
  }
  const double d = wave_sum((double)s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = d;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(D + bo, red[0] + red[1] + red[2] + red[3]);
}

// in place: dpre[m][b][t][f] = (fb[m][b][t][f] > 0) * ( sc_b * (kept ? raw : 0) - sc_b * D[bo] / Nn ),  raw = dx[t][n][W+m] as
// left by the reduce pass at the kept bins (the bins of other drop-band groups and the bins past Fo * G hold nothing: not read)
template <typename T>
__global__ __launch_bounds__(256) void sb_bwd_scatter_kernel(const T* __restrict__ fb, const float* __restrict__ scale,
                                                             const double* __restrict__ D, T* __restrict__ dpre, int B, int F,
                                                             int Tp, int ldF, long strideFb, int G, int Fo, double Nn) {
  const int t = blockIdx.x, b = blockIdx.y;
  int g = 0, bo = b;
  if (G > 1) {
    g = b % G;
    int start = 0;
    for (int gg = 0; gg < g; ++gg) start += (B - gg + G - 1) / G;
    bo = start + b / G;
  }
  const float sc = scale[b];
  const float mterm = (float)((double)sc * D[bo] / Nn);
  const size_t row = (size_t)b * Tp + t;
  for (int e = threadIdx.x; e < 3 * F; e += 256) {
    const int m = e / F, f = e % F;
    const size_t o = (size_t)m * strideFb + row * ldF + f;
    const bool kept = G > 1 ? ((f % G) == g && f < Fo * G) : true;
    const float gv = kept ? to_f32<T>(dpre[o]) : 0.f;
    const float v = to_f32<T>(fb[o]) > 0.f ? sc * gv - mterm : 0.f;
    dpre[o] = from_f32<T>(v);
  }
}

// ---------------------------------------------------------------- slab reduction into parameter gradients
// dst[map(r)][c] (+)= sum_s slabs[s][r][col0 + c],  r < rows, c < ncols
// permH > 0: LSTM gate-row un-permutation, packed row r = u*4 + g' (i,g,f,o) -> torch row {0,2,1,3}[g']*permH + u
__global__ void reduce_slabs_kernel(const float* __restrict__ slabs, int S, long slab_stride, long ld, float* __restrict__ dst,
                                    long dst_ld, int rows, int col0, int ncols, int permH, int accumulate, long sSlab,
                                    long sDst) {
  slabs += (size_t)blockIdx.z * sSlab;
  dst += (size_t)blockIdx.z * sDst;
  const long total = (long)rows * ncols;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int r = (int)(e / ncols), c = (int)(e % ncols);
    float s = 0.f;
    for (int k = 0; k < S; ++k) s += slabs[(size_t)k * slab_stride + (size_t)r * ld + col0 + c];
    int rr = r;
    if (permH > 0) {
      const int u = r >> 2, gp = r & 3;
      const int tg = gp == 0 ? 0 : (gp == 1 ? 2 : (gp == 2 ? 1 : 3));
      rr = tg * permH + u;
    }
    float* d = dst + (size_t)rr * dst_ld + c;
    *d = accumulate ? *d + s : s;
  }
}

// dst[b][r][c] = sum_s slabs[b][s][c][r]   (transposed reduction: the slabs hold the gradient's transpose)
__global__ void reduce_slabs_t_kernel(const float* __restrict__ slabs, int S, long slab_stride, long ld, float* __restrict__ dst,
                                      long dst_ld, int rows, int ncols, long sSlab, long sDst) {
  slabs += (size_t)blockIdx.z * sSlab;
  dst += (size_t)blockIdx.z * sDst;
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 256 threads: 8 rows of 32 per pass
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + tx;                          // slab element [c][r]
    float a = 0.f;
    if (c < ncols && r < rows)
      for (int s = 0; s < S; ++s) a += slabs[(size_t)s * slab_stride + (size_t)c * ld + r];
    tile[i][tx] = a;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + tx;
    if (r < rows && c < ncols) dst[(size_t)r * dst_ld + c] = tile[tx][i];
  }
}

// ---------------------------------------------------------------- Adam (torch.optim.Adam semantics, no amsgrad)
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            long n, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt,
                            float gscale, const double* __restrict__ sumsq, float max_norm,
                            const unsigned* const* __restrict__ guards = nullptr, int n_guards = 0, float* poison = nullptr) {
  // guards: sticky hand-off time-out counters of the cooperative LSTM kernels (csrc/lstm_coop.hip, lstm_ws.hip).  A
  // non-zero one means some kernel of this step gave up on a partner and carried on with wrong numbers: the update is
  // SKIPPED (parameters and moments stay bit-identical) and the step's objective is poisoned with NaN; the host raises
  // at its next check.  Every workgroup reads the same few words: they cannot change while this kernel runs.
  for (int i = 0; i < n_guards; ++i)
    if (*guards[i] != 0u) {
      if (poison && blockIdx.x == 0 && threadIdx.x == 0) *poison = __builtin_nanf("");
      return;
    }
  if (sumsq) {   // torch.nn.utils.clip_grad_norm_: coefficient max_norm / (||g|| + 1e-6), clamped to 1
    const float total = (float)sqrt(sumsq[0]) * gscale;
    const float coef = max_norm / (total + 1e-6f);
    gscale *= coef < 1.f ? coef : 1.f;
  }
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float gi = g[i] * gscale;
    const float pi = p[i];
    if (wd != 0.f) gi += wd * pi;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = pi - (lr / bc1) * (mi / denom);
  }
}

}  // namespace

extern "C" {

int nppc_transpose(int prec, const void* in, void* out, int rows, int cols, long ld_in, long ld_out, long sIn, long sOut,
                   int relu, int batch, void* stream) {
  if (!in || !out || rows <= 0 || cols <= 0 || batch <= 0) return NPPC_EBADARG;
  dim3 grid(ceil_div(rows, 32), ceil_div(cols, 32), batch);
  hipStream_t s = (hipStream_t)stream;
  if (prec == NPPC_PREC_BF16)
    hipLaunchKernelGGL(transpose_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)in, (bf16_t*)out, rows, cols, ld_in,
                       ld_out, sIn, sOut, relu);
  else
    hipLaunchKernelGGL(transpose_kernel<float>, grid, dim3(256), 0, s, (const float*)in, (float*)out, rows, cols, ld_in, ld_out,
                       sIn, sOut, relu);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

// row blocking of nppc_colsum: (aligned 16-byte path?, rows per block, row blocks)
static void colsum_plan(int prec, const void* M, int rows, int cols, long ld, long sM, bool& vec8, int& rpb, int& nblk) {
  const size_t esz = prec == NPPC_PREC_BF16 ? 2 : 4;
  vec8 = ld >= (long)round_up(cols, 8) && ld % 8 == 0 && sM % 8 == 0 && ((uintptr_t)M) % 16 == 0 && (ld * esz) % 16 == 0;
  if (vec8) {
    const int colblk = ceil_div(cols, 256);
    rpb = CS_ROWS;                                         // aim at <= ~1024 workgroups per batch entry
    while ((long)ceil_div(rows, rpb) * colblk > 1024) rpb *= 2;
  } else {
    rpb = 128;
    while (ceil_div(rows, rpb) > 4096) rpb *= 2;
  }
  nblk = ceil_div(rows, rpb);
}

/* floats of scratch nppc_colsum needs (an upper bound that does not depend on the alignment of M) */
int nppc_colsum_scratch_elems(int rows, int cols, int batch, long* n) {
  if (!n || rows <= 0 || cols <= 0 || batch <= 0) return NPPC_EBADARG;
  int nb8 = CS_ROWS, nb1 = 128;
  while ((long)ceil_div(rows, nb8) * ceil_div(cols, 256) > 1024) nb8 *= 2;
  while (ceil_div(rows, nb1) > 4096) nb1 *= 2;
  const long a = ceil_div(rows, nb8), b = ceil_div(rows, nb1);
  *n = (long)batch * (a > b ? a : b) * cols;
  return NPPC_OK;
}

int nppc_colsum(int prec, const void* M, float* out, int rows, int cols, long ld, long sM, long sOut, int batch,
                float* scratch, long scratch_elems, void* stream) {
  if (!M || !out || !scratch || rows <= 0 || cols <= 0 || batch <= 0) return NPPC_EBADARG;
  hipStream_t s = (hipStream_t)stream;
  bool vec8; int rpb, nblk;
  colsum_plan(prec, M, rows, cols, ld, sM, vec8, rpb, nblk);
  if (scratch_elems < (long)batch * nblk * cols) return NPPC_EBADARG;
  dim3 grid(ceil_div(cols, 256), nblk, batch);
  if (vec8) {
    if (prec == NPPC_PREC_BF16)
      hipLaunchKernelGGL(colsum8_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)M, scratch, rows, cols, ld, sM, rpb);
    else
      hipLaunchKernelGGL(colsum8_kernel<float>, grid, dim3(256), 0, s, (const float*)M, scratch, rows, cols, ld, sM, rpb);
  } else {
    if (prec == NPPC_PREC_BF16)
      hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)M, scratch, rows, cols, ld, sM, rpb);
    else
      hipLaunchKernelGGL(colsum_kernel<float>, grid, dim3(256), 0, s, (const float*)M, scratch, rows, cols, ld, sM, rpb);
  }
  hipLaunchKernelGGL(colsum_finish_kernel, dim3(ceil_div(cols, 4), batch), dim3(256), 0, s, scratch, out, nblk, cols, sOut);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

int nppc_sb_head_bwd(int prec, const float* dout, const void* whT, const void* h2, void* dh2, float* dWh, float* dbh,
                     long Nseq, int Tn, int la, int Hd, int O, int Fo, void* stream) {
  if (!dout || !whT || !h2 || !dh2 || !dWh || !dbh || O > 32 || O < 1 || Nseq % Fo) return NPPC_EBADARG;
  hipStream_t s = (hipStream_t)stream;
  const long tiles = ((Nseq + 15) / 16) * Tn;
  const int chunks = ceil_div(Nseq, HB_ROWS);
  dim3 gw(chunks < 4 ? chunks : 4, Tn - la);
  const int bw = round_up(Hd, 64);
  if (prec == NPPC_PREC_BF16) {
    hipLaunchKernelGGL(head_bwd_dh_kernel<bf16_t>, dim3(ceil_div(tiles, 4)), dim3(256), 0, s, dout, (const bf16_t*)whT,
                       (bf16_t*)dh2, Nseq, Tn, la, Hd, O, Fo);
    launch_head_bwd_w<bf16_t>(gw, bw, s, dout, (const bf16_t*)h2, dWh, dbh, Nseq, Tn, la, Hd, O, Fo);
  } else {
    hipLaunchKernelGGL(head_bwd_dh_kernel<float>, dim3(ceil_div(tiles, 4)), dim3(256), 0, s, dout, (const float*)whT,
                       (float*)dh2, Nseq, Tn, la, Hd, O, Fo);
    launch_head_bwd_w<float>(gw, bw, s, dout, (const float*)h2, dWh, dbh, Nseq, Tn, la, Hd, O, Fo);
  }
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

int nppc_head_dy_gather(const float* dout, void* dyt, long Nseq, int Tn, int la, int O, int Fo, void* stream) {
  if (!dout || !dyt || Nseq <= 0 || Tn <= la || O < 1 || O > 16 || Nseq % Fo) return NPPC_EBADARG;
  hipLaunchKernelGGL(head_dy_gather_kernel, dim3(ceil_div(Nseq * Tn, 256L)), dim3(256), 0, (hipStream_t)stream, dout,
                     (bf16_t*)dyt, Nseq, Tn, la, O, Fo);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

int nppc_sb_head_bwd_w(int prec, const float* dout, const void* h2, float* dWh, float* dbh, long Nseq, int Tn, int la, int Hd,
                       int O, int Fo, void* stream) {
  if (!dout || !h2 || !dWh || !dbh || O > 32 || O < 1 || Nseq % Fo) return NPPC_EBADARG;
  hipStream_t s = (hipStream_t)stream;
  const int chunks = ceil_div(Nseq, HB_ROWS);
  dim3 gw(chunks < 4 ? chunks : 4, Tn - la);
  const int bw = round_up(Hd, 64);
  if (prec == NPPC_PREC_BF16) launch_head_bwd_w<bf16_t>(gw, bw, s, dout, (const bf16_t*)h2, dWh, dbh, Nseq, Tn, la, Hd, O, Fo);
  else launch_head_bwd_w<float>(gw, bw, s, dout, (const float*)h2, dWh, dbh, Nseq, Tn, la, Hd, O, Fo);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

int nppc_subband_stage_bwd(int prec, const void* dx, const void* x, const void* fb, const float* scale, double* D,
                           void* dpre, int B, int F, int Tp, int Tv, int ldF, long strideFb, int nb, int G, int KX,
                           void* stream) {
  if (!dx || !x || !fb || !scale || !D || !dpre || B <= 0 || KX % 8) return NPPC_EBADARG;
  const int Geff = B > 1 ? G : 1;
  const int Fo = Geff <= 1 ? F : (F - F % Geff) / Geff;
  const long Nseq = (long)B * Fo;
  const int nfeat = 2 * nb + 4;
  const double Nn = (double)F * nfeat * Tv;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(D, 0, sizeof(double) * B, s) != hipSuccess) return NPPC_ELAUNCH;
  dim3 grid(Tv, B);
  const int W = 2 * nb + 1;
  if (W + 2 >= nfeat) return NPPC_EBADARG;
  if (prec == NPPC_PREC_BF16) {
    hipLaunchKernelGGL(sb_bwd_reduce_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)dx, (const bf16_t*)x, D, (bf16_t*)dpre, B,
                       Tp, ldF, strideFb, W, Geff, Fo, KX, nfeat, Nseq);
    hipLaunchKernelGGL(sb_bwd_scatter_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)fb, scale, D, (bf16_t*)dpre, B, F, Tp,
                       ldF, strideFb, Geff, Fo, Nn);
  } else {
    hipLaunchKernelGGL(sb_bwd_reduce_kernel<float>, grid, dim3(256), 0, s, (const float*)dx, (const float*)x, D, (float*)dpre, B, Tp,
                       ldF, strideFb, W, Geff, Fo, KX, nfeat, Nseq);
    hipLaunchKernelGGL(sb_bwd_scatter_kernel<float>, grid, dim3(256), 0, s, (const float*)fb, scale, D, (float*)dpre, B, F, Tp, ldF,
                       strideFb, Geff, Fo, Nn);
  }
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

int nppc_reduce_slabs(const float* slabs, int S, long slab_stride, long ld, float* dst, long dst_ld, int rows, int col0,
                      int ncols, int permH, int accumulate, long sSlab, long sDst, int batch, void* stream) {
  if (!slabs || !dst || S < 1 || rows <= 0 || ncols <= 0) return NPPC_EBADARG;
  const long total = (long)rows * ncols;
  const int gx = (int)((total + 255) / 256 > 1024 ? 1024 : (total + 255) / 256);
  hipLaunchKernelGGL(reduce_slabs_kernel, dim3(gx, 1, batch), dim3(256), 0, (hipStream_t)stream, slabs, S, slab_stride, ld,
                     dst, dst_ld, rows, col0, ncols, permH, accumulate, sSlab, sDst);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

int nppc_reduce_slabs_t(const float* slabs, int S, long slab_stride, long ld, float* dst, long dst_ld, int rows, int ncols,
                        long sSlab, long sDst, int batch, void* stream) {
  if (!slabs || !dst || S < 1 || rows <= 0 || ncols <= 0 || batch < 1) return NPPC_EBADARG;
  hipLaunchKernelGGL(reduce_slabs_t_kernel, dim3(ceil_div(ncols, 32), ceil_div(rows, 32), batch), dim3(256), 0,
                     (hipStream_t)stream, slabs, S, slab_stride, ld, dst, dst_ld, rows, ncols, sSlab, sDst);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

int nppc_adam_step(float* p, const float* g, float* m, float* v, long n, double lr, double b1, double b2, double eps,
                   double wd, int step, double gscale, void* stream) {
  if (!p || !g || !m || !v || n <= 0 || step < 1) return NPPC_EBADARG;
  const double bc1 = 1.0 - pow(b1, (double)step);          // host fp64, like torch's python-side bias corrections
  const double bc2s = sqrt(1.0 - pow(b2, (double)step));
  const int grid = (int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
  hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, (float)lr, (float)b1, (float)b2,
                     (float)eps, (float)wd, (float)bc1, (float)bc2s, (float)gscale, (const double*)nullptr, 0.f);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

// nppc_adam_step with the update guarded by device counters (guards: device array of n_guards device pointers to u32; the
// update is skipped and *poison (nullable) set to NaN when any of them is non-zero): a hand-off time-out of a cooperative LSTM
// kernel must not reach the weights in the steps between two host checks
int nppc_adam_step_guarded(float* p, const float* g, float* m, float* v, long n, double lr, double b1, double b2, double eps,
                           double wd, int step, double gscale, const void* guards, int n_guards, float* poison, void* stream) {
  if (!p || !g || !m || !v || n <= 0 || step < 1 || n_guards < 0 || (n_guards > 0 && !guards)) return NPPC_EBADARG;
  const double bc1 = 1.0 - pow(b1, (double)step);
  const double bc2s = sqrt(1.0 - pow(b2, (double)step));
  const int grid = (int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
  hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, (float)lr, (float)b1, (float)b2,
                     (float)eps, (float)wd, (float)bc1, (float)bc2s, (float)gscale, (const double*)nullptr, 0.f,
                     (const unsigned* const*)guards, n_guards, poison);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

// Adam preceded by clip_grad_norm_(max_norm): `sumsq` is a device scalar holding sum(g^2) of the UNSCALED gradient
// (nppc_sumsq), so the clip needs no host round trip (inpainting/trainer/nppc_trainer.py:149-154).
int nppc_adam_step_clip(float* p, const float* g, float* m, float* v, long n, double lr, double b1, double b2, double eps,
                        double wd, int step, double gscale, const double* sumsq, double max_norm, void* stream) {
  if (!p || !g || !m || !v || !sumsq || n <= 0 || step < 1 || !(max_norm > 0)) return NPPC_EBADARG;
  const double bc1 = 1.0 - pow(b1, (double)step);
  const double bc2s = sqrt(1.0 - pow(b2, (double)step));
  const int grid = (int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
  hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, (float)lr, (float)b1, (float)b2,
                     (float)eps, (float)wd, (float)bc1, (float)bc2s, (float)gscale, sumsq, (float)max_norm);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

}  // extern "C"
