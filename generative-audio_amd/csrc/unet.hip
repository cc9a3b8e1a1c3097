// U-Net of the inpainting sibling.
// Reference: nppc_audio/inpainting/networks/unet.py:247-313 (UNet, RestorationWrapper), tmp_utils.py:8-99
// (double_conv = [conv3x3 pad 1, BatchNorm2d, LeakyReLU(0.2)] x 2, down = MaxPool2d(2) + double_conv,
// up = bilinear x2 (align_corners) + pad + cat([skip, up]) + double_conv, outconv = conv1x1).
//
// Layout: every activation is a haloed NHWC matrix  X[(b*(H+2) + y)*(W+2) + x][ld]  (channels contiguous, one
// zero pixel all round each image, zero guard rows before and after the buffer).  A 3x3 convolution is then nine
// row-shifted GEMMs accumulated in registers (implicit GEMM, no im2col):
//   raw[p][co] = bias[co] + sum_tap sum_ci X[p + off(tap)][ci] * W[co][tap][ci],   off = dy*(W+2) + dx
// computed for every haloed row p and stored for interior pixels only, so halos stay zero for the next layer and
// for the batch statistics.  The transposed convolution (input gradient) is the same kernel on flipped/transposed
// packed weights; the weight gradient is nine row-shifted TN GEMMs (tcn.hip) or the exact-f32 kernel below.
// The concatenations are channel slices of one buffer (producers write with a channel offset).
#include <algorithm>
#include "common.h"
#include "nppc_hip.h"

namespace {

// 32-bit arithmetic: the host entry points reject P >= 2^31 (64-bit integer division has no hardware support)
__device__ __forceinline__ bool interior(long p, long P, int H, int W) {
  if (p >= P) return false;
  const unsigned Wp = W + 2, Hp = H + 2, pu = (unsigned)p;
  const unsigned row = pu / Wp;
  const unsigned x = pu - row * Wp;
  const unsigned y = row % Hp;
  return x >= 1 && x <= (unsigned)W && y >= 1 && y <= (unsigned)H;
}

// ------------------------------------------------------------------------------------------------ convolution
struct ConvArgs {
  const void* A; long lda;            // haloed NHWC input, pointer at pixel row 0 (guard rows exist before it)
  const void* Wp;                     // packed weights [Np][ntap*Cin], K contiguous
  void* C; long ldc;                  // haloed NHWC output; interior rows, columns < Cout are written
  const float* bias;                  // [Cout] or null
  const float* scale;                 // optional folded BatchNorm (eval mode): v = leaky(v * scale[c] + shift[c])
  const float* shift;
  float slope;
  long P;                             // haloed pixel rows B*(H+2)*(W+2)
  int H, W, Cin, Cout, ntap;          // Cin % 32 == 0
  // optional (conv_tiled_kernel): per-tile column sums of the STORED output, [tile = row / 128][2: sum, sum of squares][ldp]
  // fp32 -- the batch statistics of the train-mode BatchNorm that follows, without a pass of its own over the tensor
  float* stat_part; int ldp;
};

// row offset of tap t: (dy, dx) = (t/3 - 1, t%3 - 1) for a 3x3 kernel, 0 for 1x1 (arithmetic on a uniform value:
// a dynamically indexed array in the by-value argument struct would be copied to scratch / LDS)
__device__ __forceinline__ int tap_offset(const ConvArgs& g, int t) {
  return g.ntap == 9 ? (t / 3 - 1) * (g.W + 2) + (t % 3 - 1) : 0;
}

// 256 threads = 4 waves stacked over rows; wave tile 32 x 64, block tile 128 rows x 64 output channels.
template <typename T>
__global__ __launch_bounds__(256) void conv_kernel(ConvArgs g) {
  typedef typename Frag<T>::type frag;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, q = lane >> 4;
  const long r0 = (long)blockIdx.x * 128 + wave * 32;
  const int c0 = blockIdx.y * 64;
  const int kc = g.Cin / 32, nk = g.ntap * kc;
  const long ldw = (long)g.ntap * g.Cin;
  const T* ap = reinterpret_cast<const T*>(g.A) + (r0 + n) * g.lda + 8 * q;
  const T* bp = reinterpret_cast<const T*>(g.Wp) + (long)(c0 + n) * ldw + 8 * q;
  f32x4 acc[2][4];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};
  frag a0[2], b0[4], a1[2], b1[4];
  int tap = 0, cc = 0;   // position of the NEXT k-step to load
  auto ld = [&](frag(&a)[2], frag(&b)[4], int kk) {
    const long aoff = (long)tap_offset(g, tap) * g.lda + 32 * cc;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) a[mi] = load_frag<T>(ap + (long)16 * mi * g.lda + aoff);
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) b[ni] = load_frag<T>(bp + (long)16 * ni * ldw + 32 * kk);
    if (++cc == kc) { cc = 0; ++tap; }
  };
  auto mm = [&](frag(&a)[2], frag(&b)[4]) {
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
  // epilogue: element (row r0 + 16 mi + 4 q + j, col c0 + 16 ni + n)
  T* C = reinterpret_cast<T*>(g.C);
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const long row = r0 + 16 * mi + 4 * q + j;
      if (!interior(row, g.P, g.H, g.W)) continue;
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const int col = c0 + 16 * ni + n;
        if (col >= g.Cout) continue;
        float v = acc[mi][ni][j] + (g.bias ? g.bias[col] : 0.f);
        if (g.scale) {
          v = v * g.scale[col] + g.shift[col];
          v = v > 0.f ? v : g.slope * v;
        }
        C[row * g.ldc + col] = from_f32<T>(v);
      }
    }
}

// LDS-staged variant (Cin a multiple of the 128-byte stage: 64 bf16 / 32 f32).  256 threads = 2 x 2 waves, block tile
// 128 rows x BN output channels, one 128-byte K slice of ONE tap per stage, double-buffered LDS with register staging
// (next stage's global loads are issued before the MFMAs and written to LDS after them), one barrier per stage.
// LDS rows are 128 B = 8 chunks of 16 B stored at slot (chunk ^ (row & 7)): staging writes and fragment reads are
// bank-conflict free.  The A tile of a stage is the pixel rows shifted by that tap's offset.
template <typename T, int BN>
__global__ __launch_bounds__(256, 2) void conv_tiled_kernel(ConvArgs g) {
  typedef typename Frag<T>::type frag;
  constexpr int EPC = 16 / (int)sizeof(T);
  constexpr int BK = 8 * EPC;
  constexpr int KS = BK / 32;
  constexpr int NJ = BN / 32;                                      // 16-column MFMA tiles per wave
  constexpr int NBI = BN / 32;                                     // B staging chunks per thread
  __shared__ __attribute__((aligned(16))) unsigned char lds[2][(128 + BN) * 128];   // [buffer][A rows | B rows][128 B]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, q = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware order (grid.x is a multiple of 8): the gridDim.y channel tiles of one pixel tile run back to back on the
  // SAME XCD, so the pixel rows are fetched from HBM once and re-read from that XCD's L2
  const unsigned lin = blockIdx.x + gridDim.x * blockIdx.y;
  const unsigned xcd = lin & 7, jj = lin >> 3;
  const long m0 = (long)((jj / gridDim.y) * 8 + xcd) * 128;
  const int n0 = (int)(jj % gridDim.y) * BN;
  if (m0 >= g.P) return;
  const long ldw = (long)g.ntap * g.Cin;
  const int srow = tid >> 3, sc = tid & 7;
  const T* ga = reinterpret_cast<const T*>(g.A) + (m0 + srow) * g.lda + sc * EPC;
  const T* gb = reinterpret_cast<const T*>(g.Wp) + (long)(n0 + srow) * ldw + sc * EPC;
  const int soff = srow * 128 + ((sc ^ (srow & 7)) << 4);
  const int kc = g.Cin / BK;
  const int nstage = g.ntap * kc;
  uint4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;   // named registers: an indexed array would be demoted to scratch / LDS
  int tap = 0, cc = 0;
#define CGLOAD(sidx)                                                                   \
  {                                                                                    \
    const T* pa = ga + (long)tap_offset(g, tap) * g.lda + cc * BK;                     \
    ra0 = *reinterpret_cast<const uint4*>(pa);                                         \
    ra1 = *reinterpret_cast<const uint4*>(pa + 32 * g.lda);                            \
    ra2 = *reinterpret_cast<const uint4*>(pa + 64 * g.lda);                            \
    ra3 = *reinterpret_cast<const uint4*>(pa + 96 * g.lda);                            \
    const T* pb = gb + (long)(sidx) * BK;                                              \
    rb0 = *reinterpret_cast<const uint4*>(pb);                                         \
    rb1 = *reinterpret_cast<const uint4*>(pb + 32 * ldw);                              \
    if constexpr (NBI == 4) {                                                          \
      rb2 = *reinterpret_cast<const uint4*>(pb + 64 * ldw);                            \
      rb3 = *reinterpret_cast<const uint4*>(pb + 96 * ldw);                            \
    }                                                                                  \
    if (++cc == kc) { cc = 0; ++tap; }                                                 \
  }
#define CLSTORE(buf)                                                                   \
  {                                                                                    \
    *reinterpret_cast<uint4*>(&lds[buf][soff]) = ra0;                                  \
    *reinterpret_cast<uint4*>(&lds[buf][soff + 32 * 128]) = ra1;                       \
    *reinterpret_cast<uint4*>(&lds[buf][soff + 64 * 128]) = ra2;                       \
    *reinterpret_cast<uint4*>(&lds[buf][soff + 96 * 128]) = ra3;                       \
    *reinterpret_cast<uint4*>(&lds[buf][128 * 128 + soff]) = rb0;                      \
    *reinterpret_cast<uint4*>(&lds[buf][128 * 128 + soff + 32 * 128]) = rb1;           \
    if constexpr (NBI == 4) {                                                          \
      *reinterpret_cast<uint4*>(&lds[buf][128 * 128 + soff + 64 * 128]) = rb2;         \
      *reinterpret_cast<uint4*>(&lds[buf][128 * 128 + soff + 96 * 128]) = rb3;         \
    }                                                                                  \
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
  CGLOAD(0)
  CLSTORE(0)
  __syncthreads();
  for (int s = 0; s < nstage; ++s) {
    const int buf = s & 1;
    if (s + 1 < nstage) CGLOAD(s + 1)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      frag af[4], bf[NJ];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = lfrag(lds[buf], wm * 64 + 16 * i + n, ks);
#pragma unroll
      for (int j = 0; j < NJ; ++j) bf[j] = lfrag(lds[buf] + 128 * 128, wn * (BN / 2) + 16 * j + n, ks);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = mma16(af[i], bf[j], acc[i][j]);
    }
    if (s + 1 < nstage) CLSTORE(buf ^ 1)
    __syncthreads();
  }
#undef CGLOAD
#undef CLSTORE
  T* C = reinterpret_cast<T*>(g.C);
  float st1[NJ], st2[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) st1[j] = st2[j] = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long row = m0 + wm * 64 + 16 * i + 4 * q + r;
      if (!interior(row, g.P, g.H, g.W)) continue;
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int col = n0 + wn * (BN / 2) + 16 * j + n;
        if (col >= g.Cout) continue;
        float v = acc[i][j][r] + (g.bias ? g.bias[col] : 0.f);
        if (g.scale) {
          v = v * g.scale[col] + g.shift[col];
          v = v > 0.f ? v : g.slope * v;
        }
        const T o = from_f32<T>(v);
        C[row * g.ldc + col] = o;
        const float vo = to_f32<T>(o);                 // statistics of the stored (rounded) activations, as bn_stats_kernel read them
        st1[j] += vo;
        st2[j] += vo * vo;
      }
    }
  if (g.stat_part) {
    // column sums of this 128-row tile: 16 rows per lane -> the four row groups of a wave (lanes n, n+16, n+32, n+48) -> the
    // two waves along M through LDS (the operand buffers are free: the main loop ended with a barrier) -> one store per column
    float* sp = reinterpret_cast<float*>(&lds[0][0]);           // [wm][stat][BN]
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      float a = st1[j], b = st2[j];
      a += __shfl_xor(a, 16); a += __shfl_xor(a, 32);
      b += __shfl_xor(b, 16); b += __shfl_xor(b, 32);
      if (q == 0) {
        sp[(wm * 2 + 0) * BN + wn * (BN / 2) + 16 * j + n] = a;
        sp[(wm * 2 + 1) * BN + wn * (BN / 2) + 16 * j + n] = b;
      }
    }
    __syncthreads();
    if (tid < BN) {
      float* pt = g.stat_part + (size_t)(m0 / 128) * 2 * g.ldp + n0 + tid;
      pt[0] = sp[(0 * 2 + 0) * BN + tid] + sp[(1 * 2 + 0) * BN + tid];
      pt[g.ldp] = sp[(0 * 2 + 1) * BN + tid] + sp[(1 * 2 + 1) * BN + tid];
    }
  }
}

// LDS-DMA ring variant of conv_tiled_kernel (bf16, round 4): 256 pixel rows x BN channels per workgroup; EIGHT COMPUTE WAVES
// (4 x 2, wave tile 64 x BN/2) AND TWO DMA WAVES (640 threads), a ring of STAGES slots of one 128-byte K slice of one tap.
//   * Operands go global -> LDS by `global_load_lds_dwordx4` (no staging registers, no ds_write), two stages in flight while one
//     is read.  A hand-counted REGISTER ring is not safe (the register allocator may copy a destination that is still in flight,
//     tools/check/asm_rings.py); an LDS-DMA has no register destination, its completion is counted by hand (`s_waitcnt vmcnt(N)`,
//     N = DMA instructions the wave issued after the stage it needs; vector-memory operations retire in order).
//   * The DMA waves do nothing else: in-kernel stamps (tools/diag/conv_stamp.py) showed a stage's 48 requests cost the issuing
//     waves 800-900 cycles (address arithmetic, M0 set-up, ~50 cycles of issue per 64-address DMA) -- as much as the stage's matrix
//     work.  Issued by all compute waves they stopped every matrix pipe behind each stage barrier; issued by the second compute
//     wave of every SIMD they serialised that SIMD's two waves (each alone on the pipe runs at ~25 cycles per MFMA: its fragment
//     reads leave issue bubbles that only ANOTHER wave's MFMAs fill).  Two extra waves with no accumulators take the requests off
//     the compute waves altogether.
//   * Compute waves: fragments are read into registers HALF A STAGE ahead of their MFMAs (set 0 = first 32-deep k-step of a stage,
//     set 1 = second), one read between every two matrix instructions, one barrier per stage.
// A DMA instruction moves 1 KB = 8 LDS rows of 128 B; lane l fills physical 16-byte slot (l & 7) of row (l >> 3), so it FETCHES
// the logical chunk (l & 7) ^ (row & 7): the LDS image is the XOR-swizzled one conv_tiled_kernel's fragment reads expect
// (conflict-free under the ds_read_b128 lane groups of CDNA4).  Source rows are clamped to the last pixel row (rows >= P are
// never stored), so the guard rows the 128-row kernel needs are enough.
template <int BN, int STAGES>
__global__ __launch_bounds__(640, 1) void conv_dma_kernel(ConvArgs g) {
  typedef bf16_t T;
  typedef typename Frag<T>::type frag;
  constexpr int BM = 256, NJ = BN / 32;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE_BYTES = A_BYTES + B_BYTES;
  constexpr int NDW = 2;                                           // DMA waves (waves 8, 9)
  constexpr int NPIECE = STAGE_BYTES / 1024, NDMA = NPIECE / NDW;  // 1 KB pieces per stage / per DMA wave and stage
  constexpr int APIECE = A_BYTES / 1024;
  static_assert(NPIECE % NDW == 0 && (STAGES == 2 || STAGES == 3) && (STAGES - 1) * NDMA <= 63, "ring shape (vmcnt is a six-bit count)");
  extern __shared__ __attribute__((aligned(1024))) unsigned char cd_lds[];
  typedef __attribute__((address_space(3))) void lds_void;
  const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool dma_wave = wave >= 8;
  const int wm = (wave & 7) >> 1, wn = wave & 1;
  const unsigned lin = blockIdx.x + gridDim.x * blockIdx.y;        // XCD-aware order, as in conv_tiled_kernel
  const unsigned xcd = lin & 7, jj = lin >> 3;
  const long m0 = (long)((jj / gridDim.y) * 8 + xcd) * BM;
  const int n0 = (int)(jj % gridDim.y) * BN;
  if (m0 >= g.P) return;
  const int kc = g.Cin / 64;
  const int nstage = g.ntap * kc;
  f32x4 acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (dma_wave) {
    // ------------------------------------------------------------------------------------------ the two DMA waves
    const int dw = wave - 8;
    const long ldw = (long)g.ntap * g.Cin;
    const int r_in = lane >> 3, pc = lane & 7, lc = pc ^ r_in;     // (8 * piece is a multiple of 8: row & 7 == r_in)
    const unsigned char* src[NDMA];                                // this wave's pieces: NDMA * dw + h; pieces < APIECE are A rows
#pragma unroll
    for (int h = 0; h < NDMA; ++h) {
      const int pz = NDMA * dw + h;                                // (wave-uniform, but not a compile-time constant: both branches below)
      long row = m0 + 8 * pz + r_in;
      row = row < g.P ? row : g.P - 1;
      const unsigned char* pa = reinterpret_cast<const unsigned char*>(reinterpret_cast<const T*>(g.A) + row * g.lda + lc * 8);
      const unsigned char* pb = reinterpret_cast<const unsigned char*>(reinterpret_cast<const T*>(g.Wp) + (long)(n0 + 8 * (pz - APIECE) + r_in) * ldw + lc * 8);
      src[h] = pz < APIECE ? pa : pb;
    }
    const unsigned lds_base = (unsigned)(unsigned long)(lds_void*)cd_lds;
    auto dma = [&](const unsigned char* sp, unsigned ldst) {
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(sp), "s"(ldst) : "memory");
    };
    int ftap = 0, fcc = 0;                                         // (tap, K slice) of the next stage to request
    auto fill = [&](int slot) {
      const unsigned l = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)slot * STAGE_BYTES);
      const long aoff = ((long)tap_offset(g, ftap) * g.lda + fcc * 64) * 2;
      const long boff = ((long)(ftap * kc + fcc) * 64) * 2;
#pragma unroll
      for (int h = 0; h < NDMA; ++h) {
        const int pz = NDMA * dw + h;
        dma(src[h] + (pz < APIECE ? aoff : boff), l + pz * 1024);  // (the stage image is [A pieces | B pieces]: piece pz at pz KB)
      }
      if (++fcc == kc) { fcc = 0; ++ftap; }
    };
#pragma unroll
    for (int s0 = 0; s0 < STAGES; ++s0)
      if (s0 < nstage) fill(s0);
    // stage 0: requested after it: stages 1 .. STAGES-1 (where they exist)
    if (STAGES == 3 && nstage > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NDMA) : "memory");
    else if (nstage >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");                        // (B0) stage 0 visible
    int slot = 0;
    for (int st = 0; st < nstage; ++st) {
      if (st + 1 < nstage) {                                       // stage st+1 landed? (STAGES == 2: st+2 is requested behind the barrier)
        if (STAGES == 3 && st + 2 < nstage) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      asm volatile("s_barrier" ::: "memory");                      // (B1..) stage st+1 visible; every compute wave has read all of stage st
#ifndef CONV_DIAG_NOFILL   /* sizing build (timing only): no operand fetch behind the prologue stages */
      if (st + STAGES < nstage) fill(slot);
#endif
      slot = slot == STAGES - 1 ? 0 : slot + 1;
    }
  } else {
    // ------------------------------------------------------------------------------------------ the eight compute waves
    auto lfrag = [&](const unsigned char* base, int row, int ks) -> frag {
      const int c = 4 * ks + q;
      return *reinterpret_cast<const frag*>(base + row * 128 + ((c ^ (row & 7)) << 4));
    };
    constexpr int LGKM0 = 0xC07F;                                  // s_waitcnt lgkmcnt(0); vmcnt / expcnt untouched (gfx9 encoding)
    auto read_set = [&](frag (&a)[4], frag (&b)[NJ], int slot, int ks) {
      const unsigned char* ta = cd_lds + (size_t)slot * STAGE_BYTES;
      const unsigned char* tb = ta + A_BYTES;
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = lfrag(ta, wm * 64 + 16 * i + n, ks);
#pragma unroll
      for (int j = 0; j < NJ; ++j) b[j] = lfrag(tb, wn * (BN / 2) + 16 * j + n, ks);
    };
    auto mma_set = [&](const frag (&a)[4], const frag (&b)[NJ]) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = mma16(a[i], b[j], acc[i][j]);
    };
    frag a0[4], b0[NJ], a1[4], b1[NJ];
    asm volatile("s_barrier" ::: "memory");                        // (B0) stage 0 visible (the DMA waves waited for it)
    read_set(a0, b0, 0, 0);
    __builtin_amdgcn_s_waitcnt(LGKM0);
    int slot = 0;
#ifdef CONV_STAMP   /* diagnostic build (tools/diag/conv_stamp.py): cycles per phase of wave 0 of workgroup 0 */
    unsigned long long sacc[5] = {0, 0, 0, 0, 0}, slast = __builtin_readcyclecounter();
#define CST(i) { const unsigned long long nw = __builtin_readcyclecounter(); sacc[i] += nw - slast; slast = nw; }
#else
#define CST(i)
#endif
    for (int st = 0; st < nstage; ++st) {
      const int nslot = slot == STAGES - 1 ? 0 : slot + 1;
      // ---- half 1: read set 1 <- (st, k-step 1) between the 16 MFMAs on set 0 = (st, k-step 0)
      __builtin_amdgcn_sched_barrier(0);
      read_set(a1, b1, slot, 1);
      mma_set(a0, b0);                             // (set 0 was waited for at the end of the previous half: no wait here)
#pragma unroll
      for (int k = 0; k < 4 + NJ; ++k) {           // one fragment read between every two matrix instructions
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, (4 * NJ) / (4 + NJ), 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      CST(0)
      // ---- half 2: everything of stage st is in registers -> barrier -> read set 0 <- (st+1, k-step 0) between the MFMAs on set 1
      __builtin_amdgcn_s_waitcnt(LGKM0);
      CST(1)
      asm volatile("s_barrier" ::: "memory");      // (B1..) stage st+1 visible; the slot of stage st may be refilled
      CST(2)
      __builtin_amdgcn_sched_barrier(0);
      read_set(a0, b0, nslot, 0);                  // (behind the last stage: a harmless read of a stale slot -- unconditional, so that
      mma_set(a1, b1);                             //  the reads and the matrix instructions share a basic block and interleave)
#pragma unroll
      for (int k = 0; k < 4 + NJ; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, (4 * NJ) / (4 + NJ), 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      // set 0 has had sixteen MFMAs to land: an explicit wait HERE costs nothing and leaves hipcc's wait-count pass with nothing
      // pending across the loop edge (without it the pass put lgkmcnt(0) between half 1's reads and its MFMAs)
      __builtin_amdgcn_s_waitcnt(LGKM0);
      CST(4)
      slot = nslot;
    }
#ifdef CONV_STAMP
    if (blockIdx.x == 0 && blockIdx.y == 0 && g.stat_part && tid == 0) {
      unsigned long long* dbgp = reinterpret_cast<unsigned long long*>(g.stat_part + (size_t)((g.P + 127) / 128) * 2 * g.ldp);   // behind the last tile's sums
      for (int i = 0; i < 5; ++i) dbgp[i] = sacc[i];
      dbgp[5] = (unsigned long long)nstage;
    }
#endif
#undef CST
  }
  T* C = reinterpret_cast<T*>(g.C);
  float st1[NJ], st2[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) st1[j] = st2[j] = 0.f;
  if (!dma_wave) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long row = m0 + wm * 64 + 16 * i + 4 * q + r;
        if (!interior(row, g.P, g.H, g.W)) continue;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int col = n0 + wn * (BN / 2) + 16 * j + n;
          if (col >= g.Cout) continue;
          float v = acc[i][j][r] + (g.bias ? g.bias[col] : 0.f);
          if (g.scale) {
            v = v * g.scale[col] + g.shift[col];
            v = v > 0.f ? v : g.slope * v;
          }
          const T o = from_f32<T>(v);
          C[row * g.ldc + col] = o;
          const float vo = to_f32<T>(o);
          st1[j] += vo;
          st2[j] += vo * vo;
        }
      }
  }
  if (g.stat_part) {
    // the workgroup covers TWO 128-row statistics tiles (waves wm = 0,1 and wm = 2,3): same sums in the same order as
    // conv_tiled_kernel's, tile by tile.  (All ten waves take the two barriers.)
    __syncthreads();                                             // the ring is free: every wave has left the main loop
    float* sp = reinterpret_cast<float*>(cd_lds);                // [wm][stat][BN]
    if (!dma_wave) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        float a = st1[j], b = st2[j];
        a += __shfl_xor(a, 16); a += __shfl_xor(a, 32);
        b += __shfl_xor(b, 16); b += __shfl_xor(b, 32);
        if (q == 0) {
          sp[(wm * 2 + 0) * BN + wn * (BN / 2) + 16 * j + n] = a;
          sp[(wm * 2 + 1) * BN + wn * (BN / 2) + 16 * j + n] = b;
        }
      }
    }
    __syncthreads();
    if (tid < 2 * BN) {
      const int t = tid / BN, c = tid % BN;
      if (m0 + 128 * t < g.P) {
        float* pt = g.stat_part + (size_t)(m0 / 128 + t) * 2 * g.ldp + n0 + c;
        pt[0] = sp[((2 * t) * 2 + 0) * BN + c] + sp[((2 * t + 1) * 2 + 0) * BN + c];
        pt[g.ldp] = sp[((2 * t) * 2 + 1) * BN + c] + sp[((2 * t + 1) * 2 + 1) * BN + c];
      }
    }
  }
}

// the per-tile column sums of conv_tiled_kernel -> BatchNorm batch statistics st[c] = sum x, st[C + c] = sum x^2 (fp64), in two
// small launches with a fixed summation order: slices of the tile list (4 row lanes x 64 channels per workgroup), then the slices
__global__ __launch_bounds__(256) void bn_parts_reduce_kernel(const float* __restrict__ part, long ntiles, int ldp, int C, int NS,
                                                              double* __restrict__ scratch) {
  __shared__ double red[2][4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6, c = blockIdx.y * 64 + cl, sl = blockIdx.x;
  const long per = (ntiles + NS - 1) / NS, t0 = sl * per, t1 = t0 + per < ntiles ? t0 + per : ntiles;
  double a = 0.0, b = 0.0;
  if (c < C)
    for (long t = t0 + rl; t < t1; t += 4) {
      a += (double)part[(size_t)t * 2 * ldp + c];
      b += (double)part[((size_t)t * 2 + 1) * ldp + c];
    }
  red[0][rl][cl] = a; red[1][rl][cl] = b;
  __syncthreads();
  if (rl == 0 && c < C) {
    scratch[((size_t)sl * 2 + 0) * C + c] = ((red[0][0][cl] + red[0][1][cl]) + red[0][2][cl]) + red[0][3][cl];
    scratch[((size_t)sl * 2 + 1) * C + c] = ((red[1][0][cl] + red[1][1][cl]) + red[1][2][cl]) + red[1][3][cl];
  }
}
// 32 lanes per element (e = stat * C + c): lane l adds slices l, l + 32, ... in order, then a fixed xor tree over the lanes --
// one thread per element walked the NS = 128 slices one dependent load after the other: 30 us per call, 0.5 ms per C3 step
__global__ __launch_bounds__(256) void bn_parts_final_kernel(const double* __restrict__ scratch, int NS, int C, double* __restrict__ st) {
  const int e = blockIdx.x * 8 + (threadIdx.x >> 5), l = threadIdx.x & 31;
  const bool live = e < 2 * C;
  const int k = live ? e / C : 0, c = live ? e % C : 0;
  double a = 0.0;
  if (live)
    for (int sl = l; sl < NS; sl += 32) a += scratch[((size_t)sl * 2 + k) * C + c];
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);      // (stays inside the 32-lane half: o < 32)
  if (live && l == 0) st[e] = a;
}

// weight [Cout][Cin][kh][kw] fp32 -> forward pack  Wf[Np][ntap][Cinp]       (Wf[co][t][ci] = w[co][ci][t])
//                                  -> backward pack Wb[Cinp64][ntap][Coutp]  (Wb[ci][t][co] = w[co][ci][ntap-1-t])
template <typename T>
__global__ void conv_pack_kernel(const float* __restrict__ w, T* __restrict__ wf, T* __restrict__ wb, int Cout, int Cin,
                                 int ntap, int Np, int Cinp, int Mb, int Coutp) {
  const long nf = (long)Np * ntap * Cinp, nb = wb ? (long)Mb * ntap * Coutp : 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nf + nb; i += (long)gridDim.x * blockDim.x) {
    if (i < nf) {
      const int ci = (int)(i % Cinp), t = (int)((i / Cinp) % ntap), co = (int)(i / ((long)Cinp * ntap));
      wf[i] = from_f32<T>((co < Cout && ci < Cin) ? w[((long)co * Cin + ci) * ntap + t] : 0.f);
    } else {
      const long k = i - nf;
      const int co = (int)(k % Coutp), t = (int)((k / Coutp) % ntap), ci = (int)(k / ((long)Coutp * ntap));
      wb[k] = from_f32<T>((co < Cout && ci < Cin) ? w[((long)co * Cin + ci) * ntap + (ntap - 1 - t)] : 0.f);
    }
  }
}

// ------------------------------------------------------------------------------------------------ BatchNorm
// column sums of a haloed matrix (halo rows are zero): st[c] += sum x, st[C + c] += sum x^2   (fp64 atomics)
template <typename T>
__global__ __launch_bounds__(256) void bn_stats_kernel(const T* __restrict__ X, long ld, long P, int C, int rows_per_block,
                                                       double* __restrict__ st) {
  __shared__ double red[2][256][8];
  const int cg = C / 8, rl = 256 / cg;               // channel groups of 8, row lanes
  const int g8 = threadIdx.x % cg, r = threadIdx.x / cg;
  double s1[8], s2[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) s1[i] = s2[i] = 0.0;
  const long p0 = (long)blockIdx.x * rows_per_block;
  const long p1 = p0 + rows_per_block < P ? p0 + rows_per_block : P;
  if (r < rl)
    for (long p = p0 + r; p < p1; p += 4 * rl) {          // four rows in flight per thread (halo rows are zero: no test)
      float xv[4][8];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (p + j * rl < p1) load8<T>(X + (p + j * rl) * ld + g8 * 8, xv[j]);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (p + j * rl < p1) {
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const double v = (double)xv[j][i];
            s1[i] += v;
            s2[i] += v * v;
          }
        }
    }
#pragma unroll
  for (int i = 0; i < 8; ++i) { red[0][threadIdx.x][i] = s1[i]; red[1][threadIdx.x][i] = s2[i]; }
  __syncthreads();
  // thread t < C reduces channel t over the row lanes
  for (int c = threadIdx.x; c < C; c += 256) {
    const int gg = c / 8, i = c % 8;
    double a = 0.0, b = 0.0;
    for (int rr = 0; rr < rl; ++rr) { a += red[0][rr * cg + gg][i]; b += red[1][rr * cg + gg][i]; }
    atomicAdd(st + c, a);
    atomicAdd(st + C + c, b);
  }
}

// train: batch statistics -> (scale, shift, mean, rstd) and the momentum update of the running buffers
// eval : running statistics -> (scale, shift)
__global__ void bn_finalize_kernel(const double* __restrict__ st, const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ rmean, float* __restrict__ rvar, float* __restrict__ ss, int C, double n,
                                   float eps, float momentum, int train) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double mean, var;
  if (train) {
    mean = st[c] / n;
    var = st[C + c] / n - mean * mean;
    if (var < 0) var = 0;
    rmean[c] = (float)((1.0 - momentum) * (double)rmean[c] + momentum * mean);
    rvar[c] = (float)((1.0 - momentum) * (double)rvar[c] + momentum * var * (n / (n > 1 ? n - 1 : 1)));
  } else {
    mean = rmean[c];
    var = rvar[c];
  }
  const double rstd = 1.0 / sqrt(var + (double)eps);
  const double sc = (double)gamma[c] * rstd;
  ss[c] = (float)sc;
  ss[C + c] = (float)((double)beta[c] - mean * sc);
  ss[2 * C + c] = (float)mean;
  ss[3 * C + c] = (float)rstd;
}

// y = leaky(scale * x + shift) on interior pixels, 8 channels per thread; y may be a channel slice (ldy, coff)
template <typename T>
__global__ __launch_bounds__(256) void bn_act_kernel(const T* __restrict__ X, long ldx, T* __restrict__ Y, long ldy,
                                                     const float* __restrict__ ss, int C, long P, int H, int W, float slope) {
  const int cg = C / 8;
  const unsigned i = blockIdx.x * 256u + threadIdx.x;      // host guarantees P * C / 8 < 2^32
  const long p = i / (unsigned)cg;
  const int c8 = (int)(i % (unsigned)cg) * 8;
  if (!interior(p, P, H, W)) return;
  float xv[8], yv[8];
  load8<T>(X + p * ldx + c8, xv);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float v = xv[k] * ss[c8 + k] + ss[C + c8 + k];
    yv[k] = v > 0.f ? v : slope * v;
  }
  store8<T>(Y + p * ldy + c8, yv);
}

// g = (dyA + dyB) * leaky'(y);  S[c] += sum g,  S[C + c] += sum g * xhat      (fp64 atomics), xhat = (x - mean) rstd
// Round 4: the LeakyReLU mask y > 0 is RECOMPUTED from the raw convolution output, y = leaky(x scale + shift) with the very
// expression bn_act_kernel used, instead of read from the stored activation: one tensor read less in each of the two backward
// passes (Y is accepted for the ABI and not touched)
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* __restrict__ dyA, long ldA, const T* __restrict__ dyB,
                                                            long ldB, const T* __restrict__ Y, long ldy,
                                                            const T* __restrict__ X, long ldx, const float* __restrict__ ss,
                                                            int C, long P, int H, int W, float slope, int rows_per_block,
                                                            double* __restrict__ S) {
  __shared__ float red[2][256][8];
  const int cg = C / 8, rl = 256 / cg;
  const int g8 = threadIdx.x % cg, r = threadIdx.x / cg;
  float s1[8], s2[8], mean[8], rstd[8], sc[8], sh[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    s1[i] = s2[i] = 0.f;
    sc[i] = ss[g8 * 8 + i];
    sh[i] = ss[C + g8 * 8 + i];
    mean[i] = ss[2 * C + g8 * 8 + i];
    rstd[i] = ss[3 * C + g8 * 8 + i];
  }
  const long p0 = (long)blockIdx.x * rows_per_block;
  const long p1 = p0 + rows_per_block < P ? p0 + rows_per_block : P;
  if (r < rl)
    for (long p = p0 + r; p < p1; p += 2 * rl) {          // two rows (6-8 loads of 16 bytes) in flight per thread
      float av[2][8], bv[2][8], xv[2][8];
      bool ok[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const long q = p + j * rl;
        ok[j] = q < p1 && interior(q, P, H, W);
        if (ok[j]) {
          load8<T>(dyA + q * ldA + g8 * 8, av[j]);
          if (dyB) load8<T>(dyB + q * ldB + g8 * 8, bv[j]);
          load8<T>(X + q * ldx + g8 * 8, xv[j]);
        }
      }
#pragma unroll
      for (int j = 0; j < 2; ++j)
        if (ok[j]) {
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            float gq = dyB ? av[j][i] + bv[j][i] : av[j][i];
            if (!(xv[j][i] * sc[i] + sh[i] > 0.f)) gq *= slope;
            s1[i] += gq;
            s2[i] += gq * ((xv[j][i] - mean[i]) * rstd[i]);
          }
        }
    }
#pragma unroll
  for (int i = 0; i < 8; ++i) { red[0][threadIdx.x][i] = s1[i]; red[1][threadIdx.x][i] = s2[i]; }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    const int gg = c / 8, i = c % 8;
    double a = 0.0, b = 0.0;
    for (int rr = 0; rr < rl; ++rr) { a += (double)red[0][rr * cg + gg][i]; b += (double)red[1][rr * cg + gg][i]; }
    atomicAdd(S + c, a);
    atomicAdd(S + C + c, b);
  }
}

// dx = scale * (g - S1/n - xhat * S2/n) on interior pixels;  block 0 also writes dgamma = S2, dbeta = S1
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dyA, long ldA, const T* __restrict__ dyB,
                                                           long ldB, const T* __restrict__ Y, long ldy,
                                                           const T* __restrict__ X, long ldx, const float* __restrict__ ss,
                                                           const double* __restrict__ S, T* __restrict__ dX, long lddx,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta, int C, long P,
                                                           int H, int W, float slope, double n) {
  if (blockIdx.x == 0)
    for (int c = threadIdx.x; c < C; c += 256) {
      dgamma[c] = (float)S[C + c];
      dbeta[c] = (float)S[c];
    }
  const int cg = C / 8;
  const unsigned i = blockIdx.x * 256u + threadIdx.x;      // host guarantees P * C / 8 < 2^32
  const long p = i / (unsigned)cg;
  const int c8 = (int)(i % (unsigned)cg) * 8;
  if (!interior(p, P, H, W)) return;
  const double invn = 1.0 / n;
  float av[8], bv[8], xv[8], dv[8];
  load8<T>(dyA + p * ldA + c8, av);
  if (dyB) {
    load8<T>(dyB + p * ldB + c8, bv);
#pragma unroll
    for (int k = 0; k < 8; ++k) av[k] += bv[k];
  }
  load8<T>(X + p * ldx + c8, xv);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = c8 + k;
    float gq = av[k];
    if (!(xv[k] * ss[c] + ss[C + c] > 0.f)) gq *= slope;
    const float xh = (xv[k] - ss[2 * C + c]) * ss[3 * C + c];
    const float m1 = (float)(S[c] * invn), m2 = (float)(S[C + c] * invn);
    dv[k] = ss[c] * (gq - m1 - xh * m2);
  }
  store8<T>(dX + p * lddx + c8, dv);
}

// ------------------------------------------------------------------------------------------------ pooling
// MaxPool2d(2) (floor mode): first maximum in window scan order wins (strict >), index kept for the backward
template <typename T>
__global__ __launch_bounds__(256) void maxpool_kernel(const T* __restrict__ X, long ldx, T* __restrict__ Y, long ldy,
                                                      unsigned char* __restrict__ idx, int C, int B, int H, int W) {
  const int Ho = H / 2, Wo = W / 2, cg = C / 8;
  const unsigned i = blockIdx.x * 256u + threadIdx.x;
  const int c8 = (int)(i % (unsigned)cg) * 8;
  unsigned o = i / (unsigned)cg;
  if (o >= (long)B * Ho * Wo) return;
  const int xo = (int)(o % (unsigned)Wo), yo = (int)((o / (unsigned)Wo) % (unsigned)Ho), b = (int)(o / ((unsigned)Wo * (unsigned)Ho));
  const long pin = ((long)b * (H + 2) + 2 * yo + 1) * (W + 2) + 2 * xo + 1;
  const long pout = ((long)b * (Ho + 2) + yo + 1) * (Wo + 2) + xo + 1;
  const long offs[4] = {0, 1, (long)(W + 2), (long)(W + 3)};
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    float best = to_f32<T>(X[pin * ldx + c8 + k]);
    int bi = 0;
#pragma unroll
    for (int t = 1; t < 4; ++t) {
      const float v = to_f32<T>(X[(pin + offs[t]) * ldx + c8 + k]);
      if (v > best) { best = v; bi = t; }
    }
    Y[pout * ldy + c8 + k] = from_f32<T>(best);
    idx[pout * C + c8 + k] = (unsigned char)bi;
  }
}

// dX[window] = dY routed to the recorded position, zero at the other three (rows/columns outside any window stay 0)
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ dY, long ldy, const unsigned char* __restrict__ idx,
                                                          T* __restrict__ dX, long ldx, int C, int B, int H, int W) {
  const int Ho = H / 2, Wo = W / 2, cg = C / 8;
  const unsigned i = blockIdx.x * 256u + threadIdx.x;
  const int c8 = (int)(i % (unsigned)cg) * 8;
  unsigned o = i / (unsigned)cg;
  if (o >= (long)B * Ho * Wo) return;
  const int xo = (int)(o % (unsigned)Wo), yo = (int)((o / (unsigned)Wo) % (unsigned)Ho), b = (int)(o / ((unsigned)Wo * (unsigned)Ho));
  const long pin = ((long)b * (H + 2) + 2 * yo + 1) * (W + 2) + 2 * xo + 1;
  const long pout = ((long)b * (Ho + 2) + yo + 1) * (Wo + 2) + xo + 1;
  const long offs[4] = {0, 1, (long)(W + 2), (long)(W + 3)};
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const T gq = dY[pout * ldy + c8 + k];
    const int bi = idx[pout * C + c8 + k];
#pragma unroll
    for (int t = 0; t < 4; ++t) dX[(pin + offs[t]) * ldx + c8 + k] = t == bi ? gq : from_f32<T>(0.f);
  }
}

// ------------------------------------------------------------------------------------------------ upsampling
// nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True): src = dst * (in-1)/(out-1) in fp32,
// i0 = (int)src, lambda1 = src - i0.  The 2H x 2W result is placed at offset (py, px) inside the (Ht, Wt) target
// (F.pad to the skip's size, tmp_utils.py:79-84) and written into the channel slice [coff, coff+C) of the target.
__device__ __forceinline__ void bil(int o, int nin, int nout, int& i0, int& i1, float& l0, float& l1) {
  const float sc = nout > 1 ? (float)(nin - 1) / (float)(nout - 1) : 0.f;
  const float src = sc * (float)o;
  i0 = (int)src;
  i1 = i0 + (i0 < nin - 1 ? 1 : 0);
  l1 = src - (float)i0;
  l0 = 1.f - l1;
}

template <typename T>
__global__ __launch_bounds__(256) void upsample_kernel(const T* __restrict__ X, long ldx, T* __restrict__ Y, long ldy, int C,
                                                       int B, int Hi, int Wi, int Ht, int Wt, int py, int px) {
  const int Ho = 2 * Hi, Wo = 2 * Wi, cg = C / 8;
  const unsigned i = blockIdx.x * 256u + threadIdx.x;
  const int c8 = (int)(i % (unsigned)cg) * 8;
  unsigned o = i / (unsigned)cg;
  if (o >= (long)B * Ho * Wo) return;
  const int xo = (int)(o % (unsigned)Wo), yo = (int)((o / (unsigned)Wo) % (unsigned)Ho), b = (int)(o / ((unsigned)Wo * (unsigned)Ho));
  int y0, y1, x0, x1;
  float ly0, ly1, lx0, lx1;
  bil(yo, Hi, Ho, y0, y1, ly0, ly1);
  bil(xo, Wi, Wo, x0, x1, lx0, lx1);
  const long rb = (long)b * (Hi + 2);
  const long p00 = ((rb + y0 + 1) * (Wi + 2) + x0 + 1) * ldx, p01 = ((rb + y0 + 1) * (Wi + 2) + x1 + 1) * ldx;
  const long p10 = ((rb + y1 + 1) * (Wi + 2) + x0 + 1) * ldx, p11 = ((rb + y1 + 1) * (Wi + 2) + x1 + 1) * ldx;
  const long po = (((long)b * (Ht + 2) + yo + py + 1) * (Wt + 2) + xo + px + 1) * ldy;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float v = ly0 * (lx0 * to_f32<T>(X[p00 + c8 + k]) + lx1 * to_f32<T>(X[p01 + c8 + k])) +
                    ly1 * (lx0 * to_f32<T>(X[p10 + c8 + k]) + lx1 * to_f32<T>(X[p11 + c8 + k]));
    Y[po + c8 + k] = from_f32<T>(v);
  }
}

// adjoint, as a gather: input pixel (yi, xi) collects every output pixel whose two taps touch it
template <typename T>
__global__ __launch_bounds__(256) void upsample_bwd_kernel(const T* __restrict__ dY, long ldy, T* __restrict__ dX, long ldx,
                                                           int C, int B, int Hi, int Wi, int Ht, int Wt, int py, int px) {
  const int Ho = 2 * Hi, Wo = 2 * Wi, cg = C / 8;
  const unsigned i = blockIdx.x * 256u + threadIdx.x;
  const int c8 = (int)(i % (unsigned)cg) * 8;
  unsigned o = i / (unsigned)cg;
  if (o >= (long)B * Hi * Wi) return;
  const int xi = (int)(o % (unsigned)Wi), yi = (int)((o / (unsigned)Wi) % (unsigned)Hi), b = (int)(o / ((unsigned)Wi * (unsigned)Hi));
  float wy[6], wx[6];
  int oy[6], ox[6];
#pragma unroll
  for (int t = 0; t < 6; ++t) {
    int a0, a1;
    float l0, l1;
    oy[t] = 2 * yi - 2 + t;
    wy[t] = 0.f;
    if (oy[t] >= 0 && oy[t] < Ho) {
      bil(oy[t], Hi, Ho, a0, a1, l0, l1);
      wy[t] = (a0 == yi ? l0 : 0.f) + (a1 == yi ? l1 : 0.f);
    }
    ox[t] = 2 * xi - 2 + t;
    wx[t] = 0.f;
    if (ox[t] >= 0 && ox[t] < Wo) {
      bil(ox[t], Wi, Wo, a0, a1, l0, l1);
      wx[t] = (a0 == xi ? l0 : 0.f) + (a1 == xi ? l1 : 0.f);
    }
  }
  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.f;
  for (int ty = 0; ty < 6; ++ty) {
    if (wy[ty] == 0.f) continue;
    for (int tx = 0; tx < 6; ++tx) {
      if (wx[tx] == 0.f) continue;
      const float w = wy[ty] * wx[tx];
      const T* g = dY + (((long)b * (Ht + 2) + oy[ty] + py + 1) * (Wt + 2) + ox[tx] + px + 1) * ldy + c8;
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += w * to_f32<T>(g[k]);
    }
  }
  T* d = dX + (((long)b * (Hi + 2) + yi + 1) * (Wi + 2) + xi + 1) * ldx + c8;
#pragma unroll
  for (int k = 0; k < 8; ++k) d[k] = from_f32<T>(acc[k]);
}

// ------------------------------------------------------------------------------------------------ weight gradient
// exact-f32 / generic fallback:  slab[z][m][n] = sum_{p in slice z} A[p][m] * Bm[p + shift][n]   (64 x 64 tile per block)
template <typename T>
__global__ __launch_bounds__(256) void wgrad_simple_kernel(const T* __restrict__ A, long lda, const T* __restrict__ Bm, long ldb,
                                                           long shift, float* __restrict__ slab, long ldc, long slab_stride,
                                                           long P, long rows_per_slice) {
  __shared__ float sa[16][64], sb[16][64];
  const int tm = threadIdx.x / 16, tn = threadIdx.x % 16;          // 4 x 4 outputs per thread
  const int m0 = blockIdx.x * 64, n0 = blockIdx.y * 64;
  const long p0 = (long)blockIdx.z * rows_per_slice;
  const long p1 = p0 + rows_per_slice < P ? p0 + rows_per_slice : P;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  const int lr = threadIdx.x / 16, lc = (threadIdx.x % 16) * 4;
  for (long p = p0; p < p1; p += 16) {
    const long pr = p + lr;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      sa[lr][lc + k] = pr < p1 ? to_f32<T>(A[pr * lda + m0 + lc + k]) : 0.f;
      sb[lr][lc + k] = pr < p1 ? to_f32<T>(Bm[(pr + shift) * ldb + n0 + lc + k]) : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float av[4], bv[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) { av[k] = sa[r][tm * 4 + k]; bv[k] = sb[r][tn * 4 + k]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
    }
    __syncthreads();
  }
  float* out = slab + (size_t)blockIdx.z * slab_stride;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) out[(long)(m0 + tm * 4 + i) * ldc + n0 + tn * 4 + j] = acc[i][j];
}

// dW[co][ci][t] = sum_s slabs[t][s][co][ci]     (torch layout [Cout][Cin][kh][kw])
__global__ void wgrad_reduce_kernel(const float* __restrict__ slabs, int S, long slab_stride, long ldc, float* __restrict__ dW,
                                    int Cout, int Cin, int ntap, int transposed) {
  const long total = (long)Cout * Cin * ntap;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ci = (int)(i % Cin), co = (int)((i / Cin) % Cout), t = (int)(i / ((long)Cin * Cout));
    const float* s = slabs + (size_t)t * S * slab_stride + (transposed ? (long)ci * ldc + co : (long)co * ldc + ci);
    double a = 0.0;
    for (int k = 0; k < S; ++k) a += (double)s[(size_t)k * slab_stride];
    dW[((long)co * Cin + ci) * ntap + t] = (float)a;
  }
}


// ------------------------------------------------------------------------------------------------ thin convolutions
// The two ends of the U-Net are memory-bound products that the MFMA kernels above compute 32-64 channels wide:
//   first layer   conv3x3(1 or 2 -> 64)  (unet.py:247-262 `inc`, tmp_utils.py:8-37): 18 MACs per output, 128 B written per pixel
//   last layer    conv1x1(64 -> 1 or K)  (`outc`): 128 B read per pixel
// At C3 they took 2.6 of the 20 ms of convolution time (4-13 TFLOP/s).  Direct kernels: one pass over the wide tensor,
// fp32 accumulation, weights (fp32, torch layout, unpacked) in LDS.

// raw[p][co] = bias[co] + sum_{tap, ci < CIN} X[p + off(tap)][ci] * w[co][ci][tap], interior pixels only; optional folded
// eval-BatchNorm + LeakyReLU like the MFMA kernels.  Thread = (pixel, 8 consecutive output channels): 16-byte stores.
template <typename T, int CIN>
__global__ __launch_bounds__(256) void conv3x3_thin_fwd_kernel(const T* __restrict__ A, long lda, const float* __restrict__ w,
                                                               const float* __restrict__ bias, const float* __restrict__ scale,
                                                               const float* __restrict__ shift, float slope, T* __restrict__ C,
                                                               long ldc, long P, int H, int W, int Cout) {
  // a thread owns FOUR consecutive haloed pixel rows p0 .. p0+3 (neighbours along x) and 8 output channels: the 18 input
  // columns (3 kernel rows x 6 positions) are loaded once for the four, and each weight vector read from LDS serves
  // four pixels -- one pixel per thread was bound by the 18-36 LDS weight reads and 9-18 two-byte loads per pixel
  extern __shared__ float wl[];                        // [9 * CIN][Cout]
  for (int i = threadIdx.x; i < 9 * CIN * Cout; i += 256) {
    const int co = i % Cout, k = i / Cout, tap = k / CIN, ci = k % CIN;
    wl[i] = w[((long)co * CIN + ci) * 9 + tap];
  }
  __syncthreads();
  const int chunks = Cout / 8, ppb = 256 / chunks;
  const int ch = threadIdx.x % chunks, pl = threadIdx.x / chunks;
  if (pl >= ppb) return;
  float bv[8], sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    bv[j] = bias ? bias[ch * 8 + j] : 0.f;
    sc[j] = scale ? scale[ch * 8 + j] : 1.f;
    sh[j] = scale ? shift[ch * 8 + j] : 0.f;
  }
  for (long p0 = ((long)blockIdx.x * ppb + pl) * 4; p0 < P; p0 += (long)gridDim.x * ppb * 4) {
    bool in[4];
    bool any = false;
#pragma unroll
    for (int u = 0; u < 4; ++u) { in[u] = interior(p0 + u, P, H, W); any |= in[u]; }
    if (!any) continue;
    float x[3][6][CIN];                                // rows dy = -1..1, positions p0-1 .. p0+4 (guard rows make every read legal)
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 6; ++c)
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) x[r][c][ci] = to_f32<T>(A[(p0 + (long)(r - 1) * (W + 2) + (c - 1)) * lda + ci]);
    float acc[4][8];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[u][j] = bv[j];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int ci = 0; ci < CIN; ++ci) {
        const float4 w0 = *reinterpret_cast<const float4*>(wl + (tap * CIN + ci) * Cout + ch * 8);
        const float4 w1 = *reinterpret_cast<const float4*>(wl + (tap * CIN + ci) * Cout + ch * 8 + 4);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float xv = x[tap / 3][u + tap % 3][ci];
          acc[u][0] += xv * w0.x; acc[u][1] += xv * w0.y; acc[u][2] += xv * w0.z; acc[u][3] += xv * w0.w;
          acc[u][4] += xv * w1.x; acc[u][5] += xv * w1.y; acc[u][6] += xv * w1.z; acc[u][7] += xv * w1.w;
        }
      }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (!in[u]) continue;
      if (scale) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float v = acc[u][j] * sc[j] + sh[j];
          acc[u][j] = v > 0.f ? v : slope * v;
        }
      }
      store8<T>(C + (p0 + u) * ldc + ch * 8, acc[u]);
    }
  }
}

// part[block][co][ci][tap] = sum over the block's pixel rows of dY[p][co] * X[p + off(tap)][ci]   (all haloed rows: dY's halo
// is zero).  Thread = (pixel group, kernel row dy, 8 consecutive output channels): one 16-byte dY read and 3 x CIN input
// values per pixel, 24 x CIN accumulators.  (A first version with one channel PAIR per thread issued 19 loads per thread and
// pixel -- 32 lanes fetching the same 18 input values -- and was address-bound: slower than the MFMA path it replaced.)
template <typename T, int CIN>
__global__ __launch_bounds__(256) void conv3x3_thin_wgrad_kernel(const T* __restrict__ dY, long lddy, const T* __restrict__ X,
                                                                 long ldx, float* __restrict__ part, long P, int W, int Cout) {
  extern __shared__ float red[];                       // [groups][Cout][9 * CIN]
  const int chunks = Cout / 8, tpp = 3 * chunks, groups = 256 / tpp;
  const int ch = threadIdx.x % chunks, tg = (threadIdx.x / chunks) % 3, pg = threadIdx.x / tpp;
  const int n = Cout * 9 * CIN;
  float acc[3][CIN][8];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[t][ci][j] = 0.f;
  if (pg < groups) {
    // two pixel rows per iteration (loads of both in flight before the FMAs: the loop is latency-bound otherwise); the second
    // one is clamped to the last row and weighted 0 past the end
    const long stride = (long)gridDim.x * groups;
    for (long p = (long)blockIdx.x * groups + pg; p < P; p += 2 * stride) {
      const long p2 = p + stride < P ? p + stride : P - 1;
      const float m2 = p + stride < P ? 1.f : 0.f;
      float d[8], e[8];
      load8<T>(dY + p * lddy + ch * 8, d);
      load8<T>(dY + p2 * lddy + ch * 8, e);
      const T* xp = X + (p + (long)(tg - 1) * (W + 2) - 1) * ldx;
      const T* xq = X + (p2 + (long)(tg - 1) * (W + 2) - 1) * ldx;
      float xa[3][CIN], xb[3][CIN];
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) {
          xa[t][ci] = to_f32<T>(xp[t * ldx + ci]);
          xb[t][ci] = to_f32<T>(xq[t * ldx + ci]) * m2;
        }
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[t][ci][j] += d[j] * xa[t][ci] + e[j] * xb[t][ci];
    }
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
        for (int j = 0; j < 8; ++j) red[(size_t)pg * n + ((ch * 8 + j) * CIN + ci) * 9 + 3 * tg + t] = acc[t][ci][j];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += 256) {          // i = (co * CIN + ci) * 9 + tap: torch layout [co][ci][tap]
    float s = 0.f;
    for (int g = 0; g < groups; ++g) s += red[(size_t)g * n + i];
    part[(size_t)blockIdx.x * n + i] = s;
  }
}

// dW[i] = sum_blocks part[b][i]: 64 columns x 16 slices of the blocks per workgroup (one thread per column walking all the
// blocks took longer than the pass over the activations)
__global__ __launch_bounds__(1024) void thin_wgrad_finish_kernel(const float* __restrict__ part, int nblocks, int n, float* __restrict__ dW) {
  __shared__ double red[16][64];
  const int col = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + col;
  double a = 0.0;
  if (i < n) {
#pragma unroll 8
    for (int b = slice; b < nblocks; b += 16) a += (double)part[(size_t)b * n + i];
  }
  red[slice][col] = a;
  __syncthreads();
  if (slice == 0 && i < n) {
#pragma unroll
    for (int k = 1; k < 16; ++k) a += red[k][col];
    dW[i] = (float)a;
  }
}

// out[p][k] = bias[k] + sum_ci X[p][ci] * w[k][ci],  k < K <= 8, Cin = 64: 8 lanes per pixel (16 bytes of the row each),
// partial dot products reduced over the 8 lanes.  Interior pixels only, columns K..ld-1 untouched.
template <typename T>
__global__ __launch_bounds__(256) void conv1x1_thin_fwd_kernel(const T* __restrict__ X, long ldx, const float* __restrict__ w,
                                                               const float* __restrict__ bias, T* __restrict__ C, long ldc, long P,
                                                               int H, int W, int K) {
  __shared__ float wl[8 * 64];
  for (int i = threadIdx.x; i < 8 * 64; i += 256) wl[i] = i < K * 64 ? w[i] : 0.f;
  __syncthreads();
  const int ch = threadIdx.x & 7, pl = threadIdx.x >> 3;
  for (long p = (long)blockIdx.x * 32 + pl; p < P; p += (long)gridDim.x * 32) {
    const bool in = interior(p, P, H, W);               // (whole 8-lane groups agree: no divergence inside a shuffle group)
    float xv[8];
    if (in) load8<T>(X + p * ldx + ch * 8, xv);
    else {
#pragma unroll
      for (int j = 0; j < 8; ++j) xv[j] = 0.f;
    }
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float a = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) a += xv[j] * wl[k * 64 + ch * 8 + j];
      a += __shfl_xor(a, 1);
      a += __shfl_xor(a, 2);
      a += __shfl_xor(a, 4);
      acc[k] = a;
    }
    if (in && ch < K) {
      float v = acc[0];
#pragma unroll
      for (int k = 1; k < 8; ++k) v = ch == k ? acc[k] : v;
      C[p * ldc + ch] = from_f32<T>(v + (bias ? bias[ch] : 0.f));
    }
  }
}

// dX[p][ci] = sum_{k < K} dY[p][k] * w[k][ci]   (Cin = 64; interior pixels; dY's halo is zero anyway)
template <typename T>
__global__ __launch_bounds__(256) void conv1x1_thin_bwd_data_kernel(const T* __restrict__ dY, long lddy, const float* __restrict__ w,
                                                                    T* __restrict__ dX, long lddx, long P, int H, int W, int K) {
  __shared__ float wl[8 * 64];
  for (int i = threadIdx.x; i < 8 * 64; i += 256) wl[i] = i < K * 64 ? w[i] : 0.f;
  __syncthreads();
  const int ch = threadIdx.x & 7, pl = threadIdx.x >> 3;
  for (long p = (long)blockIdx.x * 32 + pl; p < P; p += (long)gridDim.x * 32) {
    if (!interior(p, P, H, W)) continue;
    float d[8];
    load8<T>(dY + p * lddy, d);                         // lddy >= 8: columns K..7 are zero in the gradient buffer or masked here
#pragma unroll
    for (int k = 0; k < 8; ++k) d[k] = k < K ? d[k] : 0.f;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) a += d[k] * wl[k * 64 + ch * 8 + j];
      acc[j] = a;
    }
    store8<T>(dX + p * lddx + ch * 8, acc);
  }
}

// part[block][k][ci] = sum over the block's rows of dY[p][k] * X[p][ci]   (K <= 8, Cin = 64; all haloed rows)
template <typename T>
__global__ __launch_bounds__(256) void conv1x1_thin_wgrad_kernel(const T* __restrict__ dY, long lddy, const T* __restrict__ X,
                                                                 long ldx, float* __restrict__ part, long P, int K) {
  __shared__ float red[32][8 * 64 / 8 + 1][8];          // [pixel lane][k * 8 + chunk][8 channels of the chunk] (padded)
  const int ch = threadIdx.x & 7, pl = threadIdx.x >> 3;
  float acc[8][8];
#pragma unroll
  for (int k = 0; k < 8; ++k)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[k][j] = 0.f;
  const long stride = (long)gridDim.x * 32;
  for (long p = (long)blockIdx.x * 32 + pl; p < P; p += 2 * stride) {     // two rows in flight per thread (latency-bound otherwise)
    const long p2 = p + stride < P ? p + stride : P - 1;
    const float m2 = p + stride < P ? 1.f : 0.f;
    float xv[8], yv[8], d[8], e[8];
    load8<T>(X + p * ldx + ch * 8, xv);
    load8<T>(X + p2 * ldx + ch * 8, yv);
    load8<T>(dY + p * lddy, d);                        // the K <= 8 gradient columns of the row: one 16-byte read (lddy >= 8)
    load8<T>(dY + p2 * lddy, e);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float dk = k < K ? d[k] : 0.f, ek = k < K ? e[k] * m2 : 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[k][j] += dk * xv[j] + ek * yv[j];
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k)
#pragma unroll
    for (int j = 0; j < 8; ++j) red[pl][k * 8 + ch][j] = acc[k][j];
  __syncthreads();
  for (int i = threadIdx.x; i < K * 64; i += 256) {      // i = k * 64 + ci
    const int k = i / 64, ci = i % 64;
    float s = 0.f;
    for (int g = 0; g < 32; ++g) s += red[g][k * 8 + ci / 8][ci % 8];
    part[(size_t)blockIdx.x * (K * 64) + i] = s;
  }
}

// ------------------------------------------------------------------------------------------------ boundary maps
// log-magnitude of a [B,2,F,T] STFT pair + fp64 sum / sum of squares (utils.py:273-306)
__global__ __launch_bounds__(256) void logmag_kernel(const float* __restrict__ spec, float* __restrict__ out, long FT, long total,
                                                     long obs, double* __restrict__ st) {
  __shared__ double red[2][4];
  double s1 = 0.0, s2 = 0.0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long b = i / FT, r = i % FT;
    const float re = spec[(b * 2) * FT + r], im = spec[(b * 2 + 1) * FT + r];
    const float v = logf(sqrtf(re * re + im * im) + 1e-6f);
    out[b * obs + r] = v;
    s1 += (double)v;
    s2 += (double)v * (double)v;
  }
  s1 = wave_sum(s1);
  s2 = wave_sum(s2);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red[0][wave] = s1; red[1][wave] = s2; }
  __syncthreads();
  if (threadIdx.x == 0 && st) {
    atomicAdd(st, red[0][0] + red[0][1] + red[0][2] + red[0][3]);
    atomicAdd(st + 1, red[1][0] + red[1][1] + red[1][2] + red[1][3]);
  }
}

// x <- (x - mean) / std with the batch-global mean and UNBIASED std from st = (sum, sumsq) over n values
__global__ void standardize_kernel(float* __restrict__ a, float* __restrict__ b, long total, long FT, long bs,
                                   const double* __restrict__ st, double n, float* __restrict__ mean_std) {
  const double mean = st[0] / n;
  double var = (st[1] - n * mean * mean) / (n - 1.0);
  if (var < 0) var = 0;
  const float m = (float)mean, s = (float)sqrt(var);
  if (blockIdx.x == 0 && threadIdx.x == 0 && mean_std) { mean_std[0] = m; mean_std[1] = s; }
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long j = (i / FT) * bs + i % FT;
    a[j] = (a[j] - m) / s;
    if (b) b[j] = (b[j] - m) / s;
  }
}

// NCHW fp32 maps [B][F][T] -> channel c of the haloed NHWC input
template <typename T>
__global__ void stage_map_kernel(const float* __restrict__ src, long sbs, T* __restrict__ dst, long ld, int c, int B, int H,
                                 int W) {
  const long total = (long)B * H * W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % W), y = (int)((i / W) % H), b = (int)(i / ((long)W * H));
    dst[(((long)b * (H + 2) + y + 1) * (W + 2) + x + 1) * ld + c] = from_f32<T>(src[(long)b * sbs + (long)y * W + x]);
  }
}

// mode 0: out[b][k] = raw[p][k] * (1 - mask[b][t])                           (pc_wrapper.py:78-83)
// mode 1: out[b][0] = x_in * mask + raw[p][0] * (1 - mask)                    (unet.py:305-312, single channel)
template <typename T>
__global__ void unet_out_kernel(const T* __restrict__ raw, long ld, const float* __restrict__ mask, const float* __restrict__ xin,
                                long xbs, float* __restrict__ out, long ps, int K, int B, int H, int W, int mode) {
  const long total = (long)B * K * H * W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % W), y = (int)((i / W) % H), k = (int)((i / ((long)W * H)) % K), b = (int)(i / ((long)W * H * K));
    const float m = mask[(long)b * W + x];
    const float r = to_f32<T>(raw[(((long)b * (H + 2) + y + 1) * (W + 2) + x + 1) * ld + k]);
    out[((long)b * K + k) * ps + (long)y * W + x] = mode ? xin[(long)b * xbs + (long)y * W + x] * m + r * (1.f - m) : r * (1.f - m);
  }
}

// dRaw[p][k] = dOut[b][k][y][x] * (1 - mask[b][x]),  columns K..ld-1 zero
template <typename T>
__global__ void unet_out_bwd_kernel(const float* __restrict__ dout, long ps, const float* __restrict__ mask, T* __restrict__ draw,
                                    long ld, int K, int B, int H, int W) {
  const long total = (long)B * H * W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % W), y = (int)((i / W) % H), b = (int)(i / ((long)W * H));
    const float m = 1.f - mask[(long)b * W + x];
    T* d = draw + (((long)b * (H + 2) + y + 1) * (W + 2) + x + 1) * ld;
    for (int k = 0; k < K; ++k) d[k] = from_f32<T>(dout[((long)b * K + k) * ps + (long)y * W + x] * m);
  }
}

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long n, double* __restrict__ out) {
  __shared__ double red[4];
  double s = 0.0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) s += (double)g[i] * (double)g[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

// rows per workgroup of the BatchNorm column reductions: every workgroup ends with 2*C fp64 atomics on the same
// addresses, so few large workgroups (about two per CU) beat many small ones (2000 workgroups: the atomics serialise)
static inline int bn_rows_per_block(long P) {
  long r = (P + 511) / 512;
  r = (r + 255) / 256 * 256;
  return (int)(r < 1024 ? 1024 : r);
}

static inline int grid_for(long total, int cap = 4096) {
  long g = (total + 255) / 256;
  return (int)(g > cap ? cap : (g < 1 ? 1 : g));
}

}  // namespace

// typed launch: KERNEL<bf16_t> or KERNEL<float> with the void* operands cast by the ARGS expression (uses `TT`)
#define LAUNCH_T(prec, KERNEL, GRID, ...)                                                        \
  do {                                                                                            \
    if ((prec) == NPPC_PREC_BF16) {                                                               \
      typedef bf16_t TT;                                                                          \
      hipLaunchKernelGGL(KERNEL<TT>, GRID, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__);       \
    } else if ((prec) == NPPC_PREC_F32) {                                                         \
      typedef float TT;                                                                           \
      hipLaunchKernelGGL(KERNEL<TT>, GRID, dim3(256), 0, (hipStream_t)stream, __VA_ARGS__);       \
    } else {                                                                                      \
      return NPPC_EBADARG;                                                                        \
    }                                                                                             \
    NPPC_CHECK_LAUNCH();                                                                          \
  } while (0)

extern "C" {

static int conv_fwd_impl(int prec, const void* A, long lda, const void* Wp, void* C, long ldc, const float* bias, const float* scale,
                         const float* shift, float slope, int B, int H, int W, int Cin, int Cout, int Np, int ksize, float* stat_part,
                         void* stream) {
  if (!A || !Wp || !C || B <= 0 || H <= 0 || W <= 0) return NPPC_EBADARG;
  if (Cin % 32 || Np % 64 || Cout > Np || (ksize != 1 && ksize != 3) || (scale && !shift)) return NPPC_EUNSUPPORTED;
  if ((long)B * (H + 2) * (W + 2) >= (1L << 31)) return NPPC_EUNSUPPORTED;
  ConvArgs g;
  g.A = A; g.lda = lda; g.Wp = Wp; g.C = C; g.ldc = ldc; g.bias = bias; g.scale = scale; g.shift = shift; g.slope = slope;
  g.P = (long)B * (H + 2) * (W + 2);
  g.H = H; g.W = W; g.Cin = Cin; g.Cout = Cout; g.ntap = ksize * ksize;
  g.stat_part = stat_part; g.ldp = Np;
  const int bk = prec == NPPC_PREC_BF16 ? 64 : 32;
  hipStream_t st = (hipStream_t)stream;
  if (stat_part && !(Cin % bk == 0 && (prec == NPPC_PREC_BF16 || prec == NPPC_PREC_F32))) return NPPC_EUNSUPPORTED;   // tiled kernel only
  if (Cin % bk == 0 && (prec == NPPC_PREC_BF16 || prec == NPPC_PREC_F32)) {
    // bf16: the LDS-DMA ring kernel (256-row tiles, one workgroup per CU, software-pipelined fragments, producer waves) takes the
    // layers whose channel count gives 128-wide tiles (default, NPPC_CONV_DMA unset or 3): bit-identical to the register-staged
    // kernel (tests/test_inpaint_gpu.py), 5-7 % faster on the >= 128-channel layers of C3, step 26.05 -> 25.85 ms.  The
    // 64-channel tiles stay on the register-staged kernel (the ring kernel is 20-28 % slower there: one 8-wave workgroup per CU
    // against two 4-wave ones).  NPPC_CONV_DMA=0: register-staged everywhere; 1: ring kernel wherever its grid fills the chip
    // (profiles/r04_c3_conv_dma_ab.txt).
    const char* cd = getenv("NPPC_CONV_DMA");
    if (!cd || !cd[0]) cd = "3";
    const bool use_dma = cd[0] == '1' || (cd[0] == '3' && Np % 128 == 0);
    if (use_dma && prec == NPPC_PREC_BF16 && Cin % 64 == 0) {
      const int bn = Np % 128 == 0 ? 128 : 64;
      const long tiles = (long)ceil_div(g.P, 256) * (Np / bn);
      const char* mt = getenv("NPPC_CONV_DMA_MIN_TILES");          // (tests lower it to run small, ragged shapes through the ring kernel)
      if (tiles >= (mt ? atol(mt) : 256L)) {
        dim3 grid(round_up(ceil_div(g.P, 256), 8), Np / bn);
        if (bn == 128) {
          constexpr int smem = 3 * (256 + 128) * 128;
          static bool attr = false;
          if (!attr) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_dma_kernel<128, 3>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    smem) != hipSuccess)
              return NPPC_ELAUNCH;
            attr = true;
          }
          hipLaunchKernelGGL((conv_dma_kernel<128, 3>), grid, dim3(640), smem, st, g);
        } else {
          constexpr int smem = 3 * (256 + 64) * 128;
          static bool attr = false;
          if (!attr) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(conv_dma_kernel<64, 3>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    smem) != hipSuccess)
              return NPPC_ELAUNCH;
            attr = true;
          }
          hipLaunchKernelGGL((conv_dma_kernel<64, 3>), grid, dim3(640), smem, st, g);
        }
        NPPC_CHECK_LAUNCH();
        return NPPC_OK;
      }
    }
    if (Np % 128 == 0) {
      dim3 grid(round_up(ceil_div(g.P, 128), 8), Np / 128);
      if (prec == NPPC_PREC_BF16) hipLaunchKernelGGL((conv_tiled_kernel<bf16_t, 128>), grid, dim3(256), 0, st, g);
      else hipLaunchKernelGGL((conv_tiled_kernel<float, 128>), grid, dim3(256), 0, st, g);
    } else {
      dim3 grid(round_up(ceil_div(g.P, 128), 8), Np / 64);
      if (prec == NPPC_PREC_BF16) hipLaunchKernelGGL((conv_tiled_kernel<bf16_t, 64>), grid, dim3(256), 0, st, g);
      else hipLaunchKernelGGL((conv_tiled_kernel<float, 64>), grid, dim3(256), 0, st, g);
    }
    NPPC_CHECK_LAUNCH();
    return NPPC_OK;
  }
  dim3 grid(ceil_div(g.P, 128), Np / 64);
  LAUNCH_T(prec, conv_kernel, grid, g);
  return NPPC_OK;
}

int nppc_conv_fwd(int prec, const void* A, long lda, const void* Wp, void* C, long ldc, const float* bias, const float* scale,
                  const float* shift, float slope, int B, int H, int W, int Cin, int Cout, int Np, int ksize, void* stream) {
  return conv_fwd_impl(prec, A, lda, Wp, C, ldc, bias, scale, shift, slope, B, H, W, Cin, Cout, Np, ksize, nullptr, stream);
}

/* the same convolution (no folded BatchNorm), which also leaves the per-tile column sums of its stored output in stat_part
 * ([ceil(P / 128)][2][Np] floats, P = B (H+2) (W+2)): nppc_bn_stats_from_parts turns them into the batch statistics of the
 * train-mode BatchNorm that follows (tmp_utils.py:8-37) without a pass over the tensor.  Tiled kernel shapes only
 * (Cin % 64 == 0 in bf16, % 32 in fp32): NPPC_EUNSUPPORTED otherwise. */
int nppc_conv_fwd_stats(int prec, const void* A, long lda, const void* Wp, void* C, long ldc, const float* bias, int B, int H, int W,
                        int Cin, int Cout, int Np, int ksize, float* stat_part, void* stream) {
  if (!stat_part) return NPPC_EBADARG;
  return conv_fwd_impl(prec, A, lda, Wp, C, ldc, bias, nullptr, nullptr, 0.f, B, H, W, Cin, Cout, Np, ksize, stat_part, stream);
}

/* st[c] = sum x, st[C + c] = sum x^2 (fp64, plain stores: no initial state) from the per-tile sums of nppc_conv_fwd_stats;
 * scratch: 2 * C * 128 doubles.  Fixed summation order: repeated runs are bit-identical. */
int nppc_bn_stats_from_parts(const float* stat_part, int B, int H, int W, int Np, int C, double* st, double* scratch, void* stream) {
  if (!stat_part || !st || !scratch || C <= 0 || C > Np || B <= 0) return NPPC_EBADARG;
  const long ntiles = ceil_div((long)B * (H + 2) * (W + 2), 128);
  const int NS = ntiles < 128 ? (int)ntiles : 128;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(bn_parts_reduce_kernel, dim3(NS, ceil_div(C, 64)), dim3(256), 0, s, stat_part, ntiles, Np, C, NS, scratch);
  hipLaunchKernelGGL(bn_parts_final_kernel, dim3(ceil_div(2 * C, 8)), dim3(256), 0, s, scratch, NS, C, st);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

/* thin ends of the U-Net (csrc/unet.hip "thin convolutions"): w are the fp32 parameter tensors in torch layout, no packing.
 * part: workspace of nppc_conv_thin_part_elems() floats. */
int nppc_conv_thin_part_elems(long* n) {
  if (!n) return NPPC_EBADARG;
  *n = 1024L * 64 * 9 * 2;       // 1024 blocks x the largest gradient tensor (64 x 2 x 3 x 3), or 2048 x the 1x1 head's 8 x 64
  return NPPC_OK;
}

int nppc_conv3x3_thin_fwd(int prec, const void* A, long lda, const float* w, const float* bias, const float* scale,
                          const float* shift, float slope, void* C, long ldc, int B, int H, int W, int Cin, int Cout, void* stream) {
  if (!A || !w || !C || B <= 0 || H <= 0 || W <= 0 || (scale && !shift)) return NPPC_EBADARG;
  if (Cin < 1 || Cin > 2 || Cout % 8 || Cout > 64 || 256 % (Cout / 8)) return NPPC_EUNSUPPORTED;
  const long P = (long)B * (H + 2) * (W + 2);
  if (P >= (1L << 31)) return NPPC_EUNSUPPORTED;
  const int ppb = 256 / (Cout / 8);
  const dim3 grid((unsigned)std::min<long>(ceil_div(P, (long)ppb * 4), 256L * 16));
  const size_t sm = sizeof(float) * 9 * Cin * Cout;
  hipStream_t st = (hipStream_t)stream;
#define THIN_FWD(TT, CI) hipLaunchKernelGGL((conv3x3_thin_fwd_kernel<TT, CI>), grid, dim3(256), sm, st, (const TT*)A, lda, w, bias, scale, shift, slope, (TT*)C, ldc, P, H, W, Cout)
  if (prec == NPPC_PREC_BF16) { if (Cin == 1) THIN_FWD(bf16_t, 1); else THIN_FWD(bf16_t, 2); }
  else if (prec == NPPC_PREC_F32) { if (Cin == 1) THIN_FWD(float, 1); else THIN_FWD(float, 2); }
  else return NPPC_EBADARG;
#undef THIN_FWD
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

int nppc_conv3x3_thin_wgrad(int prec, const void* dY, long lddy, const void* X, long ldx, float* part, float* dW, int B, int H,
                            int W, int Cin, int Cout, void* stream) {
  if (!dY || !X || !part || !dW || B <= 0 || H <= 0 || W <= 0) return NPPC_EBADARG;
  if (Cin < 1 || Cin > 2 || Cout % 8 || Cout > 64) return NPPC_EUNSUPPORTED;
  const long P = (long)B * (H + 2) * (W + 2);
  const int groups = 256 / (3 * (Cout / 8)), n = Cout * 9 * Cin;
  const int nblocks = (int)std::min<long>(ceil_div(P, (long)groups * 64), 1024L);
  const size_t sm = sizeof(float) * groups * n;
  hipStream_t st = (hipStream_t)stream;
#define THIN_WG(TT, CI) hipLaunchKernelGGL((conv3x3_thin_wgrad_kernel<TT, CI>), dim3(nblocks), dim3(256), sm, st, (const TT*)dY, lddy, (const TT*)X, ldx, part, P, W, Cout)
  if (prec == NPPC_PREC_BF16) { if (Cin == 1) THIN_WG(bf16_t, 1); else THIN_WG(bf16_t, 2); }
  else if (prec == NPPC_PREC_F32) { if (Cin == 1) THIN_WG(float, 1); else THIN_WG(float, 2); }
  else return NPPC_EBADARG;
#undef THIN_WG
  NPPC_CHECK_LAUNCH();
  hipLaunchKernelGGL(thin_wgrad_finish_kernel, dim3(ceil_div(n, 64)), dim3(1024), 0, st, part, nblocks, n, dW);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

int nppc_conv1x1_thin_fwd(int prec, const void* X, long ldx, const float* w, const float* bias, void* C, long ldc, int B, int H,
                          int W, int Cin, int K, void* stream) {
  if (!X || !w || !C || B <= 0 || H <= 0 || W <= 0) return NPPC_EBADARG;
  if (Cin != 64 || K < 1 || K > 8 || ldx % 8) return NPPC_EUNSUPPORTED;
  const long P = (long)B * (H + 2) * (W + 2);
  if (P >= (1L << 31)) return NPPC_EUNSUPPORTED;
  const dim3 grid((unsigned)std::min<long>(ceil_div(P, 32), 256L * 16));
  LAUNCH_T(prec, conv1x1_thin_fwd_kernel, grid, (const TT*)X, ldx, w, bias, (TT*)C, ldc, P, H, W, K);
  return NPPC_OK;
}

int nppc_conv1x1_thin_bwd_data(int prec, const void* dY, long lddy, const float* w, void* dX, long lddx, int B, int H, int W,
                               int Cin, int K, void* stream) {
  if (!dY || !w || !dX || B <= 0 || H <= 0 || W <= 0) return NPPC_EBADARG;
  if (Cin != 64 || K < 1 || K > 8 || lddx % 8 || lddy % 8) return NPPC_EUNSUPPORTED;
  const long P = (long)B * (H + 2) * (W + 2);
  if (P >= (1L << 31)) return NPPC_EUNSUPPORTED;
  const dim3 grid((unsigned)std::min<long>(ceil_div(P, 32), 256L * 16));
  LAUNCH_T(prec, conv1x1_thin_bwd_data_kernel, grid, (const TT*)dY, lddy, w, (TT*)dX, lddx, P, H, W, K);
  return NPPC_OK;
}

int nppc_conv1x1_thin_wgrad(int prec, const void* dY, long lddy, const void* X, long ldx, float* part, float* dW, int B, int H,
                            int W, int Cin, int K, void* stream) {
  if (!dY || !X || !part || !dW || B <= 0 || H <= 0 || W <= 0) return NPPC_EBADARG;
  if (Cin != 64 || K < 1 || K > 8 || ldx % 8 || lddy % 8) return NPPC_EUNSUPPORTED;
  const long P = (long)B * (H + 2) * (W + 2);
  const int nblocks = (int)std::min<long>(ceil_div(P, 32L * 16), 2048L);
  LAUNCH_T(prec, conv1x1_thin_wgrad_kernel, dim3(nblocks), (const TT*)dY, lddy, (const TT*)X, ldx, part, P, K);
  hipLaunchKernelGGL(thin_wgrad_finish_kernel, dim3(ceil_div(K * 64, 64)), dim3(1024), 0, (hipStream_t)stream, part, nblocks, K * 64, dW);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

int nppc_conv_pack(int prec, const float* w, void* wf, void* wb, int Cout, int Cin, int ksize, int Np, int Cinp, int Mb,
                   int Coutp, void* stream) {
  if (!w || !wf || Cout > Np || Cin > Cinp || (wb && (Cin > Mb || Cout > Coutp))) return NPPC_EBADARG;
  const int ntap = ksize * ksize;
  const long total = (long)Np * ntap * Cinp + (wb ? (long)Mb * ntap * Coutp : 0);
  LAUNCH_T(prec, conv_pack_kernel, dim3(grid_for(total)), w, (TT*)wf, (TT*)wb, Cout, Cin, ntap, Np, Cinp, Mb, Coutp);
  return NPPC_OK;
}

int nppc_bn_stats(int prec, const void* X, long ld, long P, int C, double* st, void* stream) {
  if (!X || !st || C <= 0 || C % 8 || C / 8 > 256) return NPPC_EBADARG;
  const int rpb = bn_rows_per_block(P);
  LAUNCH_T(prec, bn_stats_kernel, dim3(ceil_div(P, rpb)), (const TT*)X, ld, P, C, rpb, st);
  return NPPC_OK;
}

int nppc_bn_finalize(const double* st, const float* gamma, const float* beta, float* rmean, float* rvar, float* ss, int C,
                     double n, float eps, float momentum, int train, void* stream) {
  if (!gamma || !beta || !rmean || !rvar || !ss || (train && !st) || C <= 0) return NPPC_EBADARG;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, (hipStream_t)stream, st, gamma, beta, rmean, rvar,
                     ss, C, n, eps, momentum, train);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

int nppc_bn_act(int prec, const void* X, long ldx, void* Y, long ldy, const float* ss, int C, int B, int H, int W, float slope,
                void* stream) {
  if (!X || !Y || !ss || C % 8) return NPPC_EBADARG;
  const long P = (long)B * (H + 2) * (W + 2);
  if (P >= (1L << 31)) return NPPC_EUNSUPPORTED;
  LAUNCH_T(prec, bn_act_kernel, dim3(ceil_div(P * (C / 8), 256)), (const TT*)X, ldx, (TT*)Y, ldy, ss, C, P, H, W, slope);
  return NPPC_OK;
}

int nppc_bn_bwd(int prec, const void* dyA, long ldA, const void* dyB, long ldB, const void* Y, long ldy, const void* X, long ldx,
                const float* ss, double* S, void* dX, long lddx, float* dgamma, float* dbeta, int C, int B, int H, int W,
                float slope, void* stream) {
  if (!dyA || !Y || !X || !ss || !S || !dX || !dgamma || !dbeta || C % 8 || C / 8 > 256) return NPPC_EBADARG;
  const long P = (long)B * (H + 2) * (W + 2);
  if (P >= (1L << 31)) return NPPC_EUNSUPPORTED;
  const int rpb = bn_rows_per_block(P);
  const double n = (double)B * H * W;
  if (hipMemsetAsync(S, 0, sizeof(double) * 2 * C, (hipStream_t)stream) != hipSuccess) return NPPC_ELAUNCH;
  LAUNCH_T(prec, bn_bwd_reduce_kernel, dim3(ceil_div(P, rpb)), (const TT*)dyA, ldA, (const TT*)dyB, ldB, (const TT*)Y, ldy,
           (const TT*)X, ldx, ss, C, P, H, W, slope, rpb, S);
  LAUNCH_T(prec, bn_bwd_apply_kernel, dim3(ceil_div(P * (C / 8), 256)), (const TT*)dyA, ldA, (const TT*)dyB, ldB, (const TT*)Y,
           ldy, (const TT*)X, ldx, ss, (const double*)S, (TT*)dX, lddx, dgamma, dbeta, C, P, H, W, slope, n);
  return NPPC_OK;
}

int nppc_maxpool2(int prec, const void* X, long ldx, void* Y, long ldy, unsigned char* idx, int C, int B, int H, int W,
                  void* stream) {
  if (!X || !Y || !idx || C % 8 || H < 2 || W < 2) return NPPC_EBADARG;
  const long total = (long)B * (H / 2) * (W / 2) * (C / 8);
  LAUNCH_T(prec, maxpool_kernel, dim3(ceil_div(total, 256)), (const TT*)X, ldx, (TT*)Y, ldy, idx, C, B, H, W);
  return NPPC_OK;
}

int nppc_maxpool2_bwd(int prec, const void* dY, long ldy, const unsigned char* idx, void* dX, long ldx, int C, int B, int H, int W,
                      void* stream) {
  if (!dY || !dX || !idx || C % 8 || H < 2 || W < 2) return NPPC_EBADARG;
  const long total = (long)B * (H / 2) * (W / 2) * (C / 8);
  LAUNCH_T(prec, maxpool_bwd_kernel, dim3(ceil_div(total, 256)), (const TT*)dY, ldy, idx, (TT*)dX, ldx, C, B, H, W);
  return NPPC_OK;
}

int nppc_upsample2(int prec, const void* X, long ldx, void* Y, long ldy, int C, int B, int Hi, int Wi, int Ht, int Wt,
                   void* stream) {
  if (!X || !Y || C % 8 || Ht < 2 * Hi || Wt < 2 * Wi) return NPPC_EBADARG;
  const int py = (Ht - 2 * Hi) / 2, px = (Wt - 2 * Wi) / 2;
  const long total = (long)B * 4 * Hi * Wi * (C / 8);
  LAUNCH_T(prec, upsample_kernel, dim3(ceil_div(total, 256)), (const TT*)X, ldx, (TT*)Y, ldy, C, B, Hi, Wi, Ht, Wt, py, px);
  return NPPC_OK;
}

int nppc_upsample2_bwd(int prec, const void* dY, long ldy, void* dX, long ldx, int C, int B, int Hi, int Wi, int Ht, int Wt,
                       void* stream) {
  if (!dY || !dX || C % 8 || Ht < 2 * Hi || Wt < 2 * Wi) return NPPC_EBADARG;
  const int py = (Ht - 2 * Hi) / 2, px = (Wt - 2 * Wi) / 2;
  const long total = (long)B * Hi * Wi * (C / 8);
  LAUNCH_T(prec, upsample_bwd_kernel, dim3(ceil_div(total, 256)), (const TT*)dY, ldy, (TT*)dX, ldx, C, B, Hi, Wi, Ht, Wt, py, px);
  return NPPC_OK;
}

// Weight gradient of a ksize x ksize convolution:  slabs[t][s][m][n] = sum_{p in slice s} dY[p][m] * X[p + off(t)][n].
// bf16 with M % 128 == 0 (or M == 64: computed as 128 with the upper half discarded), N % 64 == 0 runs on the MFMA
// TN GEMM; everything else on the exact-f32 tile kernel.  Rows beyond P must be readable and dY zero there.
// slab orientation chosen by nppc_conv_wgrad (bf16, 3x3): 1 = slabs hold dW^T [N = Cin][M = Cout]
int nppc_conv_wgrad_transposed(int M, int N) { return (M % 128 != 0 && N % 128 == 0) ? 1 : 0; }

int nppc_conv_wgrad(int prec, const void* dY, long lddy, const void* X, long ldx, float* slabs, int M, int N, int B, int H, int W,
                    int ksize, int ksplit, void* stream) {
  if (!dY || !X || !slabs || M % 64 || N % 64 || ksplit < 1 || (ksize != 1 && ksize != 3)) return NPPC_EBADARG;
  const long P = (long)B * (H + 2) * (W + 2);
  const int ntap = ksize * ksize;
  const int Mr = (M + 127) / 128 * 128;
  const long slab_stride = (long)Mr * N;
  if (prec == NPPC_PREC_BF16) {
    const long R = (P + 64L * ksplit - 1) / (64L * ksplit) * (64L * ksplit);
    if (ksize == 3 && nppc_conv_wgrad_transposed(M, N))   // 64 output channels: put the wide side on the 128-row tile axis
      return nppc_gemm_tn_splitk_taps(X, ldx, dY, lddy, slabs, M, N, M, R, ksplit, W + 2, 1, stream);
    return ksize == 3 ? nppc_gemm_tn_splitk_taps(dY, lddy, X, ldx, slabs, N, Mr, N, R, ksplit, W + 2, 0, stream)
                      : nppc_gemm_tn_splitk(dY, lddy, X, ldx, slabs, N, Mr, N, R, ksplit, stream);
  }
  if (prec != NPPC_PREC_F32) return NPPC_EBADARG;
  for (int t = 0; t < ntap; ++t) {
    const long off = ksize == 3 ? (long)(t / 3 - 1) * (W + 2) + (t % 3 - 1) : 0;
    float* out = slabs + (size_t)t * ksplit * slab_stride;
    const long rps = ((P + ksplit - 1) / ksplit + 15) / 16 * 16;
    hipLaunchKernelGGL(wgrad_simple_kernel<float>, dim3(M / 64, N / 64, ksplit), dim3(256), 0, (hipStream_t)stream,
                       (const float*)dY, lddy, (const float*)X, ldx, off, out, (long)N, slab_stride, P, rps);
    NPPC_CHECK_LAUNCH();
  }
  return NPPC_OK;
}

int nppc_conv_wgrad_reduce(int prec, const float* slabs, int ksplit, int M, int N, float* dW, int Cout, int Cin, int ksize,
                           void* stream) {
  if (!slabs || !dW || Cout > M || Cin > N) return NPPC_EBADARG;
  const int Mr = (M + 127) / 128 * 128;
  const long total = (long)Cout * Cin * ksize * ksize;
  const int tr = prec == NPPC_PREC_BF16 && ksize == 3 && nppc_conv_wgrad_transposed(M, N);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, slabs, ksplit,
                     tr ? (long)N * M : (long)Mr * N, tr ? (long)M : (long)N, dW, Cout, Cin, ksize * ksize, tr);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

int nppc_logmag(const float* spec, float* out, long out_bstride, int B, long FT, double* st, void* stream) {
  if (!spec || !out || B <= 0 || FT <= 0 || out_bstride < FT) return NPPC_EBADARG;
  const long total = (long)B * FT;
  hipLaunchKernelGGL(logmag_kernel, dim3(grid_for(total, 1024)), dim3(256), 0, (hipStream_t)stream, spec, out, FT, total,
                     out_bstride, st);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

int nppc_standardize(float* a, float* b_or_null, long bstride, int B, long FT, const double* st, float* mean_std, void* stream) {
  const long total = (long)B * FT;
  if (!a || !st || total < 2 || bstride < FT) return NPPC_EBADARG;
  hipLaunchKernelGGL(standardize_kernel, dim3(grid_for(total, 1024)), dim3(256), 0, (hipStream_t)stream, a, b_or_null, total, FT,
                     bstride, st, (double)total, mean_std);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

int nppc_unet_stage_map(int prec, const float* src, long src_bstride, void* dst, long ld, int c, int B, int H, int W,
                        void* stream) {
  if (!src || !dst || c < 0 || c >= ld || src_bstride < (long)H * W) return NPPC_EBADARG;
  LAUNCH_T(prec, stage_map_kernel, dim3(grid_for((long)B * H * W)), src, src_bstride, (TT*)dst, ld, c, B, H, W);
  return NPPC_OK;
}

int nppc_unet_out(int prec, const void* raw, long ld, const float* mask, const float* xin, long xin_bstride, float* out,
                  long out_pstride, int K, int B, int H, int W, int mode, void* stream) {
  if (!raw || !mask || !out || (mode && (!xin || K != 1)) || out_pstride < (long)H * W) return NPPC_EBADARG;
  LAUNCH_T(prec, unet_out_kernel, dim3(grid_for((long)B * K * H * W)), (const TT*)raw, ld, mask, xin, xin_bstride, out,
           out_pstride, K, B, H, W, mode);
  return NPPC_OK;
}

int nppc_unet_out_bwd(int prec, const float* dout, long dout_pstride, const float* mask, void* draw, long ld, int K, int B,
                      int H, int W, void* stream) {
  if (!dout || !mask || !draw || K > ld || dout_pstride < (long)H * W) return NPPC_EBADARG;
  LAUNCH_T(prec, unet_out_bwd_kernel, dim3(grid_for((long)B * H * W)), dout, dout_pstride, mask, (TT*)draw, ld, K, B, H, W);
  return NPPC_OK;
}

int nppc_sumsq(const float* g, long n, double* out, void* stream) {
  if (!g || !out || n <= 0) return NPPC_EBADARG;
  hipLaunchKernelGGL(sumsq_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, (hipStream_t)stream, g, n, out);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

}  // extern "C"
