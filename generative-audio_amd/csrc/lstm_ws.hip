// Weight-STATIONARY cluster forward of the fused 2-layer LSTM (bf16, H = 384, input <= 64 columns):
// nn.LSTM(34, 384, num_layers=2) of audio_zen/model/module/sequence_model.py:113-123 for N independent sequences.
//
// The streaming kernels (lstm.hip, lstm_coop.hip) re-read their weight slice from L2 on every time step and are bound by
// that stream plus ~20 us of fixed per-step cost.  Here a cluster of 12 CUs holds ALL 3.7 MB of weights in REGISTERS for
// the whole launch: CU `cu` owns hidden units [32 cu, 32 cu + 32) of both layers; inside it waves 0-3 own 8 units each of
// layer 1 (28 k-steps x 4 VGPRs of A fragments) and waves 4-7 the same units of layer 2 (48 k-steps x 4 VGPRs = 192 of
// the 256 registers a wave has at two waves per SIMD).  The product is TRANSPOSED with respect to the streaming kernels:
// the MFMA A operand is the weight block [32 gate rows = 8 units x (i,g,f,o)][16 k], the B operand the activations
// [16 k][32 sequences], so a lane of the 32x32 accumulator holds the four gates of four units of ONE sequence and the
// cell update needs no cross-lane traffic.
//
// What moves instead of weights is the hidden state: a cluster walks its sequences in chunks of 32, round robin, and
// every CU gathers the full [x_s | h1_{s-1} | h2_{s-2}] tile of a chunk (32 x 832 bf16 = 52 KB) into LDS.  One item
// (chunk c, fused step s) computes layer 1 of time s AND layer 2 of time s-1 from that ONE tile (both only need h1_{s-1}
// and older state), so there is one hand-off per time step, not two, and h1 is gathered once.  A CU publishes 2 x 2 KB per
// item (its 32 units of h1_s and h2_{s-1}); the consumer of those bytes is item (c, s+1), nch - 1 items later: the
// hand-off latency is hidden by the other chunks of the cluster as long as nch >= ~4.
//
// Hand-off protocol (cdna_hip_programming.md Guideline 16, recipe R1; MI355X_MICROARCH visibility table, row 1):
//   producer: every payload store is a 16-byte write-through (sc1) buffer store; one item LATER every wave passes a counted
//             s_waitcnt that covers those stores, then the workgroup barrier, then ONE lane stores the chunk's epoch flag
//             (relaxed, agent scope).
//   consumer: each staging wave polls the 12 flags of the chunk (one sc1 load instruction, bounded spin) and only then
//             issues its sc1 buffer loads of the payload.
// Results never depend on placement or dispatch order; what must hold is co-residency of a cluster's 12 workgroups, which
// the launcher guarantees (one workgroup per CU: > 80 KB of LDS, grid <= CU count, occupancy checked).  Every spin is
// bounded; a time-out adds 1 to the sticky counter behind the flag words (same contract as lstm_coop.hip).
#include <stdlib.h>
#include "common.h"
#include "nppc_hip.h"

namespace {

typedef __attribute__((address_space(1))) unsigned int gu32;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef unsigned char lds_u8;   // LDS bytes (the compiler infers the address space from `smem`)

constexpr int WS_H = 384, WS_G = 12, WS_UC = 32, WS_KX = 64, WS_MC = 32, WS_NT = 512;
constexpr int WS_NK1 = (WS_KX + WS_H) / 16;              // 28 k-steps of 16: [x | h1]
constexpr int WS_NK2 = 2 * WS_H / 16;                    // 48 k-steps: [h1 | h2]
#ifndef WS_MFMA16
#define WS_MFMA16 0      // 1: 16x16x32 MFMAs (2 x 2 blocks per wave and k-step) instead of one 32x32x16 per k-step of 16
#endif
#if WS_MFMA16
constexpr int WS_ROWB = (WS_KX + 2 * WS_H) * 2 + 32;     // 1696 bytes per tile row (= 32 mod 64): conflict-free b128 reads of 16 rows x 4 k-groups
#else
constexpr int WS_ROWB = (WS_KX + 2 * WS_H) * 2 + 16;     // 1680 bytes per tile row: odd multiple of 16 -> conflict-free b128 reads
#endif
constexpr int WS_TILEB = WS_MC * WS_ROWB;                // 53760
constexpr int WS_OFF_BIAS = 2 * WS_TILEB;                // [2 layers][32 units][4 gates (i,g,f,o)] fp32
constexpr int WS_NKL = 4;                                // layer-2 A fragments (16 k each) that stay in LDS (register budget)
constexpr int WS_OFF_W2L = WS_OFF_BIAS + 2 * WS_UC * 16; // [4 ug][WS_NKL][64 lanes][8] bf16
constexpr int WS_OFF_CNT = WS_OFF_W2L + 4 * WS_NKL * 1024; // readers-done counter of the output staging (training), 16 bytes
constexpr int WS_OFF_STG = WS_OFF_CNT + 16;              // output staging [2 layers][32 seq][STROW] (inference: x 2 item parities)
constexpr int WS_STROW_INF = 64 + 16;                    // inference: the CU's 32 units of h (64 B) + pad
constexpr int WS_STROW_TRN = 256 + 64 + 64 + 16;         // training: gates (256 B) | c (64 B) | h (64 B) + pad
constexpr int ws_smem(bool train) { return WS_OFF_STG + (train ? 2 * WS_MC * WS_STROW_TRN : 2 * 2 * WS_MC * WS_STROW_INF); }
constexpr int WS_PF = 6;                                 // B fragments (LDS reads) in flight per wave
constexpr int WS_HB = WS_H * 2;                          // bytes of one h row
constexpr unsigned WS_SPIN_LIMIT = 1u << 22;
static_assert(ws_smem(false) > 80 * 1024 && ws_smem(true) <= 160 * 1024, "one workgroup per CU");

struct WsArgs {
  const void* x;        // [Tn][N][64] bf16
  const void* wp1;      // [12 cu][4 ug][28][64 lanes][8] bf16   (lstm_ws_pack_kernel)
  const void* wp2;      // [12 cu][4 ug][48][64 lanes][8]
  const float* bias1;   // [4][H] torch gate order (i,f,g,o), b_ih + b_hh
  const float* bias2;
  void* h1; void* h2;   // TRAIN: [Tn][N][H] (saved state AND exchange medium); else rings [2][N][H]
  void* g1; void* g2;   // TRAIN: [Tn][N][H][4] (i,g,f,o)
  void* c1; void* c2;   // TRAIN: [Tn][N][H]
  float* cst;           // fp32 cell state [clusters][12 cu][nch_max][2 layers][4 ug][64 lanes][4]
  unsigned* flags;      // [clusters][nch_max][16] epochs (zeroed by the launcher), then the sticky time-out counter
  const void* whp;      // head weights [16][H] bf16 row-major (rows >= O zero); nullptr: no head
  float* hpart;         // [Tn][N][O] fp32 head sums (bias and re-layout: nppc_sb_head_finalize with G = 1)
  long N; int Tn; int O; int clusters; int nch_max; int nchunks;
  int prio;             // 1: the layer-2 waves (the longer GEMM) run at raised priority
};

// workgroup -> (cluster, cu).  Blocks are dealt round-robin to the 8 XCDs (b % 8 labels the blocks that share one); with the
// full grid of 256 the first 24 blocks of every XCD form two whole clusters (their gather re-reads hit that XCD's L2) and
// the remaining 8 per XCD are pooled into five more.  Placement is a speed matter only.
__device__ __forceinline__ bool ws_ids(int clusters, int& cluster, int& cu) {
  const int b = blockIdx.x;
  if (gridDim.x == 256) {
    const int xcd = b & 7, slot = b >> 3;
    if (slot < 24) {
      cluster = xcd * 2 + slot / WS_G;
      cu = slot % WS_G;
    } else {
      const int p = xcd * 8 + slot - 24;
      cluster = 16 + p / WS_G;
      cu = p % WS_G;
    }
  } else {
    cluster = b / WS_G;
    cu = b % WS_G;
  }
  return cluster < clusters;
}

__device__ __forceinline__ unsigned pack2(float a, float b) { return (unsigned)f2bf(a) | ((unsigned)f2bf(b) << 16); }

// LDS-only workgroup barrier: does not drain the wave's global stores (the publish stores of THIS item stay in flight)
#ifdef WS_DIAG_NOBARRIER
__device__ __forceinline__ void ws_barrier() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }   // diagnostic: waves run free
#else
__device__ __forceinline__ void ws_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#endif

// diagnostic phase timers (tools/diag/stamp_ws.py builds with -DWS_STAMP): wave 0 (layer 1) and wave 4 (layer 2) of the
// workgroup (cluster 0, cu 0) accumulate cycles per phase and leave them behind the time-out words of the flag block
#ifdef WS_STAMP
#define WST_INIT unsigned long long st_last = __builtin_readcyclecounter(), st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define WST(i) if (st_on) { __builtin_amdgcn_sched_barrier(0); const unsigned long long nw = __builtin_readcyclecounter(); st_acc[i] += nw - st_last; st_last = nw; __builtin_amdgcn_sched_barrier(0); }
#define WST_FINI(base) if (st_on && lane == 0) { for (int i_ = 0; i_ < 8; ++i_) ((unsigned long long*)(a.flags + (size_t)a.clusters * a.nch_max * 16 + 4))[(base) + i_] = st_acc[i_]; }
#else
#define WST_INIT
#define WST(i)
#define WST_FINI(base)
#endif

struct WsItem {
  int s, c;             // fused step, local chunk
  int row0;             // first sequence of the chunk
};

// per-time-slot buffer descriptor: base + slot * N * row_bytes, N * row_bytes records (rows >= N and invalid times: range
// check drops stores and returns zeros for loads)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ws_rsrc(const void* base, long slot, long N, int row_bytes, bool valid) {
  const char* p = reinterpret_cast<const char*>(base) + (size_t)slot * N * row_bytes;
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p), 0, valid ? (unsigned)(N * row_bytes) : 0u, 0x00020000);
}

// cell update of one layer for this lane's 4 cells.  acc[4m + g]: cell m, gate g in (i, g, f, o).  32x32 MFMA: cell m = unit
// 4 hh + m of sequence lane & 31; 16x16 MFMAs: cell m = 2 sb + rb = unit 2 q + rb of sequence 16 sb + (lane & 15).  Either
// way cells 2k, 2k + 1 are neighbouring units of one sequence: hpk[k] / gpk[k] / cpk[k] pack those pairs.
template <bool TRAIN>
__device__ __forceinline__ void ws_cell(const f32x16& acc, f32x4& c, unsigned (&hpk)[2], u32x4 (&gpk)[2], unsigned (&cpk)[2]) {
  float hv[4];
#ifdef WS_DIAG_NOCELL
  // diagnostic build (timing only, results are garbage): no transcendentals, a handful of VALU instructions
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    c[m] = acc[4 * m] + acc[4 * m + 1] * c[m];
    hv[m] = acc[4 * m + 2] + acc[4 * m + 3];
    if (TRAIN) { gpk[m >> 1][2 * (m & 1)] = pack2(acc[4 * m], acc[4 * m + 1]); gpk[m >> 1][2 * (m & 1) + 1] = pack2(acc[4 * m + 2], acc[4 * m + 3]); }
  }
  hpk[0] = pack2(hv[0], hv[1]); hpk[1] = pack2(hv[2], hv[3]);
  if (TRAIN) { cpk[0] = pack2(c[0], c[1]); cpk[1] = pack2(c[2], c[3]); }
  return;
#endif
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const float iv = sigmoid_f(acc[4 * m + 0]);
    const float gv = tanh_f(acc[4 * m + 1]);
    const float fv = sigmoid_f(acc[4 * m + 2]);
    const float ov = sigmoid_f(acc[4 * m + 3]);
    const float cn = fv * c[m] + iv * gv;
    c[m] = cn;
    hv[m] = ov * tanh_f(cn);
    if (TRAIN) {
      gpk[m >> 1][2 * (m & 1)] = pack2(iv, gv);
      gpk[m >> 1][2 * (m & 1) + 1] = pack2(fv, ov);
    }
  }
  hpk[0] = pack2(hv[0], hv[1]);
  hpk[1] = pack2(hv[2], hv[3]);
  if (TRAIN) {
    cpk[0] = pack2(c[0], c[1]);
    cpk[1] = pack2(c[2], c[3]);
  }
}

// Outputs of an item leave through LDS: behind its cell update every wave drops its 8 units x 32 sequences into the layer's
// staging rows (ws_stage_out), the item's barrier follows, and at the top of the NEXT iteration wave w stores rows [4 w, 4 w + 4)
// of both layers as whole 64 / 256-byte row segments (ws_flush).  Stored straight from the accumulator lanes, every store
// instruction touched 32 different 128-byte lines with 16-32 bytes each, and the vector-memory path is bound by lines, not
// bytes (1.8k of 6.6k cycles per item in training).  h leaves write-through (sc1: the hand-off payload), gates and c
// non-temporal.  The flag of the item still rises behind the NEXT barrier, as before: the stores are now the oldest of that
// iteration instead of the youngest of the previous one.
// Inference stages into two buffers by item parity (the barrier between an item's flush and the second-next item's writes
// orders them); training has room for one: its writers wait for the readers-done counter (4 x generations flushed), which the
// flushing waves bump behind their reads -- LDS operations execute in issue order -- and which practically never waits.
template <bool TRAIN>
__device__ __forceinline__ unsigned char* ws_stg(unsigned char* smem, int layer, int par) {
  return smem + WS_OFF_STG + (TRAIN ? layer * WS_MC * WS_STROW_TRN : (par * 2 + layer) * WS_MC * WS_STROW_INF);
}
template <bool TRAIN>
__device__ __forceinline__ void ws_stage_out(unsigned char* smem, int layer, int ug, int lane, int i, const unsigned (&hpk)[2],
                                             const u32x4 (&gpk)[2], const unsigned (&cpk)[2]) {
  constexpr int STROW = TRAIN ? WS_STROW_TRN : WS_STROW_INF;
  constexpr int HOFF = TRAIN ? 320 : 0;
  asm volatile("" : "+v"(lane));        // re-derive the addresses below per item: hoisted, they would live in VGPRs across the GEMMs
  if (TRAIN) {
    volatile unsigned* cnt = reinterpret_cast<volatile unsigned*>(smem + WS_OFF_CNT);
    for (unsigned spins = 0; *cnt < 4u * (unsigned)(i + 1) && spins < (1u << 20); ++spins) {}   // generation i - 1 has been read
  }
#if WS_MFMA16
  const int n = lane & 15, q = lane >> 4;
#pragma unroll
  for (int sb = 0; sb < 2; ++sb) {      // cells 2 sb, 2 sb + 1: units 2 q, 2 q + 1 of sequence 16 sb + n
    unsigned char* wrow = ws_stg<TRAIN>(smem, layer, i & 1) + (16 * sb + n) * STROW;
    if (TRAIN) {
      *reinterpret_cast<u32x4*>(wrow + ug * 64 + q * 16) = gpk[sb];
      *reinterpret_cast<unsigned*>(wrow + 256 + ug * 16 + q * 4) = cpk[sb];
    }
    *reinterpret_cast<unsigned*>(wrow + HOFF + ug * 16 + q * 4) = hpk[sb];
  }
#else
  const int seq = lane & 31, hh = lane >> 5;
  unsigned char* wrow = ws_stg<TRAIN>(smem, layer, i & 1) + seq * STROW;
  if (TRAIN) {
    *reinterpret_cast<u32x4*>(wrow + ug * 64 + hh * 32) = gpk[0];
    *reinterpret_cast<u32x4*>(wrow + ug * 64 + hh * 32 + 16) = gpk[1];
    *reinterpret_cast<u32x2*>(wrow + 256 + ug * 16 + hh * 8) = u32x2{cpk[0], cpk[1]};
  }
  *reinterpret_cast<u32x2*>(wrow + HOFF + ug * 16 + hh * 8) = u32x2{hpk[0], hpk[1]};
#endif
}
// flush of staging generation g (iteration g): layer-2 outputs of item g, layer-1 outputs of item g - 1 (layer 1 runs its
// cell update one iteration late, see the kernel).  Done by the four layer-2 waves, wave ug rows [8 ug, 8 ug + 8) of both
// layers: reads first (ws_flush_read: LDS -> registers, bumps the readers-done counter), then the stores
// (ws_flush_store: per layer 1 (h) + TRAIN 3 (gates x 2, c) store instructions, on every path).
struct WsFlush { u32x4 g[2][2]; u32x4 c[2]; u32x4 h[2]; };
template <bool TRAIN>
__device__ __forceinline__ void ws_flush_read(WsFlush& f, unsigned char* smem, int ug, int lane, int gen) {
  constexpr int STROW = TRAIN ? WS_STROW_TRN : WS_STROW_INF;
  constexpr int HOFF = TRAIN ? 320 : 0;
  asm volatile("" : "+v"(lane));        // (see ws_stage_out)
#pragma unroll
  for (int layer = 0; layer < 2; ++layer) {
    const unsigned char* stg = ws_stg<TRAIN>(smem, layer, gen & 1);
    if (TRAIN) {
#pragma unroll
      for (int k = 0; k < 2; ++k)
        f.g[layer][k] = *reinterpret_cast<const u32x4*>(stg + (8 * ug + 4 * k + (lane >> 4)) * STROW + (lane & 15) * 16);
    }
    const int r = 8 * ug + ((lane & 31) >> 2), col = lane & 3;
    if (TRAIN) f.c[layer] = *reinterpret_cast<const u32x4*>(stg + r * STROW + 256 + col * 16);
    f.h[layer] = *reinterpret_cast<const u32x4*>(stg + r * STROW + HOFF + col * 16);
  }
  if (TRAIN && lane == 0) atomicAdd(reinterpret_cast<unsigned*>(smem + WS_OFF_CNT), 1u);
}
template <bool TRAIN>
__device__ __forceinline__ void ws_flush_store(const WsFlush& f, int layer, int ug, int lane, int row0, int cu, __amdgpu_buffer_rsrc_t hr,
                                               __amdgpu_buffer_rsrc_t gr, __amdgpu_buffer_rsrc_t cr) {
  asm volatile("" : "+v"(lane));
  if (TRAIN) {
#pragma unroll
    for (int k = 0; k < 2; ++k)
      __builtin_amdgcn_raw_buffer_store_b128(f.g[layer][k], gr,
                                             (row0 + 8 * ug + 4 * k + (lane >> 4)) * (WS_H * 8) + cu * 256 + (lane & 15) * 16, 0, 2);
  }
  const int r = 8 * ug + ((lane & 31) >> 2), col = lane & 3;
  const int goff = lane < 32 ? (row0 + r) * WS_HB + cu * 64 + col * 16 : 0x7ffffff0;     // lanes 32-63: out of range, dropped
  if (TRAIN) __builtin_amdgcn_raw_buffer_store_b128(f.c[layer], cr, goff, 0, 2);
  __builtin_amdgcn_raw_buffer_store_b128(f.h[layer], hr, goff, 0, 16);
}

__device__ __forceinline__ bf16x8 lds_frag(const lds_u8* p) { return *reinterpret_cast<const bf16x8*>(p); }
#if WS_MFMA16
// one k-step of 32 on the 2 x 2 blocks of 16x16: a[rb] = weights (gate rows 16 rb ..), b[sb] = activations (sequences 16 sb ..);
// accumulator element block m = 2 sb + rb (see ws_cell)
__device__ __forceinline__ void ws_mfma16x4(f32x16& acc, const bf16x8& a0, const bf16x8& a1, const bf16x8& b0, const bf16x8& b1) {
  f32x4 c[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) c[m] = f32x4{acc[4 * m], acc[4 * m + 1], acc[4 * m + 2], acc[4 * m + 3]};
  c[0] = mma16(a0, b0, c[0]);
  c[1] = mma16(a1, b0, c[1]);
  c[2] = mma16(a0, b1, c[2]);
  c[3] = mma16(a1, b1, c[3]);
#pragma unroll
  for (int m = 0; m < 4; ++m) { acc[4 * m] = c[m][0]; acc[4 * m + 1] = c[m][1]; acc[4 * m + 2] = c[m][2]; acc[4 * m + 3] = c[m][3]; }
}
#endif
#ifdef WS_DIAG_NOMFMA
// diagnostic build: the matrix pipe is not used (one VALU instruction keeps the operands alive)
__device__ __forceinline__ f32x16 ws_mfma(const bf16x8& a, const bf16x8& b, f32x16 c) {
  c[0] += __builtin_bit_cast(f32x4, a)[0] * __builtin_bit_cast(f32x4, b)[0];
  return c;
}
#else
__device__ __forceinline__ f32x16 ws_mfma(const bf16x8& a, const bf16x8& b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
#endif

template <bool TRAIN>
__global__ __launch_bounds__(WS_NT, 2) void lstm2_ws_fwd_kernel(WsArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int cluster, cu;
  if (!ws_ids(a.clusters, cluster, cu)) return;
  const int nch = (a.nchunks - cluster + a.clusters - 1) / a.clusters;   // chunks cluster, cluster + clusters, ...
  if (nch <= 0) return;
  const int layer = wave >> 2, ug = wave & 3;
  const long N = a.N;
  const int Tn = a.Tn;
  const int nsteps = Tn + (a.whp ? 2 : 1);
  const int nitems = nch * nsteps;
  const int seq = lane & 31, hh = lane >> 5;
  const int unit0 = cu * WS_UC + ug * 8 + 4 * hh;          // this lane's 4 units

  // ---- one-time LDS set-up: biases of the own 32 units, head weights as A fragments
  {
    float* bl = reinterpret_cast<float*>(smem + WS_OFF_BIAS);
    for (int e = tid; e < 2 * WS_UC * 4; e += WS_NT) {
      const int g = e & 3, u = (e >> 2) % WS_UC, l = e / (4 * WS_UC);
      const int tg = g == 0 ? 0 : (g == 1 ? 2 : (g == 2 ? 1 : 3));
      bl[e] = (l ? a.bias2 : a.bias1)[tg * WS_H + cu * WS_UC + u];
    }
    if (tid < 4) reinterpret_cast<unsigned*>(smem + WS_OFF_CNT)[tid] = 0u;
  }
  gu32* flags = (gu32*)(a.flags + (size_t)cluster * a.nch_max * 16);
  gu32* tmo = (gu32*)(a.flags + (size_t)a.clusters * a.nch_max * 16);
  const lds_u8* ltile = reinterpret_cast<const lds_u8*>(smem);
  // B fragment (activations) of this lane: row seq, k = 8 hh + j of each 16-wide k-step
#if WS_MFMA16
  // 16x16x32: lane (n = lane & 15, q = lane >> 4) reads row 16 sb + n, k = 32 kk + 8 q + j
  const int boff = (lane & 15) * WS_ROWB + (lane >> 4) * 16 + (layer ? WS_KX * 2 : 0);
  const int bias_unit = ug * 8 + 2 * (lane >> 4);                       // cells 2 sb + rb: unit 2 q + rb
#else
  const int boff = seq * WS_ROWB + hh * 16 + (layer ? WS_KX * 2 : 0);
#endif
  // fp32 cell state of (chunk, layer, unit group): 16 bytes per lane, private to this wave (plain accesses, L2-resident);
  // addressed through a descriptor: scalar chunk offset + the lane offset, no 64-bit pointer per lane
  const __amdgpu_buffer_rsrc_t cstr = __builtin_amdgcn_make_buffer_rsrc(
      a.cst + (((size_t)cluster * WS_G + cu) * a.nch_max * 2 + layer) * 4 * 256 + ug * 256, 0, (unsigned)a.nch_max * 2 * 4 * 256 * 4, 0x00020000);

  // accumulators start from the biases of the lane's cells (LDS: [layer][32 units][i,g,f,o])
  auto bias_init = [&](f32x16& acc, int l) {
#if WS_MFMA16
    const lds_u8* bl = reinterpret_cast<const lds_u8*>(smem) + WS_OFF_BIAS + (l * WS_UC + bias_unit) * 16;
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(bl + rb * 16);
#pragma unroll
      for (int sb = 0; sb < 2; ++sb) {
        const int m = 2 * sb + rb;
        acc[4 * m] = b[0]; acc[4 * m + 1] = b[1]; acc[4 * m + 2] = b[2]; acc[4 * m + 3] = b[3];
      }
    }
#else
    const lds_u8* bl = reinterpret_cast<const lds_u8*>(smem) + WS_OFF_BIAS + (l * WS_UC + ug * 8 + 4 * hh) * 16;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const f32x4 b = *reinterpret_cast<const f32x4*>(bl + m * 16);
      acc[4 * m] = b[0]; acc[4 * m + 1] = b[1]; acc[4 * m + 2] = b[2]; acc[4 * m + 3] = b[3];
    }
#endif
  };
  auto mk_item = [&](int s_, int c_) {
    WsItem it;                                                     // readfirstlane: provably wave-uniform (SGPRs, no waterfall
    it.s = __builtin_amdgcn_readfirstlane(s_);                     // loops around the buffer accesses that take them as offsets)
    it.c = __builtin_amdgcn_readfirstlane(c_);
    it.row0 = __builtin_amdgcn_readfirstlane((cluster + c_ * a.clusters) * WS_MC);
    return it;
  };
  auto next_item = [&](const WsItem& it) { return it.c + 1 < nch ? mk_item(it.s, it.c + 1) : mk_item(it.s + 1, 0); };
  auto slot = [&](int t) -> long { return TRAIN ? t : (t & 1); };
  auto cst_load = [&](const WsItem& it) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(cstr, lane * 16, it.c * 2 * 4 * 256 * 4, 0));
  };
  auto cst_store = [&](const WsItem& it, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), cstr, lane * 16, it.c * 2 * 4 * 256 * 4, 0);
  };

#ifdef WS_STAMP
  const bool st_on = cluster == 0 && cu == 0 && ug == 0;
#endif
  // flush of generation g: layer 1 staged item i1 (h1 / gates / c of time s), layer 2 item i2 (state of time s - 1)
  auto flush_store = [&](const WsFlush& f, const WsItem& i1, const WsItem& i2, bool v1, bool v2) {
    const bool a1 = v1 && i1.s < Tn, a2 = v2 && i2.s >= 1 && i2.s <= Tn;
    ws_flush_store<TRAIN>(f, 0, ug, lane, i1.row0, cu, ws_rsrc(a.h1, slot(i1.s), N, WS_HB, a1),
                          ws_rsrc(a.g1, i1.s, N, WS_H * 8, a1 && TRAIN), ws_rsrc(a.c1, i1.s, N, WS_HB, a1 && TRAIN));
    ws_flush_store<TRAIN>(f, 1, ug, lane, i2.row0, cu, ws_rsrc(a.h2, slot(i2.s - 1), N, WS_HB, a2),
                          ws_rsrc(a.g2, i2.s - 1, N, WS_H * 8, a2 && TRAIN), ws_rsrc(a.c2, i2.s - 1, N, WS_HB, a2 && TRAIN));
  };
  // Schedule of one iteration i (between two workgroup barriers), the two waves of a SIMD alternating between the matrix
  // pipe and the vector / memory pipes:
  //   layer-2 wave:  flush(i-1) | GEMM_2(i) ................... | poll, cell update_2(i), stage-out
  //   layer-1 wave:  gather(i+1) issue, cell update_1(i-1), stage-out, head | GEMM_1(i) | tile(i+1) -> LDS
  // i.e. layer 1 runs its cell update one iteration late (its accumulators survive the barrier): it overlaps the other wave's
  // GEMM instead of competing with it for the vector issue, and its own GEMM runs beside the other wave's cell update.
  // Outputs: staging generation i holds layer 2 of item i and layer 1 of item i - 1; the layer-2 waves flush it at the top of
  // iteration i + 1 and wait for those stores in front of that iteration's barrier: the stores of item j are complete before
  // barrier j + 2 and flag j rises behind it; the flags of item i + 2 are polled in iteration i: a cluster needs at least 5
  // chunks.
  if (layer == 0) {
    // =============================== layer-1 waves: gather + layer 1 + head ===============================
    bf16x8 w[WS_NK1];
    {
      const bf16_t* wp = reinterpret_cast<const bf16_t*>(a.wp1) + ((size_t)(cu * 4 + ug) * WS_NK1) * 512 + lane * 8;
#pragma unroll
      for (int ks = 0; ks < WS_NK1; ++ks) w[ks] = *reinterpret_cast<const bf16x8*>(wp + ks * 512);
      __builtin_amdgcn_s_waitcnt(0x0F70);                          // vmcnt(0): no per-item waits for the weight loads
    }
    // gather pattern: waves 0,1 take h1 rows [16 (ug & 1), +16), waves 2,3 the same rows of h2: 768 16-byte chunks =
    // 12 per lane, chunk idx = lane + 64 j -> (row idx / 48, column idx % 48); the pattern repeats every 3 j with 4 rows more
    int sv[3], dv[3];
#pragma unroll
    for (int jm = 0; jm < 3; ++jm) {
      const int idx = lane + 64 * jm;
      sv[jm] = (idx / 48) * WS_HB + (idx % 48) * 16;
      dv[jm] = (16 * (ug & 1) + idx / 48) * WS_ROWB + WS_KX * 2 + (ug >> 1) * WS_HB + (idx % 48) * 16;
    }
    const int xdv = (8 * ug + (lane >> 3)) * WS_ROWB + (lane & 7) * 16;

    u32x4 st[13];
    auto stage_write = [&](int buf) {
#ifdef WS_DIAG_NOTILEWRITE
      asm volatile("" :: "v"(st[0]), "v"(st[12]));
      return;
#endif
      unsigned char* t = smem + buf * WS_TILEB;
#pragma unroll
      for (int j = 0; j < 12; ++j) *reinterpret_cast<u32x4*>(t + dv[j % 3] + (j / 3) * 4 * WS_ROWB) = st[j];
      *reinterpret_cast<u32x4*>(t + xdv) = st[12];
    };

    __syncthreads();                                               // biases / head weights in LDS
    WsItem it = mk_item(0, 0), itn = next_item(it), itp = it, itpp = it;
    {
      // tile of item 0: x_0 and zeros
      const __amdgpu_buffer_rsrc_t xr = ws_rsrc(a.x, 0, N, WS_KX * 2, true);
#pragma unroll
      for (int j = 0; j < 12; ++j) st[j] = u32x4{0u, 0u, 0u, 0u};
      // (the row part of every offset sits in the VECTOR offset: the scalar offset of a raw buffer access is excluded from the
      // range check, and the rows >= N of a ragged last chunk must read as zeros, not as the next time slot's rows)
      st[12] = __builtin_amdgcn_raw_buffer_load_b128(xr, lane * 16 + (it.row0 + 8 * ug) * WS_KX * 2, 0, 0);
      stage_write(0);
    }
    f32x16 acc;                                                    // gates of the previous item (cell update pending)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    f32x4 ccp = {0.f, 0.f, 0.f, 0.f};                              // its cell state
    __syncthreads();
    WST_INIT
#pragma unroll 1
    for (int i = 0; i < nitems; ++i) {
      const int buf = i & 1;
      const bool more = i + 1 < nitems;
      WST(5)
      // ---- gather of the NEXT item (its flags were seen by wave 4 before the barrier just passed): 13 x 1 KB per wave, in
      // flight during the cell update and the GEMM below
      {
        const int th = (ug >> 1) ? itn.s - 2 : itn.s - 1;            // time of the h rows this wave gathers
        const __amdgpu_buffer_rsrc_t ghr = ws_rsrc((ug >> 1) ? a.h2 : a.h1, slot(th), N, WS_HB, more && th >= 0 && th < Tn);
        const __amdgpu_buffer_rsrc_t gxr = ws_rsrc(a.x, itn.s, N, WS_KX * 2, more && itn.s < Tn);
#ifdef WS_DIAG_GATHERL2
        const int gso = (16 * (ug & 1)) * WS_HB;      // diagnostic: always the same rows (L2 hits instead of freshly published lines)
#else
        const int gso = (itn.row0 + 16 * (ug & 1)) * WS_HB;
#endif
#ifndef WS_DIAG_NOGATHER
#pragma unroll
        for (int j = 0; j < 12; ++j) st[j] = __builtin_amdgcn_raw_buffer_load_b128(ghr, sv[j % 3] + gso + (j / 3) * 4 * WS_HB, 0, 16);
        st[12] = __builtin_amdgcn_raw_buffer_load_b128(gxr, lane * 16 + (itn.row0 + 8 * ug) * WS_KX * 2, 0, 0);
#else
#pragma unroll
        for (int j = 0; j < 13; ++j) st[j] = u32x4{(unsigned)gso, 0u, 0u, 0u};
#endif
      }
#ifdef WS_STAMP
      __builtin_amdgcn_sched_barrier(0);
      WST(6)
#endif
      // ---- cell update of item i - 1 (layer 1 of its time s), executed for every item: the steps behind the last time
      // step compute on zeros and are stored through an empty descriptor (no store sits in a conditional path)
      {
        unsigned hpk[2], cpk[2];
        u32x4 gpk[2];
        if (itp.s == 0) ccp = f32x4{0.f, 0.f, 0.f, 0.f};
        ws_cell<TRAIN>(acc, ccp, hpk, gpk, cpk);
        cst_store(itp, ccp);
        ws_stage_out<TRAIN>(smem, 0, ug, lane, i, hpk, gpk, cpk);
      }
#ifndef WS_STAMP
      // spread the 13 gather loads over the cell update (issued in one burst, the 52 loads of the four gathering waves fill the
      // CU's vector-memory queue and every wave blocks at the issue for most of the 830 cycles the 52 KB take at 64 B/clk)
#pragma unroll
      for (int j = 0; j < 13; ++j) {
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
      }
#endif
      __builtin_amdgcn_sched_barrier(0);
      WST(0)
      // ---- head of time s - 2 from the full h2_{s-2} in the tile: waves 2, 3 of the chunk's designated CU.  Its
      // (conditional) stores must be older than the gather below, so that the waits for the gather can be counted ones
      if (a.whp && ug >= 2 && it.s >= 2 && (it.c + it.s) % WS_G == cu) {
        const int n16 = lane & 15, q = lane >> 4, half = ug - 2;
        const lds_u8* bp = ltile + buf * WS_TILEB + (16 * half + n16) * WS_ROWB + (WS_KX + WS_H) * 2 + q * 16;
        const bf16_t* ap = reinterpret_cast<const bf16_t*>(a.whp) + (size_t)n16 * WS_H + 8 * q;   // A fragments: L2-resident, 12 KB
        f32x4 hacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 12; ++kk)
          hacc = mma16(*reinterpret_cast<const bf16x8*>(ap + kk * 32), *reinterpret_cast<const bf16x8*>(bp + kk * 64), hacc);
        const long hrow = it.row0 + 16 * half + n16;               // rows >= N: padding of the last (ragged) chunk
        float* hp = a.hpart + ((size_t)(it.s - 2) * N + hrow) * a.O;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (4 * q + j < a.O && hrow < N) __builtin_nontemporal_store(hacc[j], hp + 4 * q + j);
      }
      __builtin_amdgcn_sched_barrier(0);
      // cell state of THIS item (its update runs in the next iteration)
      f32x4 cc = cst_load(it);
      __builtin_amdgcn_sched_barrier(0);
      WST(1)
      {
        // ---- layer 1 of time s: gates = W1 . [x_s | h1_{s-1}]^T + b
        bias_init(acc, 0);
        const lds_u8* bp = ltile + buf * WS_TILEB + boff;
        bf16x8 b[WS_PF];
#if WS_MFMA16
        // B fragment e = 2 kk + sb: rows 16 sb .., k-step kk of 32; A fragments w[2 kk + rb]
        auto rdb = [&](int e) { return lds_frag(bp + (e & 1) * 16 * WS_ROWB + (e >> 1) * 64); };
#pragma unroll
        for (int e = 0; e < WS_PF; ++e) b[e] = rdb(e);
#pragma unroll
        for (int kk = 0; kk < WS_NK1 / 2; ++kk) {
          ws_mfma16x4(acc, w[2 * kk], w[2 * kk + 1], b[(2 * kk) % WS_PF], b[(2 * kk + 1) % WS_PF]);
          if (2 * kk + WS_PF < WS_NK1) b[(2 * kk) % WS_PF] = rdb(2 * kk + WS_PF);
          if (2 * kk + 1 + WS_PF < WS_NK1) b[(2 * kk + 1) % WS_PF] = rdb(2 * kk + 1 + WS_PF);
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 2 + WS_PF, 0);
#pragma unroll
        for (int kk = 0; kk < WS_NK1 / 2; ++kk) {
          __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
#else
#pragma unroll
        for (int d = 0; d < WS_PF; ++d) b[d] = lds_frag(bp + d * 32);
#pragma unroll
        for (int ks = 0; ks < WS_NK1; ++ks) {
          acc = ws_mfma(w[ks], b[ks % WS_PF], acc);
          if (ks + WS_PF < WS_NK1) b[ks % WS_PF] = lds_frag(bp + (ks + WS_PF) * 32);
        }
        // pin the software pipeline (left alone, the scheduler sinks every LDS read to just in front of its MFMA and the
        // wave pays the LDS latency once per two MFMAs): bias + first WS_PF reads, then one read behind each MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 4 + WS_PF, 0);
#pragma unroll
        for (int ks = 0; ks < WS_NK1; ++ks) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
#endif
      }
      __builtin_amdgcn_sched_barrier(0);
      WST(2)
      stage_write(buf ^ 1);
      ccp = cc;
      WST(3)
      ws_barrier();
      WST(4)
      if (tid == 0 && i >= 2)
        __hip_atomic_store(flags + itpp.c * 16 + cu, (unsigned)(itpp.s + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      itpp = itp;
      itp = it;
      it = itn;
      itn = next_item(itn);
    }
    // (the cell update of the last item is beyond the last time step: nothing to store)
    WST_FINI(0)
  } else {
    // =============================== layer-2 waves (wave 4 also polls) ===============================
    if (a.prio == 1) __builtin_amdgcn_s_setprio(1);                 // diagnostic switch (NPPC_WS_PRIO)
    constexpr int NKR = WS_NK2 - WS_NKL;                             // k-steps with register-resident weights: NKL .. 47
    bf16x8 w[NKR];
    {
      const bf16_t* wp = reinterpret_cast<const bf16_t*>(a.wp2) + ((size_t)(cu * 4 + ug) * WS_NK2) * 512 + lane * 8;
#pragma unroll
      for (int ks = 0; ks < WS_NKL; ++ks)                            // the first k-steps' A fragments live in LDS
        *reinterpret_cast<bf16x8*>(smem + WS_OFF_W2L + ((ug * WS_NKL + ks) * 64 + lane) * 16) =
            *reinterpret_cast<const bf16x8*>(wp + ks * 512);
#pragma unroll
      for (int ks = WS_NKL; ks < WS_NK2; ++ks) w[ks - WS_NKL] = *reinterpret_cast<const bf16x8*>(wp + ks * 512);
      __builtin_amdgcn_s_waitcnt(0x0F70);
    }
    const lds_u8* wl = reinterpret_cast<const lds_u8*>(smem) + WS_OFF_W2L + (ug * WS_NKL * 64 + lane) * 16;
    __syncthreads();
    WsItem it = mk_item(0, 0), itn = next_item(it), it2 = next_item(itn), itp = it, itpp = it;
    __syncthreads();
    WST_INIT
#pragma unroll 1
    for (int i = 0; i < nitems; ++i) {
      const int buf = i & 1;
      const int t = it.s - 1;
      WST(6)
      // wave 4: the flags of item i + 2 (its gather starts right behind this iteration's barrier): requested FIRST (older than
      // the flush stores below: the wait for them never covers a write-through store), looked at behind the GEMM.
      // Every CU of the cluster must have published item (c, s - 1).
      const bool polls = ug == 0 && i + 2 < nitems && it2.s >= 1;
      unsigned pv = 0xffffffffu;
      if (polls && lane < WS_G) pv = __hip_atomic_load(flags + it2.c * 16 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      f32x4 cc = cst_load(it);
      __builtin_amdgcn_sched_barrier(0);
      // ---- outputs of the previous generation: staging -> registers -> global, by the layer-2 waves (the shorter chain of
      // an iteration).  Oldest stores of the iteration: complete long before the counted wait in front of the barrier
      {
#ifndef WS_DIAG_NOFLUSH
        WsFlush fl;
        ws_flush_read<TRAIN>(fl, smem, ug, lane, i - 1);
        flush_store(fl, itpp, itp, i >= 2, i >= 1);
#else
        if (TRAIN && lane == 0) atomicAdd(reinterpret_cast<unsigned*>(smem + WS_OFF_CNT), 1u);
#endif
      }
      __builtin_amdgcn_sched_barrier(0);
      WST(0)
      // ---- layer 2 of time s - 1: gates = W2 . [h1_{s-1} | h2_{s-2}]^T + b (every item: see layer 1)
      f32x16 acc;
      bias_init(acc, 1);
      {
        // one ring of WS_PF LDS reads in flight feeds the MFMAs: operand stream W_0 B_0 .. W_{NKL-1} B_{NKL-1} B_NKL .. B_47
        // (the first NKL k-steps take their A fragment from LDS as well: 176 + 16 + 24 registers at the peak)
        const lds_u8* bp = ltile + buf * WS_TILEB + boff;
        constexpr int E = WS_NK2 + WS_NKL;
        bf16x8 r[WS_PF];
#if WS_MFMA16
        // operand stream: k-steps 0, 1 (of 32): W(kk,0) W(kk,1) B(kk,0) B(kk,1); then B(kk,0) B(kk,1); W2L holds fragments
        // 2 kk + rb of the first two k-steps, registers w[2 kk + rb - WS_NKL] the rest
        auto rd = [&](int e) {
          if (e < 2 * WS_NKL) {
            const int kk = e >> 2, j = e & 3;
            return j < 2 ? lds_frag(wl + (2 * kk + j) * 1024) : lds_frag(bp + (j & 1) * 16 * WS_ROWB + kk * 64);
          }
          const int f = e - WS_NKL;                                  // B fragment index 2 kk + sb
          return lds_frag(bp + (f & 1) * 16 * WS_ROWB + (f >> 1) * 64);
        };
#pragma unroll
        for (int e = 0; e < WS_PF; ++e) r[e] = rd(e);
#pragma unroll
        for (int kk = 0; kk < WS_NK2 / 2; ++kk) {
          if (2 * kk < WS_NKL) {
            const int e = 4 * kk;
            ws_mfma16x4(acc, r[e % WS_PF], r[(e + 1) % WS_PF], r[(e + 2) % WS_PF], r[(e + 3) % WS_PF]);
#pragma unroll
            for (int d = 0; d < 4; ++d)
              if (e + d + WS_PF < E) r[(e + d) % WS_PF] = rd(e + d + WS_PF);
          } else {
            const int e = WS_NKL + 2 * kk;
            ws_mfma16x4(acc, w[2 * kk - WS_NKL], w[2 * kk + 1 - WS_NKL], r[e % WS_PF], r[(e + 1) % WS_PF]);
#pragma unroll
            for (int d = 0; d < 2; ++d)
              if (e + d + WS_PF < E) r[(e + d) % WS_PF] = rd(e + d + WS_PF);
          }
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 2 + WS_PF, 0);
#pragma unroll
        for (int kk = 0; kk < WS_NK2 / 2; ++kk) {
          __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
          if (2 * kk < WS_NKL) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
          else __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
#else
        auto rd = [&](int e) {
          if (e < 2 * WS_NKL) return (e & 1) ? lds_frag(bp + (e >> 1) * 32) : lds_frag(wl + (e >> 1) * 1024);
          return lds_frag(bp + (e - WS_NKL) * 32);
        };
#pragma unroll
        for (int e = 0; e < WS_PF; ++e) r[e] = rd(e);
#pragma unroll
        for (int ks = 0; ks < WS_NK2; ++ks) {
          if (ks < WS_NKL) {
            const int e = 2 * ks;
            acc = ws_mfma(r[e % WS_PF], r[(e + 1) % WS_PF], acc);
            if (e + WS_PF < E) r[e % WS_PF] = rd(e + WS_PF);
            if (e + 1 + WS_PF < E) r[(e + 1) % WS_PF] = rd(e + 1 + WS_PF);
          } else {
            const int e = WS_NKL + ks;
            acc = ws_mfma(w[ks - WS_NKL], r[e % WS_PF], acc);
            if (e + WS_PF < E) r[e % WS_PF] = rd(e + WS_PF);
          }
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 4 + WS_PF, 0);   // (see layer 1)
#pragma unroll
        for (int ks = 0; ks < WS_NK2; ++ks) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (ks < WS_NKL) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
          else __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
#endif
      }
      WST(1)
      if (polls) {
        unsigned spins = 0;
        while (__builtin_amdgcn_ballot_w64(pv < (unsigned)it2.s) != 0) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > WS_SPIN_LIMIT) {
            if (lane == 0) __hip_atomic_fetch_add(tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
          if (lane < WS_G) pv = __hip_atomic_load(flags + it2.c * 16 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");     // compiler ordering only: the payload loads are sc1
      }
      __builtin_amdgcn_sched_barrier(0);
      WST(2)
      if (a.prio == 2) __builtin_amdgcn_s_setprio(3);
      {
        unsigned hpk[2], cpk[2];
        u32x4 gpk[2];
        if (t <= 0) cc = f32x4{0.f, 0.f, 0.f, 0.f};
        ws_cell<TRAIN>(acc, cc, hpk, gpk, cpk);
        WST(7)
        cst_store(it, cc);
        ws_stage_out<TRAIN>(smem, 1, ug, lane, i, hpk, gpk, cpk);
      }
      __builtin_amdgcn_sched_barrier(0);
      // the flush stores of this iteration (generation i - 1: layer 2 of item i - 1, layer 1 of item i - 2) must be complete
      // before the barrier, the flag of item i - 2 rises behind it: everything but this iteration's cell-state store
      if (a.prio == 2) __builtin_amdgcn_s_setprio(0);
      WST(3)
      asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
      WST(4)
      ws_barrier();
      WST(5)
      itpp = itp;
      itp = it;
      it = itn;
      itn = it2;
      it2 = next_item(it2);
    }
    {
      WsFlush fl;                                                  // the last generation (the last barrier has been passed)
      ws_flush_read<TRAIN>(fl, smem, ug, lane, nitems - 1);
      flush_store(fl, itpp, itp, true, true);
    }
    WST_FINI(8)
  }
}

// packed A fragments: element (cu, ug, ks, lane l, j): gate row r = l & 31 -> unit index r >> 2 = 2 m + hh' (the accumulator
// row map of the 32x32 MFMA), local unit = 4 hh' + m, gate g = r & 3 in (i,g,f,o); k = 16 ks + 8 (l >> 5) + j
//   layer 1: k < 64 -> W_ih[row][k] (0 for k >= I), else W_hh[row][k - 64]
//   layer 2: k < 384 -> W_ih[row][k] (input h1), else W_hh[row][k - 384]
__global__ void lstm_ws_pack_kernel(const float* __restrict__ w_ih, const float* __restrict__ w_hh, bf16_t* __restrict__ out, int I,
                                    int layer) {
  const int nk = layer == 1 ? WS_NK1 : WS_NK2;
  const size_t total = (size_t)WS_G * 4 * nk * 512;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int j = e & 7, l = (e >> 3) & 63;
    size_t f = e >> 9;
    const int ks = f % nk; f /= nk;
    const int ug = f & 3;
    const int cu = (int)(f >> 2);
    const int tgmap[4] = {0, 2, 1, 3};
#if WS_MFMA16
    // fragment ks = 2 kk + rb: A operand of the 16x16x32 MFMA, row r16 = l & 15 = 4 q' + gate -> unit 2 q' + rb
    const int kk = ks >> 1, rb = ks & 1, r16 = l & 15;
    const int ul = 2 * (r16 >> 2) + rb, g = r16 & 3;
    const int k = 32 * kk + 8 * (l >> 4) + j;
#else
    const int r = l & 31, g = r & 3, ui = r >> 2;
    const int ul = 4 * (ui & 1) + (ui >> 1);
    const int k = 16 * ks + 8 * (l >> 5) + j;
#endif
    const int row = tgmap[g] * WS_H + cu * WS_UC + ug * 8 + ul;
    float v;
    if (layer == 1) v = k < WS_KX ? (k < I ? w_ih[(size_t)row * I + k] : 0.f) : w_hh[(size_t)row * WS_H + (k - WS_KX)];
    else v = k < WS_H ? w_ih[(size_t)row * WS_H + k] : w_hh[(size_t)row * WS_H + (k - WS_H)];
    out[e] = f2bf(v);
  }
}

static int ws_fits(const void* kernel, int smem, int& per_cu_cache, int& n_cu) {
  static int ncu = 0;
  if (ncu == 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return NPPC_ELAUNCH;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return NPPC_ELAUNCH;
  }
  n_cu = ncu;
  if (per_cu_cache < 0 &&
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_cache, kernel, WS_NT, smem) != hipSuccess)
    return NPPC_ELAUNCH;
  return per_cu_cache >= 1 ? NPPC_OK : NPPC_EUNSUPPORTED;
}

}  // namespace

extern "C" {

// Plan of the weight-stationary forward for N sequences: clusters of 12 CUs and the largest number of 32-sequence chunks
// one cluster walks; 0 clusters = not applicable (the caller uses the streaming kernels).  Needs at least 5 chunks per
// cluster: the flag of item (c, s) is raised two items late and polled two items early, so with fewer chunks a cluster
// would wait for itself.  N need not be a multiple of 32 (round 4): the last chunk is ragged -- every access to a row
// >= N goes through a per-time-slot buffer descriptor of N rows (loads return 0, stores are dropped), the one plain store
// (the fused head) is guarded.
int nppc_lstm2_ws_plan(int prec, long N, int H, int I, int n_cu, int* clusters, int* nch_max) {
  *clusters = 0; *nch_max = 0;
  if (prec != NPPC_PREC_BF16 || H != WS_H || I > WS_KX || N <= 0) return NPPC_OK;
  const long nchunks = (N + WS_MC - 1) / WS_MC;
  long cl = n_cu / WS_G;
  if (cl > nchunks / 5) cl = nchunks / 5;
  if (cl < 1) return NPPC_OK;
  *clusters = (int)cl;
  *nch_max = (int)((nchunks + cl - 1) / cl);
  return NPPC_OK;
}

int nppc_lstm2_ws_packed_elems(long* n1, long* n2) {
  *n1 = (long)WS_G * 4 * WS_NK1 * 512;
  *n2 = (long)WS_G * 4 * WS_NK2 * 512;
  return NPPC_OK;
}

int nppc_lstm2_ws_pack(const float* w_ih0, const float* w_hh0, const float* w_ih1, const float* w_hh1, int I, void* wp1, void* wp2,
                       void* stream) {
  if (!w_ih0 || !w_hh0 || !w_ih1 || !w_hh1 || !wp1 || !wp2 || I > WS_KX) return NPPC_EBADARG;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(lstm_ws_pack_kernel, dim3(512), dim3(256), 0, s, w_ih0, w_hh0, (bf16_t*)wp1, I, 1);
  hipLaunchKernelGGL(lstm_ws_pack_kernel, dim3(512), dim3(256), 0, s, w_ih1, w_hh1, (bf16_t*)wp2, WS_H, 2);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

// Weight-stationary forward.  train = 0: h1 / h2 are [2][N][H] exchange rings, g / c unused; train = 1: h1 / h2 / g1 / g2 /
// c1 / c2 are the saved state in the layouts of nppc_lstm2_fwd.  cst: clusters * 12 * nch_max * 2048 floats; flags: clusters *
// nch_max * 16 + 4 u32 (epochs zeroed here, the sticky time-out counter is word clusters * nch_max * 16).  whp / hpart / O:
// fused output head (hpart [Tn][N][O] fp32 without bias), or whp = nullptr.
int nppc_lstm2_fwd_ws(int train, const void* x, const void* wp1, const void* wp2, const float* bias1, const float* bias2, void* h1,
                      void* h2, void* g1, void* g2, void* c1, void* c2, float* cst, unsigned* flags, const void* whp, float* hpart,
                      int O, long N, int Tn, int clusters, int nch_max, void* stream) {
  if (!x || !wp1 || !wp2 || !bias1 || !bias2 || !h1 || !h2 || !cst || !flags || N <= 0 || Tn <= 0 || clusters < 1 || nch_max < 1)
    return NPPC_EBADARG;
  if (train && (!g1 || !g2 || !c1 || !c2)) return NPPC_EBADARG;
  if (whp && (!hpart || O < 1 || O > 16)) return NPPC_EBADARG;
  const long nchunks = (N + WS_MC - 1) / WS_MC;
  if ((long)clusters * nch_max < nchunks || nchunks / clusters < 5 || N * (long)WS_H * 8 >= (1l << 31)) return NPPC_EBADARG;
  static const int prio = [] { const char* e = getenv("NPPC_WS_PRIO"); return e ? atoi(e) : 0; }();
  WsArgs a{x, wp1, wp2, bias1, bias2, h1, h2, g1, g2, c1, c2, cst, flags, whp, hpart, N, Tn, O, clusters, nch_max, (int)nchunks, prio};
  hipStream_t s = (hipStream_t)stream;
  const void* k = train ? reinterpret_cast<const void*>(lstm2_ws_fwd_kernel<true>)
                        : reinterpret_cast<const void*>(lstm2_ws_fwd_kernel<false>);
  const int smem = ws_smem(train != 0);
  if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess) return NPPC_ELAUNCH;
  static int per_cu[2] = {-1, -1};
  int n_cu = 0;
  const int fits = ws_fits(k, smem, per_cu[train ? 1 : 0], n_cu);
  if (fits != NPPC_OK) return fits;
  if (clusters * WS_G > n_cu) return NPPC_EUNSUPPORTED;         // every workgroup of a cluster must be resident
  if (hipMemsetAsync(flags, 0, (size_t)clusters * nch_max * 16 * sizeof(unsigned), s) != hipSuccess) return NPPC_ELAUNCH;
  const int grid = n_cu == 256 ? 256 : clusters * WS_G;
  if (train) hipLaunchKernelGGL(lstm2_ws_fwd_kernel<true>, dim3(grid), dim3(WS_NT), smem, s, a);
  else hipLaunchKernelGGL(lstm2_ws_fwd_kernel<false>, dim3(grid), dim3(WS_NT), smem, s, a);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

}  // extern "C"
