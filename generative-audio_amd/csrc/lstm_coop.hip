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
// Every spin is bounded: on time-out the workgroup adds 1 to the STICKY time-out counter -- the word right behind the
// epoch words of the caller's flag block -- and carries on (wrong numbers, no hang).  The launchers zero the epoch words
// only: the counter survives later launches and is cleared by the caller alone (ops_lstm.clear_coop_timeouts), so a host
// check at any later point (trainer log interval, end of bench) still sees a time-out of ANY earlier launch.
#include <type_traits>
#include <utility>
#include "common.h"
#include "nppc_hip.h"

namespace {

typedef __attribute__((address_space(1))) unsigned int gu32;

// Workgroup -> (cluster, cu).  Workgroups are dealt round-robin to the 8 XCDs in linear id order, each XCD with its own
// 4 MB L2.  For CU pairs (G = 2, grid a multiple of 8) every cu-0 workgroup is placed on XCDs 0-3 and every cu-1
// workgroup on XCDs 4-7: an XCD then caches only ONE half of the weights (1.85 MB instead of 3.7 MB), so the fragments
// that all its CUs re-read every step stay L2-resident.  Workgroups past the last cluster (grid rounding) exit at once.
__device__ __forceinline__ void coop_ids(int G, int& cluster, int& cu) {
  const int b = blockIdx.x;
  if (G == 2 && (gridDim.x & 7) == 0) {
    const int x = b & 7;
    cu = x >> 2;
    cluster = (b >> 3) * 4 + (x & 3);
  } else {
    cluster = b / G;
    cu = b % G;
  }
}

struct CoopArgs {
  const void* x;      // [Tn][N][KX]
  const void* wp1;    // packed like lstm.hip (pair-major, 8 "waves" x 48 units): re-indexed here per 16-unit block
  const void* wp2;
  const float* bias1;
  const float* bias2;
  void* h2;           // [Tn][N][H]
  void* h1; void* g1; void* g2; void* c1; void* c2;   // train: h1 [Tn][N][H] like h2
  void* xch;          // [clusters][2 layers][2 parities][G][MC][HC]  exchange slices
  unsigned* flags;    // [clusters][2 layers][G] epochs (zeroed before every launch), then the sticky time-out counter
  long N; int Tn; int clusters;
  // fused output head (inference): whp [16][H] row-major (rows >= O zero), hpart [G][Tn][N][O] fp32 partial sums of
  // sum_u h2[t][n][u] * Wh[o][u] over the units of CU g; whp == nullptr: h2 is stored instead
  const void* whp; float* hpart; int O;
  int x_ld;           // row stride of x in elements: KX, or a packed width (multiple of 8, >= I + 1): only the first x_ld columns of a
                      // row are fetched, the LDS columns behind them stay zero (round 4: the frozen restorer's input at 40 columns)
};

constexpr unsigned SPIN_LIMIT = 1u << 22;
#ifdef CF_NO_POLL
constexpr bool CF_POLL = false;                      // diagnostic: never wait for a partner (timing only, results are garbage)
#else
constexpr bool CF_POLL = true;
#endif

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
#ifdef CF_W_L1
  return (((((gp * 8 + w8) * nk + kk) * 3 + ub3) * 2 + s) * 512 * (int)sizeof(T)) & 0x3fff;   // diagnostic: weights from L1
#endif
  return ((((gp * 8 + w8) * nk + kk) * 3 + ub3) * 2 + s) * 512 * (int)sizeof(T);
}

// The first DEPTH-1 weight stages of a segment can be fetched EARLY (before the barrier / cell update / hand-off that
// precedes the GEMM) with coop_prime and handed to coop_gemm: the weight stream then never restarts cold.
template <typename T, int DEPTH>
__device__ __forceinline__ void coop_prime(typename Frag<T>::type (&pre)[DEPTH - 1][2], int n, int koff, int nk,
                                           __amdgpu_buffer_rsrc_t wr, int gp, int ublk, int lane) {
#pragma unroll
  for (int d = 0; d < DEPTH - 1; ++d)
    if (d < n) {
#pragma unroll
      for (int s = 0; s < 2; ++s) pre[d][s] = BFrag<T>::load(wr, lane, frag_boff<T>(gp, ublk, koff + d, s, nk));
    }
}

// Row padding of the LDS A tiles (bf16 elements).  An MFMA A fragment is one ds_read_b128 per lane: lane = (row n, 16-byte chunk
// q) at n * stride + 16 q.  CDNA4 serves a ds_read_b128 in four groups of 16 lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31},
// ... -- and a group is conflict-free when (n * s + q) mod 16 is a bijection on it, s = stride / 16 bytes: true for s = 2 (mod 4),
// for NO odd s.  With "+ 8 elements" (s = 1) rows 12 (q = 0) and 11 (q = 1) meet in every group: every A read takes two LDS
// cycles (SQ_LDS_BANK_CONFLICT = 0.48 of SQ_LDS_IDX_ACTIVE in all three recurrent kernels); with + 16 the conflicts are gone
// (0.06) -- and the kernels take the SAME time (forwards +-0.5 %, the K-split backward 1.5 % SLOWER: its 8-byte gate stores
// conflict more): the LDS array is not what these kernels wait for (profiles/r04_lstm_sq_counters.txt).  Default stays 8.
#ifndef COOP_APAD
#define COOP_APAD 8
#endif
#ifndef CF_H2_LATE
#define CF_H2_LATE 1  // forward: the partner's h2 slice is polled / requested behind layer 1's first GEMM (0 = round 3: at the top of the step)
#endif
#ifndef CF_APF
#define CF_APF 1      // 80-row forward: A-operand groups read this many groups ahead of their MFMAs (0 = round 3: read, wait, multiply)
#endif
#ifndef CF_AG
#define CF_AG 1       // row tiles (A fragments) per group
#endif
template <typename T, int MT, int RS, int DEPTH, bool PRIMED = false>
__device__ __forceinline__ void coop_gemm(f32x4 (&acc)[2][MT], const T* a_lane /* row n, col 8q of the segment */, int k0, int k1,
                                          int koff /* packed k-step of segment start */, int nk, __amdgpu_buffer_rsrc_t wr, int gp,
                                          int ublk, int lane, typename Frag<T>::type (*pre)[2] = nullptr) {
  typedef typename Frag<T>::type frag;
  auto loadb = [&](frag(&b)[2], int kk) {
#pragma unroll
    for (int s = 0; s < 2; ++s) b[s] = BFrag<T>::load(wr, lane, frag_boff<T>(gp, ublk, koff + kk, s, nk));
  };
  constexpr int AG = CF_AG;
  [[maybe_unused]] frag an[CF_APF ? CF_APF : 1][AG];   // (MT > 3, CF_APF) the A groups of the next MFMAs, read CF_APF groups ahead
  if constexpr (MT > 3 && CF_APF != 0) {
    constexpr int NG0 = (MT + AG - 1) / AG;
#pragma unroll
    for (int p = 0; p < CF_APF; ++p)
#pragma unroll
      for (int i = 0; i < AG; ++i)
        if ((p % NG0) * AG + i < MT) {
          const int kp = k0 + p / NG0 < k1 ? k0 + p / NG0 : k1 - 1;
          an[p][i] = load_frag<T>(a_lane + 16 * ((p % NG0) * AG + i) * RS + 32 * kp);
        }
  }
  auto compute = [&](const frag(&b)[2], int kk) {
    if constexpr (MT <= 3) {
      frag af[MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) af[mt] = load_frag<T>(a_lane + 16 * mt * RS + 32 * kk);
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[s][mt] = mma16(af[mt], b[s], acc[s][mt]);
    } else {
      // many row tiles: A fragments in groups of AG with a scheduling fence in between, so that at most 4*AG*(1 + CF_APF)
      // (not 4*MT, twice that when the scheduler hoists the next k-step's reads) VGPRs hold A operands -- the 80-row
      // variant spilled its layer-2 cell state otherwise, and ANY scratch reload waits for every outstanding
      // memory operation of the wave (hand-off stores, HBM streams)
#if CF_APF
      // round 4: LDS reads run CF_APF groups ahead of their MFMAs (behind the last group of a k-step: the first groups of
      // the next one): a wave's own matrix work covers part of the LDS latency of its next operands, the other waves of
      // the SIMD the rest (round 3 issued a group's reads and waited for them right there: three exposed LDS round trips
      // per k-step and wave)
      constexpr int NG = (MT + AG - 1) / AG;
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        frag nx[AG];
        const int gt = (g + CF_APF) % NG;
        int kt = kk + (g + CF_APF) / NG;
        kt = kt < k1 ? kt : k1 - 1;                                          // (behind the last k-step: harmless re-reads)
#pragma unroll
        for (int i = 0; i < AG; ++i)
          if (gt * AG + i < MT) nx[i] = load_frag<T>(a_lane + 16 * (gt * AG + i) * RS + 32 * kt);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int i = 0; i < AG; ++i)
            if (g * AG + i < MT) acc[s][g * AG + i] = mma16(an[0][i], b[s], acc[s][g * AG + i]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p + 1 < CF_APF; ++p)
#pragma unroll
          for (int i = 0; i < AG; ++i) an[p][i] = an[p + 1][i];
#pragma unroll
        for (int i = 0; i < AG; ++i) an[CF_APF - 1][i] = nx[i];
      }
#else
#pragma unroll
      for (int m0 = 0; m0 < MT; m0 += AG) {
        frag af[AG];
#pragma unroll
        for (int i = 0; i < AG; ++i)
          if (m0 + i < MT) af[i] = load_frag<T>(a_lane + 16 * (m0 + i) * RS + 32 * kk);
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int i = 0; i < AG; ++i)
            if (m0 + i < MT) acc[s][m0 + i] = mma16(af[i], b[s], acc[s][m0 + i]);
        __builtin_amdgcn_sched_barrier(0);
      }
#endif
    }
  };
  frag b[DEPTH][2];
  const int n = k1 - k0;
#pragma unroll
  for (int d = 0; d < DEPTH - 1; ++d) {
    if (PRIMED) { b[d][0] = pre[d][0]; b[d][1] = pre[d][1]; }
    else if (d < n) loadb(b[d], k0 + d);
  }
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

// saved state / outputs are written once and read by a LATER kernel: non-temporal stores keep them out of the L2 that
// holds the weight fragments every step re-reads
template <typename T> __device__ __forceinline__ void store4_nt(T* p, float a, float b, float c, float d);
template <> __device__ __forceinline__ void store4_nt<float>(float* p, float a, float b, float c, float d) {
  __builtin_nontemporal_store(f32x4{a, b, c, d}, reinterpret_cast<f32x4*>(p));
}
template <> __device__ __forceinline__ void store4_nt<bf16_t>(bf16_t* p, float a, float b, float c, float d) {
  typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
  u32x2 v;
  v[0] = (uint32_t)f2bf(a) | ((uint32_t)f2bf(b) << 16);
  v[1] = (uint32_t)f2bf(c) | ((uint32_t)f2bf(d) << 16);
  __builtin_nontemporal_store(v, reinterpret_cast<u32x2*>(p));
}
template <typename T> __device__ __forceinline__ void store1_nt(T* p, float v) { __builtin_nontemporal_store(from_f32<T>(v), p); }

// workgroup barrier that orders LDS traffic only: unlike __syncthreads() it does not make the wave wait for its outstanding
// global loads / stores / LDS-DMA (hipcc emits s_waitcnt vmcnt(0) in front of the barrier whenever an LDS-DMA may be in
// flight, and for pending stores) -- used where everything still in flight is consumed behind a LATER vmcnt(0) + barrier
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// LDS row (elements): [X0 (KX) | X1 (KX) | H1 (H) | H2 (H)] + pad.  H = 384 = 12 k-steps.
// diagnostic phase timers of the forward kernel (tools/diag/stamp_fwd.py builds with -DCF_STAMP)
// diagnostic builds of the forward kernel (tools/diag/fwd_variants.py; timing only, results are garbage):
//   CF_NO_CELL  the cell update without its transcendentals (a handful of VALU instructions per cell): what the pointwise
//               phases cost a step, i.e. what ANY schedule that hides them behind the fragment stream could win at most
//   CF_NO_BAR   no workgroup barrier inside the time step (the waves of a CU run free): what the step's barriers cost
//   CF_PRIO=n   static wave priorities: 1 = older waves first (waves 0-3: 3, 4-7: 2, 8-11: 1), 2 = younger waves first
#ifdef CF_NO_CELL
#define CF_SIG(x) (0.5f * (x))
#define CF_TANH(x) (x)
#else
#define CF_SIG(x) sigmoid_f(x)
#define CF_TANH(x) tanh_f(x)
#endif
#ifdef CF_NO_BAR
#define CF_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#else
#define CF_BARRIER() __syncthreads()
#endif
#ifdef CF_STAMP
#define FT(i) if (blockIdx.x == 0 && threadIdx.x == 0) { const unsigned long long nw = __builtin_readcyclecounter(); ft_acc[i] += nw - ft_last; ft_last = nw; }
#else
#define FT(i)
#endif

template <typename T, int G, int MT, int KX, bool TRAIN, bool HEAD = false>
__global__ __launch_bounds__((384 / G / 16) * 64) void lstm2_coop_fwd_kernel(CoopArgs a) {
#ifdef CF_STAMP
  unsigned long long ft_last = __builtin_readcyclecounter(), ft_acc[13] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
  constexpr int H = 384, HC = H / G, NW = HC / 16, NT = NW * 64, MC = 16 * MT;
  constexpr int VEC = 16 / (int)sizeof(T);
  constexpr int RS = 2 * KX + 2 * H + (sizeof(T) == 2 ? COOP_APAD : VEC);
  constexpr int OX0 = 0, OX1 = KX, OH1 = 2 * KX, OH2 = 2 * KX + H;
  constexpr int NKX = KX / 32, NKH = H / 32;
  constexpr int nk1 = NKX + NKH, nk2 = 2 * NKH;
  // weight-fragment ring depth the register budget allows (80-row variant: depth 3 spills 3 registers and takes 9.5 instead of
  // 8.8 ms, depth 4 spills 24: 9.9 ms)
  constexpr int DEPTH = (MT <= 2 || G >= 4) ? 4 : 2;
  constexpr int SLICE = MC * HC;                                  // elements of one CU's h slice
  constexpr int SLICE_CH = SLICE * (int)sizeof(T) / 16;           // 16-byte chunks
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* lds = reinterpret_cast<T*>(smem_raw);
  // TRAIN: the cell states of the two layers are staged here ([MC][HC] each) and, like the h slices that are in LDS
  // anyway, leave for HBM as coalesced 16-byte non-temporal stores after the step's barriers; as 2-byte stores from the
  // accumulator lanes (32-byte pieces of a line) the saved state cost 18 % of the kernel.  Gates go straight from the
  // lanes: 8 bytes x 16 consecutive units = full 128-byte lines.
  T* sc1 = lds + (size_t)MC * RS;
  T* sc2 = sc1 + MC * HC;

  const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int cluster, cu;
  coop_ids(G, cluster, cu);
  if (cluster >= a.clusters) return;
#ifdef CF_PRIO
  if (CF_PRIO == 1) { if (wave < 4) __builtin_amdgcn_s_setprio(3); else if (wave < 8) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(1); }
  else { if (wave < 4) __builtin_amdgcn_s_setprio(1); else if (wave < 8) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(3); }
#endif
  const int ublk = cu * NW + wave;                                // global 16-unit block of this wave
  const int unit_n = ublk * 16 + n;
  const long row0 = (long)cluster * MC;
  const long N = a.N;
  const long rbase = row0 + 4 * q;

  for (int i = tid; i < MC * RS; i += NT) lds[i] = from_f32<T>(0.f);
  // biases: registers (8 VGPRs) for the small tiles; for MT > 3, where registers are short, the own units' biases live
  // in LDS ([2 layers][4 gates][HC] floats) and are read in the pointwise stages
  constexpr bool BIAS_LDS = MT > 3;
  float* bl = reinterpret_cast<float*>(sc1 + (TRAIN ? 2 * MC * HC : 0));
  float b1[4], b2[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    b1[g] = a.bias1[g * H + unit_n];
    b2[g] = a.bias2[g * H + unit_n];
    if (BIAS_LDS && q == 0) {
      bl[(0 * 4 + g) * HC + wave * 16 + n] = b1[g];
      bl[(1 * 4 + g) * HC + wave * 16 + n] = b2[g];
    }
  }
  auto bias = [&](int layer, int g) -> float {
    if constexpr (BIAS_LDS) return bl[(layer * 4 + g) * HC + wave * 16 + n];
    else return layer ? b2[g] : b1[g];
  };
  f32x4 c1[MT], c2[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) c1[mt] = c2[mt] = f32x4{0.f, 0.f, 0.f, 0.f};

  const T* xg = reinterpret_cast<const T*>(a.x);
  const int xld = a.x_ld;
  const __amdgpu_buffer_rsrc_t xrs = make_rsrc(a.x, (unsigned)((size_t)a.Tn * N * xld * sizeof(T)));
  const int cpr = xld / VEC;                                      // 16-byte chunks fetched per row (<= KX / VEC)
  __syncthreads();
  for (int ch = tid; ch < MC * cpr; ch += NT) {
    const int r = ch / cpr, cc = ch % cpr;
    if (row0 + r < N)
      *reinterpret_cast<uint4*>(lds + r * RS + OX0 + cc * VEC) =
          *reinterpret_cast<const uint4*>(xg + ((size_t)(row0 + r)) * xld + cc * VEC);
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
  T* h1o = reinterpret_cast<T*>(a.h1);
  T* h2o = reinterpret_cast<T*>(a.h2);
  T* g1o = reinterpret_cast<T*>(a.g1);
  T* g2o = reinterpret_cast<T*>(a.g2);
  T* c1o = reinterpret_cast<T*>(a.c1);
  T* c2o = reinterpret_cast<T*>(a.c2);

  // own [MC][HC] slices of h (LDS columns hoff + cu*HC) and, TRAIN, c (staging) of step t -> HBM, 16 bytes per lane
  auto flush_state = [&](T* ho, T* co, const T* sc, int hoff, int t) {
    int tv = tid;
    asm volatile("" : "+v"(tv));           // re-derive the chunk addresses each step (hoisted, they were spilled)
    for (int ch = tv; ch < SLICE_CH; ch += NT) {
      const int r = ch / (HC / VEC), cc = ch % (HC / VEC);
      if (row0 + r < N) {
        const size_t e = ((size_t)t * N + row0 + r) * H + cu * HC + cc * VEC;
        st_nt16(ho + e, *reinterpret_cast<const u32x4*>(lds + r * RS + hoff + cu * HC + cc * VEC));
        if (TRAIN) st_nt16(co + e, *reinterpret_cast<const u32x4*>(sc + r * HC + cc * VEC));
      }
    }
  };
  // publish this CU's h slice (already in LDS columns [hoff + cu*HC, +HC)) of layer `layer` for epoch `ep`
  auto publish = [&](int layer, int hoff, int ep) {
    const int par = ep & 1;
    const int base = ((layer * 2 + par) * G + cu) * SLICE * (int)sizeof(T);
    int tv = tid;
    asm volatile("" : "+v"(tv));
    for (int ch = tv; ch < SLICE_CH; ch += NT) {
      const int r = ch / (HC / VEC), cc = ch % (HC / VEC);
      const u32x4 v = *reinterpret_cast<const u32x4*>(lds + r * RS + hoff + cu * HC + cc * VEC);
      store_sc1_b128(xr, base + ch * 16, v);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // every storing wave drains its write-through stores
    CF_BARRIER();
    if (tid == 0) __hip_atomic_store(flags + layer * G + cu, (unsigned)ep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  // wait for the partners' slices of (layer, epoch) and copy them into LDS
  auto consume = [&](int layer, int hoff, int ep) {
    if (wave == 0) {
      if (lane < G && lane != cu) {
        unsigned spins = 0;
        while (CF_POLL && __hip_atomic_load(flags + layer * G + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)ep) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > SPIN_LIMIT) {
            __hip_atomic_fetch_add(tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");     // compiler ordering only: payload loads are sc1
    }
    CF_BARRIER();
    const int par = ep & 1;
#pragma unroll 1
    for (int p = 0; p < G; ++p) {
      if (p == cu) continue;
      const int base = ((layer * 2 + par) * G + p) * SLICE * (int)sizeof(T);
      int tv = tid;
      asm volatile("" : "+v"(tv));
      for (int ch = tv; ch < SLICE_CH; ch += NT) {
        const int r = ch / (HC / VEC), cc = ch % (HC / VEC);
        const u32x4 v = load_sc1_b128(xr, base + ch * 16);
        *reinterpret_cast<u32x4*>(lds + r * RS + hoff + p * HC + cc * VEC) = v;
      }
    }
  };

  // every wave polls for itself and fetches its own share of a partner slice into registers (no workgroup barrier):
  // used for the h2 hand-off, whose LDS region nobody reads during layer 1, so the transfer hides behind its GEMMs
  constexpr int HXC = (SLICE_CH * (G - 1) + NT - 1) / NT;
  auto wave_poll = [&](int layer, int ep) {
    if (lane < G && lane != cu) {
      unsigned spins = 0;
      while (CF_POLL && __hip_atomic_load(flags + layer * G + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)ep) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > SPIN_LIMIT) {
          __hip_atomic_fetch_add(tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  auto slice_addr = [&](int idx, int& partner, int& ch) {      // idx-th chunk over the G-1 partner slices
    const int pi = idx / SLICE_CH;
    partner = pi < cu ? pi : pi + 1;
    ch = idx % SLICE_CH;
  };

#pragma unroll 1
  for (int t = 0; t < a.Tn; ++t) {
    const int p = t & 1;
    const int ep = t + 1;
    const size_t ebase = ((size_t)t * N + rbase) * H + unit_n;
    // partners' h2_{t-1} slices: global -> LDS directly (LDS-DMA, `buffer_load_dwordx4 ... lds`), one row of a slice
    // (HC bf16 = HC/8 lanes x 16 bytes, landing contiguously at the row's H2 columns) per instruction.  No registers
    // are held across layer 1 (12 VGPRs per lane before: spilled in the 80-row variant, and a spill store made the
    // wave wait out the hand-off latency at the top of every step); nobody reads H2 during layer 1, and barrier (1)
    // drains the DMA (a pending LDS write on the vm counter) long after it landed.
    // Round 4 (CF_H2_LATE): polled and requested BEHIND layer 1's first GEMM, in front of its pointwise phase, instead of
    // at the top of the step -- a wave's vector-memory operations retire in order, so at the top the first weight
    // fragments of the step queued behind the poll and the cross-CU read (in-kernel stamps: the step's first GEMM phase
    // took 15.8k cycles against 7-8k for the other 12-14 k-step phases); here the partner's flag is a third of a step old
    // and the pointwise phase covers the transfer
    auto h2_fetch = [&]() {
      if (t > 0) {
        wave_poll(1, ep - 1);
        typedef __attribute__((address_space(3))) void lds_void;
#pragma unroll 1
        for (int ri = wave; ri < MC * (G - 1); ri += NW) {
          const int pi = ri / MC, r = ri % MC;
          const int pr = pi < cu ? pi : pi + 1;
          if (lane < HC / VEC)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void*)(lds + r * RS + OH2 + pr * HC), 16, lane * 16,
                                                     ((1 * 2 + ((ep - 1) & 1)) * G + pr) * SLICE * (int)sizeof(T) + r * HC * (int)sizeof(T),
                                                     0, 16);
        }
      }
    };
    if (!CF_H2_LATE) h2_fetch();
    const bool more = t + 1 < a.Tn;
    // x_{t+1} -> the X buffer this step does not read, by LDS-DMA as well (one 128-byte row per instruction,
    // non-temporal: read once).  Issued BEHIND layer 1's GEMMs, in front of its pointwise phase and the two barriers: a
    // wave's vector-memory operations retire in order, so at the top of the step this HBM read sat in front of the first
    // weight fragments of layer 1 and every wave waited out its latency there (0.3 ms per launch; here the pointwise
    // phase covers most of it: restorer forward 9.10 -> 8.93 ms, tools/diag/lstm_variants.py)
    auto x_prefetch = [&]() {
      if (more) {
        typedef __attribute__((address_space(3))) void lds_void;
#pragma unroll 1
        for (int r = wave; r < MC; r += NW)
          if (row0 + r < N && lane < cpr)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrs, (lds_void*)(lds + r * RS + (p ? OX0 : OX1)), 16, lane * 16,
                                                     (int)((((size_t)(t + 1) * N + row0 + r) * xld) * sizeof(T)), 0, 2);
      }
    };
    constexpr bool PR = MT <= 2;                                   // prime the next weight segment across sync points
    typename Frag<T>::type pre[DEPTH - 1][2];
    // ================= layer 1: [x_t | h1_{t-1}] =================
    float iv1[TRAIN ? MT : 1][4], gv1[TRAIN ? MT : 1][4];
    f32x4 ig[MT];
    {
      f32x4 acc[2][MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[0][mt] = acc[1][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
      coop_gemm<T, MT, RS, DEPTH>(acc, a_lane + (p ? OX1 : OX0), 0, NKX, 0, nk1, wr1, 0, ublk, lane);
      coop_gemm<T, MT, RS, DEPTH>(acc, a_lane + OH1, 0, NKH, NKX, nk1, wr1, 0, ublk, lane);
      if (CF_H2_LATE) h2_fetch();
      FT(0)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float iv = CF_SIG(acc[0][mt][j] + bias(0, 0));
          const float gv = CF_TANH(acc[1][mt][j] + bias(0, 2));
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
      // every vector-memory operation issued so far has completed here anyway (the ring just consumed its last, youngest
      // fragment and vmcnt retires in order): make the completion of this step's LDS-DMA transfer (partner h2) explicit
      // before the barriers that publish it to the other waves; the next x follows and is drained by barrier (1)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      x_prefetch();
      FT(1)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float fv = CF_SIG(acc[0][mt][j] + bias(0, 1));
          const float ov = CF_SIG(acc[1][mt][j] + bias(0, 3));
          const float cn = fv * c1[mt][j] + ig[mt][j];
          c1[mt][j] = cn;
          hn1[mt][j] = ov * CF_TANH(cn);
          if (TRAIN && rbase + 16 * mt + j < N) {
            const size_t e = ebase + (size_t)(16 * mt + j) * H;
#ifndef CF_NO_SAVE
            sc1[(4 * q + 16 * mt + j) * HC + wave * 16 + n] = from_f32<T>(cn);
            store4_nt<T>(g1o + e * 4, iv1[mt][j], gv1[mt][j], fv, ov);
#endif
          }
        }
    }
    if (PR) coop_prime<T, DEPTH>(pre, NKH, NKH, nk2, wr2, 0, ublk, lane);       // layer 2, pair (i,g), h2 half
    CF_BARRIER();                                              // (1) all waves done reading h1_{t-1} and x_t
    FT(2)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int j = 0; j < 4; ++j) hw_lane[OH1 + (16 * mt + j) * RS] = from_f32<T>(hn1[mt][j]);
    CF_BARRIER();                                              // (2a) own h1_t slice complete in LDS
    FT(3)
    publish(0, OH1, ep);
    FT(4)
#ifndef CF_NO_SAVE
    if (TRAIN) flush_state(h1o, c1o, sc1, OH1, t);                // after the hand-off drain, not in front of it
#endif
    // ================= layer 2 =================
    // gate pair (i,g): the h2_{t-1} half first -- it does not need the partners' h1_t, so it hides the hand-off latency
    f32x4 ig2[MT];
    float iv2[TRAIN ? MT : 1][4], gv2[TRAIN ? MT : 1][4];
    {
      f32x4 acc[2][MT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) acc[0][mt] = acc[1][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
      coop_gemm<T, MT, RS, DEPTH, PR>(acc, a_lane + OH2, 0, NKH, NKH, nk2, wr2, 0, ublk, lane, pre);
      FT(5)
      if (PR) coop_prime<T, DEPTH>(pre, NKH, 0, nk2, wr2, 0, ublk, lane);       // h1 half: in flight during the hand-off
      consume(0, OH1, ep);
      CF_BARRIER();                                            // (2c) full h1_t in LDS
      FT(6)
      coop_gemm<T, MT, RS, DEPTH, PR>(acc, a_lane + OH1, 0, NKH, 0, nk2, wr2, 0, ublk, lane, pre);
      FT(7)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float iv = CF_SIG(acc[0][mt][j] + bias(1, 0));
          const float gv = CF_TANH(acc[1][mt][j] + bias(1, 2));
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
      FT(9)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float fv = CF_SIG(acc[0][mt][j] + bias(1, 1));
          const float ov = CF_SIG(acc[1][mt][j] + bias(1, 3));
          const float cn = fv * c2[mt][j] + ig2[mt][j];
          c2[mt][j] = cn;
          hn2[mt][j] = ov * CF_TANH(cn);
          if (rbase + 16 * mt + j < N) {
            const size_t e = ebase + (size_t)(16 * mt + j) * H;
            if (TRAIN) {
#ifndef CF_NO_SAVE
              sc2[(4 * q + 16 * mt + j) * HC + wave * 16 + n] = from_f32<T>(cn);
              store4_nt<T>(g2o + e * 4, iv2[mt][j], gv2[mt][j], fv, ov);
#endif
            }
          }
        }
    }
    CF_BARRIER();                                              // (3) all waves done reading h2_{t-1} / h1_t
    FT(10)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int j = 0; j < 4; ++j) hw_lane[OH2 + (16 * mt + j) * RS] = from_f32<T>(hn2[mt][j]);
    CF_BARRIER();                                              // own h2_t slice complete in LDS
    FT(11)
    if (more) publish(1, OH2, ep);
    FT(12)
#ifndef CF_NO_SAVE
    if constexpr (HEAD) {
      // Linear(H -> O) on the own h2_t slice, still in LDS as the MFMA A operand: wave w multiplies row tile w by the
      // own units' head weights (6 k-steps, B fragments from L2) and leaves O partial sums per row; the 1.6 GB of h2
      // stores, their re-read by a head kernel and that launch disappear (the partner's half is added by the finaliser)
      if (wave < MT) {
        int nn = lane & 15, qq = lane >> 4;
        asm volatile("" : "+v"(nn), "+v"(qq));     // re-derive the addresses each step: hoisted, they were spilled
        const T* ap = lds + (16 * wave + nn) * RS + OH2 + cu * HC + 8 * qq;
        const T* bp = reinterpret_cast<const T*>(a.whp) + (size_t)nn * H + cu * HC + 8 * qq;
        f32x4 hacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < HC / 32; ++kk) hacc = mma16(load_frag<T>(ap + 32 * kk), load_frag<T>(bp + 32 * kk), hacc);
        if (nn < a.O) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const long r = row0 + 16 * wave + 4 * qq + j;
            if (r < N) a.hpart[(((size_t)cu * a.Tn + t) * N + r) * a.O + nn] = hacc[j];
          }
        }
      }
    }
    if constexpr (TRAIN || !HEAD) flush_state(h2o, c2o, sc2, OH2, t);   // training keeps h2 / c2; inference only without a fused head
#endif
  }
#ifdef CF_STAMP
  if (blockIdx.x == 0 && threadIdx.x == 0)
    for (int i = 0; i < 13; ++i) ((unsigned long long*)(a.flags + (size_t)a.clusters * 2 * G + 4))[i] = ft_acc[i];
#endif
}

// =====================================================================================================
// Cooperative backward recurrence: a CU pair shares a tile of 32 sequences; CU `cu` owns hidden units
// [192*cu, 192*cu + 192) of both layers: it runs the cell backward for them, owns the matching output columns of the
// two backward GEMMs (d h1 / d h2_prev / d h1_prev / half of d x) and streams only those columns' weights (half).
// What crosses CUs per layer and step is the other half of the dgates tile [32][768] (48 KB), the MFMA A operand.
struct CoopBwdArgs {
  const void* g1; const void* g2; const void* c1; const void* c2; const void* dh2;
  const void* wb1; const void* wb2;   // packed by lstm_coop_pack_bwd_kernel
  void* dx; void* dg1; void* dg2;     // dg [Tn][N][4H]
  void* xch;                          // [clusters][2 layers][2 parities][2 CUs][32][768]
  unsigned* flags;                    // [clusters][2 layers][2] epochs + timeout word
  long N; int Tn; int clusters;
};

constexpr int CB_G = 2, CB_MC = 32, CB_H = 384, CB_HC = 192, CB_KX = 64, CB_NW = 12, CB_NT = CB_NW * 64;
constexpr int CB_K4 = 4 * CB_H, CB_KC = 4 * CB_HC;        // gate columns: all / own
constexpr int CB_NK = CB_K4 / 32;                          // 48 k-steps

// B fragment (cu, wave, kk, slot): packed contiguously [cu][wave][kk][slot][lane][8]
__device__ __forceinline__ int cb_frag_boff(int cu, int wave, int kk, int slot) {
  return ((((cu * CB_NW + wave) * CB_NK + kk) * 2 + slot) * 512) * 2;
}

template <int NSLOT>
__device__ __forceinline__ void cb_gemm(f32x4 (&acc)[2][2], const bf16_t* a_lane, int k0, int k1, __amdgpu_buffer_rsrc_t wr, int cu,
                                        int wave, int lane, int RS) {
  constexpr int DEPTH = 4;
  bf16x8 b[DEPTH][2];
  auto loadb = [&](bf16x8(&bb)[2], int kk) {
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) bb[s] = BFrag<bf16_t>::load(wr, lane, cb_frag_boff(cu, wave, kk, s));
  };
  auto compute = [&](const bf16x8(&bb)[2], int kk) {
    bf16x8 af[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) af[mt] = load_frag<bf16_t>(a_lane + 16 * mt * RS + 32 * kk);
#pragma unroll
    for (int s = 0; s < NSLOT; ++s)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) acc[s][mt] = mma16(af[mt], bb[s], acc[s][mt]);
  };
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

__global__ __launch_bounds__(CB_NT) void lstm2_coop_bwd_kernel(CoopBwdArgs a) {
  typedef bf16_t T;
  constexpr int MC = CB_MC, H = CB_H, HC = CB_HC, KX = CB_KX, NT = CB_NT;
  constexpr int RSA = CB_K4 + COOP_APAD;                            // A tile row stride (elements)
  constexpr int TPR = NT / MC, UPT = HC / TPR;              // 24 threads per row, 8 units per thread
  constexpr int HALF_CH = MC * CB_KC * 2 / 16;              // 16-byte chunks of one dgates half (3072)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* Abuf = reinterpret_cast<T*>(smem_raw);                                         // [32][RSA]
  float* dh1buf = reinterpret_cast<float*>(smem_raw + (size_t)MC * RSA * sizeof(T)); // [32][HC]  own units
  float* dhrec2 = dh1buf + MC * HC;                                                  // [32][HC]

  const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int cluster, cu;
  coop_ids(CB_G, cluster, cu);
  if (cluster >= a.clusters) return;
  const int pcu = 1 - cu;
  const long row0 = (long)cluster * MC;
  const long N = a.N;
  for (int i = tid; i < 2 * MC * HC; i += NT) dh1buf[i] = 0.f;
  for (int i = tid; i < MC * RSA; i += NT) Abuf[i] = 0;

  const int prow = tid / TPR, ul0 = (tid % TPR) * UPT, u0 = cu * HC + ul0;   // local / global first unit
  const bool prow_ok = row0 + prow < N;
  float dc1[UPT], dc2[UPT];
#pragma unroll
  for (int i = 0; i < UPT; ++i) dc1[i] = dc2[i] = 0.f;

  const T* g1 = reinterpret_cast<const T*>(a.g1);
  const T* g2 = reinterpret_cast<const T*>(a.g2);
  const T* c1 = reinterpret_cast<const T*>(a.c1);
  const T* c2 = reinterpret_cast<const T*>(a.c2);
  const T* dh2 = reinterpret_cast<const T*>(a.dh2);
  T* dx = reinterpret_cast<T*>(a.dx);
  T* dg1T = reinterpret_cast<T*>(a.dg1);
  T* dg2T = reinterpret_cast<T*>(a.dg2);
  constexpr unsigned WB = CB_G * CB_NW * CB_NK * 2 * 512 * 2;
  const __amdgpu_buffer_rsrc_t wr1 = make_rsrc(a.wb1, WB), wr2 = make_rsrc(a.wb2, WB);
  constexpr int XSL = MC * CB_KC;                                                   // elements of one dgates half
  const __amdgpu_buffer_rsrc_t xr = make_rsrc(reinterpret_cast<const T*>(a.xch) + (size_t)cluster * 2 * 2 * 2 * XSL,
                                              (unsigned)(2 * 2 * 2 * XSL * sizeof(T)));
  gu32* flags = (gu32*)(a.flags + (size_t)cluster * 2 * CB_G);
  gu32* tmo = (gu32*)(a.flags + (size_t)a.clusters * 2 * CB_G);
  const T* a_lane = Abuf + n * RSA + 8 * q;
  __syncthreads();

  struct Saved { T g[UPT * 4]; T ct[UPT]; T cp[UPT]; T dh[UPT]; };
  auto fetch = [&](Saved& sv, const T* gs, const T* cs, const T* dh_ext, int t) {
    if (!prow_ok || t < 0) return;
    const size_t e = ((size_t)t * N + row0 + prow) * H + u0;
#pragma unroll
    for (int i = 0; i < UPT * 4 * 2 / 16; ++i) reinterpret_cast<u32x4*>(sv.g)[i] = ld_nt16(gs + e * 4 + 8 * i);
    *reinterpret_cast<u32x4*>(sv.ct) = ld_nt16(cs + e);
    if (t > 0) *reinterpret_cast<u32x4*>(sv.cp) = ld_nt16(cs + e - (size_t)N * H);
    if (dh_ext) *reinterpret_cast<u32x4*>(sv.dh) = ld_nt16(dh_ext + e);
  };
  // cell backward of this thread's (row, 8 units) -> gate gradients (bf16, k = unit*4 + gate) into the LDS A operand
  auto cell_bwd = [&](const Saved& sv, const float* dh_lds, bool has_ext, float (&dc)[UPT], int t) {
#pragma unroll
    for (int i = 0; i < UPT; ++i) {
      float di = 0.f, dg = 0.f, df = 0.f, dO = 0.f;
      if (prow_ok) {
        const float iv = bf2f(sv.g[i * 4 + 0]), gv = bf2f(sv.g[i * 4 + 1]);
        const float fv = bf2f(sv.g[i * 4 + 2]), ov = bf2f(sv.g[i * 4 + 3]);
        const float ct = bf2f(sv.ct[i]);
        const float cp = t > 0 ? bf2f(sv.cp[i]) : 0.f;
        float dh = dh_lds[prow * HC + ul0 + i];
        if (has_ext) dh += bf2f(sv.dh[i]);
        const float tc = tanh_f(ct);
        const float dct = dh * ov * (1.f - tc * tc) + dc[i];
        dO = dh * tc * ov * (1.f - ov);
        di = dct * gv * iv * (1.f - iv);
        dg = dct * iv * (1.f - gv * gv);
        df = dct * cp * fv * (1.f - fv);
        dc[i] = dct * fv;
      }
      store4<T>(Abuf + prow * RSA + (u0 + i) * 4, di, dg, df, dO);   // 8-byte LDS stores (16-byte ones measured slower)
    }
  };
  // own half of the dgates tile: columns [cu*768, +768) of Abuf -> partner (write-through, drained, flagged).  Read back
  // from LDS so that consecutive lanes store consecutive 16-byte chunks: storing each thread's own 64 bytes straight from
  // its registers was measured 34 % SLOWER for the whole kernel (partial-line write-through transactions).
  auto publish = [&](int layer, int ep) {
    const int base = ((layer * 2 + (ep & 1)) * 2 + cu) * XSL * 2;
    for (int ch = tid; ch < HALF_CH; ch += NT) {
      const int r = ch / (CB_KC / 8), cc = ch % (CB_KC / 8);
      const u32x4 v = *reinterpret_cast<const u32x4*>(Abuf + r * RSA + cu * CB_KC + cc * 8);
      store_sc1_b128(xr, base + ch * 16, v);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(flags + layer * CB_G + cu, (unsigned)ep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  // ... and -> dg (row-major [t*N + row][4H], own 768 columns) for the weight-gradient GEMMs.  Issued AFTER the hand-off
  // so that the publish drain does not wait for these stores; nothing in the kernel reads them (non-temporal).
  auto store_T = [&](T* dg, int t) {
    for (int ch = tid; ch < HALF_CH; ch += NT) {
      const int r = ch / (CB_KC / 8), cc = ch % (CB_KC / 8);
      if (row0 + r < N)
        st_nt16(dg + ((size_t)t * N + row0 + r) * CB_K4 + cu * CB_KC + cc * 8,
                *reinterpret_cast<const u32x4*>(Abuf + r * RSA + cu * CB_KC + cc * 8));
    }
  };
  auto consume = [&](int layer, int ep) {
    if (wave == 0) {
      if (lane == 0) {
        unsigned spins = 0;
        while (CF_POLL && __hip_atomic_load(flags + layer * CB_G + pcu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)ep) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > SPIN_LIMIT) {
            __hip_atomic_fetch_add(tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    __syncthreads();
    const int base = ((layer * 2 + (ep & 1)) * 2 + pcu) * XSL * 2;
    for (int ch = tid; ch < HALF_CH; ch += NT) {
      const int r = ch / (CB_KC / 8), cc = ch % (CB_KC / 8);
      *reinterpret_cast<u32x4*>(Abuf + r * RSA + pcu * CB_KC + cc * 8) = load_sc1_b128(xr, base + ch * 16);
    }
  };

  Saved sv2, sv1;
  fetch(sv2, g2, c2, dh2, a.Tn - 1);
  const int kown0 = cu * (CB_KC / 32), kpar0 = pcu * (CB_KC / 32), khalf = CB_KC / 32;   // 24 k-steps per half

#pragma unroll 1
  for (int t = a.Tn - 1; t >= 0; --t) {
    const int ep = a.Tn - t;
    // ---------------- layer 2
    cell_bwd(sv2, dhrec2, true, dc2, t);
    __syncthreads();
    publish(1, ep);
    fetch(sv1, g1, c1, nullptr, t);          // layer-1 state of this step: lands during the layer-2 GEMM
    store_T(dg2T, t);
    {
      f32x4 acc[2][2];
#pragma unroll
      for (int s = 0; s < 2; ++s) acc[s][0] = acc[s][1] = f32x4{0.f, 0.f, 0.f, 0.f};
      cb_gemm<2>(acc, a_lane, kown0, kown0 + khalf, wr2, cu, wave, lane, RSA);      // own K half: no waiting
      consume(1, ep);
      __syncthreads();
      cb_gemm<2>(acc, a_lane, kpar0, kpar0 + khalf, wr2, cu, wave, lane, RSA);
      // slot 0: d h1_t (from layer 2) for units 16*wave+n of this CU; slot 1: d h2_{t-1}
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = 16 * mt + 4 * q + j;
          dh1buf[r * HC + 16 * wave + n] += acc[0][mt][j];
          dhrec2[r * HC + 16 * wave + n] = acc[1][mt][j];
        }
    }
    __syncthreads();
    // ---------------- layer 1
    cell_bwd(sv1, dh1buf, false, dc1, t);
    __syncthreads();
    publish(0, ep);
    fetch(sv2, g2, c2, dh2, t - 1);          // layer-2 state of the next (earlier) step
    store_T(dg1T, t);
    {
      f32x4 acc[2][2];
#pragma unroll
      for (int s = 0; s < 2; ++s) acc[s][0] = acc[s][1] = f32x4{0.f, 0.f, 0.f, 0.f};
      const bool has_dx = wave < 2;        // slot 1 of waves 0,1 = this CU's 32 columns of d x
      if (has_dx) cb_gemm<2>(acc, a_lane, kown0, kown0 + khalf, wr1, cu, wave, lane, RSA);
      else cb_gemm<1>(acc, a_lane, kown0, kown0 + khalf, wr1, cu, wave, lane, RSA);
      consume(0, ep);
      __syncthreads();
      if (has_dx) cb_gemm<2>(acc, a_lane, kpar0, kpar0 + khalf, wr1, cu, wave, lane, RSA);
      else cb_gemm<1>(acc, a_lane, kpar0, kpar0 + khalf, wr1, cu, wave, lane, RSA);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int r = 16 * mt + 4 * q + j;
          dh1buf[r * HC + 16 * wave + n] = acc[0][mt][j];                    // d h1_{t-1} (recurrent)
          if (has_dx && row0 + r < N)
            store1_nt<T>(dx + ((size_t)t * N + row0 + r) * KX + cu * 32 + 16 * wave + n, acc[1][mt][j]);
        }
    }
    __syncthreads();
  }
}

// packed backward weights for the cooperative kernel: element (cu, wave, kk, slot, lane l, j)
//   k = 32*kk + 8*(l>>4) + j = u*4 + g'  (g' in i,g,f,o -> torch block {0,2,1,3}[g'])
//   layer 2: slot 0 -> input feature h1 unit cu*192 + 16*wave + (l&15);  slot 1 -> h2 unit (same index)
//   layer 1: slot 0 -> h1 unit (W_hh);  slot 1 -> x column cu*32 + 16*wave + (l&15) for wave < 2 (W_ih, col < I), else 0
__global__ void lstm_coop_pack_bwd_kernel(const float* __restrict__ w_ih, const float* __restrict__ w_hh, bf16_t* __restrict__ out,
                                          int I, int layer) {
  const size_t total = (size_t)CB_G * CB_NW * CB_NK * 2 * 512;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int j = e & 7, l = (e >> 3) & 63;
    size_t f = e >> 9;
    const int slot = f & 1; f >>= 1;
    const int kk = f % CB_NK; f /= CB_NK;
    const int wave = f % CB_NW;
    const int cu = (int)(f / CB_NW);
    const int k = 32 * kk + 8 * (l >> 4) + j;
    const int u = k >> 2, gp = k & 3;
    const int tg = gp == 0 ? 0 : (gp == 1 ? 2 : (gp == 2 ? 1 : 3));
    const int row = tg * CB_H + u;
    const int unit = cu * CB_HC + 16 * wave + (l & 15);
    float v = 0.f;
    if (layer == 2) {
      v = slot == 0 ? w_ih[(size_t)row * CB_H + unit] : w_hh[(size_t)row * CB_H + unit];
    } else {
      if (slot == 0) v = w_hh[(size_t)row * CB_H + unit];
      else if (wave < 2) {
        const int col = cu * 32 + 16 * wave + (l & 15);
        if (col < I) v = w_ih[(size_t)row * I + col];
      }
    }
    out[e] = f2bf(v);
  }
}

// =====================================================================================================
// Cooperative backward, K-split variant.  Same CU pair / 32-sequence tile / unit ownership as above, but the two
// backward GEMMs are split along K (the gate columns) instead of along their outputs:
//   CU `cu` multiplies ITS OWN dgates half [32][768] (already in its LDS: no hand-off before the GEMM) with the
//   weight rows of its own units and ALL output columns -> a partial sum of every output,
//   keeps the partials of the outputs it owns and sends the rest to the partner as bf16 (reduce-scatter).
// Per step a CU ships 32 x (384 + 224) bf16 = 38 KB instead of 2 x 48 KB of gate gradients, the GEMM never waits for
// the partner, and the LDS A operand shrinks to the own half (50 KB).  The weight bytes streamed per CU are unchanged.
//   layer 2 outputs (48 column tiles of 16): [d h1_t (384) | d h2_{t-1} (384)]
//   layer 1 outputs (28 column tiles):       [d h1_{t-1} (384) | d x_t (64)]
// Column tile tau is computed by wave tau % 12 in slot tau / 12; a tile belongs to the CU that owns its units
// (h tiles: tau % 24 / 12, x tiles: columns 0-31 -> cu 0, 32-63 -> cu 1).
struct CoopBwd2Args {
  const void* g1; const void* g2; const void* c1; const void* c2; const void* dh2;
  const void* wb1; const void* wb2;   // packed by lstm_coop_pack_bwd2_kernel
  void* dx; void* dg1; void* dg2;     // dg [Tn][N][4H]
  void* xch;                          // [clusters][2 layers][2 parities][2 CUs][32][384] bf16 partial sums
  unsigned* flags;                    // [clusters][2 layers][2 CUs][12 waves] epochs + 4 timeout words
  long N; int Tn; int clusters;
  // fused head backward: dyt [Tn][N][16] bf16 (dY rows, zero beyond O and before the look-ahead), whT [H][32] bf16
  // ([u][o]); the kernel forms d h2 += dY . Wh itself and dh2 is not read
  const void* dyt; const void* whT;
};

constexpr int C2_NKK = CB_KC / 32;                          // 24 k-steps over the own gate columns
constexpr int C2_SLOTS = 4;                                 // column tiles per wave (layer 2: 4, layer 1: 2 or 3)
constexpr int C2_FPC = 2 * CB_G * CB_NW;                     // epoch words per cluster: [layer 2][cu 2][wave 12] (every wave hands off its own tiles)
constexpr int C2_XW = 2 * CB_HC;                            // 384 partial columns exchanged per row (layer 1 uses 224)
#ifndef C2_DEPTH
#define C2_DEPTH 4                                          // fragment ring: C2_DEPTH - 1 k-steps requested ahead
#endif
#ifndef C2_KBAR
#define C2_KBAR 8                                           // k-steps between the in-pass barriers that keep the waves in step (0: none)
#endif

// B fragment (cu, wave, kk, slot): [cu][wave][kk][slot][lane][8]
__device__ __forceinline__ int c2_frag_boff(int cu, int wave, int kk, int slot) {
#ifdef CF_W_L1
  return (((((cu * CB_NW + wave) * C2_NKK + kk) * C2_SLOTS + slot) * 512) * 2) & 0x3fff;      // diagnostic: weights from L1
#endif
  return ((((cu * CB_NW + wave) * C2_NKK + kk) * C2_SLOTS + slot) * 512) * 2;
}

// ---- weight-fragment ring of the K-split backward, with hand-counted waits --------------------------------------------
// The compiler's wait-count insertion cannot follow a register ring through a loop: for the rolled loop of round 2 it
// emitted vmcnt(7,6),(5,4),(3,2),(1,0) within every four k-steps, i.e. the ring was DRAINED by the fourth k-step of each
// round (prefetch distance 3 -> 0), and any other request placed inside the pass (the partner's partials, the next
// phase's saved state) was waited for right there.  Here a layer's GEMM is one fully unrolled fragment stream (every
// position a template constant), the fragment requests go out through inline asm (invisible to the pass), and every wait
// is `s_waitcnt vmcnt(M)` with M = the number of vector-memory operations issued AFTER the fragment pair that is about to
// be used -- ring requests plus whatever else is issued at fixed positions of the stream; the counts come from a
// compile-time table (C2Sched / C4Sched).  vmcnt retires in order, so M younger operations may still be in flight.  M must
// never exceed the true count; operations the table does not know (the rare poll loop) only make a wait stricter.
__device__ __forceinline__ bf16x8 c2_ring_load(const unsigned char* frag /* wave-uniform: start of the 1 KB fragment */, int lane16) {
  bf16x8 v;
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(v) : "v"(lane16), "s"(frag) : "memory");
  return v;
}
// wait until at most m vector-memory operations are outstanding; the fragment registers pass through the statement, so
// nothing that reads them can be scheduled above it
__device__ __forceinline__ void c2_wait(int m, bf16x8& x, bf16x8& y) {
#define C2_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(x), "+v"(y)); break;
  switch (m) {
    C2_W(0) C2_W(1) C2_W(2) C2_W(3) C2_W(4) C2_W(5) C2_W(6) C2_W(7) C2_W(8) C2_W(9) C2_W(10) C2_W(11) C2_W(12) C2_W(13) C2_W(14)
    C2_W(15) C2_W(16) C2_W(17) C2_W(18) C2_W(19) C2_W(20) C2_W(21) C2_W(22) C2_W(23) C2_W(24) C2_W(25) C2_W(26) C2_W(27) C2_W(28)
    default: asm volatile("s_waitcnt vmcnt(0)" : "+v"(x), "+v"(y)); break;
  }
#undef C2_W
}
// (one fragment: the same variable must not be tied to two operands -- the second tie is a COPY made in front of the wait)
__device__ __forceinline__ void c2_wait(int m, bf16x8& x) {
#define C2_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(x)); break;
  switch (m) {
    C2_W(0) C2_W(1) C2_W(2) C2_W(3) C2_W(4) C2_W(5) C2_W(6) C2_W(7) C2_W(8) C2_W(9) C2_W(10) C2_W(11) C2_W(12) C2_W(13) C2_W(14)
    C2_W(15) C2_W(16) C2_W(17) C2_W(18) C2_W(19) C2_W(20) C2_W(21) C2_W(22) C2_W(23) C2_W(24) C2_W(25) C2_W(26) C2_W(27) C2_W(28)
    default: asm volatile("s_waitcnt vmcnt(0)" : "+v"(x)); break;
  }
#undef C2_W
}

// diagnostic phase timers (tools/diag/stamp_bwd2.py builds with -DC2_STAMP): thread 0 of workgroup 0 accumulates
// s_memtime deltas per phase and leaves them behind the flag words
#ifdef C2_STAMP
#ifndef C2_STAMP_TID
#define C2_STAMP_TID 0
#endif
#define C2_STAMP_INIT unsigned long long st_last = __builtin_readcyclecounter(), st_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define C2T(i) if (blockIdx.x == 0 && tid == C2_STAMP_TID) { const unsigned long long nw = __builtin_readcyclecounter(); st_acc[i] += nw - st_last; st_last = nw; }
#define C2_STAMP_FINI if (blockIdx.x == 0 && tid == C2_STAMP_TID) { for (int i = 0; i < 12; ++i) ((unsigned long long*)(a.flags + (size_t)a.clusters * C2_FPC + 4))[i] = st_acc[i]; }
#else
#define C2T(i)
#endif


template <int... I, typename F>
__device__ __forceinline__ void c4_static_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void c4_static_for(F&& f) { c4_static_for_impl(std::make_integer_sequence<int, N>{}, f); }

// s_waitcnt vmcnt(m) for a compile-time m
__device__ __forceinline__ void c4_wait_mem(int m) {
#define C4_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
  switch (m) {
    C4_W(0) C4_W(1) C4_W(2) C4_W(3) C4_W(4) C4_W(5) C4_W(6) C4_W(7) C4_W(8) C4_W(9) C4_W(10) C4_W(11) C4_W(12) C4_W(13) C4_W(14)
    C4_W(15) C4_W(16) C4_W(17) C4_W(18) C4_W(19) C4_W(20)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
#undef C4_W
}
__device__ __forceinline__ void c4_wait_reg(int m, unsigned& x) {
#define C4_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(x)); break;
  switch (m) {
    C4_W(0) C4_W(1) C4_W(2) C4_W(3) C4_W(4) C4_W(5) C4_W(6) C4_W(7) C4_W(8) C4_W(9) C4_W(10) C4_W(11) C4_W(12)
    default: asm volatile("s_waitcnt vmcnt(0)" : "+v"(x)); break;
  }
#undef C4_W
}
// 16 bytes per lane global -> LDS (lds_base + 16 * lane), through inline asm: invisible to the compiler's wait insertion, which
// would otherwise drain the fragment ring in front of the next LDS read; completion is covered by a later counted wait
__device__ __forceinline__ void c4_lds_dma16(const void* src_lane, unsigned lds_base /* wave-uniform */) {
  unsigned keep;
  const unsigned lb = __builtin_amdgcn_readfirstlane(lds_base);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src_lane), "s"(lb) : "memory");
}

// Compile-time operation schedule of ONE fragment stream over both passes of a layer of the CU-pair kernel (48 k-steps: 24 on
// the partner's tiles, 24 on the own ones; the ring never restarts between them and nothing is drained).  Per position g:
//   side(g)   g = 32: raise the epoch of the shipment (1 store, behind a counted wait for its stores);  g = 36: request the
//             partner's epoch (1 load);  g = 40: request its partials (NS1 loads)
//   request of the NS0 / NS1 fragments of k-step g + D;  g = 47 - D: post (P1 saved-state loads)
//   wait + MFMAs of k-step g
//   tail(g)   g = 23: ship the partner's tiles (NS0 stores; layer 2 with the fused head: + 1 dY DMA in front of them)
template <int NS0, int NS1, int DY, int P1, int D>
struct C2Sched {
  int before[49];    // vector-memory operations issued before side(pos)
  int req_last[48];  // sequence number of the LAST fragment request of k-step g
  static constexpr int ns(int g) { return g < 24 ? NS0 : NS1; }
  static constexpr int side(int pos) { return (pos == 32 || pos == 36) ? 1 : (pos == 40 ? NS1 : 0); }
  static constexpr int post(int pos) { return pos == 47 - D ? P1 : 0; }
  static constexpr int tail(int pos) { return pos == 23 ? NS0 + DY : 0; }
  constexpr C2Sched() : before{}, req_last{} {
    int c = 0;
    for (int g = 0; g < D; ++g) { c += ns(g); req_last[g] = c - 1; }
    for (int pos = 0; pos < 48; ++pos) {
      before[pos] = c;
      c += side(pos);
      if (pos + D < 48) { c += ns(pos + D); req_last[pos + D] = c - 1; }
      c += post(pos);
      c += tail(pos);
    }
    before[48] = c;
  }
  constexpr int wait_frag(int g) const { return before[g] + side(g) + (g + D < 48 ? ns(g + D) : 0) + post(g) - req_last[g] - 1; }
  constexpr int wait_raise() const { return before[32] - before[24]; }         // since the shipment's last store
  constexpr int wait_flag() const { return before[40] - before[36] - 1; }       // since the epoch request
};

template <bool HEAD>
__global__ __launch_bounds__(CB_NT) void lstm2_coop_bwd2_kernel(CoopBwd2Args a) {
  typedef bf16_t T;
  constexpr int MC = CB_MC, H = CB_H, HC = CB_HC, KX = CB_KX, NT = CB_NT;
  constexpr int RSA = CB_KC + COOP_APAD;                            // A tile row stride (elements): own gate columns only
  constexpr int TPR = NT / MC, UPT = HC / TPR;              // 24 threads per row, 8 units per thread
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* Abuf = reinterpret_cast<T*>(smem_raw);                                          // [32][RSA]   own dgates
  float* dh1buf = reinterpret_cast<float*>(smem_raw + (size_t)MC * RSA * sizeof(T));  // [32][HC]    d h1 (own units)
  float* dhrec2 = dh1buf + MC * HC;                                                   // [32][HC]    d h2 recurrent
  float* dxbuf = dhrec2 + MC * HC;                                                    // [32][32]    own d x partial
  float* dcbuf = dxbuf + MC * 32;                                                     // [2][32][HC] d c carried from step to step
  T* whs = reinterpret_cast<T*>(dcbuf + 2 * MC * HC);                                 // HEAD: [HC][16] own units' head weights
  T* dys = whs + HC * 16;                                                             // HEAD: [32][16]  dY rows of one step

#ifdef C2_STAMP
  C2_STAMP_INIT
#endif
  const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, q = lane >> 4, n_ = n, q_ = q;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int cluster, cu;
  coop_ids(CB_G, cluster, cu);
  if (cluster >= a.clusters) return;
#ifdef C2_WPRIO
  if (wave >= 9) asm volatile("s_setprio 3"); else if (wave >= 6) asm volatile("s_setprio 2"); else if (wave >= 3) asm volatile("s_setprio 1");
#endif
  const int pcu = 1 - cu;
  const long row0 = (long)cluster * MC;
  const long N = a.N;
  for (int i = tid; i < 4 * MC * HC + MC * 32; i += NT) dh1buf[i] = 0.f;      // d h1, d h2 rec, d x, d c (both layers)
  for (int i = tid; i < MC * RSA; i += NT) Abuf[i] = 0;
  if constexpr (HEAD) {
    const T* whT = reinterpret_cast<const T*>(a.whT);
    for (int i = tid; i < HC * 16; i += NT) whs[i] = whT[(size_t)(cu * HC + i / 16) * 32 + (i % 16)];
  }

  const int prow = tid / TPR;
  const bool prow_ok = row0 + prow < N;
  // (d c of the thread's 8 units per layer lives in LDS, not in 16 registers: both layers' saved state is in flight at
  // the end of a GEMM pass since round 3, and the registers are what that costs)
  float* dc1 = dcbuf;
  float* dc2 = dcbuf + MC * HC;

  const T* g1 = reinterpret_cast<const T*>(a.g1);
  const T* g2 = reinterpret_cast<const T*>(a.g2);
  const T* c1 = reinterpret_cast<const T*>(a.c1);
  const T* c2 = reinterpret_cast<const T*>(a.c2);
  const T* dh2 = reinterpret_cast<const T*>(a.dh2);
  T* dx = reinterpret_cast<T*>(a.dx);
  T* dg1T = reinterpret_cast<T*>(a.dg1);
  T* dg2T = reinterpret_cast<T*>(a.dg2);
  constexpr unsigned WB = CB_G * CB_NW * C2_NKK * C2_SLOTS * 512 * 2;
  const __amdgpu_buffer_rsrc_t wr1 = make_rsrc(a.wb1, WB), wr2 = make_rsrc(a.wb2, WB);
  constexpr int XSL = MC * C2_XW;                                                   // elements of one partial slab
  const __amdgpu_buffer_rsrc_t xr = make_rsrc(reinterpret_cast<const T*>(a.xch) + (size_t)cluster * 2 * 2 * 2 * XSL,
                                              (unsigned)(2 * 2 * 2 * XSL * sizeof(T)));
  gu32* flags = (gu32*)(a.flags + (size_t)cluster * C2_FPC);
  gu32* tmo = (gu32*)(a.flags + (size_t)a.clusters * C2_FPC);
  const T* a_lane = Abuf + n * RSA + 8 * q;
  __syncthreads();

  // Thread -> data: 24 threads per row; thread (prow, tc) owns the four 16-byte chunks c_j = tc + 24 j (j = 0..3) of the
  // row's own-unit gate block, i.e. units 2 c_j, 2 c_j + 1.  Consecutive lanes then touch consecutive 16-byte chunks in
  // LDS and in HBM, so the gate gradients go to dg straight from the registers of the cell backward (coalesced,
  // non-temporal) and their write acknowledgements return while the rest of the cell phase computes -- issued around
  // the GEMMs they stalled the weight-fragment ring, because stores and loads share the in-order vmcnt counter.
  const int tc_ = tid % TPR;
  struct Saved { u32x4 g[4]; unsigned ct[4], cp[4], dh[4]; };     // packed bf16 pairs: 28 VGPRs
  auto ld_nt4 = [](const T* p) { return __builtin_nontemporal_load(reinterpret_cast<const unsigned*>(p)); };
  // carry: sv still holds the (consumed) state of step t + 1, whose "previous cell" c_t is this step's cell state:
  // moved in registers instead of being read from HBM a second time
  auto fetch = [&](Saved& sv, const T* gs, const T* cs, const T* dh_ext, int t, bool carry) {
    if (!prow_ok || t < 0) return;
#ifdef C2_NO_FETCH
    return;                               // diagnostic: timing without the saved-state loads (results are garbage)
#endif
    int tc = tc_;
    asm volatile("" : "+v"(tc));          // keep the per-chunk addresses out of the loop-invariant (spilled) set
    const size_t e = ((size_t)t * N + row0 + prow) * H + cu * HC;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = tc + TPR * j;
      sv.g[j] = ld_nt16(gs + e * 4 + 8 * c);
      if (carry) sv.ct[j] = sv.cp[j];
      else sv.ct[j] = ld_nt4(cs + e + 2 * c);
      if (t > 0) sv.cp[j] = ld_nt4(cs + e - (size_t)N * H + 2 * c);
      if (dh_ext) sv.dh[j] = ld_nt4(dh_ext + e + 2 * c);
    }
  };
  // The same for the chunks j0 <= j < j1 only, always with carry, and UNCONDITIONAL (time and row indices are clamped into
  // the tensors; what a clamped request returns is never used): this is the form used inside the time loop, from the
  // `post` hooks of the GEMM passes -- a load that only some paths issue makes hipcc's (path-insensitive) wait-count
  // insertion drain the whole queue in front of the next fragment use.
  const long prow_c = row0 + prow < N ? row0 + prow : N - 1;
  auto fetch_part = [&](auto with_dh, Saved& sv, const T* gs, const T* cs, const T* dh_ext, int t, int j0, int j1) {
#ifdef C2_NO_FETCH
    return;
#endif
    int tc = tc_;
    asm volatile("" : "+v"(tc));
    const int tcl = t > 0 ? t : 0, tpl = t > 1 ? t - 1 : 0;
    const size_t e = ((size_t)tcl * N + prow_c) * H + cu * HC, ep = ((size_t)tpl * N + prow_c) * H + cu * HC;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j < j0 || j >= j1) continue;
      const int c = tc + TPR * j;
      sv.g[j] = ld_nt16(gs + e * 4 + 8 * c);
      sv.ct[j] = sv.cp[j];
      sv.cp[j] = ld_nt4(cs + ep + 2 * c);
      if constexpr (decltype(with_dh)::value) sv.dh[j] = ld_nt4(dh_ext + e + 2 * c);
    }
  };
  auto lo16 = [](unsigned w) { return __uint_as_float(w << 16); };
  auto hi16 = [](unsigned w) { return __uint_as_float(w & 0xffff0000u); };
  // cell backward -> gate gradients (bf16, local k = unit*4 + gate): LDS A operand + dg [t*N + row][4H] (own 768 columns)
  // The phase has NO branch on the row being valid: rows past N compute on clamped (finite) saved state, their gate
  // gradients are forced to zero for the LDS operand and their dg stores are dropped by the range check of the step's
  // buffer descriptor (base = row 0 of time t, num_records = one time step).  A store that only some paths issue made
  // hipcc wait for vmcnt(0) -- i.e. for the acknowledgement of the previous chunk's HBM store -- before every chunk.
  auto cell_bwd = [&](const Saved& sv, const float* dh_lds, bool has_ext, float* dcl, int t, T* dg) {
    int tc = tc_;
    asm volatile("" : "+v"(tc));
    const __amdgpu_buffer_rsrc_t dgr = make_rsrc(dg + (size_t)t * N * CB_K4, (unsigned)((size_t)N * CB_K4 * sizeof(T)));
    const int rowoff = (int)(((row0 + prow) * CB_K4 + cu * CB_KC) * sizeof(T));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = tc + TPR * j;
      u32x4 out;
      float2 dcv = *reinterpret_cast<const float2*>(dcl + prow * HC + 2 * c);
      const float2 dhv = *reinterpret_cast<const float2*>(dh_lds + prow * HC + 2 * c);
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const unsigned w0 = sv.g[j][2 * e], w1 = sv.g[j][2 * e + 1];   // (i, g), (f, o) of unit 2c + e
        const float iv = lo16(w0), gv = hi16(w0), fv = lo16(w1), ov = hi16(w1);
        const float ct = e ? hi16(sv.ct[j]) : lo16(sv.ct[j]);
        const float cp = t > 0 ? (e ? hi16(sv.cp[j]) : lo16(sv.cp[j])) : 0.f;
        float dh = e ? dhv.y : dhv.x;
        if (has_ext) dh += e ? hi16(sv.dh[j]) : lo16(sv.dh[j]);
        const float tch = tanh_f(ct);
        const float dct = dh * ov * (1.f - tch * tch) + (e ? dcv.y : dcv.x);
        const float dO = dh * tch * ov * (1.f - ov);
        const float di = dct * gv * iv * (1.f - iv);
        const float dgg = dct * iv * (1.f - gv * gv);
        const float df = dct * cp * fv * (1.f - fv);
        if (e) dcv.y = dct * fv; else dcv.x = dct * fv;
        out[2 * e] = (uint32_t)f2bf(di) | ((uint32_t)f2bf(dgg) << 16);
        out[2 * e + 1] = (uint32_t)f2bf(df) | ((uint32_t)f2bf(dO) << 16);
      }
      if (!prow_ok) { out = u32x4{0u, 0u, 0u, 0u}; dcv = float2{0.f, 0.f}; }
      *reinterpret_cast<float2*>(dcl + prow * HC + 2 * c) = dcv;
#ifndef C2_NO_DG
      __builtin_amdgcn_raw_buffer_store_b128(out, dgr, rowoff + 16 * c, 0, 2);       // aux 2 = nt
#endif
      *reinterpret_cast<u32x4*>(Abuf + prow * RSA + 8 * c) = out;
    }
  };
  // Exchange of partial sums, accumulator to accumulator.  Column tile tau = wave + 12 * slot is computed by the SAME wave
  // index on both CUs (pass 0 on the CU that does not own it, pass 1 on the owner), so the non-owner's wave ships its
  // accumulators as they lie in registers -- per lane and tile 2 x f32x4 -> 16 bytes of bf16, lane-contiguous = coalesced
  // write-through stores straight from the MFMA results, no LDS staging, no scatter -- and the owner's wave fetches the
  // same 16 bytes per lane during ITS pass over the own tiles and adds them in registers before its one scatter.
  // (Round 2 staged the partials row-major in LDS, published the slab, and added the partner's slab into the fp32 LDS
  // buffers in a phase of its own: poll + barrier + exposed load latency + LDS read-modify-write + barrier = 8 % of a
  // layer-step, twice per step.)  Slab (layer, parity, producing cu): [wave 12][tile-of-pass 2][lane 64][16 B] = 24 KB.
  auto pack_acc = [](const f32x4 (&acc)[2]) {
    u32x4 o;
    o[0] = (uint32_t)f2bf(acc[0][0]) | ((uint32_t)f2bf(acc[0][1]) << 16);
    o[1] = (uint32_t)f2bf(acc[0][2]) | ((uint32_t)f2bf(acc[0][3]) << 16);
    o[2] = (uint32_t)f2bf(acc[1][0]) | ((uint32_t)f2bf(acc[1][1]) << 16);
    o[3] = (uint32_t)f2bf(acc[1][2]) | ((uint32_t)f2bf(acc[1][3]) << 16);
    return o;
  };
  auto add_packed = [](f32x4 (&acc)[2], const u32x4 v) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      acc[mt][0] += __uint_as_float(v[2 * mt] << 16);
      acc[mt][1] += __uint_as_float(v[2 * mt] & 0xffff0000u);
      acc[mt][2] += __uint_as_float(v[2 * mt + 1] << 16);
      acc[mt][3] += __uint_as_float(v[2 * mt + 1] & 0xffff0000u);
    }
  };
  // one 16-column tile of accumulators (slot sl of this wave: column tile tau = wave + 12*sl, owned by this CU) -> the
  // fp32 LDS buffer of the outputs
  auto scatter = [&](const f32x4 (&acc)[2], int sl, int layer) {
    // the lane-dependent part of every address below is re-derived from an opaque copy of (n, q) on each call: as loop
    // invariants the compiler hoisted ~50 precomputed addresses out of the time loop and spilled them to scratch, whose
    // reloads (global-memory latency) made this epilogue cost more than the GEMM
    int n = n_, q = q_;
    asm volatile("" : "+v"(n), "+v"(q));
    const int tau = wave + CB_NW * sl;
    int ld_own = HC;
    float* own_dst;
    bool overwrite = true;
    if (layer == 1) {          // LSTM layer 2: tiles 0-23 d h1 (accumulate onto the recurrent part), 24-47 d h2 rec
      const int blk = tau / 24, tt = tau % 24;
      own_dst = (blk == 0 ? dh1buf : dhrec2) + (tt % 12) * 16;
      overwrite = blk == 1;
    } else if (tau < 24) {     // LSTM layer 1: tiles 0-23 d h1 rec
      own_dst = dh1buf + (tau % 12) * 16;
    } else {                   // LSTM layer 1: tiles 24-27 d x (columns 0-31 -> cu 0, 32-63 -> cu 1)
      own_dst = dxbuf + ((tau - 24) % 2) * 16;
      ld_own = 32;
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float* d = own_dst + (16 * mt + 4 * q + j) * ld_own + n;
        *d = overwrite ? acc[mt][j] : *d + acc[mt][j];
      }
  };
  // the backward GEMM of one layer in two passes: first the column tiles the PARTNER owns (shipped at once, so the
  // hand-off flies during the second pass), then the own ones, to which the partner's partials are added in registers
  // HEAD: the head's contribution to d h2_tau, dY_tau . Wh restricted to the own units, is formed here: the 32 dY rows of a
  // step (1 KB) arrive by ONE LDS-DMA instruction, every wave turns them into its 16-unit column tile with two MFMAs and
  // adds the tile to dhrec2 -- no dh2 tensor is written by a head kernel or fetched with the saved state.  The DMA is
  // issued by EVERY wave (same source, same destination, same bytes: 11 redundant L2 hits per step) and for tau = -1 too
  // (clamped): an instruction that only one wave / some steps issue is one more conditional memory operation for the
  // wait-count insertion to be pessimistic about.
  auto dy_dma = [&](int tau) {
    if constexpr (HEAD) {
      typedef __attribute__((address_space(3))) void lds_void;
      const int tcl = tau > 0 ? tau : 0;
      // per-lane clamp to the last 16 bytes of dyt [Tn][N][16]: the rows past N of a ragged last cluster are garbage
      // that nothing uses (their gate gradients are zeroed), but at the last time step they lie BEHIND the tensor -- an
      // exact-size dyt at the end of a memory segment faulted here (round 4, tests/test_lstm_gpu.py)
      size_t off = (((size_t)tcl * N + row0) * 16) * sizeof(T) + lane * 16;
      const size_t last = (size_t)a.Tn * N * 16 * sizeof(T) - 16;
      off = off < last ? off : last;
      c4_lds_dma16(reinterpret_cast<const unsigned char*>(a.dyt) + off, (unsigned)(unsigned long)(lds_void*)dys);
    }
  };
  // The two passes of a layer -- the partner's column tiles, shipped at once, then the own ones, to which the partner's
  // partials are added in registers -- run as ONE fragment stream of 48 k-steps (C2Sched): the ring is not restarted and
  // nothing is drained between them.  (The first round-3 version ended pass 0 with its stores, restarted the ring and
  // waited for vmcnt(0) in front of the flag: 1.6-2.8k cycles per layer.)  The flag of a wave's shipment goes up eight
  // k-steps into the own tiles, behind a counted wait that covers exactly the shipment's stores; the partner's partials
  // are requested sixteen k-steps in; `nsv` ..., the saved state of THIS layer's next (earlier) step, behind the last
  // fragment request (its registers are free: this step's cell phase of the layer is over), so that it has the scatter,
  // the barrier and the other layer's whole cell phase to arrive before the next stream queues up behind it.
  auto layer_gemm = [&](auto layer_c, const void* wpacked, int ep, Saved& nsv, const T* ngs, const T* ncs, const T* ndh, int nt) {
    constexpr int layer = decltype(layer_c)::value;
    constexpr bool l2 = layer == 1;
    typedef std::integral_constant<bool, l2 && !HEAD> with_dh;             // only layer 2 without the fused head reads d h2
#ifdef C2_NO_FETCH
    constexpr int P1 = 0;
#else
    constexpr int P1 = 4 * (2 + (with_dh::value ? 1 : 0));
#endif
    constexpr int RD = C2_DEPTH - 1;
    const unsigned char* wbase = reinterpret_cast<const unsigned char*>(wpacked) + (size_t)(cu * CB_NW + wave) * (C2_NKK * C2_SLOTS * 1024);
    const bool xw = !l2 && wave < 4;                   // this wave also has a d x tile (slot 2)
    const int xo = wave >> 1;                          // ... owned by CU xo
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int slab = ((layer * 2 + (ep & 1)) * 2) * XSL * 2;                      // byte offset of (layer, parity); + cu * XSL * 2
    const int my = (wave * 2 * 64 + ln) * 16;                                     // this lane's 16 bytes of tile 0; tile 1: + 1024
    gu32* myflag = flags + (layer * CB_G + cu) * CB_NW + wave;
    const gu32* pflag = flags + (layer * CB_G + pcu) * CB_NW + wave;
    int zoff = 0, lane16 = ln * 16;
    asm volatile("" : "+v"(zoff));
    auto stream = [&](auto ns0c, auto ns1c) __attribute__((always_inline)) {
      constexpr int NS0 = decltype(ns0c)::value, NS1 = decltype(ns1c)::value;
      constexpr C2Sched<NS0, NS1, (l2 && HEAD) ? 1 : 0, P1, RD> sch{};
      // slots: pass 0 = the partner's tiles (pcu, and pcu + 2 / the d x tile), pass 1 = the own ones
      const int a0 = pcu, a1 = l2 ? pcu + 2 : 2, b0 = cu, b1 = l2 ? cu + 2 : 2;
      const unsigned char* w00 = wbase + a0 * 1024;
      const unsigned char* w01 = wbase + a1 * 1024;
      const unsigned char* w10 = wbase + b0 * 1024;
      const unsigned char* w11 = wbase + b1 * 1024;
      asm volatile("" : "+s"(w00), "+s"(w01), "+s"(w10), "+s"(w11));      // opaque per layer: 96 hoisted fragment addresses were spilled
      bf16x8 b[C2_DEPTH][2];
      f32x4 acc[2][2];
      unsigned fl = 0;
      u32x4 p0 = {0u, 0u, 0u, 0u}, p1 = {0u, 0u, 0u, 0u};
      auto request = [&](auto gc) __attribute__((always_inline)) {
        constexpr int g = decltype(gc)::value, kk = g % 24;
        b[g % C2_DEPTH][0] = c2_ring_load((g < 24 ? w00 : w10) + kk * (C2_SLOTS * 1024), lane16);
        if constexpr ((g < 24 ? NS0 : NS1) == 2) b[g % C2_DEPTH][1] = c2_ring_load((g < 24 ? w01 : w11) + kk * (C2_SLOTS * 1024), lane16);
      };
      c4_static_for<RD>([&](auto dc) __attribute__((always_inline)) { request(dc); });
      c4_static_for<48>([&](auto gc) __attribute__((always_inline)) {
        constexpr int g = decltype(gc)::value, kk = g % 24, ns = g < 24 ? NS0 : NS1;
        // ---- side
        if constexpr (g == 32) {                        // this wave's shipment acknowledged -> its epoch (Guideline 16 R1 per wave)
          c4_wait_mem(sch.wait_raise());
          if (lane == 0) __hip_atomic_store(myflag, (unsigned)ep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if constexpr (g == 36) asm volatile("global_load_dword %0, %1, %2 sc1" : "=v"(fl) : "v"(zoff), "s"(pflag) : "memory");
        if constexpr (g == 40) {
          c4_wait_reg(sch.wait_flag(), fl);
          unsigned spins = 0;
          while (CF_POLL && fl < (unsigned)ep) {        // (no exit with a request still pending: the compiler would wait for it)
            if (++spins > SPIN_LIMIT) {
              if (lane == 0) __hip_atomic_fetch_add(tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              break;
            }
            __builtin_amdgcn_s_sleep(1);
            fl = __hip_atomic_load(pflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          p0 = load_sc1_b128(xr, slab + pcu * XSL * 2 + my);
          if constexpr (NS1 == 2) p1 = load_sc1_b128(xr, slab + pcu * XSL * 2 + my + 1024);
        }
        // ---- request, post, wait, MFMAs
        if constexpr (g + RD < 48) request(std::integral_constant<int, (g + RD < 48 ? g + RD : 0)>{});
        if constexpr (g == 47 - RD) fetch_part(with_dh{}, nsv, ngs, ncs, ndh, nt, 0, 4);
        if constexpr (C2_KBAR > 0 && g > 0 && g % C2_KBAR == 0) asm volatile("s_barrier" ::: "memory");   // keeps the twelve waves in step
        if constexpr (ns == 2) c2_wait(sch.wait_frag(g), b[g % C2_DEPTH][0], b[g % C2_DEPTH][1]);
        else c2_wait(sch.wait_frag(g), b[g % C2_DEPTH][0]);
        if constexpr (kk == 0) {
#pragma unroll
          for (int i = 0; i < 2; ++i) acc[i][0] = acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        bf16x8 af[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) af[mt] = load_frag<bf16_t>(a_lane + 16 * mt * RSA + 32 * kk);
#pragma unroll
        for (int sI = 0; sI < ns; ++sI)
#pragma unroll
          for (int mt = 0; mt < 2; ++mt) acc[sI][mt] = mma16(af[mt], b[g % C2_DEPTH][sI], acc[sI][mt]);
        // ---- tail of pass 0: ship the partner's tiles straight from the accumulators
        if constexpr (g == 23) {
          if constexpr (l2) dy_dma(nt);                 // HEAD: the dY rows of the next (earlier) step
          store_sc1_b128(xr, slab + cu * XSL * 2 + my, pack_acc(acc[0]));
          if constexpr (NS0 == 2) store_sc1_b128(xr, slab + cu * XSL * 2 + my + 1024, pack_acc(acc[1]));
        }
      });
      add_packed(acc[0], p0);
      scatter(acc[0], b0, layer);
      if constexpr (NS1 == 2) {
        add_packed(acc[1], p1);
        scatter(acc[1], b1, layer);
      }
    };
    typedef std::integral_constant<int, 1> one;
    typedef std::integral_constant<int, 2> two;
    if (l2) stream(two{}, two{});
    else if (!xw) stream(one{}, one{});
    else if (xo == pcu) stream(two{}, one{});
    else stream(one{}, two{});
    __syncthreads();                                   // the layer's outputs complete in LDS
  };

  auto head_add = [&]() {
    if constexpr (HEAD) {
      int nn = lane & 15, qq = lane >> 4;
      asm volatile("" : "+v"(nn), "+v"(qq));
      bf16x8 bfr = {0, 0, 0, 0, 0, 0, 0, 0};
      if (qq < 2) bfr = load_frag<T>(whs + (16 * wave + nn) * 16 + 8 * qq);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        bf16x8 af = {0, 0, 0, 0, 0, 0, 0, 0};
        if (qq < 2) af = load_frag<T>(dys + (16 * mt + nn) * 16 + 8 * qq);
        const f32x4 hv = mma16(af, bfr, f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
        for (int j = 0; j < 4; ++j) dhrec2[(16 * mt + 4 * qq + j) * HC + 16 * wave + nn] += hv[j];
      }
    }
  };
  const T* dh2_src = HEAD ? nullptr : dh2;

  Saved sv2, sv1;
  fetch(sv2, g2, c2, dh2_src, a.Tn - 1, false);
  fetch(sv1, g1, c1, nullptr, a.Tn - 1, false);
  if constexpr (HEAD) {
    dy_dma(a.Tn - 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    head_add();
    __syncthreads();
  }
#ifdef C2_STAMP
  st_last = __builtin_readcyclecounter();
#endif

#pragma unroll 1
  for (int t = a.Tn - 1; t >= 0; --t) {
    const int ep = a.Tn - t;
    // Loads, stores and the weight-fragment ring share one in-order vmcnt queue per wave, so HBM loads issued in front of
    // a GEMM make its first weight fragments wait out the HBM latency: the saved state of a phase is requested from inside
    // the same layer's GEMM of the step before (layer_gemm).  (Measured: issuing one phase's 61-84 KB of state loads
    // occupies a CU's memory queue for ~12k cycles -- per-CU miss parallelism, not chip-wide HBM contention: starting the
    // clusters an eighth of a step apart changed nothing.)
    // ---------------- LSTM layer 2 (exchange layer index 1)
    cell_bwd(sv2, dhrec2, !HEAD, dc2, t, dg2T);
    C2T(0)
#ifdef C2_RAWBAR
    lds_barrier();                                     // own dgates complete in LDS; everyone done reading dhrec2
#else
    __syncthreads();                                   // own dgates complete in LDS; everyone done reading dhrec2
#endif
    C2T(1)
    C2T(2)
    layer_gemm(std::integral_constant<int, 1>{}, a.wb2, ep, sv2, g2, c2, dh2_src, t - 1);   // d h1_t (both contributions) and d h2_{t-1} final
    if (t > 0) head_add();                             // HEAD: + dY_{t-1} . Wh (read by the next step's cell phase, barriers between)
    // ---------------- LSTM layer 1 (exchange layer index 0)
    cell_bwd(sv1, dh1buf, false, dc1, t, dg1T);
    C2T(9)
#ifdef C2_RAWBAR
    lds_barrier();
#else
    __syncthreads();
#endif
    C2T(10)
    layer_gemm(std::integral_constant<int, 0>{}, a.wb1, ep, sv1, g1, c1, nullptr, t - 1);   // d h1_{t-1} (recurrent) and this CU's 32 columns of d x final
    C2T(11)
    int ti = tid;
    asm volatile("" : "+v"(ti));                         // (a hoisted per-lane d x address was spilled: its reload sat behind the barrier)
    for (int i = ti; i < MC * 32; i += NT) {
      const int r = i / 32, c = i % 32;
      if (row0 + r < N) store1_nt<T>(dx + ((size_t)t * N + row0 + r) * KX + cu * 32 + c, dxbuf[i]);
    }
    // (dxbuf is rewritten by the next layer-1 GEMM only, several barriers later)
  }
#ifdef C2_STAMP
  C2_STAMP_FINI
#endif
}

// =====================================================================================================
// K-split backward on FOUR-CU clusters (round 3): 64 sequences per cluster, CU `cu` owns hidden units [96 cu, 96 cu + 96)
// of both layers.  Same data flow as lstm2_coop_bwd2_kernel -- every CU multiplies ITS OWN gate-gradient quarter
// [64][384] with the weight rows of its own units and all output columns, keeps the partial sums of the output tiles it
// owns and ships the rest as bf16, accumulator to accumulator, to the same wave of the owner -- but a CU streams a
// QUARTER of the weight fragments per step for 64 rows (0.93 MB instead of 1.87 MB for 32): the launches are bound by
// the fragment stream, and halving the k-steps of the pair kernel (a diagnostic build) took 7.4 -> 5.5 ms.
// Per layer and step: three partner passes (tiles of owner cu+1, cu+2, cu+3: shipped after each pass, flagged per wave
// in front of the next one), then the own pass, whose accumulators start from the first partner's partials; the second
// partner's arrive during the own pass, the third's -- shipped one pass earlier -- are waited for behind it.
//   layer 2 output tiles of owner O: wave w < 6 -> d h1 columns 96 O + 16 w, wave w >= 6 -> d h2_{t-1} columns 96 O + 16 (w - 6)
//   layer 1 output tiles of owner O: wave w < 6 -> d h1_{t-1} columns 96 O + 16 w, wave 6 -> d x columns 16 O .. 16 O + 15
constexpr int C4_G = 4, C4_MC = 64, C4_HC = 96, C4_KC = 4 * C4_HC, C4_NKK = C4_KC / 32;   // 384 own gate columns = 12 k-steps
constexpr int C4_FPC = 2 * C4_G * 3 * CB_NW;                // epoch words per cluster: [layer][src cu][shipment][wave]
constexpr int C4_SLAB = CB_NW * 64 * 32;                    // bytes of one shipment: [wave][lane][32 B] (4 row tiles x 4 values, bf16)
constexpr long C4_XCH = 2L * 2 * C4_G * 3 * C4_SLAB;        // bytes per cluster: [layer][parity][src cu][shipment]
constexpr long C4_WFRAG = (long)C4_NKK * C4_G * 1024;       // packed weights of one (cu, wave): [kk][owner][lane][8] bf16

// The four passes of a layer (three partner tiles, then the own tile: 48 k-steps) run as ONE fragment stream -- the ring never
// restarts cold at a pass boundary and nothing is drained between passes.  Program order per position g = 12 p + k:
//   side(g)   g = 20, 32, 44: raise the epoch of shipment 0, 1, 2 (1 store, behind a counted wait for the shipment's stores);
//             g = 24, 36: request the epoch of the partner whose partials are due (1 load); g = 28, 40: its partials (2 loads)
//   request of fragment g + 3
//   wait + the four MFMAs of k-step g
//   tail(g)   g = 11, 23, 35: ship the finished partner tile (2 stores; g = 11 of layer 2 with the fused head: + 2 dY DMAs)
// Every wait is vmcnt(M), M = operations issued after the one waited for (see c2_ring_load); this table holds the counts.
#ifndef C4_DEPTH
#define C4_DEPTH 8                                          // fragment ring of the four-CU kernel: C4_DEPTH - 1 k-steps (1 KB each) requested ahead
#endif
template <int DY, int D /* k-steps requested ahead */>
struct C4Sched {
  int before[49];    // vector-memory operations issued before side(pos)
  int req_seq[48];   // sequence number of the request of fragment g
  static constexpr int side(int pos) {
    return (pos == 20 || pos == 32 || pos == 44 || pos == 24 || pos == 36) ? 1 : ((pos == 28 || pos == 40) ? 2 : 0);
  }
  static constexpr int tail(int pos) { return pos == 11 ? 2 + DY : ((pos == 23 || pos == 35) ? 2 : 0); }
  constexpr C4Sched() : before{}, req_seq{} {
    int c = D;
    for (int g = 0; g < D; ++g) req_seq[g] = g;
    for (int pos = 0; pos < 48; ++pos) {
      before[pos] = c;
      c += side(pos);
      if (pos + D < 48) { req_seq[pos + D] = c; c += 1; }
      c += tail(pos);
    }
    before[48] = c;
  }
  constexpr int wait_frag(int g) const { return before[g] + side(g) + (g + D < 48 ? 1 : 0) - req_seq[g] - 1; }
  constexpr int wait_raise(int q) const { return before[12 * q + 20] - before[12 * q + 12]; }   // since the shipment's last store
  constexpr int wait_flag(int pos_f) const { return before[pos_f + 4] - before[pos_f] - 1; }     // since the epoch request
};

template <bool HEAD>
__global__ __launch_bounds__(CB_NT) void lstm2_coop_bwd4_kernel(CoopBwd2Args a) {
  typedef bf16_t T;
  constexpr int MC = C4_MC, H = CB_H, HC = C4_HC, KX = CB_KX, NT = CB_NT;
  constexpr int RSA = C4_KC + COOP_APAD;                            // A tile row stride (elements): own gate columns only
  constexpr int TPR = NT / MC, UPT = HC / TPR;              // 12 threads per row, 8 units per thread
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* Abuf = reinterpret_cast<T*>(smem_raw);                                          // [64][RSA]   own dgates
  float* dh1buf = reinterpret_cast<float*>(smem_raw + (size_t)MC * RSA * sizeof(T));  // [64][HC]    d h1 (own units)
  float* dhrec2 = dh1buf + MC * HC;                                                   // [64][HC]    d h2 recurrent
  float* dxbuf = dhrec2 + MC * HC;                                                    // [64][16]    own d x columns
  float* dcbuf = dxbuf + MC * 16;                                                     // [2][64][HC] d c carried from step to step
  T* whs = reinterpret_cast<T*>(dcbuf + 2 * MC * HC);                                 // HEAD: [HC][16] own units' head weights
  T* dys = whs + HC * 16;                                                             // HEAD: [64][16]  dY rows of one step

  const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, q = lane >> 4, n_ = n, q_ = q;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int cluster, cu;
  coop_ids(C4_G, cluster, cu);
  if (cluster >= a.clusters) return;
  const long row0 = (long)cluster * MC;
  const long N = a.N;
  for (int i = tid; i < 4 * MC * HC + MC * 16; i += NT) dh1buf[i] = 0.f;      // d h1, d h2 rec, d x, d c (both layers)
  for (int i = tid; i < MC * RSA; i += NT) Abuf[i] = 0;
  if constexpr (HEAD) {
    const T* whT = reinterpret_cast<const T*>(a.whT);
    for (int i = tid; i < HC * 16; i += NT) whs[i] = whT[(size_t)(cu * HC + i / 16) * 32 + (i % 16)];
  }

  const int prow = tid / TPR;
  const bool prow_ok = row0 + prow < N;
  // (d c of the thread's 8 units per layer lives in LDS, not in 16 registers: both layers' saved state is in flight at
  // the end of a GEMM pass since round 3, and the registers are what that costs)
  float* dc1 = dcbuf;
  float* dc2 = dcbuf + MC * HC;

  const T* g1 = reinterpret_cast<const T*>(a.g1);
  const T* g2 = reinterpret_cast<const T*>(a.g2);
  const T* c1 = reinterpret_cast<const T*>(a.c1);
  const T* c2 = reinterpret_cast<const T*>(a.c2);
  const T* dh2 = reinterpret_cast<const T*>(a.dh2);
  T* dx = reinterpret_cast<T*>(a.dx);
  T* dg1T = reinterpret_cast<T*>(a.dg1);
  T* dg2T = reinterpret_cast<T*>(a.dg2);
  const __amdgpu_buffer_rsrc_t xr = make_rsrc(reinterpret_cast<const unsigned char*>(a.xch) + (size_t)cluster * C4_XCH,
                                              (unsigned)C4_XCH);
  gu32* flags = (gu32*)(a.flags + (size_t)cluster * C4_FPC);
  gu32* tmo = (gu32*)(a.flags + (size_t)a.clusters * C4_FPC);
  const T* a_lane = Abuf + n * RSA + 8 * q;
  __syncthreads();

  // Thread -> data: 12 threads per row; thread (prow, tc) owns the four 16-byte chunks c_j = tc + 12 j (j = 0..3) of the
  // row's own-unit gate block, i.e. units 2 c_j, 2 c_j + 1.  Consecutive lanes then touch consecutive 16-byte chunks in
  // LDS and in HBM, so the gate gradients go to dg straight from the registers of the cell backward (coalesced,
  // non-temporal) and their write acknowledgements return while the rest of the cell phase computes -- issued around
  // the GEMMs they stalled the weight-fragment ring, because stores and loads share the in-order vmcnt counter.
  const int tc_ = tid % TPR;
  struct Saved { u32x4 g[4]; unsigned ct[4], cp[4], dh[4]; };     // packed bf16 pairs: 28 VGPRs
  auto ld_nt4 = [](const T* p) { return __builtin_nontemporal_load(reinterpret_cast<const unsigned*>(p)); };
  // carry: sv still holds the (consumed) state of step t + 1, whose "previous cell" c_t is this step's cell state:
  // moved in registers instead of being read from HBM a second time
  auto fetch = [&](Saved& sv, const T* gs, const T* cs, const T* dh_ext, int t, bool carry) {
    if (!prow_ok || t < 0) return;
#ifdef C2_NO_FETCH
    return;                               // diagnostic: timing without the saved-state loads (results are garbage)
#endif
    int tc = tc_;
    asm volatile("" : "+v"(tc));          // keep the per-chunk addresses out of the loop-invariant (spilled) set
    const size_t e = ((size_t)t * N + row0 + prow) * H + cu * HC;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = tc + TPR * j;
      sv.g[j] = ld_nt16(gs + e * 4 + 8 * c);
      if (carry) sv.ct[j] = sv.cp[j];
      else sv.ct[j] = ld_nt4(cs + e + 2 * c);
      if (t > 0) sv.cp[j] = ld_nt4(cs + e - (size_t)N * H + 2 * c);
      if (dh_ext) sv.dh[j] = ld_nt4(dh_ext + e + 2 * c);
    }
  };
  // The same for the chunks j0 <= j < j1 only, always with carry, and UNCONDITIONAL (time and row indices are clamped into
  // the tensors; what a clamped request returns is never used): this is the form used inside the time loop, from the
  // `post` hooks of the GEMM passes -- a load that only some paths issue makes hipcc's (path-insensitive) wait-count
  // insertion drain the whole queue in front of the next fragment use.
  const long prow_c = row0 + prow < N ? row0 + prow : N - 1;
  auto fetch_part = [&](auto with_dh, Saved& sv, const T* gs, const T* cs, const T* dh_ext, int t, int j0, int j1) {
#ifdef C2_NO_FETCH
    return;
#endif
    int tc = tc_;
    asm volatile("" : "+v"(tc));
    const int tcl = t > 0 ? t : 0, tpl = t > 1 ? t - 1 : 0;
    const size_t e = ((size_t)tcl * N + prow_c) * H + cu * HC, ep = ((size_t)tpl * N + prow_c) * H + cu * HC;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (j < j0 || j >= j1) continue;
      const int c = tc + TPR * j;
      sv.g[j] = ld_nt16(gs + e * 4 + 8 * c);
      sv.ct[j] = sv.cp[j];
      sv.cp[j] = ld_nt4(cs + ep + 2 * c);
      if constexpr (decltype(with_dh)::value) sv.dh[j] = ld_nt4(dh_ext + e + 2 * c);
    }
  };
  auto lo16 = [](unsigned w) { return __uint_as_float(w << 16); };
  auto hi16 = [](unsigned w) { return __uint_as_float(w & 0xffff0000u); };
  // cell backward -> gate gradients (bf16, local k = unit*4 + gate): LDS A operand + dg [t*N + row][4H] (own 768 columns)
  // The phase has NO branch on the row being valid: rows past N compute on clamped (finite) saved state, their gate
  // gradients are forced to zero for the LDS operand and their dg stores are dropped by the range check of the step's
  // buffer descriptor (base = row 0 of time t, num_records = one time step).  A store that only some paths issue made
  // hipcc wait for vmcnt(0) -- i.e. for the acknowledgement of the previous chunk's HBM store -- before every chunk.
  auto cell_bwd = [&](const Saved& sv, const float* dh_lds, bool has_ext, float* dcl, int t, T* dg) {
    int tc = tc_;
    asm volatile("" : "+v"(tc));
    const __amdgpu_buffer_rsrc_t dgr = make_rsrc(dg + (size_t)t * N * CB_K4, (unsigned)((size_t)N * CB_K4 * sizeof(T)));
    const int rowoff = (int)(((row0 + prow) * CB_K4 + cu * C4_KC) * sizeof(T));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = tc + TPR * j;
      u32x4 out;
      float2 dcv = *reinterpret_cast<const float2*>(dcl + prow * HC + 2 * c);
      const float2 dhv = *reinterpret_cast<const float2*>(dh_lds + prow * HC + 2 * c);
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const unsigned w0 = sv.g[j][2 * e], w1 = sv.g[j][2 * e + 1];   // (i, g), (f, o) of unit 2c + e
        const float iv = lo16(w0), gv = hi16(w0), fv = lo16(w1), ov = hi16(w1);
        const float ct = e ? hi16(sv.ct[j]) : lo16(sv.ct[j]);
        const float cp = t > 0 ? (e ? hi16(sv.cp[j]) : lo16(sv.cp[j])) : 0.f;
        float dh = e ? dhv.y : dhv.x;
        if (has_ext) dh += e ? hi16(sv.dh[j]) : lo16(sv.dh[j]);
        const float tch = tanh_f(ct);
        const float dct = dh * ov * (1.f - tch * tch) + (e ? dcv.y : dcv.x);
        const float dO = dh * tch * ov * (1.f - ov);
        const float di = dct * gv * iv * (1.f - iv);
        const float dgg = dct * iv * (1.f - gv * gv);
        const float df = dct * cp * fv * (1.f - fv);
        if (e) dcv.y = dct * fv; else dcv.x = dct * fv;
        out[2 * e] = (uint32_t)f2bf(di) | ((uint32_t)f2bf(dgg) << 16);
        out[2 * e + 1] = (uint32_t)f2bf(df) | ((uint32_t)f2bf(dO) << 16);
      }
      if (!prow_ok) { out = u32x4{0u, 0u, 0u, 0u}; dcv = float2{0.f, 0.f}; }
      *reinterpret_cast<float2*>(dcl + prow * HC + 2 * c) = dcv;
#ifndef C2_NO_DG
      __builtin_amdgcn_raw_buffer_store_b128(out, dgr, rowoff + 16 * c, 0, 2);       // aux 2 = nt
#endif
      *reinterpret_cast<u32x4*>(Abuf + prow * RSA + 8 * c) = out;
    }
  };
  auto pack4 = [](const f32x4 (&acc)[4], int half) {
    u32x4 o;
    o[0] = (uint32_t)f2bf(acc[2 * half][0]) | ((uint32_t)f2bf(acc[2 * half][1]) << 16);
    o[1] = (uint32_t)f2bf(acc[2 * half][2]) | ((uint32_t)f2bf(acc[2 * half][3]) << 16);
    o[2] = (uint32_t)f2bf(acc[2 * half + 1][0]) | ((uint32_t)f2bf(acc[2 * half + 1][1]) << 16);
    o[3] = (uint32_t)f2bf(acc[2 * half + 1][2]) | ((uint32_t)f2bf(acc[2 * half + 1][3]) << 16);
    return o;
  };
  auto add4 = [](f32x4 (&acc)[4], const u32x4 (&v)[2]) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      const unsigned w0 = v[mt >> 1][2 * (mt & 1)], w1 = v[mt >> 1][2 * (mt & 1) + 1];
      acc[mt][0] += __uint_as_float(w0 << 16);
      acc[mt][1] += __uint_as_float(w0 & 0xffff0000u);
      acc[mt][2] += __uint_as_float(w1 << 16);
      acc[mt][3] += __uint_as_float(w1 & 0xffff0000u);
    }
  };
  // this wave's own output tile (accumulators with all partials added) -> the fp32 LDS buffer of the outputs
  auto scatter = [&](const f32x4 (&acc)[4], int layer) {
    int n = n_, q = q_;
    asm volatile("" : "+v"(n), "+v"(q));
    int ld_own = HC;
    float* own_dst;
    bool overwrite = true;
    if (layer == 1) {          // LSTM layer 2: waves 0-5 d h1 (accumulate onto the recurrent part), waves 6-11 d h2 rec
      own_dst = (wave < 6 ? dh1buf : dhrec2) + (wave % 6) * 16;
      overwrite = wave >= 6;
    } else if (wave < 6) {     // LSTM layer 1: d h1 rec
      own_dst = dh1buf + wave * 16;
    } else {                   // LSTM layer 1, wave 6: the CU's 16 columns of d x
      own_dst = dxbuf;
      ld_own = 16;
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float* d = own_dst + (16 * mt + 4 * q + j) * ld_own + n;
        *d = overwrite ? acc[mt][j] : *d + acc[mt][j];
      }
  };

  // HEAD: the dY rows of a step (64 x 32 B) by two LDS-DMA instructions issued by every wave (same bytes, same place)
  auto dy_dma = [&](int tau) {
    if constexpr (HEAD) {
      typedef __attribute__((address_space(3))) void lds_void;
      const int tcl = tau > 0 ? tau : 0;
      // (per-lane clamp to the last 16 bytes of dyt: see the CU-pair kernel)
      const size_t off = (((size_t)tcl * N + row0) * 16) * sizeof(T) + lane * 16;
      const size_t last = (size_t)a.Tn * N * 16 * sizeof(T) - 16;
      const unsigned char* base = reinterpret_cast<const unsigned char*>(a.dyt);
      const unsigned lb = (unsigned)(unsigned long)(lds_void*)dys;
      c4_lds_dma16(base + (off < last ? off : last), lb);
      c4_lds_dma16(base + (off + 1024 < last ? off + 1024 : last), lb + 1024);
    }
  };

  auto layer_gemm = [&](auto layer_c, const void* wpacked, int ep, Saved& nsv, const T* ngs, const T* ncs, const T* ndh, int nt) {
    constexpr int layer = decltype(layer_c)::value;
    constexpr bool l2 = layer == 1;
    typedef std::integral_constant<bool, l2 && !HEAD> with_dh;
    constexpr int RD = C4_DEPTH - 1;
    constexpr C4Sched<(l2 && HEAD) ? 2 : 0, RD> sch{};
    const bool active = l2 || wave < 7;                // layer 1 has 7 output tiles per owner: waves 7-11 only fetch their saved state
    if (!active) {
      fetch_part(with_dh{}, nsv, ngs, ncs, ndh, nt, 0, 4);
    } else {
      const unsigned char* wbase = reinterpret_cast<const unsigned char*>(wpacked) + (size_t)(cu * CB_NW + wave) * C4_WFRAG;
      int ln = lane;
      asm volatile("" : "+v"(ln));
      const int slabs = (layer * 2 + (ep & 1)) * C4_G * 3 * C4_SLAB;        // byte offset of (layer, parity); + (src * 3 + shipment) * C4_SLAB
      const int my = (wave * 64 + ln) * 32;
      gu32* lflags = flags + layer * (C4_G * 3 * CB_NW);                     // [src cu][shipment][wave]
      int zoff = 0, lane16 = ln * 16;
      asm volatile("" : "+v"(zoff));
      // fragment bases of the four passes (owner cu + 1, cu + 2, cu + 3, cu): opaque per layer, or 48 addresses are hoisted and spilled
      const unsigned char* wp0 = wbase + ((cu + 1) & 3) * 1024;
      const unsigned char* wp1 = wbase + ((cu + 2) & 3) * 1024;
      const unsigned char* wp2 = wbase + ((cu + 3) & 3) * 1024;
      const unsigned char* wp3 = wbase + cu * 1024;
      asm volatile("" : "+s"(wp0), "+s"(wp1), "+s"(wp2), "+s"(wp3));
      auto frag_ptr = [&](int g) { return (g < 12 ? wp0 : (g < 24 ? wp1 : (g < 36 ? wp2 : wp3))) + (g % 12) * (C4_G * 1024); };
      // receive side: cu + 3 (= cu - 1) shipped my tile in ITS pass 0, cu + 2 in its pass 1, cu + 1 in its pass 2
      const gu32* pfa = lflags + (((cu + 3) & 3) * 3 + 0) * CB_NW + wave;
      const gu32* pfb = lflags + (((cu + 2) & 3) * 3 + 1) * CB_NW + wave;
      const gu32* pfc = lflags + (((cu + 1) & 3) * 3 + 2) * CB_NW + wave;
      const int offa = slabs + (((cu + 3) & 3) * 3 + 0) * C4_SLAB + my, offb = slabs + (((cu + 2) & 3) * 3 + 1) * C4_SLAB + my,
                offc = slabs + (((cu + 1) & 3) * 3 + 2) * C4_SLAB + my;
      unsigned fla = 0, flb = 0;
      u32x4 pa[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}}, pb[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
      auto poll = [&](unsigned& fl, const gu32* pf) {   // (no exit with a request pending)
#ifdef C4_DIAG_NOPOLL
        return;                                         // diagnostic: never wait for a partner (garbage results, timing only)
#endif
        unsigned spins = 0;
        while (CF_POLL && fl < (unsigned)ep) {
          if (++spins > SPIN_LIMIT) {
            if (lane == 0) __hip_atomic_fetch_add(tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
          __builtin_amdgcn_s_sleep(1);
          fl = __hip_atomic_load(pf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      };
      bf16x8 b[C4_DEPTH];
      f32x4 acc[4];
#pragma unroll
      for (int d = 0; d < RD; ++d) b[d] = c2_ring_load(frag_ptr(d), lane16);
      // (static_for: the position is a template constant of each step -- a 48-iteration `#pragma unroll` loop of this size was
      // only partially unrolled, leaving the waits as run-time switch ladders and the ring as indexed registers)
      c4_static_for<48>([&](auto gc) __attribute__((always_inline)) {
        constexpr int g = decltype(gc)::value, k = g % 12, p = g / 12;
        // ---- side
        if constexpr (g == 20 || g == 32 || g == 44) {  // shipment (g - 20) / 12 acknowledged -> this wave's epoch for it
          constexpr int q = (g - 20) / 12;
          c4_wait_mem(sch.wait_raise(q));
          if (lane == 0) __hip_atomic_store(lflags + (cu * 3 + q) * CB_NW + wave, (unsigned)ep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if constexpr (g == 24) asm volatile("global_load_dword %0, %1, %2 sc1" : "=v"(fla) : "v"(zoff), "s"(pfa) : "memory");
        if constexpr (g == 36) asm volatile("global_load_dword %0, %1, %2 sc1" : "=v"(flb) : "v"(zoff), "s"(pfb) : "memory");
        if constexpr (g == 28) {
          c4_wait_reg(sch.wait_flag(24), fla);
          poll(fla, pfa);
          pa[0] = load_sc1_b128(xr, offa);
          pa[1] = load_sc1_b128(xr, offa + 16);
        }
        if constexpr (g == 40) {
          c4_wait_reg(sch.wait_flag(36), flb);
          poll(flb, pfb);
          pb[0] = load_sc1_b128(xr, offb);
          pb[1] = load_sc1_b128(xr, offb + 16);
        }
        // ---- request, wait, MFMAs
        if constexpr (g + RD < 48) b[(g + RD) % C4_DEPTH] = c2_ring_load(frag_ptr(g + RD), lane16);
        c2_wait(sch.wait_frag(g), b[g % C4_DEPTH]);
        if constexpr (k == 0) {
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
          if constexpr (p == 3) add4(acc, pa);          // the own tile starts from cu - 1's partials
        }
        bf16x8 af[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) af[mt] = load_frag<bf16_t>(a_lane + 16 * mt * RSA + 32 * k);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt] = mma16(af[mt], b[g % C4_DEPTH], acc[mt]);
        // ---- tail: ship the finished partner tile
        if constexpr (k == 11 && p < 3) {
          if constexpr (p == 0 && l2) dy_dma(nt);
          store_sc1_b128(xr, slabs + (cu * 3 + p) * C4_SLAB + my, pack4(acc, 0));
          store_sc1_b128(xr, slabs + (cu * 3 + p) * C4_SLAB + my + 16, pack4(acc, 1));
        }
      });
      add4(acc, pb);
      // cu + 1 shipped these tiles in its pass 2 and raised the epoch two thirds into its own pass: normally there by now.  Its
      // two loads go out FIRST, then the saved state of this layer's next step (younger: the wait for the partials leaves it in flight)
#ifdef C4_DIAG_NOFINAL
      fetch_part(with_dh{}, nsv, ngs, ncs, ndh, nt, 0, 4);      // diagnostic: the last partner's partials are not fetched (garbage results)
#else
      unsigned flc = __hip_atomic_load(pfc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      poll(flc, pfc);
      u32x4 pc[2];
      pc[0] = load_sc1_b128(xr, offc);
      pc[1] = load_sc1_b128(xr, offc + 16);
      fetch_part(with_dh{}, nsv, ngs, ncs, ndh, nt, 0, 4);
      add4(acc, pc);
#endif
      scatter(acc, layer);
    }
    __syncthreads();                                   // the layer's outputs complete in LDS
  };

  // HEAD: + dY_tau . Wh restricted to the own units: waves 6-11 own the d h2 tiles
  auto head_add = [&]() {
    if constexpr (HEAD) {
      if (wave >= 6) {
        int nn = lane & 15, qq = lane >> 4;
        asm volatile("" : "+v"(nn), "+v"(qq));
        const int lt = wave - 6;
        bf16x8 bfr = {0, 0, 0, 0, 0, 0, 0, 0};
        if (qq < 2) bfr = load_frag<T>(whs + (16 * lt + nn) * 16 + 8 * qq);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          bf16x8 af = {0, 0, 0, 0, 0, 0, 0, 0};
          if (qq < 2) af = load_frag<T>(dys + (16 * mt + nn) * 16 + 8 * qq);
          const f32x4 hv = mma16(af, bfr, f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
          for (int j = 0; j < 4; ++j) dhrec2[(16 * mt + 4 * qq + j) * HC + 16 * lt + nn] += hv[j];
        }
      }
    }
  };
  const T* dh2_src = HEAD ? nullptr : dh2;

  Saved sv2, sv1;
  fetch(sv2, g2, c2, dh2_src, a.Tn - 1, false);
  fetch(sv1, g1, c1, nullptr, a.Tn - 1, false);
  if constexpr (HEAD) {
    dy_dma(a.Tn - 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    head_add();
    __syncthreads();
  }

#pragma unroll 1
  for (int t = a.Tn - 1; t >= 0; --t) {
    const int ep = a.Tn - t;
    // ---------------- LSTM layer 2 (exchange layer index 1)
    cell_bwd(sv2, dhrec2, !HEAD, dc2, t, dg2T);
    __syncthreads();                                   // own dgates complete in LDS; everyone done reading dhrec2
    layer_gemm(std::integral_constant<int, 1>{}, a.wb2, ep, sv2, g2, c2, dh2_src, t - 1);
    if (t > 0) head_add();
    // ---------------- LSTM layer 1 (exchange layer index 0)
    cell_bwd(sv1, dh1buf, false, dc1, t, dg1T);
    __syncthreads();
    layer_gemm(std::integral_constant<int, 0>{}, a.wb1, ep, sv1, g1, c1, nullptr, t - 1);
    int ti = tid;
    asm volatile("" : "+v"(ti));
    for (int i = ti; i < MC * 16; i += NT) {
      const int r = i / 16, c = i % 16;
      if (row0 + r < N) store1_nt<T>(dx + ((size_t)t * N + row0 + r) * KX + cu * 16 + c, dxbuf[i]);
    }
  }
}

// packed weights of the four-CU backward: element (cu, wave, kk, owner, lane l, j)
//   local k = 32*kk + 8*(l>>4) + j = ul*4 + g'  (unit u = cu*96 + ul, g' in i,g,f,o -> torch block {0,2,1,3}[g'])
//   column of (owner O, wave w, n = l&15):
//   LSTM layer 2: w < 6 -> W_ih[row][96 O + 16 w + n] (input feature h1);  w >= 6 -> W_hh[row][96 O + 16 (w-6) + n]
//   LSTM layer 1: w < 6 -> W_hh[row][96 O + 16 w + n];  w == 6 -> W_ih[row][16 O + n] (0 for 16 O + n >= I);  w > 6 -> 0
__global__ void lstm_coop_pack_bwd4_kernel(const float* __restrict__ w_ih, const float* __restrict__ w_hh, bf16_t* __restrict__ out,
                                           int I, int layer) {
  const size_t total = (size_t)C4_G * CB_NW * C4_NKK * C4_G * 512;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int j = (int)(e % 8), l = (int)((e / 8) % 64);
    size_t f = e / 512;
    const int owner = (int)(f % C4_G); f /= C4_G;
    const int kk = (int)(f % C4_NKK); f /= C4_NKK;
    const int wave = (int)(f % CB_NW);
    const int cu = (int)(f / CB_NW);
    const int k = 32 * kk + 8 * (l >> 4) + j, ul = k / 4, gp = k % 4;
    const int tg = gp == 0 ? 0 : (gp == 1 ? 2 : (gp == 2 ? 1 : 3));
    const int row = tg * CB_H + cu * C4_HC + ul;
    const int n = l & 15;
    float v = 0.f;
    if (layer == 2) {
      const int c = owner * C4_HC + 16 * (wave % 6) + n;
      v = wave < 6 ? w_ih[(size_t)row * CB_H + c] : w_hh[(size_t)row * CB_H + c];
    } else if (wave < 6) {
      v = w_hh[(size_t)row * CB_H + owner * C4_HC + 16 * wave + n];
    } else if (wave == 6) {
      const int xc = owner * 16 + n;
      v = xc < I ? w_ih[(size_t)row * I + xc] : 0.f;
    }
    out[e] = f2bf(v);
  }
}

// packed weights of the K-split backward: element (cu, wave, kk, slot, lane l, j)
//   local k = 32*kk + 8*(l>>4) + j = ul*4 + g'  (unit u = cu*192 + ul, g' in i,g,f,o -> torch block {0,2,1,3}[g'])
//   column tile tau = wave + 12*slot, column c = 16*tau + (l&15)
//   LSTM layer 2: c < 384 -> W_ih[row][c] (input feature h1 unit c); else W_hh[row][c-384]
//   LSTM layer 1: c < 384 -> W_hh[row][c];  384 <= c < 448 -> W_ih[row][c-384] (0 for c-384 >= I);  tau >= 28 -> 0
__global__ void lstm_coop_pack_bwd2_kernel(const float* __restrict__ w_ih, const float* __restrict__ w_hh, bf16_t* __restrict__ out,
                                           int I, int layer) {
  const size_t total = (size_t)CB_G * CB_NW * C2_NKK * C2_SLOTS * 512;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int j = e & 7, l = (e >> 3) & 63;
    size_t f = e >> 9;
    const int slot = f % C2_SLOTS; f /= C2_SLOTS;
    const int kk = f % C2_NKK; f /= C2_NKK;
    const int wave = f % CB_NW;
    const int cu = (int)(f / CB_NW);
    const int k = 32 * kk + 8 * (l >> 4) + j;
    const int ul = k >> 2, gp = k & 3;
    const int tg = gp == 0 ? 0 : (gp == 1 ? 2 : (gp == 2 ? 1 : 3));
    const int row = tg * CB_H + cu * CB_HC + ul;
    const int c = 16 * (wave + CB_NW * slot) + (l & 15);
    float v = 0.f;
    if (layer == 2) {
      v = c < CB_H ? w_ih[(size_t)row * CB_H + c] : w_hh[(size_t)row * CB_H + (c - CB_H)];
    } else if (c < CB_H) {
      v = w_hh[(size_t)row * CB_H + c];
    } else if (c < CB_H + CB_KX) {
      const int xc = c - CB_H;
      v = xc < I ? w_ih[(size_t)row * I + xc] : 0.f;
    }
    out[e] = f2bf(v);
  }
}

// All workgroups of a cooperative launch must be resident at once (partners spin on each other's flags).  The launchers
// size the grid to at most one workgroup per CU and request more than half a CU's LDS; this asks the runtime whether the
// kernel as built (registers, LDS, waves) really fits one workgroup per CU on THIS device and that the grid does not exceed
// the CU count -- checked at launch, not assumed.  What it cannot see is work of OTHER streams occupying CUs: the engine
// never has any in flight when it launches one of these kernels (FSNEngine._unjoined, pre_lstm_hook), and a partner that
// is late only costs time: every spin is bounded and counted (sticky time-out counter).
static int coop_fits(const void* kernel, int threads, size_t smem, int grid, int& per_cu_cache) {
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return NPPC_ELAUNCH;
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return NPPC_ELAUNCH;
  }
  if (per_cu_cache < 0 &&                            // per instantiation: the occupancy query is not free
      hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_cache, kernel, threads, smem) != hipSuccess)
    return NPPC_ELAUNCH;
  if (per_cu_cache < 1 || grid > n_cu) return NPPC_EUNSUPPORTED;
  return NPPC_OK;
}

template <typename T, int G, int MT, bool TRAIN, bool HEAD = false>
static int launch_coop(CoopArgs a, size_t xch_bytes, hipStream_t s) {
  constexpr int KX = 64, H = 384, MC = 16 * MT;
  constexpr int RS = 2 * KX + 2 * H + (sizeof(T) == 2 ? COOP_APAD : 16 / (int)sizeof(T));
  size_t smem = (size_t)MC * RS * sizeof(T) + (TRAIN ? (size_t)2 * MC * (H / G) * sizeof(T) : 0) +
                (MT > 3 ? (size_t)8 * (H / G) * sizeof(float) : 0);
  if (smem < 84 * 1024) smem = 84 * 1024;            // > half a CU's LDS: at most one workgroup per CU
  if (smem > 160 * 1024) return NPPC_EUNSUPPORTED;
  const long need = (long)a.clusters * 2 * 2 * G * MC * (H / G) * sizeof(T);
  if ((long)xch_bytes < need) return NPPC_EBADARG;
  auto k = lstm2_coop_fwd_kernel<T, G, MT, KX, TRAIN, HEAD>;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) !=
      hipSuccess)
    return NPPC_ELAUNCH;
  if (hipMemsetAsync(a.flags, 0, (size_t)a.clusters * 2 * G * sizeof(unsigned), s) != hipSuccess) return NPPC_ELAUNCH;
  // G = 2: grid rounded up to a multiple of 8 for the XCD-aware placement (coop_ids); spare workgroups exit at once
  const int grid = G == 2 ? round_up(a.clusters * G, 8) : a.clusters * G;
  static int per_cu = -1;
  const int fits = coop_fits(reinterpret_cast<const void*>(k), (H / G / 16) * 64, smem, a.clusters * G, per_cu);
  if (fits != NPPC_OK) return fits;
  hipLaunchKernelGGL(k, dim3(grid), dim3((H / G / 16) * 64), smem, s, a);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

}  // namespace

extern "C" {

// Rows per cluster and cluster count the cooperative kernel would use for N sequences; 0 clusters = not applicable.
int nppc_lstm2_coop_plan(int prec, int train, long N, int H, int n_cu, int* G, int* mtile, int* clusters) {
  *G = 0; *mtile = 0; *clusters = 0;
  if (prec != NPPC_PREC_BF16 || H != 384 || N <= 0) return NPPC_OK;
  // candidates (G, mtile): the weight stream per CU shrinks with G, the rows per cluster grow with mtile.  Take the one
  // that occupies the most CUs (all workgroups of a launch must be resident: clusters * G <= n_cu); ties go to the
  // earlier entry (measured at N = 4096: G = 2 beats G = 4, whose 6 waves leave two SIMDs idle).
  const int cand[6][2] = {{2, 2}, {4, 2}, {4, 4}, {2, 5}, {8, 2}, {8, 5}};
  long best = 0;
  // eight-CU clusters are built and tested but NOT planned by default: measured at BASELINE C5 they lose to four-CU clusters
  // (restorer, 2056 sequences: 87.5 vs 48.2 ms; direction net, 1024: 41.8 vs 33.1 ms) -- the seven-partner hand-off costs
  // more than the halved weight stream saves, as the four-CU clusters already do against CU pairs at C2.  NPPC_LSTM_G8=1 plans them.
  static const bool g8 = [] { const char* e = getenv("NPPC_LSTM_G8"); return e && e[0] == '1'; }();
  for (int oi = 0; oi < 6; ++oi) {
    const int g = cand[oi][0], mt = cand[oi][1];
    if (train && mt == 5) continue;
    if (g == 8 && !g8) continue;
    const long cl = (N + 16 * mt - 1) / (16 * mt);
    if (cl * g <= n_cu && cl * g > best) { best = cl * g; *G = g; *mtile = mt; *clusters = (int)cl; }
  }
  return NPPC_OK;
}

// flags: (clusters*2*G + 4) u32; xch: clusters*2*2*G*MC*(H/G) elements.  Same tensor contract as nppc_lstm2_fwd.
static int fwd_coop_impl(int prec, int train, int G, int mtile, const void* x, const void* wp1, const void* wp2,
                         const float* bias1, const float* bias2, void* h2, void* h1, void* g1, void* g2, void* c1, void* c2,
                         void* xch, long xch_bytes, unsigned* flags, long N, int Tn, int I, int H, const void* whp,
                         float* hpart, int O, void* stream, int x_ld = 0) {
  if (prec != NPPC_PREC_BF16 || H != 384 || (G != 2 && G != 4 && G != 8)) return NPPC_EUNSUPPORTED;
  if (!x || !wp1 || !wp2 || (!h2 && (!whp || train)) || !xch || !flags || N <= 0 || Tn <= 0 || I > 64) return NPPC_EBADARG;
  if (train && (!h1 || !g1 || !g2 || !c1 || !c2)) return NPPC_EBADARG;
  if (whp && (G != 2 || !hpart || O < 1 || O > 16)) return NPPC_EBADARG;
  const int MC = 16 * mtile;
  if (x_ld == 0) x_ld = 64;
  if (x_ld % 8 || x_ld > 64 || x_ld <= I) return NPPC_EBADARG;
  CoopArgs a{x, wp1, wp2, bias1, bias2, h2, h1, g1, g2, c1, c2, xch, flags, N, Tn, (int)((N + MC - 1) / MC), whp, hpart, O, x_ld};
  hipStream_t s = (hipStream_t)stream;
  if (G == 8) {
    // eight CUs per cluster (48 hidden units = 3 waves each, 1/8 of the weight stream per CU): for the few long sequences
    // of the long-clip configuration (BASELINE C5: 1024 / 2056 sequences would leave half the chip idle with G <= 4)
    if (mtile == 2)
      return train ? launch_coop<bf16_t, 8, 2, true>(a, (size_t)xch_bytes, s) : launch_coop<bf16_t, 8, 2, false>(a, (size_t)xch_bytes, s);
    if (mtile == 5 && !train) return launch_coop<bf16_t, 8, 5, false>(a, (size_t)xch_bytes, s);
    return NPPC_EUNSUPPORTED;
  }
  if (G == 4) {
    if (mtile == 4)
      return train ? launch_coop<bf16_t, 4, 4, true>(a, (size_t)xch_bytes, s) : launch_coop<bf16_t, 4, 4, false>(a, (size_t)xch_bytes, s);
    if (mtile == 2)
      return train ? launch_coop<bf16_t, 4, 2, true>(a, (size_t)xch_bytes, s) : launch_coop<bf16_t, 4, 2, false>(a, (size_t)xch_bytes, s);
    return NPPC_EUNSUPPORTED;
  }
  if (train) {
    if (mtile == 2) return whp ? launch_coop<bf16_t, 2, 2, true, true>(a, (size_t)xch_bytes, s)
                               : launch_coop<bf16_t, 2, 2, true>(a, (size_t)xch_bytes, s);
    return NPPC_EUNSUPPORTED;
  }
  if (whp) {
    if (mtile == 2) return launch_coop<bf16_t, 2, 2, false, true>(a, (size_t)xch_bytes, s);
    if (mtile == 5) return launch_coop<bf16_t, 2, 5, false, true>(a, (size_t)xch_bytes, s);
    return NPPC_EUNSUPPORTED;
  }
  if (mtile == 2) return launch_coop<bf16_t, 2, 2, false>(a, (size_t)xch_bytes, s);
  if (mtile == 5) return launch_coop<bf16_t, 2, 5, false>(a, (size_t)xch_bytes, s);
  return NPPC_EUNSUPPORTED;
}

int nppc_lstm2_fwd_coop(int prec, int train, int G, int mtile, const void* x, const void* wp1, const void* wp2,
                        const float* bias1, const float* bias2, void* h2, void* h1, void* g1, void* g2, void* c1, void* c2,
                        void* xch, long xch_bytes, unsigned* flags, long N, int Tn, int I, int H, void* stream) {
  return fwd_coop_impl(prec, train, G, mtile, x, wp1, wp2, bias1, bias2, h2, h1, g1, g2, c1, c2, xch, xch_bytes, flags, N, Tn, I,
                       H, nullptr, nullptr, 0, stream);
}

int nppc_lstm2_fwd_coop_head(int prec, int train, int mtile, const void* x, const void* wp1, const void* wp2,
                             const float* bias1, const float* bias2, void* h2, void* h1, void* g1, void* g2, void* c1, void* c2,
                             void* xch, long xch_bytes, unsigned* flags, long N, int Tn, int I, int H, const void* whp,
                             float* hpart, int O, int x_ld, void* stream) {
  if (!whp) return NPPC_EBADARG;
  return fwd_coop_impl(prec, train, 2, mtile, x, wp1, wp2, bias1, bias2, h2, h1, g1, g2, c1, c2, xch, xch_bytes, flags, N, Tn, I,
                       H, whp, hpart, O, stream, x_ld);
}

// cooperative backward (bf16, H = 384, I <= 64): packed weights of nppc_lstm2_coop_bwd_pack; xch holds
// clusters*2*2*2*32*768 bf16, flags clusters*4 + 4 u32 (zeroed by the launcher)
int nppc_lstm2_coop_bwd_packed_elems(long* n) {
  *n = (long)CB_G * CB_NW * CB_NK * 2 * 512;
  return NPPC_OK;
}

int nppc_lstm2_coop_bwd_pack(const float* w_ih0, const float* w_hh0, const float* w_ih1, const float* w_hh1, int I, void* wb1,
                             void* wb2, void* stream) {
  if (!w_ih0 || !w_hh0 || !w_ih1 || !w_hh1 || !wb1 || !wb2 || I > CB_KX) return NPPC_EBADARG;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(lstm_coop_pack_bwd_kernel, dim3(512), dim3(256), 0, s, w_ih0, w_hh0, (bf16_t*)wb1, I, 1);
  hipLaunchKernelGGL(lstm_coop_pack_bwd_kernel, dim3(512), dim3(256), 0, s, w_ih1, w_hh1, (bf16_t*)wb2, CB_H, 2);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

int nppc_lstm2_bwd_coop(const void* g1, const void* g2, const void* c1, const void* c2, const void* dh2, const void* wb1,
                        const void* wb2, void* dx, void* dg1, void* dg2, void* xch, long xch_bytes, unsigned* flags, long N,
                        int Tn, int n_cu, void* stream) {
  if (!g1 || !g2 || !c1 || !c2 || !dh2 || !wb1 || !wb2 || !dx || !dg1 || !dg2 || !xch || !flags || N <= 0 || Tn <= 0)
    return NPPC_EBADARG;
  const int clusters = (int)((N + CB_MC - 1) / CB_MC);
  if (clusters * CB_G > n_cu) return NPPC_EUNSUPPORTED;      // every workgroup of a cluster must be resident
  if (xch_bytes < (long)clusters * 2 * 2 * 2 * CB_MC * CB_KC * 2) return NPPC_EBADARG;
  CoopBwdArgs a{g1, g2, c1, c2, dh2, wb1, wb2, dx, dg1, dg2, xch, flags, N, Tn, clusters};
  hipStream_t s = (hipStream_t)stream;
  constexpr size_t smem = (size_t)CB_MC * (CB_K4 + COOP_APAD) * 2 + (size_t)2 * CB_MC * CB_HC * 4;
  static_assert(smem <= 160 * 1024 && smem > 80 * 1024, "one workgroup per CU");
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(lstm2_coop_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)smem) != hipSuccess)
    return NPPC_ELAUNCH;
  if (hipMemsetAsync(flags, 0, (size_t)clusters * 2 * CB_G * sizeof(unsigned), s) != hipSuccess) return NPPC_ELAUNCH;
  hipLaunchKernelGGL(lstm2_coop_bwd_kernel, dim3(round_up(clusters * CB_G, 8)), dim3(CB_NT), smem, s, a);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

// K-split cooperative backward (lstm2_coop_bwd2_kernel): same tensor contract as nppc_lstm2_bwd_coop; xch holds
// clusters*2*2*2*32*384 bf16 partial sums, flags clusters*4 + 4 u32 (zeroed by the launcher)
int nppc_lstm2_coop_bwd2_packed_elems(long* n) {
  *n = (long)CB_G * CB_NW * C2_NKK * C2_SLOTS * 512;
  return NPPC_OK;
}

int nppc_lstm2_coop_bwd2_pack(const float* w_ih0, const float* w_hh0, const float* w_ih1, const float* w_hh1, int I, void* wb1,
                              void* wb2, void* stream) {
  if (!w_ih0 || !w_hh0 || !w_ih1 || !w_hh1 || !wb1 || !wb2 || I > CB_KX) return NPPC_EBADARG;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(lstm_coop_pack_bwd2_kernel, dim3(512), dim3(256), 0, s, w_ih0, w_hh0, (bf16_t*)wb1, I, 1);
  hipLaunchKernelGGL(lstm_coop_pack_bwd2_kernel, dim3(512), dim3(256), 0, s, w_ih1, w_hh1, (bf16_t*)wb2, CB_H, 2);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

static int bwd_coop2_impl(const void* g1, const void* g2, const void* c1, const void* c2, const void* dh2, const void* dyt,
                          const void* whT, const void* wb1, const void* wb2, void* dx, void* dg1, void* dg2, void* xch,
                          long xch_bytes, unsigned* flags, long N, int Tn, int n_cu, void* stream) {
  if (!g1 || !g2 || !c1 || !c2 || (!dh2 && !(dyt && whT)) || !wb1 || !wb2 || !dx || !dg1 || !dg2 || !xch || !flags || N <= 0 ||
      Tn <= 0)
    return NPPC_EBADARG;
  const int clusters = (int)((N + CB_MC - 1) / CB_MC);
  if (clusters * CB_G > n_cu) return NPPC_EUNSUPPORTED;      // every workgroup of a cluster must be resident
  if (xch_bytes < (long)clusters * 2 * 2 * 2 * CB_MC * C2_XW * 2) return NPPC_EBADARG;
  CoopBwd2Args a{g1, g2, c1, c2, dh2, wb1, wb2, dx, dg1, dg2, xch, flags, N, Tn, clusters, dyt, whT};
  hipStream_t s = (hipStream_t)stream;
  constexpr size_t smem = (size_t)CB_MC * (CB_KC + COOP_APAD) * 2 + (size_t)4 * CB_MC * CB_HC * 4 + (size_t)CB_MC * 32 * 4 +
                          (size_t)(CB_HC + CB_MC) * 16 * 2;
  static_assert(smem <= 160 * 1024 && smem > 80 * 1024, "one workgroup per CU");
  const void* k = dyt ? reinterpret_cast<const void*>(lstm2_coop_bwd2_kernel<true>)
                      : reinterpret_cast<const void*>(lstm2_coop_bwd2_kernel<false>);
  if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return NPPC_ELAUNCH;
  static int per_cu[2] = {-1, -1};
  const int fits = coop_fits(k, CB_NT, smem, clusters * CB_G, per_cu[dyt ? 1 : 0]);
  if (fits != NPPC_OK) return fits;
  if (hipMemsetAsync(flags, 0, (size_t)clusters * C2_FPC * sizeof(unsigned), s) != hipSuccess) return NPPC_ELAUNCH;
  if (dyt)
    hipLaunchKernelGGL(lstm2_coop_bwd2_kernel<true>, dim3(round_up(clusters * CB_G, 8)), dim3(CB_NT), smem, s, a);
  else
    hipLaunchKernelGGL(lstm2_coop_bwd2_kernel<false>, dim3(round_up(clusters * CB_G, 8)), dim3(CB_NT), smem, s, a);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

// four-CU K-split backward (lstm2_coop_bwd4_kernel): same tensor contract as nppc_lstm2_bwd_coop2 / _head (dh2 == null
// selects the fused head: dyt + whT); workspace sizes from nppc_lstm2_coop_bwd4_sizes
int nppc_lstm2_coop_bwd4_sizes(long N, long* packed_elems, long* xch_bytes, long* flag_words) {
  if (N <= 0 || !packed_elems || !xch_bytes || !flag_words) return NPPC_EBADARG;
  const long clusters = (N + C4_MC - 1) / C4_MC;
  *packed_elems = (long)C4_G * CB_NW * C4_NKK * C4_G * 512;
  *xch_bytes = clusters * C4_XCH;
  *flag_words = clusters * C4_FPC + 4;
  return NPPC_OK;
}

int nppc_lstm2_coop_bwd4_pack(const float* w_ih0, const float* w_hh0, const float* w_ih1, const float* w_hh1, int I, void* wb1,
                              void* wb2, void* stream) {
  if (!w_ih0 || !w_hh0 || !w_ih1 || !w_hh1 || !wb1 || !wb2 || I > CB_KX) return NPPC_EBADARG;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(lstm_coop_pack_bwd4_kernel, dim3(512), dim3(256), 0, s, w_ih0, w_hh0, (bf16_t*)wb1, I, 1);
  hipLaunchKernelGGL(lstm_coop_pack_bwd4_kernel, dim3(512), dim3(256), 0, s, w_ih1, w_hh1, (bf16_t*)wb2, CB_H, 2);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

int nppc_lstm2_bwd_coop4(const void* g1, const void* g2, const void* c1, const void* c2, const void* dh2, const void* dyt,
                         const void* whT, const void* wb1, const void* wb2, void* dx, void* dg1, void* dg2, void* xch,
                         long xch_bytes, unsigned* flags, long N, int Tn, int n_cu, void* stream) {
  if (!g1 || !g2 || !c1 || !c2 || (!dh2 && !(dyt && whT)) || !wb1 || !wb2 || !dx || !dg1 || !dg2 || !xch || !flags || N <= 0 ||
      Tn <= 0)
    return NPPC_EBADARG;
  const int clusters = (int)((N + C4_MC - 1) / C4_MC);
  if (clusters * C4_G > n_cu) return NPPC_EUNSUPPORTED;      // every workgroup of a cluster must be resident
  if (xch_bytes < (long)clusters * C4_XCH) return NPPC_EBADARG;
  CoopBwd2Args a{g1, g2, c1, c2, dh2, wb1, wb2, dx, dg1, dg2, xch, flags, N, Tn, clusters, dh2 ? nullptr : dyt, dh2 ? nullptr : whT};
  hipStream_t s = (hipStream_t)stream;
  constexpr size_t smem = (size_t)C4_MC * (C4_KC + COOP_APAD) * 2 + (size_t)4 * C4_MC * C4_HC * 4 + (size_t)C4_MC * 16 * 4 +
                          (size_t)(C4_HC + C4_MC) * 16 * 2;
  static_assert(smem <= 160 * 1024 && smem > 80 * 1024, "one workgroup per CU");
  const bool head = !dh2;
  const void* k = head ? reinterpret_cast<const void*>(lstm2_coop_bwd4_kernel<true>)
                       : reinterpret_cast<const void*>(lstm2_coop_bwd4_kernel<false>);
  if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) return NPPC_ELAUNCH;
  static int per_cu[2] = {-1, -1};
  const int fits = coop_fits(k, CB_NT, smem, clusters * C4_G, per_cu[head ? 1 : 0]);
  if (fits != NPPC_OK) return fits;
  if (hipMemsetAsync(flags, 0, (size_t)clusters * C4_FPC * sizeof(unsigned), s) != hipSuccess) return NPPC_ELAUNCH;
  if (head)
    hipLaunchKernelGGL(lstm2_coop_bwd4_kernel<true>, dim3(round_up(clusters * C4_G, 8)), dim3(CB_NT), smem, s, a);
  else
    hipLaunchKernelGGL(lstm2_coop_bwd4_kernel<false>, dim3(round_up(clusters * C4_G, 8)), dim3(CB_NT), smem, s, a);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

int nppc_lstm2_bwd_coop2(const void* g1, const void* g2, const void* c1, const void* c2, const void* dh2, const void* wb1,
                         const void* wb2, void* dx, void* dg1, void* dg2, void* xch, long xch_bytes, unsigned* flags, long N,
                         int Tn, int n_cu, void* stream) {
  if (!dh2) return NPPC_EBADARG;
  return bwd_coop2_impl(g1, g2, c1, c2, dh2, nullptr, nullptr, wb1, wb2, dx, dg1, dg2, xch, xch_bytes, flags, N, Tn, n_cu,
                        stream);
}

int nppc_lstm2_bwd_coop2_head(const void* g1, const void* g2, const void* c1, const void* c2, const void* dyt, const void* whT,
                              const void* wb1, const void* wb2, void* dx, void* dg1, void* dg2, void* xch, long xch_bytes,
                              unsigned* flags, long N, int Tn, int n_cu, void* stream) {
  if (!dyt || !whT) return NPPC_EBADARG;
  return bwd_coop2_impl(g1, g2, c1, c2, nullptr, dyt, whT, wb1, wb2, dx, dg1, dg2, xch, xch_bytes, flags, N, Tn, n_cu, stream);
}

}  // extern "C"
