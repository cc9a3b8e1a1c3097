// Shared device helpers for the NPPC-audio gfx950 kernels (CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned short bf16_t;  // storage type of a bfloat16 element
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) float f32x8;

#define NPPC_OK 0
#define NPPC_EBADARG 1
#define NPPC_ELAUNCH 2
#define NPPC_EUNSUPPORTED 3

#define NPPC_PREC_BF16 0
#define NPPC_PREC_F32 1

#define NPPC_CHECK_LAUNCH()                                  \
  do {                                                       \
    hipError_t e__ = hipGetLastError();                      \
    if (e__ != hipSuccess) return NPPC_ELAUNCH;              \
  } while (0)

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32: round-to-nearest-even, NaN stays NaN
  return *reinterpret_cast<bf16_t*>(&b);
}

template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return bf2f(v); }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return f2bf(v); }

// MFMA operand fragment of 8 consecutive k per lane: lane (r = lane&15, q = lane>>4) holds
// M[r][k0 + 8q + j], j = 0..7 (A: r = row, B: r = column).  One 16x16x32 bf16 MFMA, or eight
// 16x16x4 f32 MFMAs (MFMA j sums k in {8q + j}: a consistent permutation of the same 32 k).
template <typename T> struct Frag;
template <> struct Frag<bf16_t> { typedef bf16x8 type; };
template <> struct Frag<float> { typedef f32x8 type; };

__device__ __forceinline__ f32x4 mma16(const bf16x8& a, const bf16x8& b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mma16(const f32x8& a, const f32x8& b, f32x4 c) {
#pragma unroll
  for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], c, 0, 0, 0);
  return c;
}

// B-fragment stream through a buffer descriptor: rsrc + scalar offsets live in SGPRs, the only
// VGPR is the 32-bit lane offset -> no per-fragment 64-bit address registers (guide T8/T20).
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
template <typename T> struct BFrag;
template <> struct BFrag<bf16_t> {
  static __device__ __forceinline__ bf16x8 load(__amdgpu_buffer_rsrc_t r, int lane, int soff) {
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16, soff, 0);
    return __builtin_bit_cast(bf16x8, v);
  }
};
template <> struct BFrag<float> {
  static __device__ __forceinline__ f32x8 load(__amdgpu_buffer_rsrc_t r, int lane, int soff) {
    u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(r, lane * 32, soff, 0);
    u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(r, lane * 32 + 16, soff, 0);
    f32x8 o;
    o[0] = __uint_as_float(a[0]); o[1] = __uint_as_float(a[1]); o[2] = __uint_as_float(a[2]); o[3] = __uint_as_float(a[3]);
    o[4] = __uint_as_float(b[0]); o[5] = __uint_as_float(b[1]); o[6] = __uint_as_float(b[2]); o[7] = __uint_as_float(b[3]);
    return o;
  }
};

template <typename T> __device__ __forceinline__ typename Frag<T>::type load_frag(const T* p) {
  return *reinterpret_cast<const typename Frag<T>::type*>(p);
}

// 8 consecutive channels as one (bf16) or two (f32) 16-byte accesses; callers guarantee 16-byte alignment
// (row widths and channel offsets are multiples of 8)
template <typename T> __device__ __forceinline__ void load8(const T* p, float (&v)[8]);
template <> __device__ __forceinline__ void load8<bf16_t>(const bf16_t* p, float (&v)[8]) {
  const uint4 u = *reinterpret_cast<const uint4*>(p);
  const unsigned w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[2 * i] = __uint_as_float(w[i] << 16);
    v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
  }
}
template <> __device__ __forceinline__ void load8<float>(const float* p, float (&v)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
// eight consecutive fp32 parameters (16-byte aligned)
__device__ __forceinline__ void loadf8(const float* p, float (&v)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
template <typename T> __device__ __forceinline__ void store8(T* p, const float (&v)[8]);
template <> __device__ __forceinline__ void store8<bf16_t>(bf16_t* p, const float (&v)[8]) {
  uint4 u;
  u.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
  u.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
  u.z = (unsigned)f2bf(v[4]) | ((unsigned)f2bf(v[5]) << 16);
  u.w = (unsigned)f2bf(v[6]) | ((unsigned)f2bf(v[7]) << 16);
  *reinterpret_cast<uint4*>(p) = u;
}
template <> __device__ __forceinline__ void store8<float>(float* p, const float (&v)[8]) {
  *reinterpret_cast<float4*>(p) = float4{v[0], v[1], v[2], v[3]};
  *reinterpret_cast<float4*>(p + 4) = float4{v[4], v[5], v[6], v[7]};
}

// Non-temporal 16-byte accesses for data that is touched ONCE per launch (saved LSTM state, gate gradients): keeps it
// from evicting the re-read weight fragments out of the 4 MB per-XCD L2 (MI355X_MICROARCH.md, nt-weights row: nt on
// once-read streams, never on slices every CU re-reads).
__device__ __forceinline__ u32x4 ld_nt16(const void* p) { return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p)); }
__device__ __forceinline__ void st_nt16(void* p, u32x4 v) { __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p)); }

// v_exp_f32 / v_rcp_f32 based (1 ulp each): absolute error ~1e-7, saturate cleanly for |x| large
__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_f(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }
static inline int round_up(int a, int b) { return (a + b - 1) / b * b; }
