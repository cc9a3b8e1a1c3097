// Signal front end of the NPPC train step: batched STFT, cIRM build/compress (+drop-band),
// cIRM decompress + conj-mask application, drop-band gather.  All HBM-bound, fp32.
// Reference call sites (under /root/reference):
//   STFT        utils.py:107-147 (torch.stft centre/reflect, periodic hann), nppc_audio/trainer.py:349-355
//   cIRM build  FullSubNet_plus/speech_enhance/audio_zen/acoustics/mask.py:24-54, trainer.py:357-362
//   decompress  mask.py:57-60;  mask application utils.py:241-249 -> :75-79 (real/imag swapped: conj(mask)*noisy)
//   drop_band   audio_zen/acoustics/feature.py:254-285
#include "common.h"
#include "nppc_hip.h"

namespace {

// ---------------------------------------------------------------- STFT
// One workgroup = FR consecutive frames of one clip.  Radix-2 DIT FFT in LDS (complex N points,
// imaginary input 0), twiddles/window generated in fp64 -> fp32 once per workgroup.
// Output layout [B][F][T] (F = N/2+1), written with the FR frames of a bin contiguous.
constexpr int STFT_FR = 8;

template <int LOGN>
__global__ __launch_bounds__(256) void stft_kernel(const float* __restrict__ wave, float* __restrict__ out_re,
                                                   float* __restrict__ out_im, float* __restrict__ out_mag, int L,
                                                   int hop, int T) {
  constexpr int N = 1 << LOGN;
  constexpr int F = N / 2 + 1;
  __shared__ float2 buf[STFT_FR][N + 1];
  __shared__ float2 tw[N / 2];
  __shared__ float win[N];
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * STFT_FR;
  for (int i = tid; i < N / 2; i += 256) {
    double s, c;
    sincospi(-2.0 * i / N, &s, &c);
    tw[i] = make_float2((float)c, (float)s);
  }
  for (int i = tid; i < N; i += 256) win[i] = (float)(0.5 - 0.5 * cospi(2.0 * i / N));
  __syncthreads();
  const float* wv = wave + (size_t)b * L;
  // load, window, bit-reverse
  for (int e = tid; e < STFT_FR * N; e += 256) {
    const int j = e / N, n = e % N;
    const int t = t0 + j;
    float v = 0.f;
    if (t < T) {
      int sidx = t * hop + n - N / 2;
      if (sidx < 0) sidx = -sidx;
      if (sidx >= L) sidx = 2 * (L - 1) - sidx;
      v = wv[sidx] * win[n];
    }
    const int r = __brev((unsigned)n) >> (32 - LOGN);
    buf[j][r] = make_float2(v, 0.f);
  }
  __syncthreads();
#pragma unroll 1
  for (int s = 0; s < LOGN; ++s) {
    const int half = 1 << s;
    for (int e = tid; e < STFT_FR * (N / 2); e += 256) {
      const int j = e / (N / 2), k = e % (N / 2);
      const int grp = k >> s, pos = k & (half - 1);
      const int i0 = (grp << (s + 1)) + pos, i1 = i0 + half;
      const float2 w = tw[pos << (LOGN - 1 - s)];
      const float2 a = buf[j][i0], c = buf[j][i1];
      const float xr = c.x * w.x - c.y * w.y, xi = c.x * w.y + c.y * w.x;
      buf[j][i0] = make_float2(a.x + xr, a.y + xi);
      buf[j][i1] = make_float2(a.x - xr, a.y - xi);
    }
    __syncthreads();
  }
  for (int e = tid; e < F * STFT_FR; e += 256) {
    const int f = e / STFT_FR, j = e % STFT_FR;
    const int t = t0 + j;
    if (t < T) {
      const float2 v = buf[j][f];
      const size_t o = ((size_t)b * F + f) * T + t;
      out_re[o] = v.x;
      out_im[o] = v.y;
      if (out_mag) out_mag[o] = sqrtf(v.x * v.x + v.y * v.y);
    }
  }
}

// ---------------------------------------------------------------- iSTFT
// torch.istft(center=True, periodic hann, onesided input, length=L): y[p] = sum_t w[p - t*hop] x_t[p - t*hop] / sum_t w^2[p - t*hop],
// x_t = irfft(X_t), p = n + N/2.  One workgroup = ISTFT_FR*hop consecutive output samples of one clip; it inverts the
// ISTFT_FR + N/hop - 1 frames that overlap them (radix-2 in LDS on the Hermitian-extended spectrum).
constexpr int ISTFT_FR = 4;

template <int LOGN>
__global__ __launch_bounds__(256) void istft_kernel(const float* __restrict__ re, const float* __restrict__ im,
                                                    float* __restrict__ out, int T, int hop, int L) {
  constexpr int N = 1 << LOGN;
  constexpr int F = N / 2 + 1;
  constexpr int MAXFR = ISTFT_FR + 7;            // supports hop >= N/8
  __shared__ float2 buf[MAXFR][N + 1];
  __shared__ float2 tw[N / 2];
  __shared__ float win[N];
  const int tid = threadIdx.x, b = blockIdx.y;
  const int ov = N / hop;                          // frames overlapping one sample
  const int nfr = ISTFT_FR + ov - 1;
  const int p0 = blockIdx.x * ISTFT_FR * hop;      // first padded-coordinate sample of this block
  const int tfirst = p0 / hop - (ov - 1);          // first frame that can touch [p0, p0 + FR*hop)
  for (int i = tid; i < N / 2; i += 256) {
    double s, c;
    sincospi(2.0 * i / N, &s, &c);                 // inverse transform: +i
    tw[i] = make_float2((float)c, (float)s);
  }
  for (int i = tid; i < N; i += 256) win[i] = (float)(0.5 - 0.5 * cospi(2.0 * i / N));
  __syncthreads();
  for (int e = tid; e < nfr * N; e += 256) {
    const int j = e / N, k = e % N;
    const int t = tfirst + j;
    float2 v = make_float2(0.f, 0.f);
    if (t >= 0 && t < T) {
      const int kk = k < F ? k : N - k;
      const size_t o = ((size_t)b * F + kk) * T + t;
      v = make_float2(re[o], k < F ? im[o] : -im[o]);
      if (k == 0 || k == N / 2) v.y = 0.f;         // irfft ignores the imaginary part of DC / Nyquist
    }
    buf[j][__brev((unsigned)k) >> (32 - LOGN)] = v;
  }
  __syncthreads();
#pragma unroll 1
  for (int s = 0; s < LOGN; ++s) {
    const int half = 1 << s;
    for (int e = tid; e < nfr * (N / 2); e += 256) {
      const int j = e / (N / 2), k = e % (N / 2);
      const int grp = k >> s, pos = k & (half - 1);
      const int i0 = (grp << (s + 1)) + pos, i1 = i0 + half;
      const float2 w = tw[pos << (LOGN - 1 - s)];
      const float2 a = buf[j][i0], c = buf[j][i1];
      const float xr = c.x * w.x - c.y * w.y, xi = c.x * w.y + c.y * w.x;
      buf[j][i0] = make_float2(a.x + xr, a.y + xi);
      buf[j][i1] = make_float2(a.x - xr, a.y - xi);
    }
    __syncthreads();
  }
  for (int e = tid; e < ISTFT_FR * hop; e += 256) {
    const int p = p0 + e;
    const int nidx = p - N / 2;
    if (nidx < 0 || nidx >= L) continue;
    float num = 0.f, den = 0.f;
    for (int j = 0; j < nfr; ++j) {
      const int t = tfirst + j;
      const int off = p - t * hop;
      if (t >= 0 && t < T && off >= 0 && off < N) {
        const float w = win[off];
        num += w * buf[j][off].x * (1.0f / N);
        den += w * w;
      }
    }
    out[(size_t)b * L + nidx] = den > 1e-11f ? num / den : 0.f;
  }
}

// ---------------------------------------------------------------- drop-band index math
// drop_band(x[B,C,F,T], G): F trimmed to F - F%G, output batch = groups concatenated
// (group g = samples g, g+G, ...), sample of group g keeps bins g, g+G, ...
__device__ __forceinline__ void dropband_src(int bo, int fo, int B, int G, int* b_src, int* f_src) {
  if (G <= 1) { *b_src = bo; *f_src = fo; return; }
  int g = 0, start = 0;
  for (; g < G; ++g) {
    const int cnt = (B - g + G - 1) / G;
    if (bo < start + cnt) break;
    start += cnt;
  }
  *b_src = g + (bo - start) * G;
  *f_src = g + fo * G;
}

// out[bo][c][fo][t] = in[b_src][c][f_src][t]
__global__ void dropband_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int C, int F, int T, int G,
                                int Fo) {
  const size_t total = (size_t)B * C * Fo * T;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int t = e % T;
    size_t r = e / T;
    const int fo = r % Fo; r /= Fo;
    const int c = r % C;
    const int bo = (int)(r / C);
    int bs, fs;
    dropband_src(bo, fo, B, G, &bs, &fs);
    out[e] = in[(((size_t)bs * C + c) * F + fs) * T + t];
  }
}

__device__ __forceinline__ float compress_cirm(float m) {
  m = m <= -100.f ? -100.f : m;
  const float e = expf(-0.1f * m);
  return 10.f * (1.f - e) / (1.f + e);
}

// gt[bo][{0,1}][fo][t] = compress(cIRM(noisy, clean))[b_src][f_src][t]
__global__ void cirm_build_kernel(const float* __restrict__ nr, const float* __restrict__ ni, const float* __restrict__ cr,
                                  const float* __restrict__ ci, float* __restrict__ out, int B, int F, int T, int G, int Fo,
                                  float eps) {
  const size_t total = (size_t)B * Fo * T;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int t = e % T;
    size_t r = e / T;
    const int fo = r % Fo;
    const int bo = (int)(r / Fo);
    int bs, fs;
    dropband_src(bo, fo, B, G, &bs, &fs);
    const size_t i = ((size_t)bs * F + fs) * T + t;
    const float a = nr[i], b = ni[i], c = cr[i], d = ci[i];
    const float den = a * a + b * b + eps;
    const float mr = (a * c + b * d) / den;
    const float mi = (a * d - b * c) / den;
    const size_t o = (((size_t)bo * 2) * Fo + fo) * T + t;
    out[o] = compress_cirm(mr);
    out[o + (size_t)Fo * T] = compress_cirm(mi);
  }
}

__device__ __forceinline__ float decompress_cirm(float m) {
  m = fminf(fmaxf(m, -9.9f), 9.9f);
  return -10.f * logf((10.f - m) / (10.f + m));
}

// crm [B][2][F][T] compressed -> dec [B][F][T][2] (optional), enhanced mag/re/im [B][F][T]
__global__ void decompress_apply_kernel(const float* __restrict__ crm, const float* __restrict__ nr,
                                        const float* __restrict__ ni, float* __restrict__ dec, float* __restrict__ emag,
                                        float* __restrict__ ere, float* __restrict__ eim, int B, int F, int T, int conj_mask) {
  const size_t FT = (size_t)F * T, total = (size_t)B * FT;
  for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const size_t b = e / FT, r = e % FT;
    const float mr = decompress_cirm(crm[(b * 2) * FT + r]);
    const float mi = decompress_cirm(crm[(b * 2 + 1) * FT + r]);
    const float a = nr[e], c = ni[e];
    float xr, xi;
    if (conj_mask) {       // conj(mask) * noisy: the reference's swapped call in the TRAIN path (utils.py:241-249)
      xr = mr * a + mi * c;
      xi = mr * c - mi * a;
    } else {               // mask * noisy: model_outputs_to_waveforms / validator (utils.py:37-72, :252-256)
      xr = mr * a - mi * c;
      xi = mi * a + mr * c;
    }
    if (dec) { dec[2 * e] = mr; dec[2 * e + 1] = mi; }
    ere[e] = xr;
    eim[e] = xi;
    emag[e] = sqrtf(xr * xr + xi * xi);
  }
}

static int ew_grid(size_t total) {
  size_t g = (total + 255) / 256;
  return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}


// ---------------------------------------------------------------- on-device batch synthesis
// dataset/audio_dataset.py:92-152: clean normalised to target dBFS (rms + 1e-8 inside the log), noise scaled to the
// requested SNR (power ratio, + 1e-8 in the denominator), noisy = clean + noise, both rescaled when |noisy| > 0.99.
// One workgroup per clip, three passes over the clip (sums, mix + peak, rescale); fp64 sums.
__global__ __launch_bounds__(256) void mix_snr_kernel(const float* __restrict__ clean, const float* __restrict__ noise,
                                                      const float* __restrict__ snr_db, float target_dbfs,
                                                      const float* __restrict__ target_item,
                                                      float* __restrict__ noisy_out, float* __restrict__ clean_out, int L) {
  __shared__ double red[3][4];
  __shared__ float bc[3];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* c = clean + (size_t)b * L;
  const float* n = noise + (size_t)b * L;
  double sc = 0.0, sn = 0.0;
  for (int i = tid; i < L; i += 256) {
    sc += (double)c[i] * (double)c[i];
    sn += (double)n[i] * (double)n[i];
  }
  sc = wave_sum(sc);
  sn = wave_sum(sn);
  if (lane == 0) { red[0][wave] = sc; red[1][wave] = sn; }
  __syncthreads();
  if (tid == 0) {
    const double pc = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) / L;
    const double pn = (red[1][0] + red[1][1] + red[1][2] + red[1][3]) / L;
    const float rms = (float)sqrt(pc);
    const float rms_db = 20.f * log10f(rms + 1e-8f);
    const float gain = powf(10.f, ((target_item ? target_item[b] : target_dbfs) - rms_db) / 20.f);
    const float clean_power = (float)pc * gain * gain;
    const float snr_lin = powf(10.f, snr_db[b] / 10.f);
    bc[0] = gain;
    bc[1] = sqrtf(clean_power / (snr_lin * (float)pn + 1e-8f));
  }
  __syncthreads();
  const float gain = bc[0], scale = bc[1];
  float peak = 0.f;
  for (int i = tid; i < L; i += 256) {
    const float cv = c[i] * gain;
    const float nv = cv + n[i] * scale;
    clean_out[(size_t)b * L + i] = cv;
    noisy_out[(size_t)b * L + i] = nv;
    peak = fmaxf(peak, fabsf(nv));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) peak = fmaxf(peak, __shfl_xor(peak, o, 64));
  if (lane == 0) red[2][wave] = (double)peak;
  __syncthreads();
  if (tid == 0) {
    const float m = (float)fmax(fmax(red[2][0], red[2][1]), fmax(red[2][2], red[2][3]));
    bc[2] = m > 0.99f ? 0.99f / m : 1.f;
  }
  __syncthreads();
  const float f = bc[2];
  if (f != 1.f)
    for (int i = tid; i < L; i += 256) {      // each thread rescales exactly the elements it wrote above
      clean_out[(size_t)b * L + i] *= f;
      noisy_out[(size_t)b * L + i] *= f;
    }
}

// ---- inpainting batch synthesis (dataset/audio_dataset_inpainting.py) -------------------------------------------------
// :154-168 _normalize_audio (rms + 1e-8 inside the log), :170-181 the gap mask (zeros on [g0, g1)), :223-251
// time_to_spec_mask for that gap: frame t is 1 iff its clamped window [t*hop - win/2, +win) misses the gap (an empty
// window gives 0).  One workgroup per clip; fp64 sum of squares; two passes over the clip.
__global__ __launch_bounds__(256) void inpaint_prepare_kernel(const float* __restrict__ clean, const int* __restrict__ g0s,
                                                              const int* __restrict__ g1s, int do_norm, float target_dbfs,
                                                              float* __restrict__ clean_out, float* __restrict__ masked_out,
                                                              float* __restrict__ mask_frames, int L, int win, int hop,
                                                              int T) {
  __shared__ double red[4];
  __shared__ float bc;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* c = clean + (size_t)b * L;
  float gain = 1.f;
  if (do_norm) {
    double sc = 0.0;
    for (int i = tid; i < L; i += 256) sc += (double)c[i] * (double)c[i];
    sc = wave_sum(sc);
    if (lane == 0) red[wave] = sc;
    __syncthreads();
    if (tid == 0) {
      const float rms = (float)sqrt((red[0] + red[1] + red[2] + red[3]) / L);
      bc = powf(10.f, (target_dbfs - 20.f * log10f(rms + 1e-8f)) / 20.f);
    }
    __syncthreads();
    gain = bc;
  }
  int g0 = g0s[b], g1 = g1s[b];
  g0 = g0 < 0 ? 0 : g0;                                      // python slice clamping of mask[:, start:end]
  g1 = g1 > L ? L : g1;
  for (int i = tid; i < L; i += 256) {
    const float v = c[i] * gain;
    if (clean_out) clean_out[(size_t)b * L + i] = v;
    if (masked_out) masked_out[(size_t)b * L + i] = (i >= g0 && i < g1) ? 0.f : v;
  }
  for (int t = tid; t < T; t += 256) {
    int s = t * hop - win / 2, e = s + win;
    s = s < 0 ? 0 : s;
    e = e > L ? L : e;
    float m = 0.f;
    if (e > s) m = (g1 > g0 && g0 < e && g1 > s) ? 0.f : 1.f;
    mask_frames[(size_t)b * T + t] = m;
  }
}

// time_to_spec_mask for an ARBITRARY sample mask [B, L] (:223-251): one wave per frame, (min over the window) == 1
__global__ __launch_bounds__(256) void time_to_spec_mask_kernel(const float* __restrict__ mask_time, float* __restrict__ out,
                                                                int L, int win, int hop, int center, int T, int B) {
  const int lane = threadIdx.x & 63;
  const long f = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (f >= (long)B * T) return;
  const int b = (int)(f / T), t = (int)(f % T);
  int s = t * hop - (center ? win / 2 : 0), e = s + win;
  s = s < 0 ? 0 : s;
  e = e > L ? L : e;
  int ok = 1;
  for (int i = s + lane; i < e; i += 64) ok &= mask_time[(size_t)b * L + i] == 1.f;
  ok = __all(ok);
  if (lane == 0) out[f] = (e > s && ok) ? 1.f : 0.f;
}

// utils.audio_to_stft (utils.py:150-175) for ANY nfft = win_length <= 512 (the inpainting configuration is 255 / hop 128):
// centred (reflect-padded) periodic-hann STFT written as the [B, 2, F, T] real/imag pair, plus (optionally) the
// frame-masked copy stft_clean * mask (audio_dataset_inpainting.py:307-310).  Direct DFT: 32 frames per workgroup in LDS,
// exact twiddles exp(-2 pi i (f n mod N) / N) from a table built in double, fp64 accumulation.
constexpr int DFT_FR = 32;
__global__ __launch_bounds__(256) void stft_dft_kernel(const float* __restrict__ wave, const float* __restrict__ mask_frames,
                                                       float* __restrict__ spec, float* __restrict__ masked, int L, int N,
                                                       int hop, int T) {
  extern __shared__ double dsm[];
  const int F = N / 2 + 1, RS = N | 1;                       // odd row stride: lanes = frames hit distinct banks
  double* twc = dsm;
  double* tws = dsm + N;
  float* fr = reinterpret_cast<float*>(dsm + 2 * N);         // [DFT_FR][RS] windowed frames
  const int tid = threadIdx.x, b = blockIdx.y, t0 = blockIdx.x * DFT_FR;
  for (int i = tid; i < N; i += 256) {
    double sn, cs;
    sincospi(-2.0 * i / N, &sn, &cs);
    twc[i] = cs;
    tws[i] = sn;
  }
  const float* wv = wave + (size_t)b * L;
  for (int e = tid; e < DFT_FR * N; e += 256) {
    const int j = e / N, n = e % N, t = t0 + j;
    float v = 0.f;
    if (t < T) {
      int sidx = t * hop + n - N / 2;
      if (sidx < 0) sidx = -sidx;
      if (sidx >= L) sidx = 2 * (L - 1) - sidx;
      v = wv[sidx] * (float)(0.5 - 0.5 * cospi(2.0 * n / N));
    }
    fr[j * RS + n] = v;
  }
  __syncthreads();
  const int j = tid & (DFT_FR - 1), t = t0 + j;
  if (t >= T) return;
  const float mk = mask_frames ? mask_frames[(size_t)b * T + t] : 1.f;
  for (int f = tid / DFT_FR; f < F; f += 256 / DFT_FR) {
    double re = 0.0, im = 0.0;
    int idx = 0;
    for (int n = 0; n < N; ++n) {
      const double x = (double)fr[j * RS + n];
      re += x * twc[idx];
      im += x * tws[idx];
      idx += f;
      if (idx >= N) idx -= N;
    }
    const size_t o = (((size_t)b * 2) * F + f) * T + t, oi = o + (size_t)F * T;
    spec[o] = (float)re;
    spec[oi] = (float)im;
    if (masked) {
      masked[o] = (float)re * mk;
      masked[oi] = (float)im * mk;
    }
  }
}

}  // namespace

extern "C" {

int nppc_stft(const float* wave, float* re, float* im, float* mag, int B, int L, int nfft, int hop, void* stream) {
  if (!wave || !re || !im || B <= 0 || L <= nfft / 2 || hop <= 0) return NPPC_EBADARG;
  const int T = 1 + L / hop;
  dim3 grid(ceil_div(T, STFT_FR), B);
  hipStream_t s = (hipStream_t)stream;
  switch (nfft) {
    case 64: hipLaunchKernelGGL(stft_kernel<6>, grid, dim3(256), 0, s, wave, re, im, mag, L, hop, T); break;
    case 128: hipLaunchKernelGGL(stft_kernel<7>, grid, dim3(256), 0, s, wave, re, im, mag, L, hop, T); break;
    case 256: hipLaunchKernelGGL(stft_kernel<8>, grid, dim3(256), 0, s, wave, re, im, mag, L, hop, T); break;
    case 512: hipLaunchKernelGGL(stft_kernel<9>, grid, dim3(256), 0, s, wave, re, im, mag, L, hop, T); break;
    default: return NPPC_EUNSUPPORTED;
  }
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

int nppc_dropband(const float* in, float* out, int B, int C, int F, int T, int G, void* stream) {
  if (!in || !out || B <= 0 || G < 1) return NPPC_EBADARG;
  const int Fo = G <= 1 ? F : (F - F % G) / G;
  hipLaunchKernelGGL(dropband_kernel, dim3(ew_grid((size_t)B * C * Fo * T)), dim3(256), 0, (hipStream_t)stream, in, out,
                     B, C, F, T, G, Fo);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

int nppc_cirm_build_compress(const float* nr, const float* ni, const float* cr, const float* ci, float* out, int B, int F,
                             int T, int G, float eps, void* stream) {
  if (!nr || !ni || !cr || !ci || !out || B <= 0 || G < 1) return NPPC_EBADARG;
  const int Fo = G <= 1 ? F : (F - F % G) / G;
  hipLaunchKernelGGL(cirm_build_kernel, dim3(ew_grid((size_t)B * Fo * T)), dim3(256), 0, (hipStream_t)stream, nr, ni, cr,
                     ci, out, B, F, T, G, Fo, eps);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

int nppc_cirm_decompress_apply_conj(const float* crm, const float* nr, const float* ni, float* dec, float* emag,
                                    float* ere, float* eim, int B, int F, int T, void* stream) {
  if (!crm || !nr || !ni || !emag || !ere || !eim || B <= 0) return NPPC_EBADARG;
  hipLaunchKernelGGL(decompress_apply_kernel, dim3(ew_grid((size_t)B * F * T)), dim3(256), 0, (hipStream_t)stream, crm, nr,
                     ni, dec, emag, ere, eim, B, F, T, 1);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

int nppc_cirm_decompress_apply(const float* crm, const float* nr, const float* ni, float* dec, float* emag, float* ere,
                               float* eim, int B, int F, int T, void* stream) {
  if (!crm || !nr || !ni || !emag || !ere || !eim || B <= 0) return NPPC_EBADARG;
  hipLaunchKernelGGL(decompress_apply_kernel, dim3(ew_grid((size_t)B * F * T)), dim3(256), 0, (hipStream_t)stream, crm, nr,
                     ni, dec, emag, ere, eim, B, F, T, 0);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

int nppc_istft(const float* re, const float* im, float* out, int B, int T, int nfft, int hop, int L, void* stream) {
  if (!re || !im || !out || B <= 0 || T <= 0 || hop <= 0 || L <= 0 || nfft % hop || nfft / hop > 8) return NPPC_EBADARG;
  const int total = L + nfft / 2;                     // padded samples that can reach the output
  dim3 grid(ceil_div(total, ISTFT_FR * hop), B);
  hipStream_t s = (hipStream_t)stream;
  switch (nfft) {
    case 64: hipLaunchKernelGGL(istft_kernel<6>, grid, dim3(256), 0, s, re, im, out, T, hop, L); break;
    case 128: hipLaunchKernelGGL(istft_kernel<7>, grid, dim3(256), 0, s, re, im, out, T, hop, L); break;
    case 256: hipLaunchKernelGGL(istft_kernel<8>, grid, dim3(256), 0, s, re, im, out, T, hop, L); break;
    case 512: hipLaunchKernelGGL(istft_kernel<9>, grid, dim3(256), 0, s, re, im, out, T, hop, L); break;
    default: return NPPC_EUNSUPPORTED;
  }
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

int nppc_mix_snr(const float* clean, const float* noise, const float* snr_db, float target_dbfs, const float* target_item,
                 float* noisy_out, float* clean_out, int B, int L, void* stream) {
  if (!clean || !noise || !snr_db || !noisy_out || !clean_out || B <= 0 || L <= 0) return NPPC_EBADARG;
  hipLaunchKernelGGL(mix_snr_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, clean, noise, snr_db, target_dbfs, target_item,
                     noisy_out, clean_out, L);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

int nppc_inpaint_prepare(const float* clean, const int* gap_start, const int* gap_end, int do_norm, float target_dbfs,
                         float* clean_out, float* masked_out, float* mask_frames, int B, int L, int win, int hop, int T,
                         void* stream) {
  if (!clean || !gap_start || !gap_end || !mask_frames || B <= 0 || L <= 0 || win <= 0 || hop <= 0 || T <= 0)
    return NPPC_EBADARG;
  hipLaunchKernelGGL(inpaint_prepare_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, clean, gap_start, gap_end, do_norm,
                     target_dbfs, clean_out, masked_out, mask_frames, L, win, hop, T);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

int nppc_time_to_spec_mask(const float* mask_time, float* mask_frames, int B, int L, int win, int hop, int center, int T,
                           void* stream) {
  if (!mask_time || !mask_frames || B <= 0 || L <= 0 || win <= 0 || hop <= 0 || T <= 0) return NPPC_EBADARG;
  hipLaunchKernelGGL(time_to_spec_mask_kernel, dim3(ceil_div((long)B * T, 4L)), dim3(256), 0, (hipStream_t)stream, mask_time,
                     mask_frames, L, win, hop, center, T, B);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

int nppc_stft_pair(const float* wave, const float* mask_frames, float* spec, float* masked_spec, int B, int L, int nfft,
                   int hop, void* stream) {
  if (!wave || !spec || B <= 0 || L <= nfft / 2 || hop <= 0 || nfft < 2) return NPPC_EBADARG;
  if (nfft > 512) return NPPC_EUNSUPPORTED;
  if (masked_spec && !mask_frames) return NPPC_EBADARG;
  const int T = 1 + L / hop;
  const size_t lds = 2 * sizeof(double) * nfft + sizeof(float) * DFT_FR * (nfft | 1);
  hipLaunchKernelGGL(stft_dft_kernel, dim3(ceil_div(T, DFT_FR), B), dim3(256), lds, (hipStream_t)stream, wave, mask_frames,
                     spec, masked_spec, L, nfft, hop, T);
  NPPC_CHECK_LAUNCH();
  return NPPC_OK;
}

}  // extern "C"
