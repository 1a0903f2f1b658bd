// channel.hip — SemanticEncoder power normalisation + Channel (AWGN / Rayleigh) fused, fwd + bwd.
//   s = x * sqrt(k / sum(x^2))                 (unit average power per real symbol)
//   AWGN     : y = s + sigma*n,                sigma = 10^(-snr_db/20)
//   Rayleigh : symbols paired as complex, h = (hr + j*hi)/sqrt(2), y_c = h*s_c + sigma*n_c,
//              receiver equalises with perfect CSI: out = y_c / h = s_c + sigma*n_c*conj(h)/|h|^2
// Noise/fading come from explicit fp32 tensors (parity mode) or from an in-kernel counter-based
// Philox4x32-10 (throughput mode; oracle/philox.py restates it): key = the 64-bit seed (the host puts the DP rank in its
// high word), counter = (block index lo, hi, stream id | draw hi, draw lo) where `draw` is a device-side counter the
// caller bumps after every forward — seed, rank and draw occupy separate words, so no two of them alias.
// HBM-bound and tiny: one wave per sample row, one pass, wavefront-shuffle reductions.
// Build-authored (the upstream reference has no code, SURVEY.md finding zero).
#include "msc_common.h"

// 4 standard normals from one Philox block (two Box-Muller pairs)
__device__ __forceinline__ void philox_normal4(uint64_t seed, uint64_t index, uint32_t stream_id, uint64_t draw, float (&z)[4]) {
  uint32_t r[4];
  philox4x32_10((uint32_t)index, (uint32_t)(index >> 32), stream_id + 2u * (uint32_t)(draw >> 32), (uint32_t)draw, (uint32_t)seed,
                (uint32_t)(seed >> 32), r);
  const float inv = 2.3283064365386963e-10f;  // 2^-32
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const float u1 = ((float)r[2 * i] + 0.5f) * inv;
    const float u2 = ((float)r[2 * i + 1] + 0.5f) * inv;
    const float rad = sqrtf(-2.f * logf(u1));
    float sn, cs;
    sincosf(6.283185307179586f * u2, &sn, &cs);
    z[2 * i] = rad * cs;
    z[2 * i + 1] = rad * sn;
  }
}

#define CH_MAXQ 4  // quads per lane: k <= 64*4*4 = 1024 symbols

template <typename T>
__global__ __launch_bounds__(256) void channel_fwd_kernel(const T* __restrict__ x, const float* __restrict__ noise,
                                                          const float* __restrict__ fading, const float* __restrict__ snr_db,
                                                          T* __restrict__ y, float* __restrict__ scale, int B, int k, int mode,
                                                          uint64_t seed, uint64_t draw, const int64_t* __restrict__ draw_dev) {
  // draw number = host-supplied count (eager launches) + device-side counter (bumped by the caller after every forward where
  // the launch sequence is replayed from a hipGraph and a host value would be frozen): fresh noise per forward either way
  if (draw_dev) draw += (uint64_t)*draw_dev;
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const int nq = k / 4;  // quads in the row
  float v[CH_MAXQ][4];
  float ss = 0.f;
#pragma unroll
  for (int q = 0; q < CH_MAXQ; ++q) {
    const int qi = q * 64 + lane;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v[q][j] = (qi < nq) ? msc_traits<T>::ld(x + (long)b * k + qi * 4 + j) : 0.f;
      ss += v[q][j] * v[q][j];
    }
  }
  ss = wave_sum(ss);
  const float r = sqrtf((float)k / (ss + 1e-12f));
  const float sigma = exp10f(-snr_db[b] * 0.05f);
#pragma unroll
  for (int q = 0; q < CH_MAXQ; ++q) {
    const int qi = q * 64 + lane;
    if (qi >= nq) continue;
    float n[4], h[4];
    if (noise) {
#pragma unroll
      for (int j = 0; j < 4; ++j) n[j] = noise[(long)b * k + qi * 4 + j];
    } else {
      philox_normal4(seed, (uint64_t)b * nq + qi, 0u, draw, n);
    }
    if (mode == MSC_CH_RAYLEIGH) {
      if (fading) {
#pragma unroll
        for (int j = 0; j < 4; ++j) h[j] = fading[(long)b * k + qi * 4 + j];
      } else {
        philox_normal4(seed, (uint64_t)b * nq + qi, 1u, draw, h);
      }
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {  // two complex symbols per quad
      const float sr = v[q][2 * c] * r, si = v[q][2 * c + 1] * r;
      float orr, oi;
      if (mode == MSC_CH_RAYLEIGH) {
        const float hr = h[2 * c] * 0.70710678118654752f, hi = h[2 * c + 1] * 0.70710678118654752f;
        const float nr = sigma * n[2 * c], ni = sigma * n[2 * c + 1];
        const float den = 1.f / (hr * hr + hi * hi + 1e-9f);
        // n_c * conj(h) / |h|^2
        orr = sr + (nr * hr + ni * hi) * den;
        oi = si + (ni * hr - nr * hi) * den;
      } else {
        orr = sr + sigma * n[2 * c];
        oi = si + sigma * n[2 * c + 1];
      }
      msc_traits<T>::st(y + (long)b * k + qi * 4 + 2 * c, orr);
      msc_traits<T>::st(y + (long)b * k + qi * 4 + 2 * c + 1, oi);
    }
  }
  if (lane == 0) scale[b] = r;
}

// dx = r * (dy - s * (s . dy) / k),  s = x * r     (the additive noise term carries no gradient)
template <typename T>
__global__ __launch_bounds__(256) void channel_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                          const float* __restrict__ scale, T* __restrict__ dx, int B, int k) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const int n = k;
  const float r = scale[b];
  float dot = 0.f;
  for (int i = lane; i < n; i += 64) {
    const float s = msc_traits<T>::ld(x + (long)b * k + i) * r;
    dot += s * msc_traits<T>::ld(dy + (long)b * k + i);
  }
  dot = wave_sum(dot) / (float)k;
  for (int i = lane; i < n; i += 64) {
    const float s = msc_traits<T>::ld(x + (long)b * k + i) * r;
    const float g = msc_traits<T>::ld(dy + (long)b * k + i);
    msc_traits<T>::st(dx + (long)b * k + i, r * (g - s * dot));
  }
}

// raw generator output for tests: out[i*4..i*4+3] = 4 normals of Philox block i on `stream_id`
__global__ void philox_fill_kernel(float* __restrict__ out, long nquads, uint64_t seed, uint32_t stream_id, uint64_t draw) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nquads; i += (long)gridDim.x * blockDim.x) {
    float z[4];
    philox_normal4(seed, (uint64_t)i, stream_id, draw, z);
#pragma unroll
    for (int j = 0; j < 4; ++j) out[i * 4 + j] = z[j];
  }
}

extern "C" int msc_channel_fwd(const void* x, const float* noise, const float* fading, const float* snr_db, void* y, float* scale,
                               int B, int k, int mode, uint64_t seed, uint64_t draw, const int64_t* draw_dev, int dtype, void* stream) {
  MSC_CHECK_ARG(x && snr_db && y && scale && B > 0 && k > 0 && k % 4 == 0 && k <= 64 * 4 * CH_MAXQ);
  MSC_CHECK_ARG(mode == MSC_CH_AWGN || mode == MSC_CH_RAYLEIGH);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MSC_F32)
    hipLaunchKernelGGL(channel_fwd_kernel<float>, dim3(cdiv(B, 4)), dim3(256), 0, st, (const float*)x, noise, fading, snr_db, (float*)y, scale, B, k, mode, seed, draw, draw_dev);
  else if (dtype == MSC_BF16)
    hipLaunchKernelGGL(channel_fwd_kernel<bf16_t>, dim3(cdiv(B, 4)), dim3(256), 0, st, (const bf16_t*)x, noise, fading, snr_db, (bf16_t*)y, scale, B, k, mode, seed, draw, draw_dev);
  else
    return MSC_ERR_BAD_ARG;
  return msc_launch_status();
}

extern "C" int msc_channel_bwd(const void* dy, const void* x, const float* scale, void* dx, int B, int k, int dtype, void* stream) {
  MSC_CHECK_ARG(dy && x && scale && dx && B > 0 && k > 0);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MSC_F32)
    hipLaunchKernelGGL(channel_bwd_kernel<float>, dim3(cdiv(B, 4)), dim3(256), 0, st, (const float*)dy, (const float*)x, scale, (float*)dx, B, k);
  else if (dtype == MSC_BF16)
    hipLaunchKernelGGL(channel_bwd_kernel<bf16_t>, dim3(cdiv(B, 4)), dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)x, scale, (bf16_t*)dx, B, k);
  else
    return MSC_ERR_BAD_ARG;
  return msc_launch_status();
}

__global__ void counter_add_kernel(int64_t* ctr, int64_t inc) { *ctr += inc; }

// ctr[0] += inc on the stream (device-side step / seed counters that survive hipGraph replay)
extern "C" int msc_counter_add(int64_t* ctr, int64_t inc, void* stream) {
  MSC_CHECK_ARG(ctr);
  hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, ctr, inc);
  return msc_launch_status();
}

extern "C" int msc_philox_normal(float* out, int64_t nquads, uint64_t seed, uint32_t stream_id, uint64_t draw, void* stream) {
  MSC_CHECK_ARG(out && nquads > 0);
  long b = (nquads + 255) / 256;
  if (b > 2048) b = 2048;
  hipLaunchKernelGGL(philox_fill_kernel, dim3((unsigned)b), dim3(256), 0, (hipStream_t)stream, out, (long)nquads, seed, stream_id, draw);
  return msc_launch_status();
}
