// wire.hip — wire format of the semantic symbols between SemanticEncoder and Channel:
// the k real symbols of a frame are k/2 complex I/Q pairs; on the wire each frame is one fp32 scale followed by k signed
// bytes (I0,Q0,I1,Q1,...):  scale = amax/127 (1 when the frame is all zero), q = clamp(rne(x / scale), -127, 127).
// One wave per frame, 16-byte loads, wavefront-shuffle max.  Integer output: bit-exact against oracle/wire.py
// (fp contraction off: the division and rounding are the oracle's).  Build-authored (the upstream reference has no code).
#include "msc_common.h"
#pragma clang fp contract(off)

template <typename T>
__global__ __launch_bounds__(256) void symbols_pack_kernel(const T* __restrict__ x, int8_t* __restrict__ q, float* __restrict__ scale, int B, int k) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (b >= B) return;
  const T* row = x + (long)b * k;
  float m = 0.f;
  for (int i = lane; i < k; i += 64) m = fmaxf(m, fabsf(msc_traits<T>::ld(row + i)));
  m = wave_max(m);
  const float s = m > 0.f ? m / 127.f : 1.f;
  if (lane == 0) scale[b] = s;
  for (int i = lane; i < k; i += 64) {
    float v = rintf(msc_traits<T>::ld(row + i) / s);
    v = fminf(fmaxf(v, -127.f), 127.f);
    q[(long)b * k + i] = (int8_t)(int)v;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void symbols_unpack_kernel(const int8_t* __restrict__ q, const float* __restrict__ scale, T* __restrict__ y, long n, int k) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  msc_traits<T>::st(y + i, (float)q[i] * scale[i / k]);
}

extern "C" int msc_symbols_pack(const void* x, void* q, float* scale, int B, int k, int dtype, void* stream) {
  MSC_CHECK_ARG(x && q && scale && B > 0 && k > 0 && k % 2 == 0);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MSC_F32) hipLaunchKernelGGL(symbols_pack_kernel<float>, dim3(cdiv(B, 4)), dim3(256), 0, st, (const float*)x, (int8_t*)q, scale, B, k);
  else if (dtype == MSC_BF16) hipLaunchKernelGGL(symbols_pack_kernel<bf16_t>, dim3(cdiv(B, 4)), dim3(256), 0, st, (const bf16_t*)x, (int8_t*)q, scale, B, k);
  else return MSC_ERR_BAD_ARG;
  return msc_launch_status();
}

extern "C" int msc_symbols_unpack(const void* q, const float* scale, void* y, int B, int k, int dtype, void* stream) {
  MSC_CHECK_ARG(q && scale && y && B > 0 && k > 0);
  hipStream_t st = (hipStream_t)stream;
  const long n = (long)B * k;
  if (dtype == MSC_F32) hipLaunchKernelGGL(symbols_unpack_kernel<float>, dim3(cdiv(n, 256)), dim3(256), 0, st, (const int8_t*)q, scale, (float*)y, n, k);
  else if (dtype == MSC_BF16) hipLaunchKernelGGL(symbols_unpack_kernel<bf16_t>, dim3(cdiv(n, 256)), dim3(256), 0, st, (const int8_t*)q, scale, (bf16_t*)y, n, k);
  else return MSC_ERR_BAD_ARG;
  return msc_launch_status();
}
