// fp8.hip — OCP e4m3 path of the semantic encoder / decoder GEMMs (BASELINE.json configs[4]: "fp8 MFMA encoder/decoder
// GEMMs").  Per-tensor scaling, forward contraction on v_mfma_f32_32x32x16_fp8_fp8 (gfx950 reads fp8 operands as OCP
// e4m3), straight-through backward on the de-quantised operands through the bf16/f32 kernels of gemm.hip.
//   scale = amax / 448, q = rne_e4m3(x * (448 / amax)), x^ = q * scale          (amax == 0: scale = 1, q = 0)
//   y = act(scale_x * scale_w * (xq . wq^T) + bias)
// The codec tensors are small by construction (B x 512 activations, 512 x 512 weights): the quantiser is ONE workgroup
// (amax, then quantise; nothing to synchronise across workgroups), the GEMM one wave per 32x32 output tile reading its
// operands straight from L2.  Build-authored (the upstream reference has no code); oracle: oracle/model.py fp8_qdq.
#include "msc_common.h"

#define FP8_MAX 448.0f
#define Q_THREADS 1024

template <typename T> __device__ __forceinline__ void load4(const T* p, float (&v)[4]);
template <> __device__ __forceinline__ void load4<float>(const float* p, float (&v)[4]) {
  const float4 t = *reinterpret_cast<const float4*>(p);
  v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
template <> __device__ __forceinline__ void load4<bf16_t>(const bf16_t* p, float (&v)[4]) {
  const uint2 t = *reinterpret_cast<const uint2*>(p);
  v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
  v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
}
template <typename T> __device__ __forceinline__ void store4(T* p, const float (&v)[4]);
template <> __device__ __forceinline__ void store4<float>(float* p, const float (&v)[4]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, const float (&v)[4]) {
  uint2 t;
  t.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
  t.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
  *reinterpret_cast<uint2*>(p) = t;
}

template <typename TI, typename TO>
__global__ __launch_bounds__(Q_THREADS) void fp8_quantize_kernel(const TI* __restrict__ x, uint32_t* __restrict__ q, TO* __restrict__ deq,
                                                                  float* __restrict__ scale_out, long n4) {
  __shared__ float red[Q_THREADS / 64];
  __shared__ float s_amax;
  const int tid = threadIdx.x;
  float m = 0.f;
  for (long i = tid; i < n4; i += Q_THREADS) {
    float v[4];
    load4<TI>(x + 4 * i, v);
    m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
  }
  m = wave_max(m);
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  if (tid == 0) {
    float a = 0.f;
    for (int w = 0; w < Q_THREADS / 64; ++w) a = fmaxf(a, red[w]);
    s_amax = a;
    scale_out[0] = a > 0.f ? a / FP8_MAX : 1.f;
  }
  __syncthreads();
  const float amax = s_amax;
  const float scale = amax > 0.f ? amax / FP8_MAX : 1.f;
  const float inv = amax > 0.f ? FP8_MAX / amax : 0.f;
  for (long i = tid; i < n4; i += Q_THREADS) {
    float v[4];
    load4<TI>(x + 4 * i, v);
    int pk = 0;
    pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[0] * inv, v[1] * inv, pk, false);
    pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[2] * inv, v[3] * inv, pk, true);
    q[i] = (uint32_t)pk;
    if (deq) {
      float d[4];
      d[0] = __builtin_amdgcn_cvt_f32_fp8(pk, 0) * scale;
      d[1] = __builtin_amdgcn_cvt_f32_fp8(pk, 1) * scale;
      d[2] = __builtin_amdgcn_cvt_f32_fp8(pk, 2) * scale;
      d[3] = __builtin_amdgcn_cvt_f32_fp8(pk, 3) * scale;
      store4<TO>(deq + 4 * i, d);
    }
  }
}

extern "C" int msc_fp8_quantize(const void* x, void* q, void* deq, float* scale_out, int64_t n, int in_dtype,
                                int deq_dtype, void* stream) {
  MSC_CHECK_ARG(x && q && scale_out && n > 0 && n % 4 == 0 && n <= (1L << 24));
  MSC_CHECK_ARG((in_dtype == MSC_F32 || in_dtype == MSC_BF16) && (deq_dtype == MSC_F32 || deq_dtype == MSC_BF16));
  hipStream_t st = (hipStream_t)stream;
  const long n4 = n / 4;
#define QL(TI, TO) hipLaunchKernelGGL((fp8_quantize_kernel<TI, TO>), dim3(1), dim3(Q_THREADS), 0, st, (const TI*)x, (uint32_t*)q, (TO*)deq, scale_out, n4)
  if (in_dtype == MSC_F32 && deq_dtype == MSC_F32) QL(float, float);
  else if (in_dtype == MSC_F32) QL(float, bf16_t);
  else if (deq_dtype == MSC_F32) QL(bf16_t, float);
  else QL(bf16_t, bf16_t);
#undef QL
  return msc_launch_status();
}

// One wave per 32x32 tile of y.  Lane (r = lane & 31, h = lane >> 5) holds row m0+r of xq / row n0+r of wq, 16 bytes
// at k0 + 16 h per 32-wide K step: two MFMA K steps of 8 bytes each.  (The hardware's k order inside a step is lanes
// 0-31: k 0..7, lanes 32-63: k 8..15; A and B use the same permutation of k, so the contraction is unchanged.)
template <typename T>
__global__ __launch_bounds__(64) void linear_fp8_kernel(const uint8_t* __restrict__ xq, const uint8_t* __restrict__ wq,
                                                        const float* __restrict__ sx, const float* __restrict__ sw,
                                                        const float* __restrict__ bias, T* __restrict__ y, int M, int N,
                                                        int K, int act) {
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
  const bool aok = m0 + r < M, bok = n0 + r < N;
  const uint8_t* ap = xq + (long)(aok ? m0 + r : 0) * K + 16 * h;
  const uint8_t* bp = wq + (long)(bok ? n0 + r : 0) * K + 16 * h;
  f32x16_t acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll 4
  for (int k0 = 0; k0 < K; k0 += 32) {
    uint4 a = aok ? *reinterpret_cast<const uint4*>(ap + k0) : make_uint4(0, 0, 0, 0);
    uint4 b = bok ? *reinterpret_cast<const uint4*>(bp + k0) : make_uint4(0, 0, 0, 0);
    const long a0 = (long)(((uint64_t)a.y << 32) | a.x), a1 = (long)(((uint64_t)a.w << 32) | a.z);
    const long b0 = (long)(((uint64_t)b.y << 32) | b.x), b1 = (long)(((uint64_t)b.w << 32) | b.z);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a0, b0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a1, b1, acc, 0, 0, 0);
  }
  const float s = sx[0] * sw[0];
  const int n = n0 + r;
  if (n >= N) return;
  const float bv = bias ? bias[n] : 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int m = m0 + (i & 3) + 8 * (i >> 2) + 4 * h;
    if (m < M) msc_traits<T>::st(y + (long)m * N + n, act_apply(acc[i] * s + bv, act));
  }
}

extern "C" int msc_linear_fp8_fwd(const void* xq, const void* wq, const float* sx, const float* sw, const float* bias,
                                  void* y, int M, int N, int K, int act, int dtype, void* stream) {
  MSC_CHECK_ARG(xq && wq && sx && sw && y && M > 0 && N > 0 && K > 0 && K % 32 == 0);
  MSC_CHECK_ARG(dtype == MSC_F32 || dtype == MSC_BF16);
  const dim3 grid((unsigned)cdiv(N, 32), (unsigned)cdiv(M, 32));
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MSC_F32)
    hipLaunchKernelGGL(linear_fp8_kernel<float>, grid, dim3(64), 0, st, (const uint8_t*)xq, (const uint8_t*)wq, sx, sw, bias, (float*)y, M, N, K, act);
  else
    hipLaunchKernelGGL(linear_fp8_kernel<bf16_t>, grid, dim3(64), 0, st, (const uint8_t*)xq, (const uint8_t*)wq, sx, sw, bias, (bf16_t*)y, M, N, K, act);
  return msc_launch_status();
}
