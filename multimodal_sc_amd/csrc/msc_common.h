// msc_common.h — shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels.
// Wave = 64 lanes everywhere in this tree; nothing here is meant to build for another target.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include "../../include/msc.h"

typedef uint16_t bf16_t;  // raw bf16 bits; arithmetic always happens in f32

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;   // MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(16))) float f32x16_t;  // 32x32 MFMA accumulator
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

#define MSC_WAVE 64

// Wave priority of the kernels that sit on the step's critical chain (memory-bound normalisation passes, attention, the ping-pong
// contraction): their waves share SIMDs with the side stream's weight-gradient GEMMs, and the instruction arbiter serves the higher
// s_setprio level first.  0 = off.  Measured same-box (profiles/r4_epilogue_prefetch_and_priority.txt): 7.320 / 7.321 ms per step at
// level 2 against 7.334 / 7.347 at 0 — the contention is mostly about register-file residency (a weight-gradient workgroup holds its
// registers for 50-140 us whatever its priority), so the gain is small; kept because it is free.
#ifndef MSC_BOOST_LEVEL
#define MSC_BOOST_LEVEL 2
#endif
#define MSC_BOOST() do { if (MSC_BOOST_LEVEL) __builtin_amdgcn_s_setprio(MSC_BOOST_LEVEL); } while (0)

__device__ __forceinline__ float bf16_to_f32(bf16_t v) {
  return __uint_as_float(((uint32_t)v) << 16);
}
// round-to-nearest-even; the plain cast keeps NaN a NaN (v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(bf16_t, b);
}

template <typename T> struct msc_traits;
template <> struct msc_traits<float> {
  static constexpr int VEC = 4;  // elements per 16-byte chunk
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
  static __device__ __forceinline__ float rnd(float v) { return v; }     // the value a store of v reads back as
};
template <> struct msc_traits<bf16_t> {
  static constexpr int VEC = 8;
  static __device__ __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(*p); }
  static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f32_to_bf16(v); }
  static __device__ __forceinline__ float rnd(float v) { return bf16_to_f32(f32_to_bf16(v)); }
};

// 16-byte vector of T unpacked to floats
template <typename T> struct vecf;
template <> struct vecf<float> {
  float v[4];
  __device__ __forceinline__ void load(const float* p) {
    float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  }
  __device__ __forceinline__ void store(float* p) const {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  }
  __device__ __forceinline__ void store_nt(float* p) const {
    typedef __attribute__((ext_vector_type(4))) float f32x4;
    f32x4 w = {v[0], v[1], v[2], v[3]};
    __builtin_nontemporal_store(w, reinterpret_cast<f32x4*>(p));
  }
};
template <> struct vecf<bf16_t> {
  float v[8];
  __device__ __forceinline__ void load(const bf16_t* p) {
    uint4 t = *reinterpret_cast<const uint4*>(p);
    uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[2 * i] = __uint_as_float(w[i] << 16);
      v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
  }
  __device__ __forceinline__ void store(bf16_t* p) const {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
      w[i] = (uint32_t)f32_to_bf16(v[2 * i]) | ((uint32_t)f32_to_bf16(v[2 * i + 1]) << 16);
    *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
  }
  // streaming (non-temporal) form: the line is not kept in the L2 on its way out
  __device__ __forceinline__ void store_nt(bf16_t* p) const {
    typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
    u32x4 w;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      w[i] = (uint32_t)f32_to_bf16(v[2 * i]) | ((uint32_t)f32_to_bf16(v[2 * i + 1]) << 16);
    __builtin_nontemporal_store(w, reinterpret_cast<u32x4*>(p));
  }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// block-wide sum for blockDim.x multiple of 64, up to 1024 threads; all threads get the result.
__device__ __forceinline__ float block_sum(float v, float* smem /* >= 17 floats */) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if (lane == 0) smem[w] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < nw; ++i) r += smem[i];
  return r;
}

// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, below fp32 GEMM rounding): 1 exp + 1 rcp + 6 FMA instead of
// the ~40-instruction libm erff — GELU sits in the epilogue of the widest GEMM of the transformer.
// The reciprocal is the hardware's v_rcp_f32 (1 ulp): `__frcp_rn` expands to the correctly rounded division sequence (v_div_scale x2,
// v_rcp, 4 FMA, v_div_fmas, v_div_fixup — 10 of the epilogue's 27 instructions per element, round 4), for a last bit of t that the
// approximation's own 1.5e-7 swamps.
__device__ __forceinline__ float fast_erf(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.f));
  float y = fmaf(1.061405429f, t, -1.453152027f);
  y = fmaf(y, t, 1.421413741f);
  y = fmaf(y, t, -0.284496736f);
  y = fmaf(y, t, 0.254829592f);
  y = 1.f - y * t * __expf(-ax * ax);
  return copysignf(y, x);
}

// gelu(x) and gelu'(x) = Phi(x) + x phi(x) from ONE erf evaluation: A&S 7.1.26 at x / sqrt(2) already holds exp(-x^2 / 2)
__device__ __forceinline__ void gelu_both(float x, float& y, float& dy) {
  const float ax = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.f));
  float q = fmaf(1.061405429f, t, -1.453152027f);
  q = fmaf(q, t, 1.421413741f);
  q = fmaf(q, t, -0.284496736f);
  q = fmaf(q, t, 0.254829592f);
  const float e = __expf(-ax * ax);
  const float erfv = copysignf(1.f - q * t * e, x);
  y = 0.5f * x * (1.f + erfv);                       // the same operations in the same order as act_apply(x, MSC_ACT_GELU)
  dy = fmaf(x * 0.3989422804014327f, e, 0.5f * (1.f + erfv));
}

// gelu_both on TWO elements: the polynomial / product part on the packed fp32 instructions of gfx950 (v_pk_fma_f32, v_pk_mul_f32: two lanes'
// worth per issue slot), the two transcendentals per element as before — 23 instructions per pair against 27 per element.  The epilogue of
// the widest GEMM of the transformer (fc1: GELU + derivative over 19584 x 2048 outputs per layer) is bound by exactly this arithmetic.
typedef __attribute__((ext_vector_type(2))) float msc_f32x2;
__device__ __forceinline__ void gelu_both2(msc_f32x2 x, msc_f32x2& y, msc_f32x2& dy) {
  const msc_f32x2 one = {1.f, 1.f}, half = {0.5f, 0.5f};
  const msc_f32x2 ax = __builtin_elementwise_abs(x) * 0.70710678118654752f;
  const msc_f32x2 den = __builtin_elementwise_fma(ax, (msc_f32x2){0.3275911f, 0.3275911f}, one);
  const msc_f32x2 t = {__builtin_amdgcn_rcpf(den[0]), __builtin_amdgcn_rcpf(den[1])};
  msc_f32x2 q = __builtin_elementwise_fma((msc_f32x2){1.061405429f, 1.061405429f}, t, (msc_f32x2){-1.453152027f, -1.453152027f});
  q = __builtin_elementwise_fma(q, t, (msc_f32x2){1.421413741f, 1.421413741f});
  q = __builtin_elementwise_fma(q, t, (msc_f32x2){-0.284496736f, -0.284496736f});
  q = __builtin_elementwise_fma(q, t, (msc_f32x2){0.254829592f, 0.254829592f});
  const msc_f32x2 a2 = ax * ax;
  const msc_f32x2 e = {__expf(-a2[0]), __expf(-a2[1])};
  const msc_f32x2 erfa = one - q * t * e;
  const msc_f32x2 erfv = {copysignf(erfa[0], x[0]), copysignf(erfa[1], x[1])};
  const msc_f32x2 ph = __builtin_elementwise_fma(erfv, half, half);     // Phi(x)
  y = x * ph;
  dy = __builtin_elementwise_fma(x * 0.3989422804014327f, e, ph);
}

__device__ __forceinline__ float act_apply(float x, int act) {
  switch (act) {
    case MSC_ACT_RELU: return x > 0.f ? x : 0.f;
    case MSC_ACT_GELU: return 0.5f * x * (1.f + fast_erf(x * 0.70710678118654752f));
    case MSC_ACT_TANH: return tanhf(x);
    case MSC_ACT_SIGMOID: return 1.f / (1.f + __expf(-x));
    default: return x;
  }
}

// d act(x)/dx given ref = output y (relu/tanh/sigmoid) or the pre-activation x (gelu)
__device__ __forceinline__ float act_grad(float r, int act) {
  switch (act) {
    case MSC_ACT_RELU: return r > 0.f ? 1.f : 0.f;
    case MSC_ACT_TANH: return 1.f - r * r;
    case MSC_ACT_SIGMOID: return r * (1.f - r);
    case MSC_ACT_GELU: return 0.5f * (1.f + fast_erf(r * 0.70710678118654752f)) + r * 0.3989422804014327f * __expf(-0.5f * r * r);
    case MSC_ACT_MUL_REF: return r;
    default: return 1.f;
  }
}

// Philox4x32-10 (Salmon et al., SC'11); oracle/philox.py restates it with the Random123 known-answer vector
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static inline int ilog2_exact(int v) {  // -1 when v is not a power of two
  if (v <= 0 || (v & (v - 1))) return -1;
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}

#define MSC_CHECK_ARG(cond) \
  do {                      \
    if (!(cond)) return MSC_ERR_BAD_ARG; \
  } while (0)

static inline int msc_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? MSC_OK : MSC_ERR_LAUNCH;
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
