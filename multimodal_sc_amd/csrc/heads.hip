// heads.hip — policy/value head selection + losses, reconstruction loss, attention softmax.
//   msc_argmax_rows     : arg-max with lowest-index tie-break (bit-exact integer output)
//   msc_topk_rows       : top-k action indices (k <= 8 of n <= 32), order (value desc, index asc); k = 1 == arg-max
//   msc_ppo_loss        : PPO clipped surrogate + value + entropy, returns loss terms AND d/d(head output)
//   msc_recon_loss      : MSE between sigmoid patch output and the NCHW image, per-sample SSE (-> PSNR) and gradient
//   msc_unpatchify      : patches -> NCHW fp32 image
//   msc_softmax_fwd/bwd : row softmax for the unfused attention path (fp32 scores in, T probabilities out)
// Build-authored (the upstream reference has no code, SURVEY.md finding zero).
#include "msc_common.h"

// Order used by arg-max and top-k: value descending, index ascending on exact ties (0.0 == -0.0 is a tie), NaN after
// everything else (also after -inf).  `a beats b` for a candidate at a HIGHER index than b, so ties keep the lower index.
__device__ __forceinline__ bool rank_beats(float va, float vb) {
  const bool na = va != va, nb = vb != vb;
  return na != nb ? nb : va > vb;
}

template <typename T>
__global__ void argmax_rows_kernel(const T* __restrict__ x, int64_t* __restrict__ out, long rows, int n, long ld) {
  const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  float best = msc_traits<T>::ld(x + r * ld);
  int bi = 0;
  for (int j = 1; j < n; ++j) {
    const float v = msc_traits<T>::ld(x + r * ld + j);
    if (rank_beats(v, best)) { best = v; bi = j; }   // strict: the lowest index wins ties
  }
  out[r] = bi;
}

// top-k by k selection passes over a row held in registers (n <= 32 action logits, k <= 8): pass p picks the best element
// not picked yet; bit p of `taken` marks it.  One thread per row: B rows of 9-16 logits, pure latency.
#define TOPK_MAX_N 32
#define TOPK_MAX_K 8
template <typename T>
__global__ void topk_rows_kernel(const T* __restrict__ x, int64_t* __restrict__ out_idx, float* __restrict__ out_val, long rows,
                                 int n, int k, long ld) {
  const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  float v[TOPK_MAX_N];
#pragma unroll
  for (int j = 0; j < TOPK_MAX_N; ++j) v[j] = j < n ? msc_traits<T>::ld(x + r * ld + j) : 0.f;
  uint32_t taken = 0;
  for (int p = 0; p < k; ++p) {
    int bi = -1;
    float best = 0.f;
#pragma unroll
    for (int j = 0; j < TOPK_MAX_N; ++j) {
      if (j < n && !((taken >> j) & 1u) && (bi < 0 || rank_beats(v[j], best))) { best = v[j]; bi = j; }
    }
    taken |= 1u << bi;
    out_idx[r * k + p] = bi;
    if (out_val) out_val[r * k + p] = best;
  }
}

// One block.  head: T [B][ld], cols 0..A-1 = logits, col A = value.
// terms (fp32[4]) = {total, policy, value, entropy};  dhead: T [B][ld] gradient of total wrt head.
#define PPO_MAX_A 32
template <typename T>
__global__ __launch_bounds__(256) void ppo_loss_kernel(const T* __restrict__ head, const int64_t* __restrict__ actions,
                                                       const float* __restrict__ old_logp, const float* __restrict__ adv,
                                                       const float* __restrict__ ret, float* __restrict__ terms,
                                                       T* __restrict__ dhead, int B, int A, int ld, float clip_eps,
                                                       float c_v, float c_e) {
  __shared__ float red[20];
  float pl = 0.f, vl = 0.f, en = 0.f;
  const float invB = 1.f / (float)B;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    float l[PPO_MAX_A];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < PPO_MAX_A; ++j)
      if (j < A) { l[j] = msc_traits<T>::ld(head + (long)b * ld + j); mx = fmaxf(mx, l[j]); }
    float se = 0.f;
#pragma unroll
    for (int j = 0; j < PPO_MAX_A; ++j)
      if (j < A) se += expf(l[j] - mx);
    const float lse = mx + logf(se);
    const int a = (int)actions[b];
    float H = 0.f, lpa = 0.f;
#pragma unroll
    for (int j = 0; j < PPO_MAX_A; ++j)
      if (j < A) {
        const float lp = l[j] - lse;
        H -= expf(lp) * lp;
        if (j == a) lpa = lp;
      }
    const float ratio = expf(lpa - old_logp[b]);
    const float ad = adv[b];
    const float s1 = ratio * ad;
    const float rc = fminf(fmaxf(ratio, 1.f - clip_eps), 1.f + clip_eps);
    const float s2 = rc * ad;
    const float surr = fminf(s1, s2);
    // d surr / d logp_a: s1 branch (or tie inside the clip range) -> ratio*adv, clipped branch -> 0
    const float g = (s1 <= s2) ? ratio * ad : ((rc == ratio) ? ratio * ad : 0.f);
    const float val = msc_traits<T>::ld(head + (long)b * ld + A);
    const float dv = val - ret[b];
    pl -= surr;
    vl += 0.5f * dv * dv;
    en += H;
#pragma unroll
    for (int j = 0; j < PPO_MAX_A; ++j)
      if (j < A) {
        const float lp = l[j] - lse, p = expf(lp);
        float d = -g * ((j == a ? 1.f : 0.f) - p);   // policy term
        d += c_e * p * (lp + H);                       // -c_e * entropy term
        msc_traits<T>::st(dhead + (long)b * ld + j, d * invB);
      }
    msc_traits<T>::st(dhead + (long)b * ld + A, c_v * dv * invB);
    for (int j = A + 1; j < ld; ++j) msc_traits<T>::st(dhead + (long)b * ld + j, 0.f);
  }
  pl = block_sum(pl, red);
  vl = block_sum(vl, red);
  en = block_sum(en, red);
  if (threadIdx.x == 0) {
    pl *= invB; vl *= invB; en *= invB;
    terms[0] = pl + c_v * vl - c_e * en;
    terms[1] = pl;
    terms[2] = vl;
    terms[3] = en;
  }
}

// ---------------------------------------------------------------- reconstruction loss
// recon: T [B][G*G][P*P*3] (pixel order py,px,c) ; img: fp32 NCHW [B][3][G*P][G*P]
// partial[b][chunk] = SSE of the chunk; drecon = gscale * (recon - img)
#define RL_CHUNKS 32
// A thread takes 8 consecutive pixels of one patch row: 24 consecutive recon elements (three 16-byte chunks for bf16) against 8 consecutive
// floats of each colour plane — whole-vector loads and stores, no per-element index arithmetic (the element-wise version: 42 us for 77 MB,
// on the critical path between the forward and the backward).  P % 8 == 0; other P: the element-wise loop.
template <typename T>
__global__ __launch_bounds__(256) void recon_loss_kernel(const T* __restrict__ recon, const float* __restrict__ img,
                                                         float* __restrict__ partial, T* __restrict__ drecon, int G, int P,
                                                         float gscale, int vec_ok) {
  __shared__ float red[20];
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int HW = G * P;
  const long per = (long)G * G * P * P * 3;
  float s = 0.f;
  if (vec_ok) {                                               // host: P % 8 == 0 and 16-byte aligned recon / img / drecon
    const int gpr = P >> 3;                                   // 8-pixel groups per patch row
    const long ngrp = (long)G * G * P * gpr;                  // groups of this sample
    const long lo = ngrp * chunk / RL_CHUNKS, hi = ngrp * (chunk + 1) / RL_CHUNKS;
    for (long g = lo + threadIdx.x; g < hi; g += 256) {
      const int pg = (int)(g % gpr);
      long t = g / gpr;
      const int py = (int)(t % P); t /= P;
      const int gx = (int)(t % G);
      const int gy = (int)(t / G);
      const long e0 = (long)b * per + g * 24;                  // (patch, py, px = 8 pg .. 8 pg + 7, c): 24 consecutive elements
      float r[24];
      if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          vecf<T> v;
          v.load(recon + e0 + q * 8);
#pragma unroll
          for (int k = 0; k < 8; ++k) r[q * 8 + k] = v.v[k];
        }
      } else {
#pragma unroll
        for (int q = 0; q < 6; ++q) {
          vecf<T> v;
          v.load(recon + e0 + q * 4);
#pragma unroll
          for (int k = 0; k < 4; ++k) r[q * 4 + k] = v.v[k];
        }
      }
      float d[24];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float* ip = img + (((long)b * 3 + c) * HW + gy * P + py) * HW + gx * P + pg * 8;
        const float4 a0 = *reinterpret_cast<const float4*>(ip), a1 = *reinterpret_cast<const float4*>(ip + 4);
        const float iv[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
        for (int k = 0; k < 8; ++k) d[k * 3 + c] = r[k * 3 + c] - iv[k];
      }
#pragma unroll
      for (int k = 0; k < 24; ++k) s += d[k] * d[k];
      if (drecon) {
        if constexpr (sizeof(T) == 2) {
#pragma unroll
          for (int q = 0; q < 3; ++q) {
            vecf<T> v;
#pragma unroll
            for (int k = 0; k < 8; ++k) v.v[k] = gscale * d[q * 8 + k];
            v.store(drecon + e0 + q * 8);
          }
        } else {
#pragma unroll
          for (int q = 0; q < 6; ++q) {
            vecf<T> v;
#pragma unroll
            for (int k = 0; k < 4; ++k) v.v[k] = gscale * d[q * 4 + k];
            v.store(drecon + e0 + q * 4);
          }
        }
      }
    }
  } else {
    const long lo = per * chunk / RL_CHUNKS, hi = per * (chunk + 1) / RL_CHUNKS;
    for (long e = lo + threadIdx.x; e < hi; e += 256) {
      const int c = (int)(e % 3);
      long t = e / 3;
      const int px = (int)(t % P); t /= P;
      const int py = (int)(t % P); t /= P;
      const int gx = (int)(t % G);
      const int gy = (int)(t / G);
      const float r = msc_traits<T>::ld(recon + (long)b * per + e);
      const float d = r - img[(((long)b * 3 + c) * HW + gy * P + py) * HW + gx * P + px];
      s += d * d;
      if (drecon) msc_traits<T>::st(drecon + (long)b * per + e, gscale * d);
    }
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0) partial[b * RL_CHUNKS + chunk] = s;
}
__global__ void recon_loss_finalize_kernel(const float* __restrict__ partial, float* __restrict__ sse, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float s = 0.f;
  for (int c = 0; c < RL_CHUNKS; ++c) s += partial[b * RL_CHUNKS + c];
  sse[b] = s;
}

template <typename T>
__global__ void unpatchify_kernel(const T* __restrict__ recon, float* __restrict__ img, int B, int G, int P) {
  const int HW = G * P;
  const long n = (long)B * 3 * HW * HW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int x = (int)(i % HW);
    long t = i / HW;
    const int y = (int)(t % HW); t /= HW;
    const int c = (int)(t % 3);
    const long b = t / 3;
    const int gy = y / P, py = y % P, gx = x / P, px = x % P;
    img[i] = msc_traits<T>::ld(recon + (((b * G + gy) * G + gx) * P * P + py * P + px) * 3 + c);
  }
}

// ---------------------------------------------------------------- softmax over rows (one wave per row)
// S: fp32 [rows][ld] (first n valid), P: T [rows][ld]; padded columns n..ld-1 of P are written as 0.
#define SM_MAXC 8  // n <= 512
template <typename T>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* __restrict__ S, T* __restrict__ Pm, long rows, int n, int ld) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float v[SM_MAXC];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < SM_MAXC; ++i) {
    const int c = i * 64 + lane;
    v[i] = c < n ? S[row * ld + c] : -INFINITY;
    mx = fmaxf(mx, v[i]);
  }
  mx = wave_max(mx);
  float se = 0.f;
#pragma unroll
  for (int i = 0; i < SM_MAXC; ++i) {
    v[i] = (i * 64 + lane) < n ? __expf(v[i] - mx) : 0.f;
    se += v[i];
  }
  se = 1.f / wave_sum(se);
#pragma unroll
  for (int i = 0; i < SM_MAXC; ++i) {
    const int c = i * 64 + lane;
    if (c < ld) msc_traits<T>::st(Pm + row * ld + c, v[i] * se);
  }
}
// dS = P * (dP - sum_j dP_j P_j)   (dP fp32, P T, dS T; padded columns -> 0)
template <typename T>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ dP, const T* __restrict__ Pm, T* __restrict__ dS,
                                                          long rows, int n, int ld) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float p[SM_MAXC], g[SM_MAXC];
  float dot = 0.f;
#pragma unroll
  for (int i = 0; i < SM_MAXC; ++i) {
    const int c = i * 64 + lane;
    p[i] = c < n ? msc_traits<T>::ld(Pm + row * ld + c) : 0.f;
    g[i] = c < n ? dP[row * ld + c] : 0.f;
    dot += p[i] * g[i];
  }
  dot = wave_sum(dot);
#pragma unroll
  for (int i = 0; i < SM_MAXC; ++i) {
    const int c = i * 64 + lane;
    if (c < ld) msc_traits<T>::st(dS + row * ld + c, p[i] * (g[i] - dot));
  }
}

// ================================================================ C-ABI
extern "C" int msc_argmax_rows(const void* x, int64_t* out, int64_t rows, int n, int64_t ld, int dtype, void* stream) {
  MSC_CHECK_ARG(x && out && rows > 0 && n > 0 && ld >= n);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MSC_F32)
    hipLaunchKernelGGL(argmax_rows_kernel<float>, dim3(cdiv(rows, 64)), dim3(64), 0, st, (const float*)x, out, (long)rows, n, (long)ld);
  else if (dtype == MSC_BF16)
    hipLaunchKernelGGL(argmax_rows_kernel<bf16_t>, dim3(cdiv(rows, 64)), dim3(64), 0, st, (const bf16_t*)x, out, (long)rows, n, (long)ld);
  else
    return MSC_ERR_BAD_ARG;
  return msc_launch_status();
}

extern "C" int msc_topk_rows(const void* x, int64_t* out_idx, float* out_val, int64_t rows, int n, int k, int64_t ld, int dtype,
                             void* stream) {
  MSC_CHECK_ARG(x && out_idx && rows > 0 && n > 0 && n <= TOPK_MAX_N && k > 0 && k <= TOPK_MAX_K && k <= n && ld >= n);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MSC_F32)
    hipLaunchKernelGGL(topk_rows_kernel<float>, dim3(cdiv(rows, 64)), dim3(64), 0, st, (const float*)x, out_idx, out_val, (long)rows, n, k, (long)ld);
  else if (dtype == MSC_BF16)
    hipLaunchKernelGGL(topk_rows_kernel<bf16_t>, dim3(cdiv(rows, 64)), dim3(64), 0, st, (const bf16_t*)x, out_idx, out_val, (long)rows, n, k, (long)ld);
  else
    return MSC_ERR_BAD_ARG;
  return msc_launch_status();
}

extern "C" int msc_ppo_loss(const void* head, const int64_t* actions, const float* old_logp, const float* adv, const float* ret,
                            float* terms, void* dhead, int B, int A, int ld, float clip_eps, float c_v, float c_e, int dtype, void* stream) {
  MSC_CHECK_ARG(head && actions && old_logp && adv && ret && terms && dhead && B > 0 && A > 0 && A <= PPO_MAX_A && ld > A);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MSC_F32)
    hipLaunchKernelGGL(ppo_loss_kernel<float>, dim3(1), dim3(256), 0, st, (const float*)head, actions, old_logp, adv, ret, terms, (float*)dhead, B, A, ld, clip_eps, c_v, c_e);
  else if (dtype == MSC_BF16)
    hipLaunchKernelGGL(ppo_loss_kernel<bf16_t>, dim3(1), dim3(256), 0, st, (const bf16_t*)head, actions, old_logp, adv, ret, terms, (bf16_t*)dhead, B, A, ld, clip_eps, c_v, c_e);
  else
    return MSC_ERR_BAD_ARG;
  return msc_launch_status();
}

extern "C" int64_t msc_recon_loss_workspace_bytes(int B) { return (int64_t)B * RL_CHUNKS * 4; }

extern "C" int msc_recon_loss(const void* recon, const float* img_nchw, float* sse, void* drecon, float* ws, int B, int G, int P,
                              float gscale, int dtype, void* stream) {
  MSC_CHECK_ARG(recon && img_nchw && sse && ws && B > 0 && G > 0 && P > 0);
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(RL_CHUNKS, (unsigned)B);
  // the 8-pixel vector path does 16-byte loads / stores of all three tensors: a caller's image view at a 4-byte storage offset (eager
  // mode hands the user's tensor over as it is) takes the element-wise loop instead
  const int vec_ok = P % 8 == 0 && (((uintptr_t)recon | (uintptr_t)img_nchw | (uintptr_t)drecon) & 15) == 0;
  if (dtype == MSC_F32)
    hipLaunchKernelGGL(recon_loss_kernel<float>, grid, dim3(256), 0, st, (const float*)recon, img_nchw, ws, (float*)drecon, G, P, gscale, vec_ok);
  else if (dtype == MSC_BF16)
    hipLaunchKernelGGL(recon_loss_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)recon, img_nchw, ws, (bf16_t*)drecon, G, P, gscale, vec_ok);
  else
    return MSC_ERR_BAD_ARG;
  hipLaunchKernelGGL(recon_loss_finalize_kernel, dim3(cdiv(B, 64)), dim3(64), 0, st, ws, sse, B);
  return msc_launch_status();
}

extern "C" int msc_unpatchify(const void* recon, float* img_nchw, int B, int G, int P, int dtype, void* stream) {
  MSC_CHECK_ARG(recon && img_nchw && B > 0 && G > 0 && P > 0);
  hipStream_t st = (hipStream_t)stream;
  const long n = (long)B * 3 * G * P * G * P;
  long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (dtype == MSC_F32)
    hipLaunchKernelGGL(unpatchify_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)recon, img_nchw, B, G, P);
  else if (dtype == MSC_BF16)
    hipLaunchKernelGGL(unpatchify_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, st, (const bf16_t*)recon, img_nchw, B, G, P);
  else
    return MSC_ERR_BAD_ARG;
  return msc_launch_status();
}

extern "C" int msc_softmax_fwd(const float* S, void* P, int64_t rows, int n, int ld, int dtype, void* stream) {
  MSC_CHECK_ARG(S && P && rows > 0 && n > 0 && ld >= n && ld <= 64 * SM_MAXC);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MSC_F32)
    hipLaunchKernelGGL(softmax_fwd_kernel<float>, dim3(cdiv(rows, 4)), dim3(256), 0, st, S, (float*)P, (long)rows, n, ld);
  else if (dtype == MSC_BF16)
    hipLaunchKernelGGL(softmax_fwd_kernel<bf16_t>, dim3(cdiv(rows, 4)), dim3(256), 0, st, S, (bf16_t*)P, (long)rows, n, ld);
  else
    return MSC_ERR_BAD_ARG;
  return msc_launch_status();
}

extern "C" int msc_softmax_bwd(const float* dP, const void* P, void* dS, int64_t rows, int n, int ld, int dtype, void* stream) {
  MSC_CHECK_ARG(dP && P && dS && rows > 0 && n > 0 && ld >= n && ld <= 64 * SM_MAXC);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MSC_F32)
    hipLaunchKernelGGL(softmax_bwd_kernel<float>, dim3(cdiv(rows, 4)), dim3(256), 0, st, dP, (const float*)P, (float*)dS, (long)rows, n, ld);
  else if (dtype == MSC_BF16)
    hipLaunchKernelGGL(softmax_bwd_kernel<bf16_t>, dim3(cdiv(rows, 4)), dim3(256), 0, st, dP, (const bf16_t*)P, (bf16_t*)dS, (long)rows, n, ld);
  else
    return MSC_ERR_BAD_ARG;
  return msc_launch_status();
}
