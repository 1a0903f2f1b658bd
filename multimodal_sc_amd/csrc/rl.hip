// rl.hip — the PPO loop's device-side pieces around the hot path: action sampling from the
// policy head, generalised advantage estimation over a rollout, advantage normalisation.  All tiny and latency-bound:
// one thread per environment / per row, no atomics, fixed evaluation order (bit-reproducible).
// fp contraction is OFF in this file so that the fp32 recurrences match the oracle's (NumPy/PyTorch-CPU, unfused) bit
// for bit.  Build-authored (the upstream reference has no code); oracle: oracle/rl.py.
#include "msc_common.h"
#pragma clang fp contract(off)

// adv[t][n] = delta_t + gamma*lam*(1-done_t)*adv[t+1][n],  delta_t = r_t + gamma*(1-done_t)*V[t+1][n] - V[t][n]
// ret = adv + V[t].   rewards/dones [T][N], values [T+1][N] (last row = bootstrap value).
__global__ void gae_kernel(const float* __restrict__ rew, const float* __restrict__ val, const float* __restrict__ done,
                           float* __restrict__ adv, float* __restrict__ ret, int T, int N, float gamma, float lam) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float a = 0.f;
  for (int t = T - 1; t >= 0; --t) {
    const float nd = 1.f - done[(long)t * N + n];
    const float v = val[(long)t * N + n];
    const float delta = rew[(long)t * N + n] + gamma * val[(long)(t + 1) * N + n] * nd - v;
    a = delta + gamma * lam * nd * a;
    adv[(long)t * N + n] = a;
    ret[(long)t * N + n] = a + v;
  }
}

extern "C" int msc_gae(const float* rewards, const float* values, const float* dones, float* adv, float* ret, int T, int N,
                       float gamma, float lam, void* stream) {
  MSC_CHECK_ARG(rewards && values && dones && adv && ret && T > 0 && N > 0);
  hipLaunchKernelGGL(gae_kernel, dim3(cdiv(N, 64)), dim3(64), 0, (hipStream_t)stream, rewards, values, dones, adv, ret, T, N, gamma, lam);
  return msc_launch_status();
}

// x <- (x - mean) / (std + eps), unbiased std (n-1), one workgroup, fixed-order tree (n <= 2^22: a rollout)
__global__ __launch_bounds__(1024) void normalize_kernel(float* __restrict__ x, long n, float eps) {
  __shared__ float red[1024];
  __shared__ float s_mean, s_inv;
  const int tid = threadIdx.x;
  float s = 0.f;
  for (long i = tid; i < n; i += 1024) s += x[i];
  red[tid] = s;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  if (tid == 0) s_mean = red[0] / (float)n;
  __syncthreads();
  const float mean = s_mean;
  float q = 0.f;
  for (long i = tid; i < n; i += 1024) { const float d = x[i] - mean; q += d * d; }
  red[tid] = q;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  if (tid == 0) s_inv = 1.f / (sqrtf(red[0] / (float)(n > 1 ? n - 1 : 1)) + eps);
  __syncthreads();
  const float inv = s_inv;
  for (long i = tid; i < n; i += 1024) x[i] = (x[i] - mean) * inv;
}

extern "C" int msc_normalize(float* x, int64_t n, float eps, void* stream) {
  MSC_CHECK_ARG(x && n > 0 && n <= (1L << 22));
  hipLaunchKernelGGL(normalize_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, (long)n, eps);
  return msc_launch_status();
}

// Categorical sampling from the policy head by inverse CDF: u = (philox(seed; row, step).x + 0.5) * 2^-32,
// action = first a with sum_{j<=a} p_j > u (last action if rounding leaves the total below u).
// Also emits log pi(a) and the value estimate.  head [B][ld]: logits 0..A-1, value at column A.
template <typename T>
__global__ void sample_action_kernel(const T* __restrict__ head, int64_t* __restrict__ action, float* __restrict__ logp,
                                     float* __restrict__ value, int B, int A, int ld, uint64_t seed, const int64_t* seed_dev,
                                     uint64_t step) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  if (seed_dev) seed += (uint64_t)seed_dev[0];
  const T* row = head + (long)b * ld;
  float mx = -INFINITY;
  for (int a = 0; a < A; ++a) mx = fmaxf(mx, msc_traits<T>::ld(row + a));
  float z = 0.f;
  for (int a = 0; a < A; ++a) z += expf(msc_traits<T>::ld(row + a) - mx);
  uint32_t r[4];
  philox4x32_10((uint32_t)b, (uint32_t)step, (uint32_t)(step >> 32), 2u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
  const float u = ((float)r[0] + 0.5f) * 2.3283064365386963e-10f;
  float c = 0.f;
  int pick = A - 1;
  for (int a = 0; a < A; ++a) {
    c += expf(msc_traits<T>::ld(row + a) - mx) / z;
    if (c > u) { pick = a; break; }
  }
  action[b] = pick;
  logp[b] = (msc_traits<T>::ld(row + pick) - mx) - logf(z);
  value[b] = msc_traits<T>::ld(row + A);
}

extern "C" int msc_sample_actions(const void* head, int64_t* action, float* logp, float* value, int B, int A, int ld,
                                  uint64_t seed, const int64_t* seed_dev, uint64_t step, int dtype, void* stream) {
  MSC_CHECK_ARG(head && action && logp && value && B > 0 && A > 0 && ld > A);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MSC_F32)
    hipLaunchKernelGGL(sample_action_kernel<float>, dim3(cdiv(B, 64)), dim3(64), 0, st, (const float*)head, action, logp, value, B, A, ld, seed, seed_dev, step);
  else if (dtype == MSC_BF16)
    hipLaunchKernelGGL(sample_action_kernel<bf16_t>, dim3(cdiv(B, 64)), dim3(64), 0, st, (const bf16_t*)head, action, logp, value, B, A, ld, seed, seed_dev, step);
  else
    return MSC_ERR_BAD_ARG;
  return msc_launch_status();
}
