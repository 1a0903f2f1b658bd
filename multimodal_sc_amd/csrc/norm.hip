// norm.hip — BatchNorm (training statistics, NHWC) and LayerNorm, forward and backward, plus the column-sum used
// for bias / positional-embedding gradients.
// All HBM-bound: 16-byte vector accesses, fp32 statistics, fixed-order (bitwise reproducible) two-stage column
// reductions (per-block partials -> parallel finalize), no float atomics.
// Roles: ImageEncoder / BEV conv stack (BN), FusionTransformer (LN).  Build-authored; the upstream reference has no
// code (SURVEY.md finding zero).
#include "msc_common.h"
#include "gemm_plan.h"     // mscplan::g_knobs (tuning knobs, msc_set_knob)

#define BN_MAX_BLOCKS 1024
#ifndef LN_UR
#define LN_UR 1      // rows per wave in flight in the LayerNorm backward (2: no gain inside the step)
#endif

// ------------------------------------------------------------------------------------------------
// Finalize of every two-stat column reduction: partial [nblk][2][C] -> sums over blocks.
// block = 8 channels x 32 row-lanes; lane l sums blocks l, l+32, ... in fp64, then a fixed-order LDS tree.
enum { FIN_BN_FWD = 0, FIN_SUMS = 1 };
struct FinArgs {
  const float* partial; int nblk; int C;
  const float* partial2; int nblk2;      // optional second partial buffer (FIN_SUMS: the other half-batch chain's rows), summed after the first
  // FIN_BN_FWD
  double inv_count; float eps; const float* gamma; const float* beta; float* mean; float* rstd; float* scale; float* shift;
  // FIN_SUMS
  float* out0; float* out1; float* acc0; float* acc1; int accumulate;
};

// CPB channels per block x RL = 256/CPB row lanes.  These kernels sit on the critical chain between every convolution and
// its BatchNorm apply (and between the two BatchNorm-backward passes): a few-microsecond, latency-bound reduction, so
// the host picks CPB = 2 (128 row lanes, 4x more blocks) when there are many partial rows — the per-thread chain of
// dependent load rounds is what the kernel's time consists of.
template <int MODE, int CPB>
__global__ __launch_bounds__(256) void finalize2_kernel(const FinArgs a) {
  constexpr int RL = 256 / CPB;
  __shared__ double red[2][RL][CPB];
  const int cl = threadIdx.x % CPB, rl = threadIdx.x / CPB;
  const int c = blockIdx.x * CPB + cl;
  double s1 = 0.0, s2 = 0.0;
  if (c < a.C) {
    // 4 independent load pairs per trip: the serial part is the fp64 add chain, not a chain of memory round trips
    int b = rl;
    for (; b + 3 * RL < a.nblk; b += 4 * RL) {
      const float x0 = a.partial[(long)b * 2 * a.C + c], y0 = a.partial[(long)b * 2 * a.C + a.C + c];
      const float x1 = a.partial[(long)(b + RL) * 2 * a.C + c], y1 = a.partial[(long)(b + RL) * 2 * a.C + a.C + c];
      const float x2 = a.partial[(long)(b + 2 * RL) * 2 * a.C + c], y2 = a.partial[(long)(b + 2 * RL) * 2 * a.C + a.C + c];
      const float x3 = a.partial[(long)(b + 3 * RL) * 2 * a.C + c], y3 = a.partial[(long)(b + 3 * RL) * 2 * a.C + a.C + c];
      s1 += ((double)x0 + (double)x1) + ((double)x2 + (double)x3);
      s2 += ((double)y0 + (double)y1) + ((double)y2 + (double)y3);
    }
    for (; b < a.nblk; b += RL) {
      s1 += (double)a.partial[(long)b * 2 * a.C + c];
      s2 += (double)a.partial[(long)b * 2 * a.C + a.C + c];
    }
    for (b = rl; b < a.nblk2; b += RL) {
      s1 += (double)a.partial2[(long)b * 2 * a.C + c];
      s2 += (double)a.partial2[(long)b * 2 * a.C + a.C + c];
    }
  }
  red[0][rl][cl] = s1;
  red[1][rl][cl] = s2;
  __syncthreads();
  // fixed-order tree over the row lanes (bitwise reproducible), then lane 0 of each channel finishes
  for (int h = RL / 2; h >= 1; h >>= 1) {
    if (rl < h) { red[0][rl][cl] += red[0][rl + h][cl]; red[1][rl][cl] += red[1][rl + h][cl]; }
    __syncthreads();
  }
  if (rl != 0 || c >= a.C) return;
  s1 = red[0][0][cl]; s2 = red[1][0][cl];
  if constexpr (MODE == FIN_BN_FWD) {
    // E[x^2]-E[x]^2 formed in double so the cancellation does not cost fp32 digits
    const double m = s1 * a.inv_count;
    double var = s2 * a.inv_count - m * m;
    if (var < 0.0) var = 0.0;
    const float rs = (float)(1.0 / sqrt(var + (double)a.eps));
    const float mu = (float)m;
    a.mean[c] = mu;
    a.rstd[c] = rs;
    const float sc = a.gamma[c] * rs;
    a.scale[c] = sc;
    a.shift[c] = a.beta[c] - mu * sc;
  } else {
    const float f1 = (float)s1, f2 = (float)s2;
    if (a.out0) a.out0[c] = f1;
    if (a.out1) a.out1[c] = f2;
    if (a.acc0) a.acc0[c] = (a.accumulate ? a.acc0[c] : 0.f) + f1;
    if (a.acc1) a.acc1[c] = (a.accumulate ? a.acc1[c] : 0.f) + f2;
  }
}

// load VEC consecutive per-channel floats
template <int VEC> __device__ __forceinline__ void ldc(const float* __restrict__ p, float (&o)[VEC]) {
#pragma unroll
  for (int j = 0; j < VEC; j += 4) {
    const float4 t = *reinterpret_cast<const float4*>(p + j);
    o[j] = t.x; o[j + 1] = t.y; o[j + 2] = t.z; o[j + 3] = t.w;
  }
}

// y = relu?( x*scale[c] + shift[c] + res? ).  The grid stride (gridDim * 256 vectors) is a multiple of C / VEC (a power of two <= 64),
// so a thread meets the same 16-byte channel group in every iteration: the per-channel constants are loaded once, and UN vectors per
// operand are in flight per thread (the kernel is a pure stream: the more bytes in flight per CU, the closer to the HBM rate).
template <typename T, int UN>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                                                       const T* __restrict__ res, T* __restrict__ y, long nvec, int C, int relu) {
  MSC_BOOST();
  constexpr int VEC = msc_traits<T>::VEC;
  const unsigned ncv = (unsigned)C / VEC;
  const long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long)gridDim.x * blockDim.x;
  const int c0 = (int)((unsigned)(i0 % ncv) * VEC);
  float sc[VEC], sh[VEC];
  ldc<VEC>(scale + c0, sc);
  ldc<VEC>(shift + c0, sh);
  for (long i = i0; i < nvec; i += stride * UN) {
    vecf<T> vx[UN], vr[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const long e = (i + u * stride) * VEC;
      if (i + u * stride < nvec) {
        vx[u].load(x + e);
        if (res) vr[u].load(res + e);
      }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      if (i + u * stride >= nvec) break;
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        float v = fmaf(vx[u].v[j], sc[j], sh[j]);
        if (res) v += vr[u].v[j];
        if (relu) v = v > 0.f ? v : 0.f;
        vx[u].v[j] = v;
      }
      vx[u].store(y + (i + u * stride) * VEC);
    }
  }
}

// the same for channel counts whose 16-byte groups do not divide 256 (a thread then meets another group every iteration)
template <typename T>
__global__ void bn_apply_any_kernel(const T* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                                const T* __restrict__ res, T* __restrict__ y, long nvec, int C, int relu) {
  constexpr int VEC = msc_traits<T>::VEC;
  // channel of the vector, kept incrementally: a 64-bit modulo per 16-byte vector costs more than the rest of the body
  const unsigned ncv = (unsigned)C / VEC;
  const long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long)gridDim.x * blockDim.x;
  unsigned cvi = (unsigned)(i0 % ncv);
  const unsigned cstep = (unsigned)(stride % ncv);
  for (long i = i0; i < nvec; i += stride, cvi = cvi + cstep >= ncv ? cvi + cstep - ncv : cvi + cstep) {
    const long e = i * VEC;
    const int c0 = (int)(cvi * VEC);
    vecf<T> vx, vr;
    vx.load(x + e);
    if (res) vr.load(res + e);
    float sc[VEC], sh[VEC];
    ldc<VEC>(scale + c0, sc);
    ldc<VEC>(shift + c0, sh);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float v = fmaf(vx.v[j], sc[j], sh[j]);
      if (res) v += vr.v[j];
      if (relu) v = v > 0.f ? v : 0.f;
      vx.v[j] = v;
    }
    vx.store(y + e);
  }
}

// Stage 1 of BN backward: per-block partial sums of dz and dz*xhat over a slab of rows.
// Threads: tx = channel chunk (C/VEC of them), ty = row lane.  Four rows per thread are in flight at a time (the loop is
// latency-bound otherwise: one 16-B load per operand per iteration left the kernel at ~2 TB/s).
// scale != nullptr: the ReLU mask is rebuilt as x*scale+shift > 0 (the forward's folded affine; valid when the forward had no
// residual input) and y is not read.
template <typename T, int UN>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                            const T* __restrict__ y, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, float* __restrict__ partial,
                                                            long rows, int C, int relu, int rows_per_block) {
  MSC_BOOST();
  constexpr int VEC = msc_traits<T>::VEC;
  const int ncx = C / VEC;          // threads across channels
  const int nty = 256 / ncx;        // rows handled concurrently
  const int tx = threadIdx.x % ncx, ty = threadIdx.x / ncx;
  const long r0 = (long)blockIdx.x * rows_per_block;
  const long r1 = min(rows, r0 + rows_per_block);
  const bool xmask = relu && scale != nullptr;
  const bool ymask = relu && scale == nullptr;
  float s1[VEC], s2[VEC], mu[VEC], rs[VEC], sc[VEC], sh[VEC];
  ldc<VEC>(mean + tx * VEC, mu);
  ldc<VEC>(rstd + tx * VEC, rs);
#pragma unroll
  for (int j = 0; j < VEC; ++j) { s1[j] = 0.f; s2[j] = 0.f; sc[j] = 0.f; sh[j] = 0.f; }
  if (xmask) { ldc<VEC>(scale + tx * VEC, sc); ldc<VEC>(shift + tx * VEC, sh); }
  if (ty < nty) {
    long r = r0 + ty;
    for (; r + (long)(UN - 1) * nty < r1; r += (long)UN * nty) {
      vecf<T> vd[UN], vx[UN], vy[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const long e = (r + (long)u * nty) * C + tx * VEC;
        vd[u].load(dy + e);
        vx[u].load(x + e);
        if (ymask) vy[u].load(y + e);
      }
#pragma unroll
      for (int u = 0; u < UN; ++u)
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          float dz = vd[u].v[j];
          if (ymask && !(vy[u].v[j] > 0.f)) dz = 0.f;
          if (xmask && !(fmaf(vx[u].v[j], sc[j], sh[j]) > 0.f)) dz = 0.f;
          s1[j] += dz;
          s2[j] += dz * (vx[u].v[j] - mu[j]) * rs[j];
        }
    }
    for (; r < r1; r += nty) {
      const long e = r * C + tx * VEC;
      vecf<T> vd, vx, vy;
      vd.load(dy + e);
      vx.load(x + e);
      if (ymask) vy.load(y + e);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        float dz = vd.v[j];
        if (ymask && !(vy.v[j] > 0.f)) dz = 0.f;
        if (xmask && !(fmaf(vx.v[j], sc[j], sh[j]) > 0.f)) dz = 0.f;
        s1[j] += dz;
        s2[j] += dz * (vx.v[j] - mu[j]) * rs[j];
      }
    }
  }
  __shared__ float red[2][2048];  // [stat][ty][C] with nty*C <= 256*VEC <= 2048
  float* r1p = &red[0][0];
  float* r2p = &red[1][0];
  if (ty < nty) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) { r1p[ty * C + tx * VEC + j] = s1[j]; r2p[ty * C + tx * VEC + j] = s2[j]; }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float a = 0.f, b = 0.f;
    for (int t = 0; t < nty; ++t) { a += r1p[t * C + c]; b += r2p[t * C + c]; }
    partial[(long)blockIdx.x * 2 * C + c] = a;
    partial[(long)blockIdx.x * 2 * C + C + c] = b;
  }
}

// Stage 3: dx = gamma*rstd*(dz - dbeta/M - xhat*dgamma/M); dres = dz (optional).  scale != nullptr: mask from x (see stage 1).
// As bn_apply_kernel: the thread's channel group is fixed, its constants (folded into three per channel: dx = a*dz + b*x + c) are
// computed once, UN vectors per operand are in flight.
template <typename T, int UN>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ y,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, const float* __restrict__ sum_dz,
                                                           const float* __restrict__ sum_dzx, T* __restrict__ dx, T* __restrict__ dres,
                                                           long nvec, int C, int relu, float inv_rows) {
  MSC_BOOST();
  constexpr int VEC = msc_traits<T>::VEC;
  const bool xmask = relu && scale != nullptr;
  const bool ymask = relu && scale == nullptr;
  const unsigned ncv = (unsigned)C / VEC;
  const long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long)gridDim.x * blockDim.x;
  const int c0 = (int)((unsigned)(i0 % ncv) * VEC);
  float mu[VEC], rs[VEC], ga[VEC], sa[VEC], sb[VEC], sc[VEC], sh[VEC];
  ldc<VEC>(mean + c0, mu);
  ldc<VEC>(rstd + c0, rs);
  ldc<VEC>(gamma + c0, ga);
  ldc<VEC>(sum_dz + c0, sa);
  ldc<VEC>(sum_dzx + c0, sb);
#pragma unroll
  for (int j = 0; j < VEC; ++j) { sc[j] = 0.f; sh[j] = 0.f; }
  if (xmask) { ldc<VEC>(scale + c0, sc); ldc<VEC>(shift + c0, sh); }
  for (long i = i0; i < nvec; i += stride * UN) {
    vecf<T> vd[UN], vx[UN], vy[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const long e = (i + u * stride) * VEC;
      if (i + u * stride < nvec) {
        vd[u].load(dy + e);
        vx[u].load(x + e);
        if (ymask) vy[u].load(y + e);
      }
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      if (i + u * stride >= nvec) break;
      const long e = (i + u * stride) * VEC;
      vecf<T> vo;
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        float dz = vd[u].v[j];
        if (ymask && !(vy[u].v[j] > 0.f)) dz = 0.f;
        if (xmask && !(fmaf(vx[u].v[j], sc[j], sh[j]) > 0.f)) dz = 0.f;
        const float xh = (vx[u].v[j] - mu[j]) * rs[j];
        vo.v[j] = ga[j] * rs[j] * (dz - sa[j] * inv_rows - xh * sb[j] * inv_rows);
        vd[u].v[j] = dz;
      }
      vo.store(dx + e);
      if (dres) vd[u].store(dres + e);
    }
  }
}

// stage 3 for channel counts whose 16-byte groups do not divide 256
template <typename T>
__global__ void bn_bwd_apply_any_kernel(const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ y,
                                    const float* __restrict__ mean, const float* __restrict__ rstd,
                                    const float* __restrict__ gamma, const float* __restrict__ scale,
                                    const float* __restrict__ shift, const float* __restrict__ sum_dz,
                                    const float* __restrict__ sum_dzx, T* __restrict__ dx, T* __restrict__ dres,
                                    long nvec, int C, int relu, float inv_rows) {
  constexpr int VEC = msc_traits<T>::VEC;
  const bool xmask = relu && scale != nullptr;
  const bool ymask = relu && scale == nullptr;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
    const long e = i * VEC;
    const int c0 = (int)(e % C);
    vecf<T> vd, vx, vy, vo;
    vd.load(dy + e);
    vx.load(x + e);
    if (ymask) vy.load(y + e);
    float mu[VEC], rs[VEC], ga[VEC], sa[VEC], sb[VEC], sc[VEC], sh[VEC];
    ldc<VEC>(mean + c0, mu);
    ldc<VEC>(rstd + c0, rs);
    ldc<VEC>(gamma + c0, ga);
    ldc<VEC>(sum_dz + c0, sa);
    ldc<VEC>(sum_dzx + c0, sb);
    if (xmask) { ldc<VEC>(scale + c0, sc); ldc<VEC>(shift + c0, sh); }
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float dz = vd.v[j];
      if (ymask && !(vy.v[j] > 0.f)) dz = 0.f;
      if (xmask && !(fmaf(vx.v[j], sc[j], sh[j]) > 0.f)) dz = 0.f;
      const float xh = (vx.v[j] - mu[j]) * rs[j];
      vo.v[j] = ga[j] * rs[j] * (dz - sa[j] * inv_rows - xh * sb[j] * inv_rows);
      vd.v[j] = dz;
    }
    vo.store(dx + e);
    if (dres) vd.store(dres + e);
  }
}

// ---------------------------------------------------------------- stem: BatchNorm + ReLU + max-pool fused
// ImageEncoder stem (conv 7x7/2 -> BN -> ReLU -> max-pool 3x3/2): the post-activation map (103 MB at 64 frames) is never
// written.  Forward: one pass reads the conv output, normalises, clamps, rounds to the storage type (so values and tie-breaks
// equal the unfused kernels bit for bit) and writes the pooled map + the arg-max tap ids.  Backward: the BatchNorm-backward
// passes GATHER their incoming gradient from the pooled gradient through the tap ids (what maxpool_bwd would have written out
// and they would have read back) and rebuild the ReLU mask from x*scale+shift > 0 — 412 MB less traffic and two launches
// fewer per step.
struct PoolGeo { int H, W, Ho, Wo, K, S, P; };

template <typename T> __device__ __forceinline__ float round_store(float v) { return v; }
template <> __device__ __forceinline__ float round_store<bf16_t>(float v) { return bf16_to_f32(f32_to_bf16(v)); }

template <typename T>
__global__ void bn_relu_pool_fwd_kernel(const T* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ shift,
                                        T* __restrict__ y, uint8_t* __restrict__ idx, T* __restrict__ xsel, int B, int C, PoolGeo g) {
  constexpr int VEC = msc_traits<T>::VEC;
  const int cv = C / VEC;
  const long n = (long)B * g.Ho * g.Wo * cv;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const unsigned iu = (unsigned)i;   // n < 2^31 (launcher)
    const int c0 = (int)(iu % (unsigned)cv) * VEC;
    unsigned t = iu / (unsigned)cv;
    const int wo = (int)(t % (unsigned)g.Wo); t /= (unsigned)g.Wo;
    const int ho = (int)(t % (unsigned)g.Ho);
    const int b = (int)(t / (unsigned)g.Ho);
    float sc[VEC], sh[VEC], best[VEC], bx[VEC];
    int bi[VEC];
    ldc<VEC>(scale + c0, sc);
    ldc<VEC>(shift + c0, sh);
#pragma unroll
    for (int j = 0; j < VEC; ++j) { best[j] = -INFINITY; bi[j] = 0; bx[j] = 0.f; }
    for (int kh = 0; kh < g.K; ++kh) {
      const int hi = ho * g.S - g.P + kh;
      if ((unsigned)hi >= (unsigned)g.H) continue;
      for (int kw = 0; kw < g.K; ++kw) {
        const int wi = wo * g.S - g.P + kw;
        if ((unsigned)wi >= (unsigned)g.W) continue;
        vecf<T> v;
        v.load(x + (((long)b * g.H + hi) * g.W + wi) * C + c0);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          float a = fmaf(v.v[j], sc[j], sh[j]);
          a = round_store<T>(a > 0.f ? a : 0.f);                 // the value bn_apply would have stored
          if (a > best[j]) { best[j] = a; bi[j] = kh * g.K + kw; bx[j] = v.v[j]; }
        }
      }
    }
    vecf<T> o, ox;
#pragma unroll
    for (int j = 0; j < VEC; ++j) { o.v[j] = best[j]; ox.v[j] = bx[j]; }
    const long oo = (((long)b * g.Ho + ho) * g.Wo + wo) * C + c0;
    o.store(y + oo);
    ox.store(xsel + oo);       // the conv output at the arg-max: the backward's statistics pass runs on the pooled size
#pragma unroll
    for (int j = 0; j < VEC; ++j) idx[oo + j] = (uint8_t)bi[j];
  }
}

// gradient w.r.t. the pre-pool pixel (b,h,w), channels c0..c0+VEC-1: sum of the pooled gradients of the windows whose arg-max it is
template <typename T>
__device__ __forceinline__ void pool_gather(const T* __restrict__ dpool, const uint8_t* __restrict__ idx, int b, int h, int w, int C, int c0,
                                            const PoolGeo& g, float (&acc)[msc_traits<T>::VEC]) {
  constexpr int VEC = msc_traits<T>::VEC;
#pragma unroll
  for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
  // windows (ho, kh) that contain row h: h = ho*S - P + kh  ->  kh = (h+P) mod S, +S, ...  (at most ceil(K/S) per axis)
  const int hq = h + g.P, wq = w + g.P;
  for (int kh = hq % g.S, ho = hq / g.S; kh < g.K && ho >= 0; kh += g.S, --ho) {
    if (ho >= g.Ho) continue;
    for (int kw = wq % g.S, wo = wq / g.S; kw < g.K && wo >= 0; kw += g.S, --wo) {
      if (wo >= g.Wo) continue;
      const long oo = (((long)b * g.Ho + ho) * g.Wo + wo) * C + c0;
      vecf<T> gr;
      gr.load(dpool + oo);
      uint64_t taps;
      if constexpr (VEC == 8) taps = *reinterpret_cast<const uint64_t*>(idx + oo);
      else taps = *reinterpret_cast<const uint32_t*>(idx + oo);
      const unsigned want = (unsigned)(kh * g.K + kw);
#pragma unroll
      for (int j = 0; j < VEC; ++j)
        if (((taps >> (8 * j)) & 0xffu) == want) acc[j] += gr.v[j];
    }
  }
}

// stage 3: dx = gamma*rstd*(dz - dbeta/M - xhat*dgamma/M)
template <typename T>
__global__ void bn_pool_bwd_apply_kernel(const T* __restrict__ dpool, const uint8_t* __restrict__ idx, const T* __restrict__ x,
                                         const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ gamma,
                                         const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ sum_dz,
                                         const float* __restrict__ sum_dzx, T* __restrict__ dx, long nvec, int C, float inv_rows, PoolGeo g) {
  constexpr int VEC = msc_traits<T>::VEC;
  const unsigned cv = (unsigned)(C / VEC);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
    const unsigned iu = (unsigned)i;   // nvec < 2^31 (launcher)
    const int c0 = (int)(iu % cv) * VEC;
    unsigned t = iu / cv;
    const int w = (int)(t % (unsigned)g.W); t /= (unsigned)g.W;
    const int h = (int)(t % (unsigned)g.H), b = (int)(t / (unsigned)g.H);
    float dz[VEC];
    pool_gather<T>(dpool, idx, b, h, w, C, c0, g, dz);
    vecf<T> vx, vo;
    vx.load(x + i * VEC);
    float mu[VEC], rs[VEC], ga[VEC], sa[VEC], sb[VEC], sc[VEC], sh[VEC];
    ldc<VEC>(mean + c0, mu);
    ldc<VEC>(rstd + c0, rs);
    ldc<VEC>(gamma + c0, ga);
    ldc<VEC>(sum_dz + c0, sa);
    ldc<VEC>(sum_dzx + c0, sb);
    ldc<VEC>(scale + c0, sc);
    ldc<VEC>(shift + c0, sh);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const float d = fmaf(vx.v[j], sc[j], sh[j]) > 0.f ? dz[j] : 0.f;
      const float xh = (vx.v[j] - mu[j]) * rs[j];
      vo.v[j] = ga[j] * rs[j] * (d - sa[j] * inv_rows - xh * sb[j] * inv_rows);
    }
    vo.store(dx + i * VEC);
  }
}

// ---------------------------------------------------------------- LayerNorm
// one wave per row, the row lives in registers (D <= 64*LN_MAXC); D % (64*VEC) == 0
#define LN_MAXC 16
// LN_FWD_ROWS consecutive rows per wave, all their loads issued before the first reduction: one row per wave left 16 bytes per lane in
// flight (tools/ln_bench.py, 19584 x 512 bf16, cache-warm: 1 row 9.5 us, 2 rows 8.9, 4 rows 10.2, 8 rows 12.1)
#ifndef LN_FWD_ROWS
#define LN_FWD_ROWS 2
#endif
template <typename T>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, T* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd, long rows,
                                                     int D, float eps) {
  MSC_BOOST();
  constexpr int VEC = msc_traits<T>::VEC;
  constexpr int MAXIT = LN_MAXC / VEC;
  constexpr int R = LN_FWD_ROWS;
  const int lane = threadIdx.x & 63;
  const long row0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
  if (row0 >= rows) return;
  const int nit = D / (64 * VEC);
  if (nit == 1) {       // D = 64 * VEC (the path's 512 in bf16): the R rows' loads in flight together, gamma / beta once
    vecf<T> v[R];
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (row0 + r < rows) v[r].load(x + (row0 + r) * D + lane * VEC);
    float g[VEC], b[VEC];
    ldc<VEC>(gamma + lane * VEC, g);
    ldc<VEC>(beta + lane * VEC, b);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (row0 + r >= rows) break;
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < VEC; ++j) s += v[r].v[j];
      const float mu = wave_sum(s) / (float)D;
      float q = 0.f;
#pragma unroll
      for (int j = 0; j < VEC; ++j) { const float d = v[r].v[j] - mu; q += d * d; }
      const float rs = rsqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
      for (int j = 0; j < VEC; ++j) v[r].v[j] = (v[r].v[j] - mu) * rs * g[j] + b[j];
      v[r].store(y + (row0 + r) * D + lane * VEC);
      if (lane == 0) { mean[row0 + r] = mu; rstd[row0 + r] = rs; }
    }
    return;
  }
  for (int r = 0; r < R; ++r) {
    const long row = row0 + r;
    if (row >= rows) break;
    vecf<T> v[MAXIT];
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it)
      if (it < nit) {
        v[it].load(x + row * D + (it * 64 + lane) * VEC);
#pragma unroll
        for (int j = 0; j < VEC; ++j) s += v[it].v[j];
      }
    const float mu = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it)
      if (it < nit) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) { const float d = v[it].v[j] - mu; q += d * d; }
      }
    const float rs = rsqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
    for (int it = 0; it < MAXIT; ++it)
      if (it < nit) {
        const int c0 = (it * 64 + lane) * VEC;
        float g[VEC], b[VEC];
        ldc<VEC>(gamma + c0, g);
        ldc<VEC>(beta + c0, b);
#pragma unroll
        for (int j = 0; j < VEC; ++j) v[it].v[j] = (v[it].v[j] - mu) * rs * g[j] + b[j];
        v[it].store(y + row * D + c0);
      }
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
  }
}

// block = 4 waves, LN_ROWS_PER_BLOCK rows (each wave takes every 4th); each lane owns D/64 columns
#ifndef LN_ROWS_PER_BLOCK
#define LN_ROWS_PER_BLOCK 32
#endif
template <typename T>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                     const float* __restrict__ gamma, const T* __restrict__ dadd,
                                                     T* __restrict__ dx, float* __restrict__ partial, long rows, int D) {
  MSC_BOOST();
  constexpr int VEC = msc_traits<T>::VEC;
  constexpr int MAXIT = LN_MAXC / VEC;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nit = D / (64 * VEC);
  float dg[LN_MAXC], db[LN_MAXC], ga[LN_MAXC];
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) {
    float g[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) g[j] = 0.f;
    if (it < nit) ldc<VEC>(gamma + (it * 64 + lane) * VEC, g);
#pragma unroll
    for (int j = 0; j < VEC; ++j) { dg[it * VEC + j] = 0.f; db[it * VEC + j] = 0.f; ga[it * VEC + j] = g[j]; }
  }
  const long r0 = (long)blockIdx.x * LN_ROWS_PER_BLOCK;
  // two rows per wave in flight (rows rr and rr+4): the loads of the second row hide the wave-reduce + store of the first
  constexpr int UR = LN_UR;
  for (int rr = wave; rr < LN_ROWS_PER_BLOCK; rr += 4 * UR) {
    if (r0 + rr >= rows) break;
    vecf<T> vd[UR][MAXIT], vx[UR][MAXIT], va[UR][MAXIT];
    float mu[UR], rs[UR];
    bool ok[UR];
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      const long row = r0 + rr + 4 * u;
      ok[u] = rr + 4 * u < LN_ROWS_PER_BLOCK && row < rows;
      if (ok[u]) {
        mu[u] = mean[row];
        rs[u] = rstd[row];
#pragma unroll
        for (int it = 0; it < MAXIT; ++it)
          if (it < nit) {
            const int c0 = (it * 64 + lane) * VEC;
            vd[u][it].load(dy + row * D + c0);
            vx[u][it].load(x + row * D + c0);
            if (dadd) va[u][it].load(dadd + row * D + c0);
          }
      }
    }
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      if (!ok[u]) continue;
      const long row = r0 + rr + 4 * u;
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int it = 0; it < MAXIT; ++it)
        if (it < nit) {
#pragma unroll
          for (int j = 0; j < VEC; ++j) {
            const float xh = (vx[u][it].v[j] - mu[u]) * rs[u];
            const float g = vd[u][it].v[j] * ga[it * VEC + j];
            vx[u][it].v[j] = xh;
            s1 += g;
            s2 += g * xh;
            dg[it * VEC + j] += vd[u][it].v[j] * xh;
            db[it * VEC + j] += vd[u][it].v[j];
          }
        }
      s1 = wave_sum(s1) / (float)D;
      s2 = wave_sum(s2) / (float)D;
#pragma unroll
      for (int it = 0; it < MAXIT; ++it)
        if (it < nit) {
          const int c0 = (it * 64 + lane) * VEC;
#pragma unroll
          for (int j = 0; j < VEC; ++j) {
            float o = rs[u] * (vd[u][it].v[j] * ga[it * VEC + j] - s1 - vx[u][it].v[j] * s2);
            if (dadd) o += va[u][it].v[j];
            vx[u][it].v[j] = o;
          }
          vx[u][it].store(dx + row * D + c0);
        }
    }
  }
  // cross-wave reduce of dgamma/dbeta, then one partial row per block
  __shared__ float red[2][4][1024];
#pragma unroll
  for (int it = 0; it < MAXIT; ++it)
    if (it < nit) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const int c = (it * 64 + lane) * VEC + j;
        red[0][wave][c] = dg[it * VEC + j];
        red[1][wave][c] = db[it * VEC + j];
      }
    }
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += 256) {
    partial[(long)blockIdx.x * 2 * D + c] = red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c];
    partial[(long)blockIdx.x * 2 * D + D + c] = red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c];
  }
}

// ---------------------------------------------------------------- generic column sum (bias grads, dpos)
// out[c] (+)= sum_r x[r][c];  stage 1: block = (cx column-chunks) x (256/cx row lanes) over a slab of rows
#define COLSUM_MAX_BLOCKS 1024
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ x, float* __restrict__ partial,
                                                             long rows, long C, int rows_per_block, int cx) {
  constexpr int VEC = msc_traits<T>::VEC;
  const int nty = 256 / cx;
  const int tx = threadIdx.x % cx, ty = threadIdx.x / cx;
  const long c0 = ((long)blockIdx.x * cx + tx) * VEC;
  const long r0 = (long)blockIdx.y * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  float s[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) s[j] = 0.f;
  if (c0 < C && ty < nty) {
    long r = r0 + ty;
    for (; r + 3L * nty < r1; r += 4L * nty) {   // four independent 16-byte loads in flight per lane
      vecf<T> v0, v1, v2, v3;
      v0.load(x + r * C + c0);
      v1.load(x + (r + nty) * C + c0);
      v2.load(x + (r + 2L * nty) * C + c0);
      v3.load(x + (r + 3L * nty) * C + c0);
#pragma unroll
      for (int j = 0; j < VEC; ++j) s[j] += (v0.v[j] + v1.v[j]) + (v2.v[j] + v3.v[j]);
    }
    for (; r < r1; r += nty) {
      vecf<T> v;
      v.load(x + r * C + c0);
#pragma unroll
      for (int j = 0; j < VEC; ++j) s[j] += v.v[j];
    }
  }
  __shared__ float red[2048];   // [ty][cx*VEC]
  const int W = cx * VEC;
  if (ty < nty) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) red[ty * W + tx * VEC + j] = s[j];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < W; c += 256) {
    const long cg = (long)blockIdx.x * W + c;
    if (cg >= C) break;
    float a = 0.f;
    for (int t = 0; t < nty; ++t) a += red[t * W + c];
    partial[(long)blockIdx.y * C + cg] = a;
  }
}
__global__ void colsum_finalize_kernel(const float* __restrict__ partial, int nblk, long C, float* __restrict__ out,
                                       int accumulate) {
  __shared__ float red[8][32];
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;     // 32 columns x 8 row lanes
  const long c = (long)blockIdx.x * 32 + cl;
  float s = 0.f;
  if (c < C) {
    int k = rl;
    for (; k + 24 < nblk; k += 32) {   // 4 independent loads per trip
      const float x0 = partial[(long)k * C + c], x1 = partial[(long)(k + 8) * C + c];
      const float x2 = partial[(long)(k + 16) * C + c], x3 = partial[(long)(k + 24) * C + c];
      s += (x0 + x1) + (x2 + x3);
    }
    for (; k < nblk; k += 8) s += partial[(long)k * C + c];
  }
  red[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && c < C) {
    float t = accumulate ? out[c] : 0.f;
    for (int r = 0; r < 8; ++r) t += red[r][cl];
    out[c] = t;
  }
}

static inline int ew_blocks(long n) {
  long b = (n + 255) / 256;
  return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

// grid of the BatchNorm streaming kernels: UN vectors per thread and iteration, at most 8 blocks of 256 threads per CU (2048 blocks) —
// enough waves to cover the HBM latency with UN x 16 B per operand in flight per lane, few enough that every thread iterates
#ifndef BN_APPLY_UN
#define BN_APPLY_UN 4
#endif
static inline int bn_stream_blocks(long nvec, int un) {
  long b = (nvec + 256L * un - 1) / (256L * un);
  return (int)(b > 2048 ? 2048 : (b < 1 ? 1 : b));
}

// ================================================================ C-ABI
extern "C" int msc_bn_finalize(const float* partial, int nblk, int C, int64_t count, float eps, const float* gamma,
                               const float* beta, float* mean, float* rstd, float* scale, float* shift, void* stream) {
  MSC_CHECK_ARG(partial && gamma && beta && mean && rstd && scale && shift && nblk > 0 && C > 0 && count > 0);
  FinArgs a;
  memset(&a, 0, sizeof(a));
  a.partial = partial; a.nblk = nblk; a.C = C; a.inv_count = 1.0 / (double)count; a.eps = eps;
  a.gamma = gamma; a.beta = beta; a.mean = mean; a.rstd = rstd; a.scale = scale; a.shift = shift;
  if (nblk > 128) hipLaunchKernelGGL((finalize2_kernel<FIN_BN_FWD, 2>), dim3(cdiv(C, 2)), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((finalize2_kernel<FIN_BN_FWD, 8>), dim3(cdiv(C, 8)), dim3(256), 0, (hipStream_t)stream, a);
  return msc_launch_status();
}

static void launch_fin_sums(const float* partial, int nblk, int C, float* out0, float* out1, float* acc0, float* acc1,
                            int accumulate, hipStream_t st) {
  FinArgs a;
  memset(&a, 0, sizeof(a));
  a.partial = partial; a.nblk = nblk; a.C = C; a.out0 = out0; a.out1 = out1; a.acc0 = acc0; a.acc1 = acc1; a.accumulate = accumulate;
  if (nblk > 128) hipLaunchKernelGGL((finalize2_kernel<FIN_SUMS, 2>), dim3(cdiv(C, 2)), dim3(256), 0, st, a);
  else hipLaunchKernelGGL((finalize2_kernel<FIN_SUMS, 8>), dim3(cdiv(C, 8)), dim3(256), 0, st, a);
}

extern "C" int msc_bn_apply(const void* x, const float* scale, const float* shift, const void* res, void* y, int64_t rows,
                            int C, int relu, int dtype, void* stream) {
  MSC_CHECK_ARG(x && scale && shift && y && rows > 0 && C > 0);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MSC_F32) {
    MSC_CHECK_ARG(C % 4 == 0);
    const long nvec = rows * C / 4;
    if (256 % (C / 4) == 0)
      hipLaunchKernelGGL((bn_apply_kernel<float, BN_APPLY_UN>), dim3(bn_stream_blocks(nvec, BN_APPLY_UN)), dim3(256), 0, st, (const float*)x, scale, shift,
                         (const float*)res, (float*)y, nvec, C, relu);
    else
      hipLaunchKernelGGL(bn_apply_any_kernel<float>, dim3(ew_blocks(nvec)), dim3(256), 0, st, (const float*)x, scale, shift, (const float*)res, (float*)y, nvec, C, relu);
  } else if (dtype == MSC_BF16) {
    MSC_CHECK_ARG(C % 8 == 0);
    const long nvec = rows * C / 8;
    if (256 % (C / 8) == 0)
      hipLaunchKernelGGL((bn_apply_kernel<bf16_t, BN_APPLY_UN>), dim3(bn_stream_blocks(nvec, BN_APPLY_UN)), dim3(256), 0, st, (const bf16_t*)x, scale, shift,
                         (const bf16_t*)res, (bf16_t*)y, nvec, C, relu);
    else
      hipLaunchKernelGGL(bn_apply_any_kernel<bf16_t>, dim3(ew_blocks(nvec)), dim3(256), 0, st, (const bf16_t*)x, scale, shift, (const bf16_t*)res, (bf16_t*)y, nvec, C, relu);
  } else
    return MSC_ERR_BAD_ARG;
  return msc_launch_status();
}

extern "C" int64_t msc_bn_bwd_workspace_bytes(int C) { return (int64_t)(BN_MAX_BLOCKS + 1) * 2 * C * 4; }

// Rows per block of the BatchNorm-backward reduction.  A block has nty = 256 / (C / VEC) row lanes; its threads walk their rows one after the
// other, so with 128 rows per block a 512-channel layer (4 row lanes) made 32 dependent load rounds per thread — the reduction of a 3 MB
// tensor took as long as that of a 50 MB one (the launch was 25 blocks).  At most BN_ROWS_PER_LANE rows per lane, 32..128 rows per block.
#ifndef BN_ROWS_PER_LANE
#define BN_ROWS_PER_LANE 8
#endif
static inline int bn_reduce_rows_per_block(int nty) {
  int r = BN_ROWS_PER_LANE * nty;
  return r < 32 ? 32 : (r > 128 ? 128 : r);
}

// partial_in / nblk_in: the statistics pass was done by the kernel that produced dy (msc_conv2d_dgrad_bnstat): [nblk_in][2][C] row-block sums
template <typename T>
static int bn_bwd_t(const void* dy, const void* x, const void* y, const float* mean, const float* rstd,
                    const float* gamma, const float* scale, const float* shift, void* dx, void* dres, float* dgamma, float* dbeta, float* ws, int64_t rows,
                    int C, int relu, int accumulate, hipStream_t st, const float* partial_in = nullptr, int nblk_in = 0) {
  constexpr int VEC = msc_traits<T>::VEC;
  if (C % VEC || C / VEC > 256) return MSC_ERR_BAD_ARG;
  const int ncx = C / VEC, nty = 256 / ncx;
  if (nty < 1 || (long)nty * C > 2048) return MSC_ERR_BAD_ARG;
  float* sums = ws + (long)BN_MAX_BLOCKS * 2 * C;  // [2][C]
  if (partial_in) {
    launch_fin_sums(partial_in, nblk_in, C, sums, sums + C, dbeta, dgamma, accumulate, st);
  } else {
    int nblk = cdiv(rows, bn_reduce_rows_per_block(nty));
    if (nblk > BN_MAX_BLOCKS) nblk = BN_MAX_BLOCKS;
    const int rpb = cdiv(rows, nblk);
    nblk = cdiv(rows, rpb);
    float* partial = ws;
    if (mscplan::g_knobs.bn_un >= 4)
      hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, 4>), dim3(nblk), dim3(256), 0, st, (const T*)dy, (const T*)x, (const T*)y,
                         mean, rstd, scale, shift, partial, (long)rows, C, relu, rpb);
    else
      hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, 1>), dim3(nblk), dim3(256), 0, st, (const T*)dy, (const T*)x, (const T*)y,
                         mean, rstd, scale, shift, partial, (long)rows, C, relu, rpb);
    launch_fin_sums(partial, nblk, C, sums, sums + C, dbeta, dgamma, accumulate, st);
  }
  const long nvec = rows * C / VEC;
  if (256 % ncx == 0)
    hipLaunchKernelGGL((bn_bwd_apply_kernel<T, BN_APPLY_UN>), dim3(bn_stream_blocks(nvec, BN_APPLY_UN)), dim3(256), 0, st, (const T*)dy, (const T*)x,
                       (const T*)y, mean, rstd, gamma, scale, shift, sums, sums + C, (T*)dx, (T*)dres, nvec, C, relu, 1.f / (float)rows);
  else
    hipLaunchKernelGGL(bn_bwd_apply_any_kernel<T>, dim3(ew_blocks(nvec)), dim3(256), 0, st, (const T*)dy, (const T*)x,
                       (const T*)y, mean, rstd, gamma, scale, shift, sums, sums + C, (T*)dx, (T*)dres, nvec, C, relu, 1.f / (float)rows);
  return msc_launch_status();
}

extern "C" int msc_bn_bwd(const void* dy, const void* x, const void* y, const float* mean, const float* rstd,
                          const float* gamma, const float* scale, const float* shift, void* dx, void* dres, float* dgamma,
                          float* dbeta, float* ws, int64_t rows, int C, int relu, int accumulate, int dtype, void* stream) {
  MSC_CHECK_ARG(dy && x && mean && rstd && gamma && dx && dgamma && dbeta && ws && rows > 0 && C > 0);
  MSC_CHECK_ARG((scale == nullptr) == (shift == nullptr));
  MSC_CHECK_ARG(!relu || y || scale);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MSC_F32)
    return bn_bwd_t<float>(dy, x, y, mean, rstd, gamma, scale, shift, dx, dres, dgamma, dbeta, ws, rows, C, relu, accumulate, st);
  if (dtype == MSC_BF16)
    return bn_bwd_t<bf16_t>(dy, x, y, mean, rstd, gamma, scale, shift, dx, dres, dgamma, dbeta, ws, rows, C, relu, accumulate, st);
  return MSC_ERR_BAD_ARG;
}

extern "C" int msc_bn_bwd_from_partials(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma, const float* scale,
                                        const float* shift, const float* partial, int nblk, void* dx, float* dgamma, float* dbeta, float* ws,
                                        int64_t rows, int C, int accumulate, int dtype, void* stream) {
  MSC_CHECK_ARG(dy && x && mean && rstd && gamma && scale && shift && partial && nblk > 0 && dx && dgamma && dbeta && ws && rows > 0 && C > 0);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MSC_F32)
    return bn_bwd_t<float>(dy, x, nullptr, mean, rstd, gamma, scale, shift, dx, nullptr, dgamma, dbeta, ws, rows, C, 1, accumulate, st, partial, nblk);
  if (dtype == MSC_BF16)
    return bn_bwd_t<bf16_t>(dy, x, nullptr, mean, rstd, gamma, scale, shift, dx, nullptr, dgamma, dbeta, ws, rows, C, 1, accumulate, st, partial, nblk);
  return MSC_ERR_BAD_ARG;
}

// ---- stem: BN + ReLU + max-pool fused (see the kernels above).  x: conv output [B,H,W,C]; y/idx: [B,Ho,Wo,C]
extern "C" int msc_bn_relu_pool_fwd(const void* x, const float* scale, const float* shift, void* y, uint8_t* idx, void* xsel, int B, int H,
                                    int W, int C, int K, int S, int P, int dtype, void* stream) {
  MSC_CHECK_ARG(x && scale && shift && y && idx && xsel && B > 0 && H > 0 && W > 0 && C > 0 && K > 0 && K * K <= 255 && S > 0 && P >= 0);
  MSC_CHECK_ARG(dtype == MSC_F32 || dtype == MSC_BF16);
  PoolGeo g = {H, W, (H + 2 * P - K) / S + 1, (W + 2 * P - K) / S + 1, K, S, P};
  MSC_CHECK_ARG(g.Ho > 0 && g.Wo > 0);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MSC_F32) {
    MSC_CHECK_ARG(C % 4 == 0 && (long)B * g.Ho * g.Wo * C / 4 < 0x7fffffffL);
    hipLaunchKernelGGL(bn_relu_pool_fwd_kernel<float>, dim3(ew_blocks((long)B * g.Ho * g.Wo * C / 4)), dim3(256), 0, st, (const float*)x, scale,
                       shift, (float*)y, idx, (float*)xsel, B, C, g);
  } else {
    MSC_CHECK_ARG(C % 8 == 0 && (long)B * g.Ho * g.Wo * C / 8 < 0x7fffffffL);
    hipLaunchKernelGGL(bn_relu_pool_fwd_kernel<bf16_t>, dim3(ew_blocks((long)B * g.Ho * g.Wo * C / 8)), dim3(256), 0, st, (const bf16_t*)x,
                       scale, shift, (bf16_t*)y, idx, (bf16_t*)xsel, B, C, g);
  }
  return msc_launch_status();
}

template <typename T>
static int bn_pool_bwd_t(const void* dpool, const uint8_t* idx, const void* xsel, const void* x, const float* mean, const float* rstd,
                         const float* gamma, const float* scale, const float* shift, void* dx, float* dgamma, float* dbeta, float* ws, int B,
                         int C, PoolGeo g, int accumulate, hipStream_t st) {
  constexpr int VEC = msc_traits<T>::VEC;
  if (C % VEC || C / VEC > 256) return MSC_ERR_BAD_ARG;
  const int ncx = C / VEC, nty = 256 / ncx;
  if (nty < 1 || (long)nty * C > 2048) return MSC_ERR_BAD_ARG;
  const long rows = (long)B * g.H * g.W;
  if (rows * (C / VEC) >= 0x7fffffffL) return MSC_ERR_UNSUPPORTED;
  // statistics pass on the POOLED size: sum over windows of dpool*[relu'] and dpool*[relu']*xhat at the window's arg-max (xsel) equals
  // the sum over pixels of the gathered gradient — the generic BatchNorm-backward reduction with dy = dpool, x = xsel
  const long prow = (long)B * g.Ho * g.Wo;
  int nblk = cdiv(prow, bn_reduce_rows_per_block(nty));
  if (nblk > BN_MAX_BLOCKS) nblk = BN_MAX_BLOCKS;
  const int rpb = cdiv(prow, nblk);
  nblk = cdiv(prow, rpb);
  float* partial = ws;
  float* sums = ws + (long)BN_MAX_BLOCKS * 2 * C;
  hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, 4>), dim3(nblk), dim3(256), 0, st, (const T*)dpool, (const T*)xsel, (const T*)nullptr, mean, rstd, scale,
                     shift, partial, prow, C, 1, rpb);
  launch_fin_sums(partial, nblk, C, sums, sums + C, dbeta, dgamma, accumulate, st);
  const long nvec = rows * C / VEC;
  hipLaunchKernelGGL(bn_pool_bwd_apply_kernel<T>, dim3(ew_blocks(nvec)), dim3(256), 0, st, (const T*)dpool, idx, (const T*)x, mean, rstd, gamma,
                     scale, shift, sums, sums + C, (T*)dx, nvec, C, 1.f / (float)rows, g);
  return msc_launch_status();
}

// dpool: gradient w.r.t. the pooled map [B,Ho,Wo,C]; xsel: from the forward; dx: gradient w.r.t. the conv output x [B,H,W,C]; ws as msc_bn_bwd
extern "C" int msc_bn_relu_pool_bwd(const void* dpool, const uint8_t* idx, const void* xsel, const void* x, const float* mean,
                                    const float* rstd, const float* gamma, const float* scale, const float* shift, void* dx, float* dgamma,
                                    float* dbeta, float* ws, int B, int H, int W, int C, int K, int S, int P, int accumulate, int dtype,
                                    void* stream) {
  MSC_CHECK_ARG(dpool && idx && xsel && x && mean && rstd && gamma && scale && shift && dx && dgamma && dbeta && ws);
  MSC_CHECK_ARG(B > 0 && H > 0 && W > 0 && C > 0 && K > 0 && S > 0 && P >= 0);
  PoolGeo g = {H, W, (H + 2 * P - K) / S + 1, (W + 2 * P - K) / S + 1, K, S, P};
  MSC_CHECK_ARG(g.Ho > 0 && g.Wo > 0);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == MSC_F32)
    return bn_pool_bwd_t<float>(dpool, idx, xsel, x, mean, rstd, gamma, scale, shift, dx, dgamma, dbeta, ws, B, C, g, accumulate, st);
  if (dtype == MSC_BF16)
    return bn_pool_bwd_t<bf16_t>(dpool, idx, xsel, x, mean, rstd, gamma, scale, shift, dx, dgamma, dbeta, ws, B, C, g, accumulate, st);
  return MSC_ERR_BAD_ARG;
}

extern "C" int msc_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean,
                                 float* rstd, int64_t rows, int D, float eps, int dtype, void* stream) {
  MSC_CHECK_ARG(x && gamma && beta && y && mean && rstd && rows > 0 && D > 0 && D <= 1024);
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = (unsigned)cdiv(rows, 4 * LN_FWD_ROWS);
  if (dtype == MSC_F32) {
    MSC_CHECK_ARG(D % 256 == 0);
    hipLaunchKernelGGL(ln_fwd_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)x, gamma, beta, (float*)y,
                       mean, rstd, (long)rows, D, eps);
  } else if (dtype == MSC_BF16) {
    MSC_CHECK_ARG(D % 512 == 0);
    hipLaunchKernelGGL(ln_fwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t*)x, gamma, beta,
                       (bf16_t*)y, mean, rstd, (long)rows, D, eps);
  } else
    return MSC_ERR_BAD_ARG;
  return msc_launch_status();
}

extern "C" int64_t msc_layernorm_bwd_workspace_bytes(int64_t rows, int D) {
  return (int64_t)cdiv(rows, LN_ROWS_PER_BLOCK) * 2 * D * 4;
}

extern "C" int msc_layernorm_bwd(const void* dy, const void* x, const float* mean, const float* rstd,
                                 const float* gamma, const void* dadd, void* dx, float* dgamma, float* dbeta, float* ws,
                                 int64_t rows, int D, int accumulate, int dtype, void* stream) {
  MSC_CHECK_ARG(dy && x && mean && rstd && gamma && dx && ws && rows > 0 && D > 0 && D <= 1024 && ((dgamma == nullptr) == (dbeta == nullptr)));
  hipStream_t st = (hipStream_t)stream;
  const int nblk = cdiv(rows, LN_ROWS_PER_BLOCK);
  if (dtype == MSC_F32) {
    MSC_CHECK_ARG(D % 256 == 0);
    hipLaunchKernelGGL(ln_bwd_kernel<float>, dim3(nblk), dim3(256), 0, st, (const float*)dy, (const float*)x, mean,
                       rstd, gamma, (const float*)dadd, (float*)dx, ws, (long)rows, D);
  } else if (dtype == MSC_BF16) {
    MSC_CHECK_ARG(D % 512 == 0);
    hipLaunchKernelGGL(ln_bwd_kernel<bf16_t>, dim3(nblk), dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)x,
                       mean, rstd, gamma, (const bf16_t*)dadd, (bf16_t*)dx, ws, (long)rows, D);
  } else
    return MSC_ERR_BAD_ARG;
  // ws holds [nblk][2][D]: first stat = dgamma, second = dbeta.  dgamma == dbeta == NULL: the caller finishes them later, off the
  // critical chain (msc_layernorm_bwd_params)
  if (dgamma) launch_fin_sums(ws, nblk, D, nullptr, nullptr, dgamma, dbeta, accumulate, st);
  return msc_launch_status();
}

extern "C" int msc_layernorm_bwd_params(const float* ws_a, int64_t rows_a, const float* ws_b, int64_t rows_b, int D, float* dgamma,
                                        float* dbeta, int accumulate, void* stream) {
  MSC_CHECK_ARG(ws_a && rows_a > 0 && dgamma && dbeta && D > 0 && D <= 1024 && ((ws_b != nullptr) == (rows_b > 0)));
  FinArgs a;
  memset(&a, 0, sizeof(a));
  a.partial = ws_a; a.nblk = cdiv(rows_a, LN_ROWS_PER_BLOCK); a.C = D;
  a.partial2 = ws_b; a.nblk2 = ws_b ? cdiv(rows_b, LN_ROWS_PER_BLOCK) : 0;
  a.acc0 = dgamma; a.acc1 = dbeta; a.accumulate = accumulate;
  // off the critical chain: many blocks, short per-thread chains
  hipLaunchKernelGGL((finalize2_kernel<FIN_SUMS, 2>), dim3(cdiv(D, 2)), dim3(256), 0, (hipStream_t)stream, a);
  return msc_launch_status();
}

// row-slabs: enough workgroups to fill the chip (~2048 with the column blocks), each with >= 32 rows
static int colsum_row_blocks(long rows) {
  long nblk = rows / 32;
  if (nblk > 512) nblk = 512;
  if (nblk > COLSUM_MAX_BLOCKS) nblk = COLSUM_MAX_BLOCKS;
  if (nblk < 1) nblk = 1;
  return (int)nblk;
}

// out[c] (+)= sum_k partial[k][c] — finish column sums whose per-row-block partials another kernel produced
// (msc_linear_dgrad's fused bias-gradient epilogue)
extern "C" int msc_colsum_finalize(const float* partial, int nblk, int64_t cols, float* out, int accumulate, void* stream) {
  MSC_CHECK_ARG(partial && out && nblk > 0 && cols > 0);
  hipLaunchKernelGGL(colsum_finalize_kernel, dim3(cdiv(cols, 32)), dim3(256), 0, (hipStream_t)stream, partial, nblk, (long)cols, out, accumulate);
  return msc_launch_status();
}

extern "C" int64_t msc_colsum_workspace_bytes(int64_t rows, int64_t cols) {
  return (int64_t)colsum_row_blocks(rows) * cols * 4;
}

// out[c] (+)= sum_r x[r][c]  (x is T [rows][cols]) — bias gradients, positional-embedding gradient
extern "C" int msc_colsum(const void* x, float* out, float* ws, int64_t rows, int64_t cols, int accumulate, int dtype,
                          void* stream) {
  MSC_CHECK_ARG(x && out && ws && rows > 0 && cols > 0);
  MSC_CHECK_ARG(dtype == MSC_F32 || dtype == MSC_BF16);
  hipStream_t st = (hipStream_t)stream;
  const int VEC = dtype == MSC_F32 ? 4 : 8;
  MSC_CHECK_ARG(cols % VEC == 0);
  // cx = column chunks per block: a power of two <= 64 (>= 4 row lanes per block keep loads of several rows in flight)
  int cx = 1;
  while (cx < 64 && (long)cx * VEC < cols) cx <<= 1;
  int nblk = colsum_row_blocks(rows);
  const int rpb = cdiv(rows, nblk);
  nblk = cdiv(rows, rpb);
  dim3 grid((unsigned)cdiv(cols, (long)cx * VEC), (unsigned)nblk);
  if (dtype == MSC_F32)
    hipLaunchKernelGGL(colsum_partial_kernel<float>, grid, dim3(256), 0, st, (const float*)x, ws, (long)rows, (long)cols, rpb, cx);
  else
    hipLaunchKernelGGL(colsum_partial_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)x, ws, (long)rows, (long)cols, rpb, cx);
  hipLaunchKernelGGL(colsum_finalize_kernel, dim3(cdiv(cols, 32)), dim3(256), 0, st, ws, nblk, (long)cols, out, accumulate);
  return msc_launch_status();
}
