// elementwise.hip — HBM-bound layout / pooling / activation / optimizer kernels of the path.
// 16-byte vector accesses, grid-stride loops capped at ~16 blocks per CU, no atomics.
// Build-authored (the upstream reference has no code, SURVEY.md finding zero).
#include "msc_common.h"

static inline int ew_blocks(long n, int bs = 256) {
  long b = (n + bs - 1) / bs;
  return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

// ---------------------------------------------------------------- image NCHW fp32 -> NHWC(Cpad) T
template <typename T>
__global__ void image_to_nhwc_kernel(const float* __restrict__ img, T* __restrict__ out, long npix_total, int C, long HW, int Cpad) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix_total; i += (long)gridDim.x * blockDim.x) {
    const long b = i / HW, pix = i % HW;
    if (Cpad == 8 && C <= 8) {   // the stem's layout: one pixel = 8 channels = one (bf16) or two (f32) 16-byte stores
      float v[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) v[c] = c < C ? img[(b * C + c) * HW + pix] : 0.f;
      constexpr int VEC = msc_traits<T>::VEC;
#pragma unroll
      for (int c0 = 0; c0 < 8; c0 += VEC) {
        vecf<T> o;
#pragma unroll
        for (int j = 0; j < VEC; ++j) o.v[j] = v[c0 + j];
        o.store(out + i * 8 + c0);
      }
      continue;
    }
    for (int c = 0; c < Cpad; ++c) {
      const float v = c < C ? img[(b * C + c) * HW + pix] : 0.f;
      msc_traits<T>::st(out + i * Cpad + c, v);
    }
  }
}

// ---------------------------------------------------------------- stem in space-to-depth form (see csrc/gemm.hip msc_stem_s2d_*)
// image NCHW fp32 [B][3][H][W] -> T [B][H/2][W/2][16], channel (dy*2+dx)*3 + c holds pixel (2i+dy, 2j+dx); channels 12..15 = 0
template <typename T>
__global__ void image_to_s2d_kernel(const float* __restrict__ img, T* __restrict__ out, long ncell, int H, int W) {
  const int Hs = H >> 1, Ws = W >> 1;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < ncell; i += (long)gridDim.x * blockDim.x) {
    const int j = (int)(i % Ws);
    const long t = i / Ws;
    const int r = (int)(t % Hs);
    const long b = t / Hs;
    float v[16];
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float2 px = *reinterpret_cast<const float2*>(img + ((b * 3 + c) * H + 2 * r + dy) * (long)W + 2 * j);   // dx = 0, 1
        v[(dy * 2 + 0) * 3 + c] = px.x;
        v[(dy * 2 + 1) * 3 + c] = px.y;
      }
    v[12] = v[13] = v[14] = v[15] = 0.f;
    constexpr int VEC = msc_traits<T>::VEC;
#pragma unroll
    for (int c0 = 0; c0 < 16; c0 += VEC) {
      vecf<T> o;
#pragma unroll
      for (int k = 0; k < VEC; ++k) o.v[k] = v[c0 + k];
      o.store(out + i * 16 + c0);
    }
  }
}
// conv1 weights fp32 [Cout][7][7][8] (OHWI, RGB in channels 0..2) -> T [Cout][4][4][16]: tap (i, dy) of the re-blocked kernel is
// original row kh = 2i + dy - 1 (kh = -1: the zero row that pads 7 to 8), same for columns
template <typename T>
__global__ void stem_weight_s2d_kernel(const float* __restrict__ w7, T* __restrict__ out, int Cout) {
  const int n = Cout * 256;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
    const int ch = e & 15, jj = (e >> 4) & 3, ii = (e >> 6) & 3, o = e >> 8;
    float v = 0.f;
    if (ch < 12) {
      const int q = ch / 3, c = ch - 3 * q, dy = q >> 1, dx = q & 1;
      const int kh = 2 * ii + dy - 1, kw = 2 * jj + dx - 1;
      if (kh >= 0 && kw >= 0) v = w7[((o * 7 + kh) * 7 + kw) * 8 + c];
    }
    msc_traits<T>::st(out + e, v);
  }
}
// weight gradient back: dw7 [Cout][7][7][8] (+)= gather of dw_s2d [Cout][4][4][16] (padded channels 3..7 get 0)
__global__ void stem_wgrad_unpack_kernel(const float* __restrict__ dws, float* __restrict__ dw7, int Cout, int accumulate) {
  const int n = Cout * 7 * 7 * 8;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
    const int c = e & 7, t = e >> 3, kw = t % 7, kh = (t / 7) % 7, o = t / 49;
    float v = 0.f;
    if (c < 3) {
      const int ii = (kh + 1) >> 1, dy = (kh + 1) & 1, jj = (kw + 1) >> 1, dx = (kw + 1) & 1;
      v = dws[((o * 4 + ii) * 4 + jj) * 16 + (dy * 2 + dx) * 3 + c];
    }
    dw7[e] = (accumulate ? dw7[e] : 0.f) + v;
  }
}

// ---------------------------------------------------------------- camera frames as they come off the wire / disk:
// uint8 RGB, pixel-interleaved [B][H][W][3]  ->  fp32 planes [B][3][H][W] in [0,1] (IEEE division by 255: bit-equal to
// numpy's float32(u8) / float32(255)).  One thread = 4 consecutive pixels: three aligned 4-byte loads, one float4 store per plane.
__global__ void frames_u8_to_nchw_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst, long nquads, long HW) {
  for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < nquads; q += (long)gridDim.x * blockDim.x) {
    const long pix = q * 4, b = pix / HW, p = pix - b * HW;      // HW % 4 == 0: the 4 pixels lie in one frame
    const uint32_t* s = reinterpret_cast<const uint32_t*>(src + pix * 3);
    const uint32_t w0 = s[0], w1 = s[1], w2 = s[2];               // R0 G0 B0 R1 | G1 B1 R2 G2 | B2 R3 G3 B3
    const uint32_t r[4] = {w0 & 255u, (w0 >> 24) & 255u, (w1 >> 16) & 255u, (w2 >> 8) & 255u};
    const uint32_t g[4] = {(w0 >> 8) & 255u, w1 & 255u, (w1 >> 24) & 255u, (w2 >> 16) & 255u};
    const uint32_t bl[4] = {(w0 >> 16) & 255u, (w1 >> 8) & 255u, w2 & 255u, (w2 >> 24) & 255u};
    float* o = dst + b * 3 * HW + p;
    *reinterpret_cast<float4*>(o) = make_float4(__fdiv_rn((float)r[0], 255.f), __fdiv_rn((float)r[1], 255.f), __fdiv_rn((float)r[2], 255.f), __fdiv_rn((float)r[3], 255.f));
    *reinterpret_cast<float4*>(o + HW) = make_float4(__fdiv_rn((float)g[0], 255.f), __fdiv_rn((float)g[1], 255.f), __fdiv_rn((float)g[2], 255.f), __fdiv_rn((float)g[3], 255.f));
    *reinterpret_cast<float4*>(o + 2 * HW) = make_float4(__fdiv_rn((float)bl[0], 255.f), __fdiv_rn((float)bl[1], 255.f), __fdiv_rn((float)bl[2], 255.f), __fdiv_rn((float)bl[3], 255.f));
  }
}

// ---------------------------------------------------------------- max-pool (NHWC), argmax kept as uint8 tap id
template <typename T>
__global__ void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, uint8_t* __restrict__ idx, int B, int H, int W,
                                   int C, int Ho, int Wo, int K, int S, int P) {
  constexpr int VEC = msc_traits<T>::VEC;
  const int cv = C / VEC;
  const long n = (long)B * Ho * Wo * cv;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const unsigned iu = (unsigned)i;   // n < 2^31 (launcher): 32-bit divisions
    const int c0 = (int)(iu % (unsigned)cv) * VEC;
    unsigned t = iu / (unsigned)cv;
    const int wo = (int)(t % (unsigned)Wo); t /= (unsigned)Wo;
    const int ho = (int)(t % (unsigned)Ho);
    const int b = (int)(t / (unsigned)Ho);
    float best[VEC];
    int bi[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { best[j] = -INFINITY; bi[j] = 0; }
    for (int kh = 0; kh < K; ++kh) {
      const int hi = ho * S - P + kh;
      if ((unsigned)hi >= (unsigned)H) continue;
      for (int kw = 0; kw < K; ++kw) {
        const int wi = wo * S - P + kw;
        if ((unsigned)wi >= (unsigned)W) continue;
        vecf<T> v;
        v.load(x + (((long)b * H + hi) * W + wi) * C + c0);
#pragma unroll
        for (int j = 0; j < VEC; ++j)
          if (v.v[j] > best[j]) { best[j] = v.v[j]; bi[j] = kh * K + kw; }
      }
    }
    vecf<T> o;
#pragma unroll
    for (int j = 0; j < VEC; ++j) o.v[j] = best[j];
    const long oo = (((long)b * Ho + ho) * Wo + wo) * C + c0;
    o.store(y + oo);
#pragma unroll
    for (int j = 0; j < VEC; ++j) idx[oo + j] = (uint8_t)bi[j];
  }
}

template <typename T>
__global__ void maxpool_bwd_kernel(const T* __restrict__ dy, const uint8_t* __restrict__ idx, T* __restrict__ dx, int B, int H,
                                   int W, int C, int Ho, int Wo, int K, int S, int P) {
  constexpr int VEC = msc_traits<T>::VEC;
  const int cv = C / VEC;
  const long n = (long)B * H * W * cv;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    // 32-bit index arithmetic (n < 2^31 is checked by the launcher): 64-bit divisions dominated this kernel
    const unsigned iu = (unsigned)i;
    const int c0 = (int)(iu % (unsigned)cv) * VEC;
    unsigned t = iu / (unsigned)cv;
    const int w = (int)(t % (unsigned)W); t /= (unsigned)W;
    const int h = (int)(t % (unsigned)H);
    const int b = (int)(t / (unsigned)H);
    vecf<T> acc;
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc.v[j] = 0.f;
    for (int kh = 0; kh < K; ++kh) {
      const int th = h + P - kh;
      if (th < 0 || th % S) continue;
      const int ho = th / S;
      if (ho >= Ho) continue;
      for (int kw = 0; kw < K; ++kw) {
        const int tw = w + P - kw;
        if (tw < 0 || tw % S) continue;
        const int wo = tw / S;
        if (wo >= Wo) continue;
        const long oo = (((long)b * Ho + ho) * Wo + wo) * C + c0;
        vecf<T> g;
        g.load(dy + oo);
        // the VEC tap ids of this window in one load (oo is a multiple of VEC)
        uint64_t taps;
        if constexpr (VEC == 8) taps = *reinterpret_cast<const uint64_t*>(idx + oo);
        else taps = *reinterpret_cast<const uint32_t*>(idx + oo);
        const unsigned want = (unsigned)(kh * K + kw);
#pragma unroll
        for (int j = 0; j < VEC; ++j)
          if (((taps >> (8 * j)) & 0xffu) == want) acc.v[j] += g.v[j];
      }
    }
    acc.store(dx + (((long)b * H + h) * W + w) * C + c0);
  }
}

// ---------------------------------------------------------------- token assembly: concat(modalities) + positional embedding
template <typename T>
__global__ void tokens_assemble_kernel(const T* __restrict__ s0, const T* __restrict__ s1, const T* __restrict__ s2,
                                       const float* __restrict__ pos, T* __restrict__ out, int B, int n0, int n1, int n2, int D) {
  constexpr int VEC = msc_traits<T>::VEC;
  const int Tt = n0 + n1 + n2, dv = D / VEC;
  const long n = (long)B * Tt * dv;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int d0 = (int)(i % dv) * VEC;
    const long r = i / dv;
    const int t = (int)(r % Tt);
    const long b = r / Tt;
    const T* src;
    if (t < n0) src = s0 + ((b * n0 + t) * D + d0);
    else if (t < n0 + n1) src = s1 + ((b * n1 + (t - n0)) * D + d0);
    else src = s2 + ((b * n2 + (t - n0 - n1)) * D + d0);
    vecf<T> v;
    v.load(src);
#pragma unroll
    for (int j = 0; j < VEC; ++j) v.v[j] += pos[(long)t * D + d0 + j];
    v.store(out + r * D + d0);
  }
}
// backward: scatter the rows of dout back to the three modality gradients (pure copy)
template <typename T>
__global__ void tokens_split_kernel(const T* __restrict__ dout, T* __restrict__ d0p, T* __restrict__ d1p, T* __restrict__ d2p,
                                    int B, int n0, int n1, int n2, int D) {
  constexpr int VEC = msc_traits<T>::VEC;
  const int Tt = n0 + n1 + n2, dv = D / VEC;
  const long n = (long)B * Tt * dv;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int d0 = (int)(i % dv) * VEC;
    const long r = i / dv;
    const int t = (int)(r % Tt);
    const long b = r / Tt;
    T* dst;
    if (t < n0) dst = d0p ? d0p + ((b * n0 + t) * D + d0) : nullptr;
    else if (t < n0 + n1) dst = d1p ? d1p + ((b * n1 + (t - n0)) * D + d0) : nullptr;
    else dst = d2p ? d2p + ((b * n2 + (t - n0 - n1)) * D + d0) : nullptr;
    if (dst) *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(dout + r * D + d0);
  }
}

// ---------------------------------------------------------------- mean over tokens
// block = (D/VEC column vectors, <= 64) x TG token groups for ONE batch row-block; thread (cx, g) sums tokens g, g+TG, ...
// with four independent accumulators (the loop is latency-bound otherwise: 306 dependent loads per thread), then the
// groups are combined through LDS in the order 0..TG-1 (fixed order: reproducible).
#define MEANPOOL_TG 16
template <typename T>
__global__ __launch_bounds__(64 * MEANPOOL_TG) void meanpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int Tt, int D) {
  constexpr int VEC = msc_traits<T>::VEC;
  __shared__ float red[MEANPOOL_TG][64][VEC];
  const int dv = D / VEC;
  const int cx = threadIdx.x, g = threadIdx.y;
  const int cblk = (dv + 63) / 64;
  const long b = blockIdx.x / cblk;
  const int c = (blockIdx.x % cblk) * 64 + cx;
  float s[4][VEC];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int j = 0; j < VEC; ++j) s[u][j] = 0.f;
  if (c < dv) {
    const T* base = x + (b * Tt) * D + c * VEC;
    int t = g;
    for (; t + 3 * MEANPOOL_TG < Tt; t += 4 * MEANPOOL_TG) {
      vecf<T> v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u].load(base + (long)(t + u * MEANPOOL_TG) * D);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < VEC; ++j) s[u][j] += v[u].v[j];
    }
    for (; t < Tt; t += MEANPOOL_TG) {
      vecf<T> v;
      v.load(base + (long)t * D);
#pragma unroll
      for (int j = 0; j < VEC; ++j) s[0][j] += v.v[j];
    }
  }
#pragma unroll
  for (int j = 0; j < VEC; ++j) red[g][cx][j] = (s[0][j] + s[1][j]) + (s[2][j] + s[3][j]);
  __syncthreads();
  if (g != 0 || c >= dv) return;
  vecf<T> o;
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    float a = red[0][cx][j];
    for (int r = 1; r < MEANPOOL_TG; ++r) a += red[r][cx][j];
    o.v[j] = a / (float)Tt;
  }
  o.store(y + b * D + c * VEC);
}
template <typename T>
__global__ void meanpool_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int B, int Tt, int D) {
  constexpr int VEC = msc_traits<T>::VEC;
  const int dv = D / VEC;
  const long n = (long)B * Tt * dv;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int d0 = (int)(i % dv) * VEC;
    const long r = i / dv;
    const long b = r / Tt;
    vecf<T> v;
    v.load(dy + b * D + d0);
#pragma unroll
    for (int j = 0; j < VEC; ++j) v.v[j] /= (float)Tt;
    v.store(dx + r * D + d0);
  }
}

// ---------------------------------------------------------------- activation backward
// ref = output y for relu/tanh/sigmoid, pre-activation for gelu
template <typename T>
__global__ void act_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ ref, T* __restrict__ dx, long nvec, int act) {
  constexpr int VEC = msc_traits<T>::VEC;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
    vecf<T> g, r;
    g.load(dy + i * VEC);
    r.load(ref + i * VEC);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const float y = r.v[j];
      float d;
      switch (act) {
        case MSC_ACT_RELU: d = y > 0.f ? 1.f : 0.f; break;
        case MSC_ACT_TANH: d = 1.f - y * y; break;
        case MSC_ACT_SIGMOID: d = y * (1.f - y); break;
        case MSC_ACT_GELU: d = 0.5f * (1.f + fast_erf(y * 0.70710678118654752f)) + y * 0.3989422804014327f * __expf(-0.5f * y * y); break;
        case MSC_ACT_MUL_REF: d = y; break;
        default: d = 1.f;
      }
      g.v[j] *= d;
    }
    g.store(dx + i * VEC);
  }
}

// ---------------------------------------------------------------- fp32 -> T casts / permutes of weights
template <typename T>
__global__ void cast_kernel(const float* __restrict__ src, T* __restrict__ dst, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    msc_traits<T>::st(dst + i, src[i]);
}
// dst[k][j][i] = src[i][j][k]  (src dims d0,d1,d2).  32x32 LDS tile transposes over (i,k) for each j.
template <typename T>
__global__ void permute3_cast_kernel(const float* __restrict__ src, T* __restrict__ dst, int d0, int d1, int d2) {
  __shared__ float tile[32][33];
  const int j = blockIdx.z;
  const int i0 = blockIdx.y * 32, k0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int i = i0 + r, k = k0 + tx;
    tile[r][tx] = (i < d0 && k < d2) ? src[((long)i * d1 + j) * d2 + k] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int k = k0 + r, i = i0 + tx;
    if (k < d2 && i < d0) msc_traits<T>::st(dst + ((long)k * d1 + j) * d0 + i, tile[tx][r]);
  }
}

// All weight transposes of a step in ONE launch.  table: int64 [n][6] = {src_off, dst_off, d0, d1, d2, first_block}
// (offsets in elements into the flat fp32 parameter buffer / the flat transposed buffer).
template <typename T>
__global__ void permute3_cast_batched_kernel(const float* __restrict__ P, T* __restrict__ TR, const int64_t* __restrict__ table, int n) {
  __shared__ float tile[32][33];
  __shared__ int s_e;
  // which table entry owns this block: every lane of wave 0 tests entries lane, lane+64, ... at once (a serial walk of
  // the ~60 entries was ~60 dependent loads at the head of every block)
  if (threadIdx.x < 64) {
    int cnt = 0;
    for (int t = threadIdx.x; t < n; t += 64) cnt += ((long)blockIdx.x >= table[t * 6 + 5]) ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    if (threadIdx.x == 0) s_e = cnt - 1;     // first_block is ascending and starts at 0
  }
  __syncthreads();
  const int e = s_e;
  const int64_t* d = table + e * 6;
  const float* src = P + d[0];
  T* dst = TR + d[1];
  const int d0 = (int)d[2], d1 = (int)d[3], d2 = (int)d[4];
  int local = (int)((long)blockIdx.x - d[5]);
  const int nbx = (d2 + 31) / 32, nby = (d0 + 31) / 32;
  const int bx = local % nbx; local /= nbx;
  const int by = local % nby;
  const int j = local / nby;
  const int i0 = by * 32, k0 = bx * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int i = i0 + r, k = k0 + tx;
    tile[r][tx] = (i < d0 && k < d2) ? src[((long)i * d1 + j) * d2 + k] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int k = k0 + r, i = i0 + tx;
    if (k < d2 && i < d0) msc_traits<T>::st(dst + ((long)k * d1 + j) * d0 + i, tile[tx][r]);
  }
}

// ---------------------------------------------------------------- fused Adam (+ low-precision shadow copy)
template <typename T>
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            T* __restrict__ shadow, long n, float lr, float b1, float b2, float eps, float wd, int step,
                            float gscale, const int64_t* __restrict__ step_dev) {
  // bias corrections are computed HERE from the step number in both modes (host argument for eager launches, device counter
  // under hipGraph replay), so the two launch paths are bit-identical
  const float st = step_dev ? (float)*step_dev : (float)step;
  const float bc1 = 1.f - powf(b1, st);
  const float bc2_sqrt = sqrtf(1.f - powf(b2, st));
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float gi = g[i] * gscale;
    float pi = p[i];
    if (wd != 0.f) gi += wd * pi;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    pi -= (lr / bc1) * mi / (sqrtf(vi) / bc2_sqrt + eps);
    p[i] = pi;
    if (shadow) msc_traits<T>::st(shadow + i, pi);
  }
}

// ================================================================ C-ABI
#define DISPATCH_T(dtype, CALL)              \
  if (dtype == MSC_F32) { typedef float T; CALL; }        \
  else if (dtype == MSC_BF16) { typedef bf16_t T; CALL; } \
  else return MSC_ERR_BAD_ARG;

extern "C" int msc_image_to_nhwc(const float* img_nchw, void* out, int B, int C, int H, int W, int Cpad, int dtype, void* stream) {
  MSC_CHECK_ARG(img_nchw && out && B > 0 && C > 0 && H > 0 && W > 0 && Cpad >= C);
  const long npix = (long)B * H * W;
  DISPATCH_T(dtype, hipLaunchKernelGGL(image_to_nhwc_kernel<T>, dim3(ew_blocks(npix)), dim3(256), 0, (hipStream_t)stream,
                                       img_nchw, (T*)out, npix, C, (long)H * W, Cpad));
  return msc_launch_status();
}

extern "C" int msc_image_to_s2d(const float* img_nchw, void* out, int B, int H, int W, int dtype, void* stream) {
  MSC_CHECK_ARG(img_nchw && out && B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && ((uintptr_t)img_nchw & 7) == 0);
  const long ncell = (long)B * (H / 2) * (W / 2);
  DISPATCH_T(dtype, hipLaunchKernelGGL(image_to_s2d_kernel<T>, dim3(ew_blocks(ncell)), dim3(256), 0, (hipStream_t)stream, img_nchw, (T*)out, ncell, H, W));
  return msc_launch_status();
}
extern "C" int msc_stem_weight_s2d(const float* w7, void* w_s2d, int Cout, int dtype, void* stream) {
  MSC_CHECK_ARG(w7 && w_s2d && Cout > 0);
  DISPATCH_T(dtype, hipLaunchKernelGGL(stem_weight_s2d_kernel<T>, dim3(ew_blocks((long)Cout * 256)), dim3(256), 0, (hipStream_t)stream, w7, (T*)w_s2d, Cout));
  return msc_launch_status();
}
extern "C" int msc_stem_wgrad_unpack(const float* dw_s2d, float* dw7, int Cout, int accumulate, void* stream) {
  MSC_CHECK_ARG(dw_s2d && dw7 && Cout > 0);
  hipLaunchKernelGGL(stem_wgrad_unpack_kernel, dim3(ew_blocks((long)Cout * 392)), dim3(256), 0, (hipStream_t)stream, dw_s2d, dw7, Cout, accumulate);
  return msc_launch_status();
}

extern "C" int msc_frames_u8_to_nchw(const uint8_t* frames_nhwc, float* img_nchw, int B, int H, int W, void* stream) {
  MSC_CHECK_ARG(frames_nhwc && img_nchw && B > 0 && H > 0 && W > 0 && ((long)H * W) % 4 == 0);
  MSC_CHECK_ARG(((uintptr_t)frames_nhwc & 3) == 0 && ((uintptr_t)img_nchw & 15) == 0);
  const long nquads = (long)B * H * W / 4;
  hipLaunchKernelGGL(frames_u8_to_nchw_kernel, dim3(ew_blocks(nquads)), dim3(256), 0, (hipStream_t)stream, frames_nhwc, img_nchw, nquads, (long)H * W);
  return msc_launch_status();
}

extern "C" int msc_maxpool_fwd(const void* x, void* y, uint8_t* idx, int B, int H, int W, int C, int K, int S, int P, int dtype, void* stream) {
  MSC_CHECK_ARG(x && y && idx && B > 0 && H > 0 && W > 0 && C > 0 && K > 0 && K * K <= 255 && S > 0 && P >= 0 && P < K);
  MSC_CHECK_ARG(C % (dtype == MSC_F32 ? 4 : 8) == 0 && (long)B * H * W * C < 0x7fffffffL);
  const int Ho = (H + 2 * P - K) / S + 1, Wo = (W + 2 * P - K) / S + 1;
  DISPATCH_T(dtype, hipLaunchKernelGGL(maxpool_fwd_kernel<T>, dim3(ew_blocks((long)B * Ho * Wo * C / msc_traits<T>::VEC)), dim3(256), 0,
                                       (hipStream_t)stream, (const T*)x, (T*)y, idx, B, H, W, C, Ho, Wo, K, S, P));
  return msc_launch_status();
}

extern "C" int msc_maxpool_bwd(const void* dy, const uint8_t* idx, void* dx, int B, int H, int W, int C, int K, int S, int P, int dtype, void* stream) {
  MSC_CHECK_ARG(dy && dx && idx && B > 0 && H > 0 && W > 0 && C > 0 && K > 0 && S > 0 && P >= 0 && P < K);
  MSC_CHECK_ARG(C % (dtype == MSC_F32 ? 4 : 8) == 0 && (long)B * H * W * C < 0x7fffffffL);
  const int Ho = (H + 2 * P - K) / S + 1, Wo = (W + 2 * P - K) / S + 1;
  DISPATCH_T(dtype, hipLaunchKernelGGL(maxpool_bwd_kernel<T>, dim3(ew_blocks((long)B * H * W * C / msc_traits<T>::VEC)), dim3(256), 0,
                                       (hipStream_t)stream, (const T*)dy, idx, (T*)dx, B, H, W, C, Ho, Wo, K, S, P));
  return msc_launch_status();
}

extern "C" int msc_tokens_assemble(const void* s0, const void* s1, const void* s2, const float* pos, void* out, int B, int n0, int n1,
                                   int n2, int D, int dtype, void* stream) {
  MSC_CHECK_ARG(pos && out && B > 0 && n0 >= 0 && n1 >= 0 && n2 >= 0 && n0 + n1 + n2 > 0 && D > 0);
  MSC_CHECK_ARG((n0 == 0 || s0) && (n1 == 0 || s1) && (n2 == 0 || s2));
  MSC_CHECK_ARG(D % (dtype == MSC_F32 ? 4 : 8) == 0);
  DISPATCH_T(dtype, hipLaunchKernelGGL(tokens_assemble_kernel<T>, dim3(ew_blocks((long)B * (n0 + n1 + n2) * D / msc_traits<T>::VEC)), dim3(256), 0,
                                       (hipStream_t)stream, (const T*)s0, (const T*)s1, (const T*)s2, pos, (T*)out, B, n0, n1, n2, D));
  return msc_launch_status();
}

extern "C" int msc_tokens_split(const void* dout, void* d0, void* d1, void* d2, int B, int n0, int n1, int n2, int D, int dtype, void* stream) {
  MSC_CHECK_ARG(dout && B > 0 && n0 >= 0 && n1 >= 0 && n2 >= 0 && n0 + n1 + n2 > 0 && D > 0);
  MSC_CHECK_ARG(D % (dtype == MSC_F32 ? 4 : 8) == 0);
  DISPATCH_T(dtype, hipLaunchKernelGGL(tokens_split_kernel<T>, dim3(ew_blocks((long)B * (n0 + n1 + n2) * D / msc_traits<T>::VEC)), dim3(256), 0,
                                       (hipStream_t)stream, (const T*)dout, (T*)d0, (T*)d1, (T*)d2, B, n0, n1, n2, D));
  return msc_launch_status();
}

extern "C" int msc_meanpool_fwd(const void* x, void* y, int B, int Tt, int D, int dtype, void* stream) {
  MSC_CHECK_ARG(x && y && B > 0 && Tt > 0 && D > 0 && D % (dtype == MSC_F32 ? 4 : 8) == 0);
  DISPATCH_T(dtype, hipLaunchKernelGGL(meanpool_fwd_kernel<T>, dim3((unsigned)(B * ((D / msc_traits<T>::VEC + 63) / 64))), dim3(64, MEANPOOL_TG), 0,
                                       (hipStream_t)stream, (const T*)x, (T*)y, B, Tt, D));
  return msc_launch_status();
}
extern "C" int msc_meanpool_bwd(const void* dy, void* dx, int B, int Tt, int D, int dtype, void* stream) {
  MSC_CHECK_ARG(dy && dx && B > 0 && Tt > 0 && D > 0 && D % (dtype == MSC_F32 ? 4 : 8) == 0);
  DISPATCH_T(dtype, hipLaunchKernelGGL(meanpool_bwd_kernel<T>, dim3(ew_blocks((long)B * Tt * D / msc_traits<T>::VEC)), dim3(256), 0, (hipStream_t)stream,
                                       (const T*)dy, (T*)dx, B, Tt, D));
  return msc_launch_status();
}

extern "C" int msc_act_bwd(const void* dy, const void* ref, void* dx, int64_t n, int act, int dtype, void* stream) {
  MSC_CHECK_ARG(dy && ref && dx && n > 0 && n % (dtype == MSC_F32 ? 4 : 8) == 0);
  DISPATCH_T(dtype, hipLaunchKernelGGL(act_bwd_kernel<T>, dim3(ew_blocks(n / msc_traits<T>::VEC)), dim3(256), 0, (hipStream_t)stream, (const T*)dy,
                                       (const T*)ref, (T*)dx, (long)(n / msc_traits<T>::VEC), act));
  return msc_launch_status();
}

extern "C" int msc_cast(const float* src, void* dst, int64_t n, int dtype, void* stream) {
  MSC_CHECK_ARG(src && dst && n > 0);
  DISPATCH_T(dtype, hipLaunchKernelGGL(cast_kernel<T>, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, src, (T*)dst, (long)n));
  return msc_launch_status();
}

__global__ void add2_f32_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, long n, int accumulate) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    out[i] = (accumulate ? out[i] : 0.f) + (a[i] + b[i]);
}
// out = (accumulate ? out : 0) + (a + b): the two half-batch chains of a fusion layer each finalize their LayerNorm parameter
// gradients into scratch; this adds them into G in a fixed order
extern "C" int msc_add2_f32(const float* a, const float* b, float* out, int64_t n, int accumulate, void* stream) {
  MSC_CHECK_ARG(a && b && out && n > 0);
  hipLaunchKernelGGL(add2_f32_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, a, b, out, (long)n, accumulate);
  return msc_launch_status();
}

extern "C" int msc_permute3_cast(const float* src, void* dst, int d0, int d1, int d2, int dtype, void* stream) {
  MSC_CHECK_ARG(src && dst && d0 > 0 && d1 > 0 && d2 > 0 && d1 <= 65535);
  dim3 grid((unsigned)cdiv(d2, 32), (unsigned)cdiv(d0, 32), (unsigned)d1);
  DISPATCH_T(dtype, hipLaunchKernelGGL(permute3_cast_kernel<T>, grid, dim3(256), 0, (hipStream_t)stream, src, (T*)dst, d0, d1, d2));
  return msc_launch_status();
}

extern "C" int msc_permute3_cast_batched(const float* P, void* TR, const int64_t* table, int n_entries, int total_blocks, int dtype,
                                         void* stream) {
  MSC_CHECK_ARG(P && TR && table && n_entries > 0 && total_blocks > 0);
  DISPATCH_T(dtype, hipLaunchKernelGGL(permute3_cast_batched_kernel<T>, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, P,
                                       (T*)TR, table, n_entries));
  return msc_launch_status();
}

extern "C" int msc_adam_step(float* p, const float* g, float* m, float* v, void* shadow, int64_t n, float lr, float beta1, float beta2,
                             float eps, float weight_decay, int step, const int64_t* step_dev, float grad_scale, int dtype, void* stream) {
  MSC_CHECK_ARG(p && g && m && v && n > 0 && (step >= 1 || step_dev));
  DISPATCH_T(dtype, hipLaunchKernelGGL(adam_kernel<T>, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v,
                                       (T*)(dtype == MSC_F32 ? nullptr : shadow), (long)n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, step_dev));
  return msc_launch_status();
}
