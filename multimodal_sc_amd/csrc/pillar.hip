// pillar.hip — LiDAR / BEV encoder front end: per-point MLP (4 -> C, ReLU) + scatter-max into a BEV grid.
//   feat[p][c] = relu(w[c] . point[p] + bias[c]);  grid[b][iy][ix][c] = max over points in the cell (empty = 0)
// One wave per point, lane = channel (C = 64): the 64 atomic-max of a point hit one 256-B line.
// Max is order-independent, so the int-compare atomicMax on non-negative floats is bitwise reproducible.
// Backward routes d(grid) to the arg-max point(s) and reduces dW/db with fixed-order partials.
// HBM-bound gather/scatter.  Build-authored; the upstream reference has no code.
#include "msc_common.h"

#define PIL_C 64
#define PIL_BLOCKS 2048

__device__ __forceinline__ int pillar_cell(float x, float y, float lo, float cell, int G) {
  const float fx = floorf((x - lo) / cell), fy = floorf((y - lo) / cell);
  if (!(fx >= 0.f && fx < (float)G && fy >= 0.f && fy < (float)G)) return -1;
  return (int)fy * G + (int)fx;
}

__global__ __launch_bounds__(256) void pillar_scatter_kernel(const float* __restrict__ pts, const float* __restrict__ w,
                                                             const float* __restrict__ bias, float* __restrict__ grid,
                                                             long npts_total, int Np, int G, float lo, float cell) {
  const int lane = threadIdx.x & 63;
  const float w0 = w[lane * 4 + 0], w1 = w[lane * 4 + 1], w2 = w[lane * 4 + 2], w3 = w[lane * 4 + 3], bb = bias[lane];
  const long nw = (long)gridDim.x * 4;
  for (long p = (long)blockIdx.x * 4 + (threadIdx.x >> 6); p < npts_total; p += nw) {
    const float4 pt = *reinterpret_cast<const float4*>(pts + p * 4);
    const int c = pillar_cell(pt.x, pt.y, lo, cell, G);
    if (c < 0) continue;
    const float f = fmaf(w3, pt.w, fmaf(w2, pt.z, fmaf(w1, pt.y, w0 * pt.x))) + bb;
    if (f > 0.f) {
      const long b = p / Np;
      atomicMax(reinterpret_cast<int*>(grid + ((b * G * G + c) * PIL_C + lane)), __float_as_int(f));
    }
  }
}

// partial[blk][c][5] = sum over the block's points of g * (x,y,z,i,1) where g = dgrid at the arg-max
template <typename T>
__global__ __launch_bounds__(256) void pillar_bwd_kernel(const float* __restrict__ pts, const float* __restrict__ w,
                                                         const float* __restrict__ bias, const float* __restrict__ grid,
                                                         const T* __restrict__ dgrid, float* __restrict__ partial,
                                                         long npts_total, int Np, int G, float lo, float cell) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float w0 = w[lane * 4 + 0], w1 = w[lane * 4 + 1], w2 = w[lane * 4 + 2], w3 = w[lane * 4 + 3], bb = bias[lane];
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f;
  const long nw = (long)gridDim.x * 4;
  // 4 points per iteration: their dependent point -> cell -> grid/dgrid loads overlap instead of serialising
  for (long p0 = ((long)blockIdx.x * 4 + wave) * 4; p0 < npts_total; p0 += nw * 4) {
    float4 pt[4];
    long o[4];
    float f[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long p = p0 + u;
      o[u] = -1;
      f[u] = 0.f;
      if (p < npts_total) {
        pt[u] = *reinterpret_cast<const float4*>(pts + p * 4);
        const int c = pillar_cell(pt[u].x, pt[u].y, lo, cell, G);
        if (c >= 0) {
          f[u] = fmaf(w3, pt[u].w, fmaf(w2, pt[u].z, fmaf(w1, pt[u].y, w0 * pt[u].x))) + bb;
          o[u] = ((p / Np) * G * G + c) * PIL_C + lane;
        }
      }
    }
    float gv[4], dg[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      gv[u] = o[u] >= 0 ? grid[o[u]] : -1.f;
      dg[u] = o[u] >= 0 ? msc_traits<T>::ld(dgrid + o[u]) : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (f[u] > 0.f && f[u] == gv[u]) {
        const float g = dg[u];
        a0 += g * pt[u].x; a1 += g * pt[u].y; a2 += g * pt[u].z; a3 += g * pt[u].w; a4 += g;
      }
    }
  }
  __shared__ float red[4][PIL_C][5];
  red[wave][lane][0] = a0; red[wave][lane][1] = a1; red[wave][lane][2] = a2; red[wave][lane][3] = a3; red[wave][lane][4] = a4;
  __syncthreads();
  for (int i = threadIdx.x; i < PIL_C * 5; i += 256) {
    const int c = i / 5, k = i % 5;
    partial[(long)blockIdx.x * PIL_C * 5 + i] = red[0][c][k] + red[1][c][k] + red[2][c][k] + red[3][c][k];
  }
}

__global__ __launch_bounds__(1024) void pillar_bwd_finalize_kernel(const float* __restrict__ partial, int nblk, float* __restrict__ dw, float* __restrict__ db, int accumulate) {
  // block = 32 outputs x 32 row lanes; lane r sums partial blocks r, r+32, ... with four independent accumulators (the
  // serial walk of 256 rows per lane was latency-bound: 69 us for 2.6 MB); fixed-order combine (reproducible)
  __shared__ float red[32][33];
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + cl;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (i < PIL_C * 5) {
    int k = rl;
    for (; k + 96 < nblk; k += 128) {
      s0 += partial[(long)k * PIL_C * 5 + i];
      s1 += partial[(long)(k + 32) * PIL_C * 5 + i];
      s2 += partial[(long)(k + 64) * PIL_C * 5 + i];
      s3 += partial[(long)(k + 96) * PIL_C * 5 + i];
    }
    for (; k < nblk; k += 32) s0 += partial[(long)k * PIL_C * 5 + i];
  }
  red[rl][cl] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (rl != 0 || i >= PIL_C * 5) return;
  float s = 0.f;
  for (int r = 0; r < 32; ++r) s += red[r][cl];
  const int c = i / 5, j = i % 5;
  float* dst = j < 4 ? dw + c * 4 + j : db + c;
  *dst = (accumulate ? *dst : 0.f) + s;
}

template <typename T>
__global__ void grid_cast_kernel(const float* __restrict__ src, T* __restrict__ dst, long nvec) {
  constexpr int VEC = msc_traits<T>::VEC;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long)gridDim.x * blockDim.x) {
    vecf<T> v;
#pragma unroll
    for (int j = 0; j < VEC; ++j) v.v[j] = src[i * VEC + j];
    v.store(dst + i * VEC);
  }
}

// points: fp32 [B][Np][4] (x,y,z,intensity); w: fp32 [64][4]; bias fp32 [64]
// grid_f32: fp32 [B][G][G][64] (zeroed here, kept for backward); grid_out: T copy fed to the BEV convs
extern "C" int msc_pillar_fwd(const float* points, const float* w, const float* bias, float* grid_f32, void* grid_out, int B, int Np,
                              int G, int C, float range_lo, float cell, int dtype, void* stream) {
  MSC_CHECK_ARG(points && w && bias && grid_f32 && B > 0 && Np > 0 && G > 0 && C == PIL_C && cell > 0.f);
  hipStream_t st = (hipStream_t)stream;
  const long n = (long)B * G * G * C;
  if (hipMemsetAsync(grid_f32, 0, n * 4, st) != hipSuccess) return MSC_ERR_LAUNCH;
  const long npts = (long)B * Np;
  int blocks = cdiv(npts, 4 * 8);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(pillar_scatter_kernel, dim3(blocks), dim3(256), 0, st, points, w, bias, grid_f32, npts, Np, G, range_lo, cell);
  if (grid_out && dtype == MSC_BF16) {
    long nb = (n / 8 + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(grid_cast_kernel<bf16_t>, dim3((unsigned)nb), dim3(256), 0, st, grid_f32, (bf16_t*)grid_out, n / 8);
  } else if (grid_out && dtype == MSC_F32 && grid_out != (void*)grid_f32) {
    if (hipMemcpyAsync(grid_out, grid_f32, n * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return MSC_ERR_LAUNCH;
  }
  return msc_launch_status();
}

extern "C" int64_t msc_pillar_bwd_workspace_bytes(void) { return (int64_t)PIL_BLOCKS * PIL_C * 5 * 4; }

extern "C" int msc_pillar_bwd(const float* points, const float* w, const float* bias, const float* grid_f32, const void* dgrid,
                              float* dw, float* db, float* ws, int B, int Np, int G, int C, float range_lo, float cell, int accumulate,
                              int dtype, void* stream) {
  MSC_CHECK_ARG(points && w && bias && grid_f32 && dgrid && dw && db && ws && B > 0 && Np > 0 && G > 0 && C == PIL_C);
  hipStream_t st = (hipStream_t)stream;
  const long npts = (long)B * Np;
  int blocks = cdiv(npts, 4 * 4 * 8);
  if (blocks > PIL_BLOCKS) blocks = PIL_BLOCKS;
  if (dtype == MSC_F32)
    hipLaunchKernelGGL(pillar_bwd_kernel<float>, dim3(blocks), dim3(256), 0, st, points, w, bias, grid_f32, (const float*)dgrid, ws, npts, Np, G, range_lo, cell);
  else if (dtype == MSC_BF16)
    hipLaunchKernelGGL(pillar_bwd_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, points, w, bias, grid_f32, (const bf16_t*)dgrid, ws, npts, Np, G, range_lo, cell);
  else
    return MSC_ERR_BAD_ARG;
  hipLaunchKernelGGL(pillar_bwd_finalize_kernel, dim3(cdiv(PIL_C * 5, 32)), dim3(1024), 0, st, ws, blocks, dw, db, accumulate);
  return msc_launch_status();
}
