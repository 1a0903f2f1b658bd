// pillar.hip — LiDAR / BEV encoder front end: per-point MLP (4 -> C, ReLU) + max over the points of each BEV cell.
//   feat[p][c] = relu(w[c] . point[p] + bias[c]);  grid[b][iy][ix][c] = max over points in the cell (empty = 0)
// Gather formulation (no fp32 grid, no memset, no atomics on HBM, no cast pass):
//   1. pillar_sort_kernel  — one workgroup per frame bins the frame's points by cell: LDS histogram (rank of a point inside its
//      cell by LDS atomic), exclusive scan in LDS, points written in cell order -> sorted_pts, cell_start [B][G*G+1].
//   2. pillar_gather_kernel — a wave owns runs of consecutive cells, lane = channel: max of the C features over the cell's
//      points, written straight in the storage type (dense, coalesced 128 B per cell).
//   backward: the same walk recomputes the features, finds the arg-max point(s) of every (cell, channel) and accumulates
//      dgrid * (x, y, z, i, 1) per channel; fixed-order partials -> finalize.  Only occupied cells read dgrid.
// Max is order-independent and the order of the points inside a cell (LDS atomics: arbitrary) only matters when several
// points TIE for a channel's maximum; they share the gradient evenly (as the oracle's scatter_reduce(amax) backward does) and their
// coordinates are summed in fp64 before the one fp32 multiply — exact for any realistic coordinates, hence order-independent:
// outputs and gradients are bitwise reproducible.
// Traffic per 64-frame step: 0.35 GB (was 1.37 GB with the fp32 scatter grid: memset 268 + atomics 300 + cast 403 + backward 401).
// HBM-bound gather.  Build-authored; the upstream reference has no code.
#include "msc_common.h"
#include <atomic>

#define PIL_C 64
#define PIL_BLOCKS 2048
#define PIL_MAX_CELLS 32768      // LDS histogram of one frame: 128 KB

__device__ __forceinline__ int pillar_cell(float x, float y, float lo, float cell, int G) {
  const float fx = floorf((x - lo) / cell), fy = floorf((y - lo) / cell);
  if (!(fx >= 0.f && fx < (float)G && fy >= 0.f && fy < (float)G)) return -1;
  return (int)fy * G + (int)fx;
}

// one workgroup (1024 threads) per frame.  rank: int32 scratch [B*Np] (only used when a thread owns more than PPT points).
// cell_start[b][c] = index into sorted_pts of the first point of cell c (frame b's points occupy [b*Np, b*Np + valid));
// cell_start[b][G*G] = end of the last cell.
// PPT > 0: every thread keeps its <= PPT points, their cells and ranks in registers — all the frame's loads are in flight at once
// (the loop version below walks 16 dependent load -> LDS-atomic rounds per thread and took 117 us for 64 frames).
#define HX(c) ((c) + ((c) >> 4))
template <int PPT>
__global__ __launch_bounds__(1024) void pillar_sort_kernel(const float* __restrict__ pts, float4* __restrict__ sorted_pts,
                                                           int* __restrict__ cell_start, int* __restrict__ rank, int Np, int G,
                                                           float lo, float cell) {
  extern __shared__ int hist[];          // [G*G] padded by one word per 16 (HX): the scan below walks 16-cell runs per thread
  __shared__ int wsum[16];
  const int b = blockIdx.x, ncell = G * G, tid = threadIdx.x;
  const float4* fp = reinterpret_cast<const float4*>(pts) + (long)b * Np;
  constexpr int NR = PPT > 0 ? PPT : 1;
  float4 rp[NR];
  int rc[NR], rr[NR];
  if constexpr (PPT > 0) {
#pragma unroll
    for (int u = 0; u < PPT; ++u) {
      const int p = tid + u * 1024;
      rp[u] = p < Np ? fp[p] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  for (int c = tid; c < ncell; c += 1024) hist[HX(c)] = 0;
  __syncthreads();
  if constexpr (PPT > 0) {
#pragma unroll
    for (int u = 0; u < PPT; ++u) {
      const int p = tid + u * 1024;
      rc[u] = p < Np ? pillar_cell(rp[u].x, rp[u].y, lo, cell, G) : -1;
      rr[u] = rc[u] >= 0 ? atomicAdd(&hist[HX(rc[u])], 1) : -1;
    }
  } else {
    for (int p = tid; p < Np; p += 1024) {
      const float4 pt = fp[p];
      const int c = pillar_cell(pt.x, pt.y, lo, cell, G);
      rank[(long)b * Np + p] = c >= 0 ? atomicAdd(&hist[HX(c)], 1) : -1;
    }
  }
  __syncthreads();
  // exclusive scan of hist: thread t owns cells [t*per, (t+1)*per)
  const int per = (ncell + 1023) / 1024;
  const int c0 = tid * per, c1 = min(ncell, c0 + per);
  int local = 0;
  for (int c = c0; c < c1; ++c) local += hist[HX(c)];
  int incl = local;                      // inclusive scan across the 1024 threads: wave scan, then the 16 wave totals
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(incl, o, 64);
    if ((tid & 63) >= o) incl += v;
  }
  if ((tid & 63) == 63) wsum[tid >> 6] = incl;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < (tid >> 6); ++w) base += wsum[w];
  int run = base + incl - local;         // exclusive prefix of this thread's first cell
  int* cs = cell_start + (long)b * (ncell + 1);
  const int fbase = b * Np;
  for (int c = c0; c < c1; ++c) {
    const int n = hist[HX(c)];
    hist[HX(c)] = run;
    cs[c] = fbase + run;
    run += n;
  }
  if (tid == 1023) cs[ncell] = fbase + run;    // thread 1023 owns the last cells (or none): run = number of binned points
  __syncthreads();
  if constexpr (PPT > 0) {
#pragma unroll
    for (int u = 0; u < PPT; ++u)
      if (rr[u] >= 0) sorted_pts[fbase + hist[HX(rc[u])] + rr[u]] = rp[u];
  } else {
    for (int p = tid; p < Np; p += 1024) {
      const int r = rank[(long)b * Np + p];
      if (r < 0) continue;
      const float4 pt = fp[p];
      const int c = pillar_cell(pt.x, pt.y, lo, cell, G);
      sorted_pts[fbase + hist[HX(c)] + r] = pt;
    }
  }
}

// Cell walkers: a 16-lane group owns a cell (4 channels per lane: 8-byte bf16 / 16-byte fp32 accesses), so a wave has four
// cells in flight; a wave owns PIL_CPW consecutive cells of the flattened [B*G*G] index.
#define PIL_CPW 16
struct PilW { float w[4][4]; float b[4]; };
__device__ __forceinline__ void pil_load_w(const float* __restrict__ w, const float* __restrict__ bias, int ch0, PilW& o) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float4 t = *reinterpret_cast<const float4*>(w + (ch0 + j) * 4);
    o.w[j][0] = t.x; o.w[j][1] = t.y; o.w[j][2] = t.z; o.w[j][3] = t.w;
    o.b[j] = bias[ch0 + j];
  }
}
__device__ __forceinline__ float pil_feat(const PilW& k, int j, const float4& pt) {
  return fmaf(k.w[j][3], pt.w, fmaf(k.w[j][2], pt.z, fmaf(k.w[j][1], pt.y, k.w[j][0] * pt.x))) + k.b[j];
}
template <typename T> __device__ __forceinline__ void pil_store4(T* p, const float (&v)[4]);
template <> __device__ __forceinline__ void pil_store4<float>(float* p, const float (&v)[4]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
template <> __device__ __forceinline__ void pil_store4<bf16_t>(bf16_t* p, const float (&v)[4]) {
  uint2 o;
  o.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
  o.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
  *reinterpret_cast<uint2*>(p) = o;
}
template <typename T> __device__ __forceinline__ void pil_load4(const T* p, float (&v)[4]);
template <> __device__ __forceinline__ void pil_load4<float>(const float* p, float (&v)[4]) {
  const float4 t = *reinterpret_cast<const float4*>(p);
  v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
template <> __device__ __forceinline__ void pil_load4<bf16_t>(const bf16_t* p, float (&v)[4]) {
  const uint2 t = *reinterpret_cast<const uint2*>(p);
  v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
  v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
}

template <typename T>
__global__ __launch_bounds__(256) void pillar_gather_kernel(const float4* __restrict__ sorted_pts, const int* __restrict__ cell_start,
                                                            const float* __restrict__ w, const float* __restrict__ bias,
                                                            T* __restrict__ out, long ncells_total, int ncell) {
  const int lane = threadIdx.x & 63, grp = lane >> 4, ch0 = (lane & 15) * 4;
  PilW k;
  pil_load_w(w, bias, ch0, k);
  const long cbeg = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * PIL_CPW;
  if (cbeg >= ncells_total) return;
  // boundaries of the run's cells: lane q < PIL_CPW loads cell cbeg + q's [start, end) (each frame's row has G*G + 1 entries)
  int s0 = 0, e0 = 0;
  if (lane < PIL_CPW && cbeg + lane < ncells_total) {
    const long ci = cbeg + lane, fb = ci / ncell, cc = ci - fb * ncell;
    s0 = cell_start[fb * (ncell + 1) + cc];
    e0 = cell_start[fb * (ncell + 1) + cc + 1];
  }
#pragma unroll
  for (int j = 0; j < PIL_CPW / 4; ++j) {
    const int q = 4 * j + grp;
    const int s = __shfl(s0, q, 64), e = __shfl(e0, q, 64);
    const long ci = cbeg + q;
    float m[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = s; i < e; ++i) {
      const float4 pt = sorted_pts[i];
#pragma unroll
      for (int c = 0; c < 4; ++c) m[c] = fmaxf(m[c], pil_feat(k, c, pt));
    }
    if (ci < ncells_total) pil_store4<T>(out + ci * PIL_C + ch0, m);
  }
}

// partial[blk][c][5] = sum over the block's cells of g * (x,y,z,i,1) at the arg-max point(s), g = dgrid[cell][c].
// lane = channel.  The occupied cells of a 64-cell chunk are taken four at a time: their (wave-uniform) bounds come from a
// ballot walk, then the four dgrid rows and first points are loaded back to back before any of them is used — the walk is a
// chain of dependent ~2 us loads otherwise (one cell at a time: 224 us for 64 frames).
template <typename T>
__global__ __launch_bounds__(256) void pillar_bwd_kernel(const float4* __restrict__ sorted_pts, const int* __restrict__ cell_start,
                                                         const float* __restrict__ w, const float* __restrict__ bias,
                                                         const T* __restrict__ dgrid, float* __restrict__ partial, long ncells_total,
                                                         int ncell, int cells_per_wave) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float w0 = w[lane * 4 + 0], w1 = w[lane * 4 + 1], w2 = w[lane * 4 + 2], w3 = w[lane * 4 + 3], bb = bias[lane];
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f;
  const long cbeg = ((long)blockIdx.x * 4 + wave) * cells_per_wave;
  const long cend = min(ncells_total, cbeg + cells_per_wave);
  for (long c64 = cbeg; c64 < cend; c64 += 64) {
    const long ci0 = c64 + lane;        // lane q: bounds of cell c64 + q
    int s0 = 0, e0 = 0;
    if (ci0 < cend) {
      const long fb = ci0 / ncell, cc = ci0 - fb * ncell;
      s0 = cell_start[fb * (ncell + 1) + cc];
      e0 = cell_start[fb * (ncell + 1) + cc + 1];
    }
    unsigned long long occ = __ballot(e0 > s0);
    while (occ) {
      int ss[4], nn[4];
      float g[4];
      float4 p0[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        nn[u] = 0;
        ss[u] = 0;
        if (occ) {
          const int q = __ffsll((long long)occ) - 1;
          occ &= occ - 1;
          ss[u] = __builtin_amdgcn_readfirstlane(__shfl(s0, q, 64));
          nn[u] = __builtin_amdgcn_readfirstlane(__shfl(e0, q, 64)) - ss[u];
          g[u] = msc_traits<T>::ld(dgrid + (c64 + q) * PIL_C + lane);
          p0[u] = sorted_pts[ss[u]];
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (nn[u] == 0) continue;
        const float f0 = fmaf(w3, p0[u].w, fmaf(w2, p0[u].z, fmaf(w1, p0[u].y, w0 * p0[u].x))) + bb;
        if (nn[u] == 1) {                 // the common case: one point in the cell
          if (f0 > 0.f) { a0 += g[u] * p0[u].x; a1 += g[u] * p0[u].y; a2 += g[u] * p0[u].z; a3 += g[u] * p0[u].w; a4 += g[u]; }
          continue;
        }
        float m = fmaxf(0.f, f0);
        for (int i = 1; i < nn[u]; ++i) {
          const float4 pt = sorted_pts[ss[u] + i];
          m = fmaxf(m, fmaf(w3, pt.w, fmaf(w2, pt.z, fmaf(w1, pt.y, w0 * pt.x))) + bb);
        }
        if (m > 0.f) {
          // the points that attain the maximum: their coordinates are summed in fp64 (exact, hence order-independent) ...
          double sx = 0.0, sy = 0.0, sz = 0.0, sw = 0.0;
          int cnt = 0;
          for (int i = 0; i < nn[u]; ++i) {
            const float4 pt = sorted_pts[ss[u] + i];
            const float f = fmaf(w3, pt.w, fmaf(w2, pt.z, fmaf(w1, pt.y, w0 * pt.x))) + bb;
            if (f == m) { sx += (double)pt.x; sy += (double)pt.y; sz += (double)pt.z; sw += (double)pt.w; ++cnt; }
          }
          // ... and share it evenly (the oracle's scatter_reduce(amax) backward: grad / number of maxima)
          const double inv = 1.0 / (double)cnt;
          a0 += g[u] * (float)(sx * inv); a1 += g[u] * (float)(sy * inv); a2 += g[u] * (float)(sz * inv); a3 += g[u] * (float)(sw * inv); a4 += g[u];
        }
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

// points: fp32 [B][Np][4] (x,y,z,intensity); w: fp32 [64][4]; bias fp32 [64]
// sorted_pts: fp32 [B*Np][4], cell_start: int32 [B][G*G+1], rank: int32 [B*Np] (scratch) — written here, read by the backward
// grid_out: T [B][G][G][64], fed to the BEV convs
extern "C" int64_t msc_pillar_index_bytes(int B, int G) { return (int64_t)B * ((int64_t)G * G + 1) * 4; }

extern "C" int msc_pillar_fwd(const float* points, const float* w, const float* bias, float* sorted_pts, int32_t* cell_start, int32_t* rank,
                              void* grid_out, int B, int Np, int G, int C, float range_lo, float cell, int dtype, void* stream) {
  MSC_CHECK_ARG(points && w && bias && sorted_pts && cell_start && rank && grid_out && B > 0 && Np > 0 && G > 0 && C == PIL_C && cell > 0.f);
  MSC_CHECK_ARG(dtype == MSC_F32 || dtype == MSC_BF16);
  if ((long)G * G > PIL_MAX_CELLS || (long)B * Np >= 0x7fffffffL) return MSC_ERR_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  const int ncell = G * G;
  // up to 16 points per thread stay in registers (Np <= 16384); larger frames take the loop version
  // the attribute is per device: keyed on the current device (a bitmask under an atomic, so concurrent first calls from two host
  // threads at worst both set it — the call is idempotent)
  static std::atomic<uint64_t> attr_set{0};
  int devid = 0;
  if (hipGetDevice(&devid) != hipSuccess || devid < 0 || devid > 63) return MSC_ERR_LAUNCH;
  if (!((attr_set.load(std::memory_order_acquire) >> devid) & 1ull)) {
    if (hipFuncSetAttribute((const void*)pillar_sort_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (PIL_MAX_CELLS + PIL_MAX_CELLS / 16 + 1) * 4) != hipSuccess ||
        hipFuncSetAttribute((const void*)pillar_sort_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (PIL_MAX_CELLS + PIL_MAX_CELLS / 16 + 1) * 4) != hipSuccess ||
        hipFuncSetAttribute((const void*)pillar_sort_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (PIL_MAX_CELLS + PIL_MAX_CELLS / 16 + 1) * 4) != hipSuccess)
      return MSC_ERR_LAUNCH;
    attr_set.fetch_or(1ull << devid, std::memory_order_release);
  }
  float4* sp = reinterpret_cast<float4*>(sorted_pts);
  if (Np <= 4 * 1024)
    hipLaunchKernelGGL(pillar_sort_kernel<4>, dim3(B), dim3(1024), (size_t)(ncell + ncell / 16 + 1) * 4, st, points, sp, cell_start, rank, Np, G, range_lo, cell);
  else if (Np <= 16 * 1024)
    hipLaunchKernelGGL(pillar_sort_kernel<16>, dim3(B), dim3(1024), (size_t)(ncell + ncell / 16 + 1) * 4, st, points, sp, cell_start, rank, Np, G, range_lo, cell);
  else
    hipLaunchKernelGGL(pillar_sort_kernel<0>, dim3(B), dim3(1024), (size_t)(ncell + ncell / 16 + 1) * 4, st, points, sp, cell_start, rank, Np, G, range_lo, cell);
  const long ncells_total = (long)B * ncell;
  const unsigned blocks = (unsigned)cdiv(ncells_total, 4 * PIL_CPW);
  if (dtype == MSC_F32)
    hipLaunchKernelGGL(pillar_gather_kernel<float>, dim3(blocks), dim3(256), 0, st, reinterpret_cast<const float4*>(sorted_pts), cell_start, w,
                       bias, (float*)grid_out, ncells_total, ncell);
  else
    hipLaunchKernelGGL(pillar_gather_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, reinterpret_cast<const float4*>(sorted_pts), cell_start, w,
                       bias, (bf16_t*)grid_out, ncells_total, ncell);
  return msc_launch_status();
}

extern "C" int64_t msc_pillar_bwd_workspace_bytes(void) { return (int64_t)PIL_BLOCKS * PIL_C * 5 * 4; }

extern "C" int msc_pillar_bwd(const float* sorted_pts, const int32_t* cell_start, const float* w, const float* bias, const void* dgrid,
                              float* dw, float* db, float* ws, int B, int G, int C, int accumulate, int dtype, void* stream) {
  MSC_CHECK_ARG(sorted_pts && cell_start && w && bias && dgrid && dw && db && ws && B > 0 && G > 0 && C == PIL_C);
  hipStream_t st = (hipStream_t)stream;
  const int ncell = G * G;
  const long ncells_total = (long)B * ncell;
  // waves walk multiples of 64 cells; as many blocks as keep the partial buffer's bound
  long cpw = 64;
  while (cdiv(ncells_total, cpw * 4) > PIL_BLOCKS) cpw += 64;
  const int blocks = (int)cdiv(ncells_total, cpw * 4);
  if (dtype == MSC_F32)
    hipLaunchKernelGGL(pillar_bwd_kernel<float>, dim3(blocks), dim3(256), 0, st, reinterpret_cast<const float4*>(sorted_pts), cell_start, w, bias,
                       (const float*)dgrid, ws, ncells_total, ncell, (int)cpw);
  else if (dtype == MSC_BF16)
    hipLaunchKernelGGL(pillar_bwd_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, reinterpret_cast<const float4*>(sorted_pts), cell_start, w,
                       bias, (const bf16_t*)dgrid, ws, ncells_total, ncell, (int)cpw);
  else
    return MSC_ERR_BAD_ARG;
  hipLaunchKernelGGL(pillar_bwd_finalize_kernel, dim3(cdiv(PIL_C * 5, 32)), dim3(1024), 0, st, ws, blocks, dw, db, accumulate);
  return msc_launch_status();
}
