// conv_stem.hip — weight gradient of the ImageEncoder stem in space-to-depth form (4x4 / stride 1 over 16 channels -> 64, 112 x 112 maps), gfx950.
// Build-authored (the upstream reference has no code).
//
// dW[co][kh][kw][c] = sum over pixels of dy[p][co] * xs[p + (kh - 2, kw - 2)][c]: a 64 x 256 output with a reduction over 802 816 pixels (64
// frames).  It is the LAST kernel of the backward — nothing is left on the launching stream to hide it behind — and as a gathered split-K
// GEMM (gemm.hip: 64 x 128 tiles, every dy row staged twice, the 4 x 4 taps of a pixel gathered as sixteen 32-byte runs) it took 92 us + a
// 7 us reduce.  Here both operands are staged ONCE per tile by dense LDS-DMA and the accumulators never leave the registers:
//   * persistent 4-wave workgroups, one per CU; a tile = 2 image rows x 112 pixels = 224 pixels = 7 k-steps of 32 pixels; per tile the 28 KB of
//     dy (contiguous in memory) and the 5 x 115-pixel halo of xs (18 KB; zero padding by the DMA's bounds check) are double-buffered in LDS;
//   * the reduction index is the PIXEL, i.e. both MFMA operands are reduction-major in memory: fragments come from the row-major LDS images by
//     the hardware transpose read ds_read_b64_tr_b16 (a 16-lane group turns 4 pixel rows x 16 channels into 4 consecutive k per lane);
//   * wave w owns filter row kh = w: 4 taps (kw) x all 64 output channels = 16 accumulators of 16 x 16 that live in registers across ALL the
//     workgroup's tiles (14 at 64 frames); per k-step 8 + 8 transposed reads feed 16 MFMAs;
//   * K-slot (lane group g, element e) of a k-step <-> pixel 16 (e >> 2) + 4 g + (e & 3): the two groups of a 32-lane half read 8
//     consecutive pixels, conflict-free on the 32-byte-pitch halo; the dy image (128-byte pitch) swaps its four 32-byte channel groups by
//     (pixel ^ (pixel >> 2)) & 3, applied on the DMA's source side;
//   * at the end each workgroup stores its 64 x 256 fp32 partial; the split-K reduce of gemm.hip sums them in workgroup order.
#include "msc_common.h"
#include "gemm_plan.h"
#include <type_traits>

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 sw_bf16x8;
typedef __attribute__((ext_vector_type(4))) float sw_f32x4;
typedef __attribute__((ext_vector_type(4))) short sw_s16x4;
typedef __attribute__((ext_vector_type(8))) short sw_s16x8;

#define SW_OOB 0x80000000u
#define SW_LDS(p) ((__attribute__((address_space(3))) void*)(p))

constexpr int SW_W = 112, SW_R = 2;                       // map width, image rows per tile
constexpr int SW_PIX = SW_W * SW_R;                       // 224 pixels = 7 k-steps of 32
constexpr int SW_KS = SW_PIX / 32;
constexpr int SW_HW = SW_W + 3, SW_HR = SW_R + 3;         // halo: 115 x 5 pixels (2 before, 1 after)
constexpr int SW_HPIX = SW_HW * SW_HR;                    // 575
constexpr int SW_NH = (SW_HPIX * 32 + 1023) / 1024;       // 18 DMA instructions (32 halo pixels each)
constexpr int SW_ND = SW_PIX * 128 / 1024;                // 28 DMA instructions (8 dy pixels each)
constexpr int SW_HALO_BYTES = SW_NH * 1024, SW_DY_BYTES = SW_ND * 1024;
constexpr int SW_BUF_BYTES = SW_HALO_BYTES + SW_DY_BYTES; // 47104

struct SWParams {
  const void* xs; const void* dy; float* slab;            // slab: [gridDim.x][64][256] fp32
  int B, H;                                               // W = 112, H % 2 == 0
};

__device__ __forceinline__ int sw_key(int q) { return (q ^ (q >> 2)) & 3; }

__global__ __launch_bounds__(256, 1) void stem_wgrad_kernel(const SWParams p) {
  __shared__ __attribute__((aligned(1024))) char lds[2 * SW_BUF_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // = filter row kh
  const int g = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
  const int tiles_per_img = p.H / SW_R;
  const int ntiles = p.B * tiles_per_img;

  // ---- DMA side.  xs descriptor base = xs - (2 rows + 2 pixels): halo pixel (hr, hc) of the tile at image row r0 sits at ((b H + r0) W + hr W + hc) * 32.
  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(p.xs) - (2 * SW_W + 2) * 32), 0, (int)0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, (int)0x7ffffff0, 0x00020000);
  constexpr int NHW = (SW_NH + 3) / 4;                    // halo instructions per wave (waves 0, 1: 5; waves 2, 3: 4)
  uint32_t hrel[NHW];
  int hrow[NHW];
#pragma unroll
  for (int i = 0; i < NHW; ++i) {
    const int s = 64 * (wave + 4 * i) + lane;             // 16-byte slot of the halo image: pixel s >> 1, half s & 1
    const int P = s >> 1, hr = P / SW_HW, hc = P - hr * SW_HW;
    const bool ok = P < SW_HPIX && hc >= 2 && hc < SW_W + 2;
    hrel[i] = ok ? (uint32_t)((hr * SW_W + hc) * 32 + (s & 1) * 16) : SW_OOB;
    hrow[i] = hr;
  }
  constexpr int NDW = SW_ND / 4;                          // dy instructions per wave: 7
  uint32_t drel[NDW];
#pragma unroll
  for (int i = 0; i < NDW; ++i) {
    const int s = 64 * (wave + 4 * i) + lane;             // 16-byte slot of the dy image: pixel s >> 3, position s & 7
    const int q = s >> 3, pos = s & 7;
    const int chunk = ((((pos >> 1) ^ sw_key(q)) << 1) | (pos & 1));     // the source chunk that belongs at this position
    drel[i] = (uint32_t)(q * 128 + chunk * 16);
  }
  auto stage = [&](int tile, int buf) __attribute__((always_inline)) {
    const int b = tile / tiles_per_img, r0 = (tile - b * tiles_per_img) * SW_R;
    const uint32_t pix0 = (uint32_t)((b * p.H + r0) * SW_W);
    char* base = lds + buf * SW_BUF_BYTES;
#pragma unroll
    for (int i = 0; i < NHW; ++i) {
      if (wave + 4 * i < SW_NH) {
        const int row = r0 - 2 + hrow[i];
        const bool dead = row < 0 || row >= p.H;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, SW_LDS(base + (wave + 4 * i) * 1024), 16, dead ? SW_OOB : hrel[i], pix0 * 32u, 0, 0);
      }
    }
#pragma unroll
    for (int i = 0; i < NDW; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsd, SW_LDS(base + SW_HALO_BYTES + (wave + 4 * i) * 1024), 16, drel[i], pix0 * 128u, 0, 0);
  };

  // ---- fragment addresses (tile-invariant).  Block blk (0 / 1) of k-step ks covers pixels q = 32 ks + 16 blk + 4 g + qq; the lane supplies the
  // address of pixel row qq of its group's block, 8 bytes at channel 4 pp (transposed read: lane i of the group receives channel i of the 4 rows)
  uint32_t abase[SW_KS][2];                               // dy image: q * 128 + 8 pp   (+ ((j ^ key) << 5) per channel group j)
  int akey[SW_KS][2];
  uint32_t bbase[SW_KS][2];                               // halo image: ((r + kh) * 115 + c) * 32 + 8 pp   (+ 32 kw)
#pragma unroll
  for (int ks = 0; ks < SW_KS; ++ks)
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      const int q = 32 * ks + 16 * blk + 4 * g + qq;
      const int r = q / SW_W, c = q - r * SW_W;
      abase[ks][blk] = (uint32_t)(SW_HALO_BYTES + q * 128 + 8 * pp);
      akey[ks][blk] = sw_key(q);
      bbase[ks][blk] = (uint32_t)(((r + wave) * SW_HW + c) * 32 + 8 * pp);
    }

  sw_f32x4 acc[4][4];                                     // [channel group j][kw]
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int kw = 0; kw < 4; ++kw) acc[j][kw] = sw_f32x4{0.f, 0.f, 0.f, 0.f};

  int tile = (int)blockIdx.x;
  if (tile < ntiles) stage(tile, 0);
  int buf = 0;
  for (; tile < ntiles; tile += (int)gridDim.x, buf ^= 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this tile's images have landed
    __syncthreads();                                      // ... for every wave; and nobody still reads the other buffer
    if (tile + (int)gridDim.x < ntiles) stage(tile + (int)gridDim.x, buf ^ 1);
    const char* img = lds + buf * SW_BUF_BYTES;
    // Transposed reads as inline assembly (see csrc/gemm.hip ds_read_tr_asm: through the builtin the compiler waits for the NEXT tile's
    // LDS-DMA, issued just above, before the first read — staging and MFMAs of a tile one after the other).  The reads' completion is waited
    // for by hand: the halo operand of tap kw + 1 is in flight under the MFMAs of tap kw.
    auto tr_issue = [&](sw_s16x4& lo, sw_s16x4& hi, const char* a0, const char* a1) __attribute__((always_inline)) {
      const uint32_t u0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)a0;
      const uint32_t u1 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)a1;
      asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(u0));
      asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"(u1));
    };
    auto join = [](const sw_s16x4& lo, const sw_s16x4& hi) __attribute__((always_inline)) {
      return __builtin_bit_cast(sw_bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
#pragma unroll
    for (int ks = 0; ks < SW_KS; ++ks) {
      sw_s16x4 alo[4], ahi[4], blo[2], bhi[2];
      sw_bf16x8 af[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) tr_issue(alo[j], ahi[j], img + abase[ks][0] + ((j ^ akey[ks][0]) << 5), img + abase[ks][1] + ((j ^ akey[ks][1]) << 5));
      tr_issue(blo[0], bhi[0], img + bbase[ks][0], img + bbase[ks][1]);
#pragma unroll
      for (int kw = 0; kw < 4; ++kw) {
        if (kw < 3) {
          tr_issue(blo[(kw + 1) & 1], bhi[(kw + 1) & 1], img + bbase[ks][0] + 32 * (kw + 1), img + bbase[ks][1] + 32 * (kw + 1));
          asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        asm volatile("" : "+v"(blo[kw & 1]), "+v"(bhi[kw & 1]));
        if (kw == 0) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            asm volatile("" : "+v"(alo[j]), "+v"(ahi[j]));
            af[j] = join(alo[j], ahi[j]);
          }
        }
        const sw_bf16x8 bf = join(blo[kw & 1], bhi[kw & 1]);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j][kw] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[j], bf, acc[j][kw], 0, 0, 0);
      }
    }
  }

  // ---- the workgroup's partial: accumulator [j][kw] register r of lane l = dW[co = 16 j + 4 (l >> 4) + r][kh = wave][kw][c = l & 15]
  float* out = p.slab + (long)blockIdx.x * 64 * 256;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int kw = 0; kw < 4; ++kw)
#pragma unroll
      for (int r = 0; r < 4; ++r) out[(16 * j + 4 * g + r) * 256 + (wave * 4 + kw) * 16 + (lane & 15)] = acc[j][kw][r];
}

// ================================================================================================ forward
// y[p][co] = sum over the 4 x 4 taps and 16 channels of xs[p + (kh - 2, kw - 2)][c] * w[co][kh][kw][c] (+ per-tile BatchNorm partial sums).  As an
// implicit GEMM (M = 802 816 pixels, N = 64, K = 256) it was 77 us for 26 GFLOP, every pixel's 32 bytes gathered sixteen times.  Scheme of
// conv_c64.hip: the wave's 32 x 256 weight block in 64 registers for the whole kernel, the tile's halo staged once (the same 5 x 115-pixel halo
// as the weight gradient above), v_mfma_f32_16x16x32_bf16 issued transposed with permuted weight rows (a lane stores 8 consecutive channels).
// A k-step is two taps x 16 channels: lane group lg reads tap 2 ks + (lg >> 1), channels 8 (lg & 1) .. + 7 of its pixel — one ds_read_b128 at
// base(pixel, lg) + a per-k-step immediate (the halo has no swizzle: 32-byte pixels, 2-way bank conflicts accepted).
constexpr int SF_NF = SW_PIX / 16, SF_FW = SF_NF / 2;     // 14 pixel fragments per tile, 7 per wave
constexpr int SF_KS = 8;                                  // k-steps of 32 = 2 taps x 16 channels

struct SFParams {
  const void* xs; const void* w; void* y;
  float* part; int part_rows;                             // [part_rows][2][64]
  int B, H;
};

typedef __attribute__((ext_vector_type(4))) uint32_t sw_u32x4;

__device__ __forceinline__ float sf_row_sum(float v) {
  auto dpp = [](float x, auto ctrl) __attribute__((always_inline)) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xF, 0xF, false));
  };
  v += dpp(v, std::integral_constant<int, 0xB1>{});
  v += dpp(v, std::integral_constant<int, 0x4E>{});
  v += dpp(v, std::integral_constant<int, 0x124>{});
  v += dpp(v, std::integral_constant<int, 0x128>{});
  return v;
}

__global__ __launch_bounds__(256, 2) void stem_fwd_kernel(const SFParams p) {
  __shared__ __attribute__((aligned(1024))) char lds[2 * SW_HALO_BYTES];
  __shared__ float red[2][2][64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 15, lg = lane >> 4;
  const int tiles_per_img = p.H / SW_R;
  const int ntiles = p.B * tiles_per_img;

  // ---- halo DMA (as in the weight-gradient kernel)
  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(p.xs) - (2 * SW_W + 2) * 32), 0, (int)0x7ffffff0, 0x00020000);
  constexpr int NHW = (SW_NH + 3) / 4;
  uint32_t hrel[NHW];
  int hrow[NHW];
#pragma unroll
  for (int i = 0; i < NHW; ++i) {
    const int s = 64 * (wave + 4 * i) + lane;
    const int P = s >> 1, hr = P / SW_HW, hc = P - hr * SW_HW;
    const bool ok = P < SW_HPIX && hc >= 2 && hc < SW_W + 2;
    hrel[i] = ok ? (uint32_t)((hr * SW_W + hc) * 32 + (s & 1) * 16) : SW_OOB;
    hrow[i] = hr;
  }
  auto stage = [&](int tile, int buf) __attribute__((always_inline)) {
    const int b = tile / tiles_per_img, r0 = (tile - b * tiles_per_img) * SW_R;
    const uint32_t pix0 = (uint32_t)((b * p.H + r0) * SW_W);
#pragma unroll
    for (int i = 0; i < NHW; ++i) {
      if (wave + 4 * i < SW_NH) {
        const int row = r0 - 2 + hrow[i];
        const bool dead = row < 0 || row >= p.H;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, SW_LDS(lds + buf * SW_HALO_BYTES + (wave + 4 * i) * 1024), 16, dead ? SW_OOB : hrel[i], pix0 * 32u, 0, 0);
      }
    }
  };
  int tile = (int)blockIdx.x;
  if (tile < ntiles) stage(tile, 0);

  // ---- weights -> registers: MFMA row i of channel fragment j <- channel 32 wn + 8 (i >> 2) + 4 j + (i & 3)
  sw_bf16x8 wreg[2][SF_KS];
  {
    const bf16_t* W = reinterpret_cast<const bf16_t*>(p.w);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int ch = 32 * wn + 8 * (lr >> 2) + 4 * j + (lr & 3);
#pragma unroll
      for (int ks = 0; ks < SF_KS; ++ks) wreg[j][ks] = *reinterpret_cast<const sw_bf16x8*>(W + ch * 256 + ks * 32 + 8 * lg);
    }
  }
  // ---- pixel fragment bases: pixel q = 16 (7 wm + f) + lr -> halo byte ((q / 112) * 115 + q % 112) * 32 + 16 lg  (tap (0, 0); + (kh * 115 + kw) * 32 per k-step)
  uint32_t fbase[SF_FW];
#pragma unroll
  for (int f = 0; f < SF_FW; ++f) {
    const int q = 16 * (SF_FW * wm + f) + lr;
    const int r = q / SW_W, c = q - r * SW_W;
    fbase[f] = (uint32_t)((r * SW_HW + c) * 32 + 16 * lg);
  }
  const uint32_t out_lane = (uint32_t)(32 * wn + 8 * lg) * 2u;

  int buf = 0;
  bool first = true;
  for (; tile < ntiles; tile += (int)gridDim.x, buf ^= 1) {
    if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(7)" ::: "memory");          // the previous tile's 7 (+1) stores may stay in flight: the halo DMA is older
    first = false;
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) stage(tile + (int)gridDim.x, buf ^ 1);
    const char* img = lds + buf * SW_HALO_BYTES;

    sw_f32x4 acc[SF_FW][2];
#pragma unroll
    for (int f = 0; f < SF_FW; ++f) { acc[f][0] = sw_f32x4{0.f, 0.f, 0.f, 0.f}; acc[f][1] = sw_f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int ks = 0; ks < SF_KS; ++ks) {
      const int koff = ((ks >> 1) * SW_HW + 2 * (ks & 1)) * 32;      // taps (kh = ks >> 1, kw = 2 (ks & 1) + (lg >> 1))
#pragma unroll
      for (int f = 0; f < SF_FW; ++f) {
        const sw_bf16x8 xf = *reinterpret_cast<const sw_bf16x8*>(img + fbase[f] + koff);
        acc[f][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[0][ks], xf, acc[f][0], 0, 0, 0);
        acc[f][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[1][ks], xf, acc[f][1], 0, 0, 0);
      }
    }

    // ---- epilogue: the lane holds channels 32 wn + 8 lg .. + 7 of pixel 16 (7 wm + f) + lr
    const int b = tile / tiles_per_img, r0 = (tile - b * tiles_per_img) * SW_R;
    const long pix0 = ((long)b * p.H + r0) * SW_W;
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
#pragma unroll
    for (int f = 0; f < SF_FW; ++f) {
      const long o = (pix0 + 16 * (SF_FW * wm + f) + lr) * 128 + out_lane;
      float v[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = acc[f][0][e]; v[4 + e] = acc[f][1][e]; }
      if (p.part) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { s1[e] += v[e]; s2[e] = fmaf(v[e], v[e], s2[e]); }
      }
      sw_u32x4 wv;
#pragma unroll
      for (int e = 0; e < 4; ++e) wv[e] = (uint32_t)f32_to_bf16(v[2 * e]) | ((uint32_t)f32_to_bf16(v[2 * e + 1]) << 16);
      *reinterpret_cast<sw_u32x4*>(reinterpret_cast<char*>(p.y) + o) = wv;
    }
    if (p.part) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { s1[e] = sf_row_sum(s1[e]); s2[e] = sf_row_sum(s2[e]); }
      if (lr == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { red[wm][0][32 * wn + 8 * lg + e] = s1[e]; red[wm][1][32 * wn + 8 * lg + e] = s2[e]; }
      }
      __syncthreads();
      if (tid < 128) {
        const int st = tid >> 6, ch = tid & 63;
        p.part[(long)tile * 128 + st * 64 + ch] = red[0][st][ch] + red[1][st][ch];
      }
    }
  }
  if (p.part) {       // rows of the caller's partial buffer this kernel has no tile for: zero (msc_bn_finalize sums every row)
    const long n0 = (long)ntiles * 128, n1 = (long)p.part_rows * 128;
    for (long i = n0 + (long)blockIdx.x * 256 + tid; i < n1; i += (long)gridDim.x * 256) p.part[i] = 0.f;
  }
}

}  // namespace

bool msc_stem_wgrad_halo_ok(int B, int Hs, int Ws, int Cout, int dtype, int64_t ws_bytes) {
  if (!mscplan::g_knobs.stem_wgrad_halo || mscplan::g_knobs.tile_bm) return false;
  if (dtype != MSC_BF16 || Cout != 64 || Ws != SW_W || Hs % SW_R || Hs < SW_R || B < 1) return false;
  if ((long)B * Hs * Ws * 128 >= 0x7ffffff0L) return false;
  return ws_bytes >= (int64_t)64 * 256 * 4;               // room for at least one partial
}

// launches the kernel with as many workgroups as CUs, tiles and slab room allow; returns that count (< 0: error code)
int msc_stem_wgrad_halo_launch(const void* xs, const void* dy, float* slab, int64_t slab_bytes, int B, int Hs, int ncu, hipStream_t st) {
  if ((((uintptr_t)xs | (uintptr_t)dy | (uintptr_t)slab) & 15) != 0) return -MSC_ERR_BAD_ARG;
  const int ntiles = B * (Hs / SW_R);
  long nwg = ntiles < ncu ? ntiles : ncu;
  const long room = slab_bytes / ((long)64 * 256 * 4);
  if (nwg > room) nwg = room;
  if (nwg > 0) { const long rounds = (ntiles + nwg - 1) / nwg; nwg = (ntiles + rounds - 1) / rounds; }     // fewest workgroups for that many rounds of tiles
  SWParams p;
  p.xs = xs; p.dy = dy; p.slab = slab; p.B = B; p.H = Hs;
  hipLaunchKernelGGL(stem_wgrad_kernel, dim3((unsigned)nwg), dim3(256), 0, st, p);
  return msc_launch_status() == MSC_OK ? (int)nwg : -MSC_ERR_LAUNCH;
}

bool msc_stem_fwd_halo_ok(int B, int Hs, int Ws, int Cout, int dtype) {
  if (!mscplan::g_knobs.stem_fwd_halo || mscplan::g_knobs.tile_bm) return false;
  if (dtype != MSC_BF16 || Cout != 64 || Ws != SW_W || Hs % SW_R || Hs < SW_R || B < 1) return false;
  return (long)B * Hs * Ws * 128 < 0x7ffffff0L;
}

int msc_stem_fwd_halo_launch(const void* xs, const void* w, void* y, float* part, int part_rows, int B, int Hs, int ncu, hipStream_t st) {
  if ((((uintptr_t)xs | (uintptr_t)w | (uintptr_t)y) & 15) != 0) return MSC_ERR_BAD_ARG;
  const int ntiles = B * (Hs / SW_R);
  if (part && part_rows < ntiles) return MSC_ERR_BAD_ARG;
  SFParams p;
  p.xs = xs; p.w = w; p.y = y; p.part = part; p.part_rows = part_rows; p.B = B; p.H = Hs;
  const int nwg = ntiles < 2 * ncu ? ntiles : 2 * ncu;      // two workgroups per CU (36 KB of LDS, <= 128 registers each)
  hipLaunchKernelGGL(stem_fwd_kernel, dim3((unsigned)nwg), dim3(256), 0, st, p);
  return msc_launch_status();
}
