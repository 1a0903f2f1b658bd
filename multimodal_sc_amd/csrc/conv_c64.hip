// conv_c64.hip — 3x3 / stride 1 / pad 1 convolution over 64 -> 64 channels on 56-pixel-wide NHWC bf16 maps (gfx950): forward and data gradient of
// the four convolutions of ResNet-18's layer 1.  Build-authored (the upstream reference has no code).
//
// As an implicit GEMM (gemm.hip) these launches are M = 200704 pixels x N = 64 x K = 576: the 64-wide output caps the tile at 43 FLOP per
// staged byte, every pixel's 128 bytes are staged nine times (once per filter tap), and a workgroup lives for nine K steps — 34 us forward,
// 39 us data gradient per launch, 26 % of the workgroup's life in prologue and epilogue (clock stamps: profiles/r4_conv_stamps.txt).
// Here the input is staged ONCE and the weights never pass through LDS at all:
//   * a workgroup (4 waves, one per SIMD, persistent over the tiles) owns an output tile of 4 image rows x 56 pixels = 224 pixels = 14
//     fragments of 16; its input is the 6 x 58 pixel halo of that tile (44 KB, zero padding supplied by the buffer bounds check of the
//     LDS-DMA), double-buffered: the next tile's halo lands while this one is multiplied;
//   * wave (wm, wn) computes pixel fragments 7 wm .. 7 wm + 6 for output channels 32 wn .. 32 wn + 31 and keeps the WHOLE 32 x 576 weight
//     block of those channels in registers for the lifetime of the kernel (144 VGPRs in MFMA A-fragment layout), so the K loop is
//     "for each tap, each half of the 64 input channels: 7 ds_read_b128 of shifted halo pixels, 14 MFMAs";
//   * v_mfma_f32_16x16x32_bf16 is issued transposed (A = weights, B = pixels), and the weight rows are assigned to MFMA rows so that a lane
//     ends up with 8 CONSECUTIVE output channels of one pixel: one 16-byte store per fragment;
//   * LDS halo layout [pixel][8 chunks of 16 B], chunk position = chunk ^ (pixel & 7): the 16 lanes one ds_read_b128 cycle serves (8 consecutive
//     pixels at one logical chunk, 8 at the next: MI355X_MICROARCH.md, LDS) hit 16 different 16-byte slots of the 256-byte LDS row for every
//     starting pixel (a fragment that straddles two image rows: 2-way; 2 of the 14) — with ((pixel >> 1) & 7), the row swizzle of gemm_pp.hip,
//     this access pattern is 2-way on average.  Realised on the DMA side by permuting the SOURCE chunk per lane.
// Forward epilogue: bf16 store + per-tile BatchNorm partial sums (sum x, sum x^2 from the fp32 accumulators) in the layout msc_bn_finalize
// reads; rows of the caller's partial buffer beyond this kernel's tile count are zeroed.  Data-gradient epilogue: + residual, x relu'(ref).
// The data gradient is the same kernel with the transposed weight layout the library keeps for dgrads and the tap offsets negated.
#include "msc_common.h"
#include "gemm_plan.h"
#include <type_traits>
#include <utility>

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 c6_bf16x8;
typedef __attribute__((ext_vector_type(4))) float c6_f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t c6_u32x4;

#define C6_OOB 0x80000000u
#define C6_LDS(p) ((__attribute__((address_space(3))) void*)(p))

constexpr int C6_W = 56, C6_TR = 4;                       // image width, output rows per tile
constexpr int C6_HW = C6_W + 2, C6_HR = C6_TR + 2;        // halo: 58 x 6 pixels
constexpr int C6_HPIX = C6_HW * C6_HR;                    // 348
constexpr int C6_NDMA = (C6_HPIX * 128 + 1023) / 1024;    // 44 one-KiB DMA instructions per halo (11 per wave)
constexpr int C6_HALO_BYTES = C6_NDMA * 1024;             // 45056
constexpr int C6_NF = C6_TR * C6_W / 16;                  // 14 pixel fragments per tile
constexpr int C6_FW = C6_NF / 2;                          // 7 per wave
constexpr int C6_KS = 18;                                 // k-steps of 32: 9 taps x 2 halves of the 64 input channels
constexpr int C6_EB = 7;                                  // fragments per group of epilogue loads
constexpr int C6_WX_BYTES = 64 * 576 * 2;                 // image of the weight block during the prologue: 72 KB (>= one halo buffer)
static_assert(C6_WX_BYTES >= C6_HALO_BYTES, "halo buffer 1 lives inside the exchange area");

// C6_STAMP: diagnostic build only (make alt ALT=c6stamp ALT_SRC=conv_c64 ALT_FLAGS=-DC6_STAMP; tools/c64_stamps.py).  Wave 0 of every workgroup
// sums shader-clock cycles per phase into a debug buffer nothing else reads: [0] weights -> registers, [1] wait for the halo + barrier,
// [2] K loop, [3] epilogue, [4] tiles, [5] whole kernel.
#ifdef C6_STAMP
__device__ unsigned long long g_c6_stamps[1024 * 8];
#define C6_T(v) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define C6_T(v) ((void)0)
#endif

// sum over the 16 lanes of a DPP row (= the 16 pixels of a fragment): every lane ends with the total, in an order that depends on the lane only
__device__ __forceinline__ float c6_row_sum(float v) {
  auto dpp = [](float x, auto ctrl) __attribute__((always_inline)) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xF, 0xF, false));
  };
  v += dpp(v, std::integral_constant<int, 0xB1>{});       // quad_perm [1,0,3,2]
  v += dpp(v, std::integral_constant<int, 0x4E>{});       // quad_perm [2,3,0,1]
  v += dpp(v, std::integral_constant<int, 0x124>{});      // row_ror:4
  v += dpp(v, std::integral_constant<int, 0x128>{});      // row_ror:8
  return v;
}

struct C6Params {
  const void* x; const void* w; void* y;
  const void* res; const void* relu_ref;   // data gradient only
  float* part; int part_rows;              // forward only: [part_rows][2][64]
  int B, H;                                // W = 56
  int flip;                                // 1: data gradient (tap offsets negated)
};

template <int FLIP>
__global__ __launch_bounds__(256, 1) void conv_c64_kernel(const C6Params p) {
  // LDS: halo buffer 0 | halo buffer 1; the weight exchange of the prologue (72 KB) starts at buffer 1 and is dead before buffer 1 is first filled
  __shared__ __attribute__((aligned(1024))) char lds[C6_HALO_BYTES + C6_WX_BYTES];
  __shared__ float red[2][2][64];                        // [pixel half][sum | sum of squares][channel]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 15, lg = lane >> 4;
  const int tiles_per_img = p.H / C6_TR;
  const int ntiles = p.B * tiles_per_img;
#ifdef C6_STAMP
  unsigned long long t0, t1, ta, tb, tc, td, tp1 = 0, tp2 = 0, s_wait = 0, s_k = 0, s_ep = 0, n_t = 0;
  C6_T(t0);
#endif

  // ---- DMA side.  Descriptor base = x - (W + 1) pixels, so that halo pixel (hr, hc) of the tile whose first output row is r0 of image b
  // sits at byte ((b H + r0) W + hr W + hc) * 128 >= 0: the per-lane part (hr W + hc) * 128 + source chunk * 16 never changes, the tile part
  // goes into the scalar offset.  Halo pixels outside the image (left / right columns, the row above the first / below the last image row)
  // get the out-of-range marker: the DMA writes zeros.
  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(p.x) - (C6_W + 1) * 128), 0, (int)0x7ffffff0, 0x00020000);
  constexpr int NDW = C6_NDMA / 4;                        // DMA instructions per wave
  uint32_t drel[NDW];
  uint32_t dtop = 0, dbot = 0;                            // bit i: instruction i's pixel is in the halo's first / last row
#pragma unroll
  for (int i = 0; i < NDW; ++i) {
    const int P = 8 * (wave + 4 * i) + (lane >> 3);       // halo pixel this lane fills (LDS is written lane-linearly)
    const int hr = P / C6_HW, hc = P - hr * C6_HW;
    const int src_chunk = (lane & 7) ^ (P & 7);
    const bool ok = P < C6_HPIX && hc >= 1 && hc <= C6_W;
    drel[i] = ok ? (uint32_t)((hr * C6_W + hc) * 128 + src_chunk * 16) : C6_OOB;
    if (hr == 0) dtop |= 1u << i;
    if (hr == C6_HR - 1) dbot |= 1u << i;
  }
  // instruction i of this wave's share of a halo (scalar offset / border flags of the tile: stage_tile)
  uint32_t st_soff = 0;
  bool st_top = false, st_bot = false;
  auto stage_tile = [&](int tile) __attribute__((always_inline)) {
    const int b = tile / tiles_per_img, r0 = (tile - b * tiles_per_img) * C6_TR;
    st_soff = (uint32_t)((b * p.H + r0) * C6_W) * 128u;
    st_top = r0 == 0; st_bot = r0 + C6_TR == p.H;
  };
  auto stage_one = [&](int buf, auto i_c) __attribute__((always_inline)) {
    constexpr int i = decltype(i_c)::value;
    const bool dead = (st_top && ((dtop >> i) & 1)) || (st_bot && ((dbot >> i) & 1));
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, C6_LDS(lds + buf * C6_HALO_BYTES + (wave + 4 * i) * 1024), 16, dead ? C6_OOB : drel[i], st_soff, 0, 0);
  };
  auto stage = [&](int tile, int buf) __attribute__((always_inline)) {
    stage_tile(tile);
    [&]<int... I>(std::integer_sequence<int, I...>) __attribute__((always_inline)) { (stage_one(buf, std::integral_constant<int, I>{}), ...); }(std::make_integer_sequence<int, NDW>{});
  };

  // ---- fragment read side: pixel q = 16 (7 wm + f) + lr of the tile -> halo pixel (q / 56 + 1) * 58 + q % 56 + 1
  int hp[C6_FW];
#pragma unroll
  for (int f = 0; f < C6_FW; ++f) {
    const int q = 16 * (C6_FW * wm + f) + lr;
    const int r = q / C6_W, c = q - r * C6_W;
    hp[f] = (r + 1) * C6_HW + c + 1;
  }
  const uint32_t out_lane = (uint32_t)(32 * wn + 8 * lg) * 2u;      // byte offset of the lane's 8 channels inside a pixel

  int tile = (int)blockIdx.x;
  if (tile < ntiles) stage(tile, 0);
  // ---- weights -> registers, once.  MFMA row i of co-fragment j <- channel 32 wn + 8 (i >> 2) + 4 j + (i & 3): the accumulator rows a lane
  // holds (4 lg + r of both fragments) are then channels 32 wn + 8 lg .. + 7.  The [64][576] block is one contiguous 72 KB array: it is
  // copied into the exchange area by LDS-DMA as it lies (full 128-byte lines, 18 one-KiB instructions per wave next to the first tile's halo),
  // and every wave then picks its 36 fragments out of the image with ds_read_b128 (rows 1152 B apart; position of chunk c in row ch =
  // c ^ ((ch >> 1) & 7), applied on the DMA's source side: 2-way instead of 8-way bank conflicts).
  // Fragment-shaped loads from global memory (16 rows x 64 B per instruction; to registers or by DMA alike) cost 9-11 k cycles of the
  // workgroup's ~45 k: profiles/r4_conv_c64.txt.
  c6_bf16x8 wreg[2][C6_KS];
  {
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, 64 * 576 * 2, 0x00020000);
    char* wx = lds + C6_HALO_BYTES;
#pragma unroll
    for (int i = 0; i < 64 * 576 * 2 / 1024 / 4; ++i) {
      const int t = wave + 4 * i;                            // KiB t of the image: 16-byte slot q = 64 t + lane = (row ch, position c of 72)
      const int q = 64 * t + lane, ch = q / 72, c = q - 72 * ch;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, C6_LDS(wx + t * 1024), 16, (uint32_t)(ch * 1152 + (c ^ ((ch >> 1) & 7)) * 16), 0, 0, 0);
    }
#ifdef C6_STAMP
    C6_T(tp1);
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#ifdef C6_STAMP
    C6_T(tp2);
#endif
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int ch = 32 * wn + 8 * (lr >> 2) + 4 * j + (lr & 3);
#pragma unroll
      for (int ks = 0; ks < C6_KS; ++ks)
        wreg[j][ks] = *reinterpret_cast<const c6_bf16x8*>(wx + ch * 1152 + ((4 * ks + lg) ^ ((ch >> 1) & 7)) * 16);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the exchange area is dead from here on (buffer 1 is filled after the next barrier)
  }

#ifdef C6_STAMP
  C6_T(t1);
#endif
  int buf = 0;
  bool first = true;
  for (; tile < ntiles; tile += (int)gridDim.x, buf ^= 1) {
    // this tile's halo has landed.  Its DMA was issued before the previous tile's epilogue, whose 7 (+1: statistics row) stores are the only
    // younger vector-memory operations: they may stay in flight (the queue retires in order) — waiting for them costs a store round trip per tile
    C6_T(ta);
    if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    first = false;
    __syncthreads();                                      // ... for every wave; and nobody still reads the other buffer
    C6_T(tb);
    // the next tile's halo: one DMA instruction per k-step of this tile's K loop (an LDS-DMA instruction holds the issuing wave for 60-180
    // cycles: among the MFMAs the matrix pipe keeps working on what is queued; eleven in a row in front of the loop it does not)
    const bool more_tiles = tile + (int)gridDim.x < ntiles;
    if (more_tiles) stage_tile(tile + (int)gridDim.x);
    const uint32_t hbase = (uint32_t)(uintptr_t)C6_LDS(lds + buf * C6_HALO_BYTES);
    // (the 7 x 9 fragment addresses are cheap to form and expensive to keep: without this the compiler hoists all of them out of the tile
    // loop into ~90 registers and the epilogue's operand loads spill)
#pragma unroll
    for (int f = 0; f < C6_FW; ++f) asm volatile("" : "+v"(hp[f]));

    c6_f32x4 acc[C6_FW][2];
#pragma unroll
    for (int f = 0; f < C6_FW; ++f) { acc[f][0] = c6_f32x4{0.f, 0.f, 0.f, 0.f}; acc[f][1] = c6_f32x4{0.f, 0.f, 0.f, 0.f}; }

    // k-step ks = 2 tap + half.  One wave per SIMD: nothing but this wave's own later instructions can cover an LDS read, so the pixel
    // fragments of step ks + 1 are issued (inline assembly: the compiler would wait lgkmcnt(0) behind every read it knows) before the
    // MFMAs of step ks, and the counted wait in between names the fragments it releases (so no MFMA moves above it).
    c6_u32x4 xb[2][C6_FW];
    auto read_step = [&](int ks, c6_u32x4 (&dst)[C6_FW]) __attribute__((always_inline)) {
      const int tap = ks >> 1, half = ks & 1;
      const int kh = tap / 3, kw = tap - 3 * kh;
      const int dh = FLIP ? 1 - kh : kh - 1, dw = FLIP ? 1 - kw : kw - 1;
      const int toff = dh * C6_HW + dw;
#pragma unroll
      for (int f = 0; f < C6_FW; ++f) {
        const int P = hp[f] + toff;
        const uint32_t a = hbase + (uint32_t)(P * 128 + (((lg + 4 * half) ^ (P & 7)) << 4));
        asm volatile("ds_read_b128 %0, %1" : "=v"(dst[f]) : "v"(a));
      }
    };
    auto release = [&](c6_u32x4 (&b)[C6_FW], bool more) __attribute__((always_inline)) {
      static_assert(C6_FW == 7, "operand list below");
      if (more)
        asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]));
      else
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]));
    };
    auto dma_step = [&](int b2, int i) __attribute__((always_inline)) {
      // (i is a constant after unrolling; the chain folds to one call)
      if (i == 0) stage_one(b2, std::integral_constant<int, 0>{});
      if (i == 1) stage_one(b2, std::integral_constant<int, 1>{});
      if (i == 2) stage_one(b2, std::integral_constant<int, 2>{});
      if (i == 3) stage_one(b2, std::integral_constant<int, 3>{});
      if (i == 4) stage_one(b2, std::integral_constant<int, 4>{});
      if (i == 5) stage_one(b2, std::integral_constant<int, 5>{});
      if (i == 6) stage_one(b2, std::integral_constant<int, 6>{});
      if (i == 7) stage_one(b2, std::integral_constant<int, 7>{});
      if (i == 8) stage_one(b2, std::integral_constant<int, 8>{});
      if (i == 9) stage_one(b2, std::integral_constant<int, 9>{});
      if (i == 10) stage_one(b2, std::integral_constant<int, 10>{});
      static_assert(NDW == 11, "chain above");
    };
    // (reading fragment f of step ks + 1 right behind the MFMA pair of fragment f of step ks — one counted wait per pair — was 12 % slower:
    // profiles/r4_conv_c64.txt)
    read_step(0, xb[0]);
#pragma unroll
    for (int ks = 0; ks < C6_KS; ++ks) {
      if (ks + 1 < C6_KS) read_step(ks + 1, xb[(ks + 1) & 1]);
      release(xb[ks & 1], ks + 1 < C6_KS);
      if (ks < NDW && more_tiles) dma_step(buf ^ 1, ks);
#pragma unroll
      for (int f = 0; f < C6_FW; ++f) {
        const c6_bf16x8 xf = __builtin_bit_cast(c6_bf16x8, xb[ks & 1][f]);
        acc[f][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[0][ks], xf, acc[f][0], 0, 0, 0);
        acc[f][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[1][ks], xf, acc[f][1], 0, 0, 0);
      }
    }

    C6_T(tc);
    // ---- epilogue: the lane holds channels 32 wn + 8 lg .. + 7 of pixel 16 (7 wm + f) + lr, f = 0..6
    const int b = tile / tiles_per_img, r0 = (tile - b * tiles_per_img) * C6_TR;
    const long pix0 = ((long)b * p.H + r0) * C6_W;        // first pixel of the tile (NHWC pixel index)
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
    // operands of the data-gradient epilogue in groups of C6_EB fragments: the group's loads first, i.e. one memory round trip per group
    // instead of one per fragment (all 7 at once do not fit: 256 + 256 registers are in use)
#pragma unroll
    for (int f0 = 0; f0 < C6_FW; f0 += C6_EB) {
    c6_u32x4 rres[C6_EB], rref[C6_EB];
    if (p.res) {
#pragma unroll
      for (int g = 0; g < C6_EB; ++g)
        if (f0 + g < C6_FW)
          rres[g] = *reinterpret_cast<const c6_u32x4*>(reinterpret_cast<const char*>(p.res) + (pix0 + 16 * (C6_FW * wm + f0 + g) + lr) * 128 + out_lane);
    }
    if (p.relu_ref) {
#pragma unroll
      for (int g = 0; g < C6_EB; ++g)
        if (f0 + g < C6_FW)
          rref[g] = *reinterpret_cast<const c6_u32x4*>(reinterpret_cast<const char*>(p.relu_ref) + (pix0 + 16 * (C6_FW * wm + f0 + g) + lr) * 128 + out_lane);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int f = f0; f < f0 + C6_EB && f < C6_FW; ++f) {
      const long o = (pix0 + 16 * (C6_FW * wm + f) + lr) * 128 + out_lane;      // byte offset
      float v[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = acc[f][0][e]; v[4 + e] = acc[f][1][e]; }
      if (p.part) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { s1[e] += v[e]; s2[e] = fmaf(v[e], v[e], s2[e]); }
      }
      if (p.res) {
        const c6_u32x4 r = rres[f - f0];
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[2 * e] += __uint_as_float(r[e] << 16); v[2 * e + 1] += __uint_as_float(r[e] & 0xffff0000u); }
      }
      if (p.relu_ref) {
        const c6_u32x4 r = rref[f - f0];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (!(__uint_as_float(r[e] << 16) > 0.f)) v[2 * e] = 0.f;
          if (!(__uint_as_float(r[e] & 0xffff0000u) > 0.f)) v[2 * e + 1] = 0.f;
        }
      }
      c6_u32x4 wv;
#pragma unroll
      for (int e = 0; e < 4; ++e) wv[e] = (uint32_t)f32_to_bf16(v[2 * e]) | ((uint32_t)f32_to_bf16(v[2 * e + 1]) << 16);
      *reinterpret_cast<c6_u32x4*>(reinterpret_cast<char*>(p.y) + o) = wv;
    }
    }
    if (p.part) {
      // sums over the 16 pixel lanes of a lane group (fixed butterfly order), then over the two pixel halves through LDS
#pragma unroll
      for (int e = 0; e < 8; ++e) { s1[e] = c6_row_sum(s1[e]); s2[e] = c6_row_sum(s2[e]); }
      if (lr == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { red[wm][0][32 * wn + 8 * lg + e] = s1[e]; red[wm][1][32 * wn + 8 * lg + e] = s2[e]; }
      }
      __syncthreads();
      if (tid < 128) {
        const int st = tid >> 6, ch = tid & 63;
        p.part[(long)tile * 128 + st * 64 + ch] = red[0][st][ch] + red[1][st][ch];
      }
      // (red is rewritten only after the next tile's first __syncthreads)
    }
#ifdef C6_STAMP
    C6_T(td);
    s_wait += tb - ta; s_k += tc - tb; s_ep += td - tc; ++n_t;
#endif
  }
#ifdef C6_STAMP
  C6_T(td);
  if (tid == 0 && blockIdx.x < 1024) {
    unsigned long long* o = g_c6_stamps + (size_t)blockIdx.x * 8;
    o[0] = t1 - t0; o[1] = s_wait; o[2] = s_k; o[3] = s_ep; o[4] = n_t; o[5] = td - t0; o[6] = tp1 - t0; o[7] = tp2 - tp1;
  }
#endif
  // rows of the caller's partial buffer this kernel has no tile for: zero (msc_bn_finalize sums every row)
  if (p.part) {
    const long n0 = (long)ntiles * 128, n1 = (long)p.part_rows * 128;
    for (long i = n0 + (long)blockIdx.x * 256 + tid; i < n1; i += (long)gridDim.x * 256) p.part[i] = 0.f;
  }
}

}  // namespace

#ifdef C6_STAMP
extern "C" int msc_debug_c64_stamps(unsigned long long* host_out, int nwg) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_c6_stamps), (size_t)nwg * 8 * sizeof(unsigned long long)) == hipSuccess ? MSC_OK : MSC_ERR_LAUNCH;
}
#endif

// Host side.  Eligibility: bf16, 64 -> 64 channels, 3x3, stride 1, pad 1, 56-pixel-wide maps whose height is a multiple of 4.
bool msc_conv_c64_ok(int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int dtype) {
  if (!mscplan::g_knobs.conv_c64 || mscplan::g_knobs.tile_bm) return false;
  if (dtype != MSC_BF16 || Cin != 64 || Cout != 64 || KH != 3 || KW != 3 || stride != 1 || pad != 1) return false;
  if (W != C6_W || H % C6_TR || H < C6_TR || B < 1) return false;
  if ((long)B * H * W * 128 >= 0x7ffffff0L - (C6_W + 1) * 128) return false;
  return true;
}

int msc_conv_c64_launch(const void* x, const void* w, void* y, const void* res, const void* relu_ref, float* part, int part_rows, int B, int H,
                        int flip, int ncu, hipStream_t st) {
  if ((((uintptr_t)x | (uintptr_t)w | (uintptr_t)y | (uintptr_t)res | (uintptr_t)relu_ref) & 15) != 0) return MSC_ERR_BAD_ARG;
  C6Params p;
  p.x = x; p.w = w; p.y = y; p.res = res; p.relu_ref = relu_ref; p.part = part; p.part_rows = part_rows; p.B = B; p.H = H; p.flip = flip;
  const int ntiles = B * (H / C6_TR);
  if (part && part_rows < ntiles) return MSC_ERR_BAD_ARG;
  // Workgroups: the smallest count that needs no more rounds of tiles than one per CU would (64 frames: 896 tiles = 4 rounds on 256 CUs, 3.5 tiles
  // each; on 224 workgroups 4 tiles each — the same time, and 32 CUs stay free for whatever runs beside the launch).  knob c64_wgs overrides.
  int nwg = ntiles < ncu ? ntiles : ncu;
  if (mscplan::g_knobs.c64_wgs > 0) nwg = mscplan::g_knobs.c64_wgs < nwg ? mscplan::g_knobs.c64_wgs : nwg;
  else if (nwg == ncu) { const int rounds = (ntiles + ncu - 1) / ncu; nwg = (ntiles + rounds - 1) / rounds; }
  const dim3 grid((unsigned)nwg);
  if (flip) hipLaunchKernelGGL(conv_c64_kernel<1>, grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL(conv_c64_kernel<0>, grid, dim3(256), 0, st, p);
  return msc_launch_status();
}
