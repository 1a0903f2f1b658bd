// gemm_pp.hip — persistent "ping-pong" contraction kernel for the dense linears of the FusionTransformer (BASELINE.json: fusion
// attention / MLP projections), bf16, both operands k-major:   C[m][n] = epilogue( sum_k A[m][k] * B[n][k] )
// Build-authored: the upstream reference has no code (SURVEY.md finding zero).
//
// Why a second kernel: the tiled family of gemm.hip stages (BM + BN) x 128 B per K step for BM x BN x 64 MACs.  A CU
// takes in ~30-37 B/clk from its L2 (MI355X_MICROARCH.md "Indexed rows: gather into LDS"; measured 88 GB/s per CU in round 2), so a
// 128 x 128 tile (64 FLOP per staged byte) caps the matrix pipes below 50 % whatever its K loop does, and its 16 waves per CU were the
// only latency hiding it had.  This kernel is built the other way round:
//   * 256 x 256 output tile (BM = 32 * (MF0 + MF1) rows; 128 FLOP per staged byte), ONE 8-wave workgroup per CU, persistent
//     (gridDim.x = min(#tiles, #CUs); a workgroup walks tiles w, w + gridDim.x, ...; XCD-aware tile order as in gemm.hip).
//   * waves 2 (M) x 4 (N), wave tile (16 * MI) x 64 on v_mfma_f32_16x16x32_bf16, accumulators 128 registers.
//   * the K loop is the 8-phase schedule of cdna_hip_programming.md §5 ("The 256^2 8-phase template"): a K tile of 64 is cut into four
//     pieces (two halves of A, two of B) that are consumed by four phases, each phase = { fragment reads + ONE piece of LDS-DMA
//     prefetch | barrier | 16 MFMAs of one accumulator quadrant | barrier }.  Waves 4-7 run one barrier interval behind waves 0-3, so on
//     every SIMD one wave issues MFMAs while its partner issues LDS reads / DMA / stores: instruction issue of everything that is not an
//     MFMA is hidden behind the partner's MFMA cluster.  vmcnt is counted (never 0 in the loop): three pieces stay in flight across the
//     barriers; the DMA stream simply continues into the workgroup's NEXT tile.
//   * no LDS slab epilogue.  The MFMA is issued "transposed" (A operand = fragment of B, B operand = fragment of A), so a lane ends up
//     with 4 consecutive output COLUMNS of one row per accumulator; the B rows are dealt to the fragments so that two accumulators give a
//     lane 8 consecutive columns = one 16-byte store (16 rows x 64 B per store instruction).  The permutation costs nothing: it is in the
//     DMA's per-lane source row.  Each accumulator quadrant is finished (bias / activation / residual / store) in the READ part of the
//     phase after its last MFMA — beside the partner wave's MFMAs — and is not needed again before the next tile's same phase, so
//     the epilogue needs no second accumulator set and no barrier of its own.
#include "msc_common.h"
#include "gemm_params.h"
#include "gemm_plan.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 pp_bf16x8;
typedef __attribute__((ext_vector_type(4))) float pp_f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t pp_u32x4;

enum { PP_RES = 1, PP_C2 = 2, PP_DREF = 4 };
#define PP_OOB 0x80000000u          // >= num_records of every descriptor built here: loads return / DMA writes zeros, stores are dropped
#define PP_LDS(p) ((__attribute__((address_space(3))) void*)(p))
#define PP_MAX_N 8192               // bias vector staged in LDS once per launch

__device__ __forceinline__ void pp_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

template <int N> __device__ __forceinline__ void pp_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ uint32_t pp_pack2(float a, float b) {
  return (uint32_t)f32_to_bf16(a) | ((uint32_t)f32_to_bf16(b) << 16);
}
__device__ __forceinline__ void pp_unpack8(const pp_u32x4 w, float (&v)[8]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[2 * i] = __uint_as_float(w[i] << 16);
    v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
  }
}

// activations this kernel's epilogue knows (the tanh / sigmoid layers of the path are 64-row head GEMMs, which stay on gemm.hip)
__device__ __forceinline__ float pp_act(float x, int act) {
  if (act == MSC_ACT_RELU) return x > 0.f ? x : 0.f;
  if (act == MSC_ACT_GELU) return 0.5f * x * (1.f + fast_erf(x * 0.70710678118654752f));
  return x;
}
__device__ __forceinline__ float pp_act_grad(float r, int act) {
  if (act == MSC_ACT_MUL_REF) return r;
  if (act == MSC_ACT_RELU) return r > 0.f ? 1.f : 0.f;
  if (act == MSC_ACT_GELU) return 0.5f * (1.f + fast_erf(r * 0.70710678118654752f)) + r * 0.3989422804014327f * __expf(-0.5f * r * r);
  return 1.f;
}

// ACT: p.act of the launch (EPI & PP_DREF: p.dact) as a compile-time constant — the epilogue is unrolled 5 x MF x 8 times
template <int MF0, int MF1, int EPI, int ACT>
__global__ __launch_bounds__(512, 2) void gemm_pp_kernel(const GemmParams p) {
  constexpr int MI = MF0 + MF1, BM = 32 * MI, BN = 256;
  constexpr int PA0_ROWS = 32 * MF0, PA1_ROWS = 32 * MF1;
  constexpr int PB_BYTES = 128 * 128, PA0_BYTES = PA0_ROWS * 128, PA1_BYTES = PA1_ROWS * 128;
  constexpr int OFF_PB0 = 0, OFF_PA0 = PB_BYTES, OFF_PB1 = OFF_PA0 + PA0_BYTES, OFF_PA1 = OFF_PB1 + PB_BYTES;
  constexpr int BUF_BYTES = OFF_PA1 + PA1_BYTES;
  constexpr int BIAS_OFF = 2 * BUF_BYTES;
  constexpr int SMEM_BYTES = BIAS_OFF + PP_MAX_N * 4;
  static_assert(SMEM_BYTES <= 163840, "LDS budget of a CU");
  // 1-KiB DMA instructions per piece; wave w issues instructions w, w + 8 of a piece: NSx slots, the second only for waves < NIx - 8
  constexpr int NIA0 = PA0_ROWS / 8, NIA1 = PA1_ROWS / 8;
  static_assert(NIA0 >= 8 && NIA0 <= 16 && NIA1 >= 8 && NIA1 <= 16, "2 <= MF <= 4");
  // DMA instructions of {B0, A0, B1, A1} = any four consecutive pieces of the stream, for waves 0-3 (group 0) / 4-7 (group 1)
  constexpr int INFL_G0 = 4 + (NIA0 > 8 ? 2 : 1) + (NIA1 > 8 ? 2 : 1);
  constexpr int INFL_G1 = 4 + (NIA0 > 12 ? 2 : 1) + (NIA1 > 12 ? 2 : 1);
  static_assert(NIA0 == 8 || NIA0 == 12 || NIA0 == 16, "A piece: 8, 12 or 16 DMA instructions (waves 0-3 / 4-7 issue the same number each)");
  static_assert(NIA1 == 8 || NIA1 == 12 || NIA1 == 16, "A piece: 8, 12 or 16 DMA instructions");
  __shared__ __attribute__((aligned(1024))) char smem[SMEM_BYTES];

  MSC_BOOST();
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;            // wm = ping-pong group
  const int lr = lane & 15, lg = lane >> 4;
  const int ntn = p.N / BN;                           // host guarantees N % 256 == 0
  const int ntiles = ((p.M + BM - 1) / BM) * ntn;
  const int NK = p.K / 64;                            // host guarantees K % 128 == 0, K >= 256: K tiles come in (buffer 0, buffer 1) pairs
  const int gstep = (int)gridDim.x;

  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc((void*)p.C, 0, (int)0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsC2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.C2, 0, (int)0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((void*)p.res, 0, (int)0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc((void*)p.dref, 0, (int)0x7ffffff0, 0x00020000);

  // ---- bias -> LDS (before any DMA is in flight: these are ordinary loads)
  {
    float* bl = reinterpret_cast<float*>(smem + BIAS_OFF);
    for (int i = tid; i < p.N; i += 512) bl[i] = p.bias ? p.bias[i] : 0.f;
    __syncthreads();
  }

  const uint32_t lda2 = (uint32_t)p.lda * 2u, ldb2 = (uint32_t)p.ldb * 2u;
  // work item -> first row / column of its tile (wave-uniform); it >= ntiles: an empty tile at (M, N): every A row out of range, every store dropped
  auto tile_origin = [&](int it, int& m0, int& n0) __attribute__((always_inline)) {
    const bool live = it < ntiles;
    const int bid = mscplan::xcd_remap(live ? it : 0, ntiles);
    const int tm = bid / ntn;
    m0 = __builtin_amdgcn_readfirstlane(live ? tm * BM : p.M);
    n0 = __builtin_amdgcn_readfirstlane(live ? (bid - tm * ntn) * BN : 0);
  };

  // ---- DMA side.  LDS row rho of a piece <- row of the tile:
  //   B piece b : rho = wn'*32 + j*16 + i  <-  n - n0 = wn'*64 + b*32 + 8*(i>>2) + 4*j + (i&3)     (see the epilogue)
  //   A piece a : rho = wm'*(16*MFa) + i'  <-  m - m0 = wm'*(16*MI) + (a ? 16*MF0 : 0) + i'
  // One DMA instruction = 8 LDS rows x 128 B; lane -> row 8q + (lane>>3), LDS chunk position lane&7 holds the row's logical 16-byte
  // chunk (lane&7) ^ ((rho>>1)&7) (conflict-free ds_read_b128 for the 16x16x32 operand pattern: tools/lds_bank_check.py).
  // The tile base and the K tile go into the instruction's scalar offset.  B: one per-lane offset (the other half / the second
  // instruction are a constant number of rows further: scalar); columns always exist (N % 256 == 0).  A: per (half, instruction) the
  // per-lane offset with "row >= M" already folded in (out-of-range marker: the DMA writes zeros), refreshed when the half's stream
  // moves on to the next tile.
  uint32_t relB;
  {
    const int rho = 8 * wave + (lane >> 3);
    const int ch = (lane & 7) ^ ((rho >> 1) & 7);
    const int wnp = rho >> 5, j = (rho >> 4) & 1, ii = rho & 15;
    relB = (uint32_t)(wnp * 64 + 8 * (ii >> 2) + 4 * j + (ii & 3)) * ldb2 + ch * 16;
  }
  uint32_t voffA[2][2];
  auto set_voffA = [&](auto sel_c, int m0) __attribute__((always_inline)) {
    constexpr int SEL = decltype(sel_c)::value;
    constexpr int MFS = SEL ? MF1 : MF0;
    const int rows_left = p.M - m0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int rho = 8 * (wave + 8 * i) + (lane >> 3);
      const int ch = (lane & 7) ^ ((rho >> 1) & 7);
      const int wmp = rho / (16 * MFS), ip = rho - wmp * (16 * MFS);
      const int mrel = wmp * (16 * MI) + (SEL ? 16 * MF0 : 0) + ip;
      voffA[SEL][i] = mrel < rows_left ? (uint32_t)mrel * lda2 + ch * 16 : PP_OOB;
    }
  };

  int c_m0, c_n0, n_m0, n_n0;          // current / next tile of this workgroup
  // issue one piece of K tile `kts` (counted from the current tile's first K tile; >= NK: the next tile's) into LDS buffer `buf`
  auto stage = [&](int kts, int buf, auto is_a_c, auto sel_c) __attribute__((always_inline)) {
    constexpr bool IS_A = decltype(is_a_c)::value;
    constexpr int SEL = decltype(sel_c)::value;
    constexpr int NI = IS_A ? (SEL ? NIA1 : NIA0) : 16;
    constexpr int OFF = IS_A ? (SEL ? OFF_PA1 : OFF_PA0) : (SEL ? OFF_PB1 : OFF_PB0);
    const bool wrap = kts >= NK;
    const uint32_t kb = (uint32_t)(wrap ? kts - NK : kts) * 128u;
    char* dst = smem + buf * BUF_BYTES + OFF;
    if constexpr (IS_A) {
      const uint32_t soff = (uint32_t)(wrap ? n_m0 : c_m0) * lda2 + kb;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, PP_LDS(dst + wave * 1024), 16, voffA[SEL][0], soff, 0, 0);
      if (NI > 8 && (NI == 16 || wave < NI - 8))
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, PP_LDS(dst + (wave + 8) * 1024), 16, voffA[SEL][1], soff, 0, 0);
    } else {
      const uint32_t soff = (uint32_t)((wrap ? n_n0 : c_n0) + SEL * 32) * ldb2 + kb;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, PP_LDS(dst + wave * 1024), 16, relB, soff, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, PP_LDS(dst + (wave + 8) * 1024), 16, relB, soff + 128u * ldb2, 0, 0);
    }
  };
  using TA = std::true_type;
  using TB = std::false_type;
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;

  // ---- fragment read addresses (bytes inside a K-tile buffer).  Row rho = wave part + frag*16 + lr; chunk (4s + lg) ^ ((lr>>1)&7)
  const int xk = (lr >> 1) & 7;
  const int lane_s0 = lr * 128 + (((0 + lg) ^ xk) << 4);
  const int lane_s1 = lr * 128 + (((4 + lg) ^ xk) << 4);
  const int baseB = wn * 32 * 128;                     // + j*16*128
  const int baseA0 = wm * (16 * MF0) * 128;            // + f*16*128
  const int baseA1 = wm * (16 * MF1) * 128;

  pp_f32x4 acc[4][MI];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < MI; ++j) acc[i][j] = pp_f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int MFX = MF0 > MF1 ? MF0 : MF1;
  pp_bf16x8 fa[MFX][2], fb[2][2][2];    // A fragments of the current half (f, s); B fragments (b, j, s)

  auto read_b = [&](int buf, auto b_c) __attribute__((always_inline)) {
    constexpr int Bh = decltype(b_c)::value;
    const char* base = smem + buf * BUF_BYTES + (Bh ? OFF_PB1 : OFF_PB0) + baseB;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      fb[Bh][j][0] = *reinterpret_cast<const pp_bf16x8*>(base + j * 2048 + lane_s0);
      fb[Bh][j][1] = *reinterpret_cast<const pp_bf16x8*>(base + j * 2048 + lane_s1);
    }
  };
  auto read_a = [&](int buf, auto a_c) __attribute__((always_inline)) {
    constexpr int Ah = decltype(a_c)::value;
    constexpr int MFS = Ah ? MF1 : MF0;
    const char* base = smem + buf * BUF_BYTES + (Ah ? OFF_PA1 : OFF_PA0) + (Ah ? baseA1 : baseA0);
#pragma unroll
    for (int f = 0; f < MFS; ++f) {
      fa[f][0] = *reinterpret_cast<const pp_bf16x8*>(base + f * 2048 + lane_s0);
      fa[f][1] = *reinterpret_cast<const pp_bf16x8*>(base + f * 2048 + lane_s1);
    }
  };
  // MFMAs of quadrant (a, b): 4 * MFa instructions
  auto mma = [&](auto a_c, auto b_c) __attribute__((always_inline)) {
    constexpr int Ah = decltype(a_c)::value, Bh = decltype(b_c)::value;
    constexpr int MFS = Ah ? MF1 : MF0, MB = Ah ? MF0 : 0;
    __builtin_amdgcn_s_setprio(MSC_BOOST_LEVEL ? 3 : 1);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int f = 0; f < MFS; ++f)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[2 * Bh + j][MB + f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[Bh][j][s], fa[f][s], acc[2 * Bh + j][MB + f], 0, 0, 0);
    __builtin_amdgcn_s_setprio(MSC_BOOST_LEVEL);
  };

  // ---- epilogue of quadrant (a, b) of the tile at (m0, n0): lane = row lr of each 16-row fragment, 8 consecutive columns; the
  // quadrant's accumulators are cleared for the next tile.  Every store / load carries its whole address in the VECTOR offset (scalar
  // offset 0): with an SGPR in the soffset field hipcc (ROCm 7.2) does not pad the "VALU overwrites the data registers of a 16-byte
  // store in the next instruction" hazard, and gfx950 has it (measured: lanes 12-15 of each row group stored the NEXT fragment's bits).
  const float* biasl = reinterpret_cast<const float*>(smem + BIAS_OFF);
  const uint32_t ldc2 = (uint32_t)p.ldc * 2u;
  const int ep_row = wm * (16 * MI) + lr;                                                    // row inside the tile (+ 16 per fragment)
  const uint32_t ep_lane_off = (uint32_t)ep_row * ldc2 + (uint32_t)(wn * 64 + 8 * lg) * 2u; // bytes from the tile's first element
  // Epilogue operands read from memory (residual / stored activation derivative): an ordinary load inside the epilogue costs a full
  // drain of the vector-memory queue (the compiler waits vmcnt(0) for it beside LDS-DMA) and, inside the training step, the latency of
  // HBM — the 80 MB derivative tensor of fc2's dgrad is cold there: 95 us per launch against 64 from a warm cache, round 4.  PREF: the
  // tile's 2 * MI 16-byte loads are issued as inline assembly (the compiler does not wait for what it does not know) in phase 1 of the
  // tile's SECOND-TO-LAST K tile, five phases before the first quadrant epilogue; the counted waits order them like any DMA piece: they
  // are older than that phase's piece, so the wait of phase 1 of the last K tile ("everything older than the four youngest pieces") is
  // also the wait for them, and the four waits in between allow them in flight (+NPRE).  40 registers at MI = 5; the 192- and 256-row
  // tiles have none to spare and keep the in-epilogue loads.
  constexpr bool PREF = (EPI & (PP_RES | PP_DREF)) != 0 && MI <= 5;
  constexpr int NPRE = PREF ? 2 * MI : 0;
  pp_u32x4 pre[2][PREF ? MI : 1];
  auto prefetch = [&](int m0, int n0) __attribute__((always_inline)) {
    if constexpr (PREF) {
      const uint64_t base = (uint64_t)((EPI & PP_DREF) ? p.dref : p.res);
      const pp_u32x4 dsc = {(uint32_t)base, (uint32_t)(base >> 32) & 0xffffu, 0x7ffffff0u, 0x00020000u};
      const int rows_left = p.M - m0;
#pragma unroll
      for (int bh = 0; bh < 2; ++bh) {
        const uint32_t tile_off = (uint32_t)m0 * ldc2 + (uint32_t)(n0 + bh * 32) * 2u;
#pragma unroll
        for (int f = 0; f < MI; ++f) {
          const uint32_t off = (ep_row + f * 16 < rows_left) ? ep_lane_off + (tile_off + (uint32_t)(f * 16) * ldc2) : PP_OOB;
          asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(pre[bh][f]) : "v"(off), "s"(dsc) : "memory");
        }
      }
    }
  };
  auto epilogue = [&](int m0, int n0, auto a_c, auto b_c) __attribute__((always_inline)) {
    constexpr int Ah = decltype(a_c)::value, Bh = decltype(b_c)::value;
    constexpr int MFS = Ah ? MF1 : MF0, MB = Ah ? MF0 : 0;
    float bv[8];
    {
      const int bc = n0 + wn * 64 + Bh * 32 + 8 * lg;      // first of the lane's 8 columns (always < N: N % 256 == 0)
      const pp_f32x4 b0 = *reinterpret_cast<const pp_f32x4*>(biasl + bc);
      const pp_f32x4 b1 = *reinterpret_cast<const pp_f32x4*>(biasl + bc + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { bv[e] = b0[e]; bv[4 + e] = b1[e]; }
    }
    const int rows_left = p.M - m0;                       // scalar
    const uint32_t tile_off = (uint32_t)m0 * ldc2 + (uint32_t)(n0 + Bh * 32) * 2u;
#pragma unroll
    for (int f = 0; f < MFS; ++f) {
      const uint32_t off = (ep_row + (MB + f) * 16 < rows_left) ? ep_lane_off + (tile_off + (uint32_t)((MB + f) * 16) * ldc2) : PP_OOB;
      float v[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = acc[2 * Bh][MB + f][e] + bv[e]; v[4 + e] = acc[2 * Bh + 1][MB + f][e] + bv[4 + e]; }
      acc[2 * Bh][MB + f] = pp_f32x4{0.f, 0.f, 0.f, 0.f};
      acc[2 * Bh + 1][MB + f] = pp_f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (EPI & PP_C2) {
        pp_u32x4 w2;
        if constexpr (ACT == MSC_ACT_GELU_DSAVE) {
          float d[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) {          // two elements per call: packed fp32 arithmetic (msc_common.h gelu_both2)
            msc_f32x2 yy, dd;
            gelu_both2((msc_f32x2){v[2 * e], v[2 * e + 1]}, yy, dd);
            v[2 * e] = yy[0]; v[2 * e + 1] = yy[1];
            d[2 * e] = dd[0]; d[2 * e + 1] = dd[1];
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) w2[e] = pp_pack2(d[2 * e], d[2 * e + 1]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) w2[e] = pp_pack2(v[2 * e], v[2 * e + 1]);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = pp_act(v[e], ACT);
        }
        __builtin_amdgcn_raw_buffer_store_b128(w2, rsC2, off, 0, 2);
      } else if constexpr (!(EPI & PP_DREF)) {
        if constexpr (ACT != MSC_ACT_NONE) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = pp_act(v[e], ACT);
        }
      }
      if constexpr (EPI & PP_RES) {
        pp_u32x4 r;
        if constexpr (PREF) r = pre[Bh][MB + f]; else r = __builtin_amdgcn_raw_buffer_load_b128(rsR, off, 0, 0);
        float rv[8];
        pp_unpack8(r, rv);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += rv[e];
      }
      if constexpr (EPI & PP_DREF) {
        pp_u32x4 r;
        if constexpr (PREF) r = pre[Bh][MB + f]; else r = __builtin_amdgcn_raw_buffer_load_b128(rsD, off, 0, 0);
        float rv[8];
        pp_unpack8(r, rv);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= pp_act_grad(rv[e], ACT);
      }
      pp_u32x4 w;
#pragma unroll
      for (int e = 0; e < 4; ++e) w[e] = pp_pack2(v[2 * e], v[2 * e + 1]);
      __builtin_amdgcn_raw_buffer_store_b128(w, rsC, off, 0, 2);
      // (256-row tiles: one fragment's erf arithmetic at a time — registers; the smaller tiles let the compiler interleave fragments: fc1 -3 %)
      if constexpr (ACT == MSC_ACT_GELU_DSAVE && MI > 6) __builtin_amdgcn_sched_barrier(0);
    }
  };
  // vector-memory instructions one fragment of an epilogue issues and leaves in flight (its loads are consumed inside): the stores
  constexpr int EP_ST = 1 + ((EPI & PP_C2) ? 1 : 0);
  constexpr bool EP_LD = (EPI & (PP_RES | PP_DREF)) != 0 && !PREF;     // epilogues that still load inside
  constexpr int MFN = MF0 < MF1 ? MF0 : MF1;          // (counting the smaller half for every quadrant only makes a wait stricter)

  // Counted wait of a phase, after its DMA piece was issued: everything older than the FOUR youngest pieces has landed — the piece
  // issued four phases ago is the one the NEXT phase reads ("read a staged buffer one phase after the wait that retires it").  XQ:
  // epilogue stores (in quadrant-epilogue units) known to be younger than that piece; they may stay in flight too.
  auto wait_phase_n = [&](auto x_c) __attribute__((always_inline)) {
    constexpr int X = decltype(x_c)::value;
    if constexpr (INFL_G0 == INFL_G1) {
      pp_wait_vm<INFL_G0 + X>();
    } else {
      if (wm == 0) pp_wait_vm<INFL_G0 + X>();
      else pp_wait_vm<INFL_G1 + X>();
    }
  };
  auto wait_phase = [&](auto xq_c) __attribute__((always_inline)) {
    wait_phase_n(std::integral_constant<int, decltype(xq_c)::value * MFN * EP_ST>{});
  };
  // steady-state wait of a phase; `pf`: this K tile is the one whose phase 1 issued the epilogue operand prefetch (they stay in flight)
  auto wait_steady = [&](bool pf) __attribute__((always_inline)) {
    if (PREF && pf) wait_phase_n(std::integral_constant<int, NPRE>{});
    else wait_phase_n(std::integral_constant<int, 0>{});
  };
  using X0 = std::integral_constant<int, 0>;
  using X1 = std::integral_constant<int, 1>;
  using X2 = std::integral_constant<int, 2>;
  using X3 = std::integral_constant<int, 3>;
  using X4 = std::integral_constant<int, 4>;

  // ---- prologue: the stream of pieces is B0(0) A0(0) B1(0) A1(0) B0(1) A0(1) | then one piece per phase: B1(s+1) A1(s+1) B0(s+2) A0(s+2)
  int it = (int)blockIdx.x;
  tile_origin(it, c_m0, c_n0);
  tile_origin(it + gstep, n_m0, n_n0);
  set_voffA(S0{}, c_m0);
  set_voffA(S1{}, c_m0);
  stage(0, 0, TB{}, S0{}); stage(0, 0, TA{}, S0{}); stage(0, 0, TB{}, S1{}); stage(0, 0, TA{}, S1{});
  stage(1, 1, TB{}, S0{}); stage(1, 1, TA{}, S0{});
  wait_phase(X0{});              // B0(0), A0(0) have landed
  if (wm == 1) pp_barrier();     // group 1 runs one barrier interval behind group 0
  pp_barrier();

  int pm0 = p.M, pn0 = 0;        // previous tile (its last quadrant is finished in the first phase of the next one)

  // One K tile s (buffer U) = 4 phases, each { LDS reads | one DMA piece | counted wait | barrier | MFMAs of one quadrant | barrier }:
  //   1: read B0, A0 | DMA B1(s+1) | quadrant (0,0)        3: read A1 | DMA B0(s+2) | quadrant (1,1)
  //   2: read B1     | DMA A1(s+1) | quadrant (0,1)        4:         | DMA A0(s+2) | quadrant (1,0)
  // A piece is overwritten (two K tiles on) no earlier than two phases after the phase that read it, i.e. after BOTH groups' reads
  // of it have been waited for: group 1 issues its reads one barrier interval after group 0 and retires them (lgkmcnt) before its
  // MFMAs, one interval later still.
  // In a tile's closing K tile the read parts of phases 2, 3, 4 also finish quadrants (0,0), (0,1), (1,1); quadrant (1,0) is finished in
  // phase 1 of the next tile's first K tile (or after the loop).  Epilogues with loads run BEFORE the phase's DMA (the compiler waits
  // vmcnt(0) for an ordinary load beside LDS-DMA: cdna_hip_programming.md §5 trap (b)), store-only epilogues AFTER it (their stores stay
  // in flight).  first / last / after_first: position of the K tile in its output tile (wave-uniform).
  // Where a quadrant's epilogue runs: epilogues with loads in the READ part of the phase, before its DMA piece (above).  Store-only
  // epilogues: group 0 in the read part behind the DMA piece, group 1 one barrier interval later, behind the MFMAs of the same phase —
  // which is the interval of group 0's NEXT read part: the two waves of a SIMD then run their epilogue arithmetic (fc1: exact-erf GELU +
  // derivative, as long as the K loop) side by side instead of one after the other, each stalling its partner at the barrier.
  // Counted waits: XG0 / XG1 = quadrant epilogues whose stores are younger than the piece issued four phases ago, per group (group 1
  // has not issued the current phase's stores yet when it waits).
  // only where the epilogue arithmetic is long (GELU + derivative: -8 us on fc1); on the plain store epilogues the longer MFMA parts cost 4-5 us
  // per launch (profiles/r4_pp_late_epilogue.txt); the 192- and 256-row tiles have no registers to spare for the longer live ranges
  constexpr bool LATE = (EPI & PP_C2) != 0 && MI <= 5;
  auto epi_read = [&](bool on, int m0, int n0, auto a_c, auto b_c, bool before_dma) __attribute__((always_inline)) {
    if (on && (EP_LD ? before_dma : (!before_dma && (!LATE || wm == 0)))) {
      if (!before_dma) __builtin_amdgcn_sched_barrier(0);
      epilogue(m0, n0, a_c, b_c);
      if (before_dma) __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto epi_late = [&](bool on, int m0, int n0, auto a_c, auto b_c) __attribute__((always_inline)) {
    if constexpr (LATE) {
      if (on && wm == 1) epilogue(m0, n0, a_c, b_c);
    }
  };
  auto wait2 = [&](auto xg0_c, auto xg1_c) __attribute__((always_inline)) {
    if (!LATE || wm == 0) wait_phase(xg0_c); else wait_phase(xg1_c);
  };
  auto ktile = [&](int kt, bool first, bool after_first, bool last_pair, bool last, auto u_c) __attribute__((always_inline)) {
    constexpr int U = decltype(u_c)::value;
    constexpr bool U0 = U == 0, U1 = U == 1;
    using XL = std::conditional_t<EP_LD, X3, X4>;       // (epilogues with loads sit BEFORE their phase's piece: one fewer is younger)
    const bool e1 = U0 && first, e2 = U1 && last;
    const bool pf = U0 && last_pair;     // the tile's second-to-last K tile (never its first: K >= 256 = 4 K tiles)
    // ---- phase 1
    read_b(U, S0{});
    read_a(U, S0{});
    epi_read(e1, pm0, pn0, S1{}, S0{}, true);
    if (PREF && pf) prefetch(c_m0, c_n0);
    stage(kt + 1, U ^ 1, TB{}, S1{});
    epi_read(e1, pm0, pn0, S1{}, S0{}, false);
    if (e1) wait2(X4{}, X3{});
    else if (U1 && after_first && !EP_LD) wait_phase(X1{});
    else wait_steady(pf);
    pp_barrier();
    mma(S0{}, S0{});
    epi_late(e1, pm0, pn0, S1{}, S0{});
    pp_barrier();
    // ---- phase 2
    read_b(U, S1{});
    epi_read(e2, c_m0, c_n0, S0{}, S0{}, true);
    if (e2) set_voffA(S1{}, n_m0);       // this half's DMA stream moves on to the next tile with the piece below
    stage(kt + 1, U ^ 1, TA{}, S1{});
    epi_read(e2, c_m0, c_n0, S0{}, S0{}, false);
    if (e2) wait2(X1{}, X0{});
    else if (e1) wait2(XL{}, X4{});
    else wait_steady(pf);
    pp_barrier();
    mma(S0{}, S1{});
    epi_late(e2, c_m0, c_n0, S0{}, S0{});
    pp_barrier();
    // ---- phase 3
    read_a(U, S1{});
    epi_read(e2, c_m0, c_n0, S0{}, S1{}, true);
    stage(kt + 2, U, TB{}, S0{});
    epi_read(e2, c_m0, c_n0, S0{}, S1{}, false);
    if (e2) wait2(X2{}, X1{});
    else if (e1) wait2(std::conditional_t<EP_LD, X2, X3>{}, X3{});
    else wait_steady(pf);
    pp_barrier();
    mma(S1{}, S1{});
    epi_late(e2, c_m0, c_n0, S0{}, S1{});
    pp_barrier();
    // ---- phase 4
    epi_read(e2, c_m0, c_n0, S1{}, S1{}, true);
    if (U0 && last_pair) set_voffA(S0{}, n_m0);  // A0(s+2) is the next tile's from here on
    stage(kt + 2, U, TA{}, S0{});
    epi_read(e2, c_m0, c_n0, S1{}, S1{}, false);
    if (e2) wait2(X3{}, X2{});
    else if (e1) wait2(std::conditional_t<EP_LD, X1, X2>{}, X2{});
    else wait_steady(pf);
    pp_barrier();
    mma(S1{}, S0{});
    epi_late(e2, c_m0, c_n0, S1{}, S1{});
    pp_barrier();
  };

  const int npairs = NK / 2;
  for (; it < ntiles; it += gstep) {
    // `first` / `after_first` stand for "the previous tile's epilogue stores are in the vector-memory queue": not so in this
    // workgroup's FIRST tile — the counted waits of its first K tiles must not make room for stores that were never issued (they
    // would let the prologue's own DMA pieces stay in flight past the phase that reads them)
    const bool have_prev = it != (int)blockIdx.x;
    for (int kp = 0; kp < npairs; ++kp) {
      ktile(2 * kp, kp == 0 && have_prev, false, kp == npairs - 1, false, S0{});
      ktile(2 * kp + 1, false, kp == 0 && have_prev, false, kp == npairs - 1, S1{});
    }
    pm0 = c_m0; pn0 = c_n0;
    c_m0 = n_m0; c_n0 = n_n0;
    tile_origin(it + 2 * gstep, n_m0, n_n0);
  }
  // last quadrant of the last tile; the DMA side is past the last tile (all-OOB A pieces, B pieces nobody reads)
  epilogue(pm0, pn0, S1{}, S0{});
  if (wm == 0) pp_barrier();     // matches group 1's extra barrier at the start
  pp_wait_vm<0>();
}

template <int EPI, int ACT> int launch_pp_epi(const GemmParams& p, int bm, hipStream_t st, int ncu) {
  const int ntiles = cdiv(p.M, bm) * (p.N / 256);
  dim3 grid((unsigned)(ntiles < ncu ? ntiles : ncu), 1, 1);
  if (bm == 256) hipLaunchKernelGGL((gemm_pp_kernel<4, 4, EPI, ACT>), grid, dim3(512), 0, st, p);
  else if (bm == 192) hipLaunchKernelGGL((gemm_pp_kernel<3, 3, EPI, ACT>), grid, dim3(512), 0, st, p);
  else if (bm == 160) hipLaunchKernelGGL((gemm_pp_kernel<3, 2, EPI, ACT>), grid, dim3(512), 0, st, p);
  else if (bm == 128) hipLaunchKernelGGL((gemm_pp_kernel<2, 2, EPI, ACT>), grid, dim3(512), 0, st, p);
  else return MSC_ERR_UNSUPPORTED;
  return msc_launch_status();
}

}  // namespace

// Host side: does the persistent ping-pong kernel take this dense bf16 launch?  (k-major operands, no K split, no batch, bf16 output,
// one of the epilogues the path's large linears use.)  Returns the tile height it would use, 0 = not eligible.
int msc_pp_tile_rows(const GemmParams& p, int ncu) {
  if (!mscplan::g_knobs.gemm_pp || mscplan::g_knobs.tile_bm) return 0;      // (a forced tile means the tiled family)
  if (p.nsplit != 1 || p.xcd_split || p.stats || p.colsum || p.bnpart || p.rowsum || p.out_f32 || p.alpha != 1.f) return 0;
  if (p.K < 256 || p.K % 128 || p.N % 256 || p.N > PP_MAX_N || p.M < 2048) return 0;
  if (p.lda % 8 || p.ldb % 8 || p.ldc % 8) return 0;
  if ((long)p.M * p.ldc * 2 >= 0x7ffffff0L) return 0;
  const int mask = mscplan::g_knobs.gemm_pp_mask;
  if (mscplan::g_knobs.gemm_pp_k && mscplan::g_knobs.gemm_pp_k != p.K) return 0;
  if (p.dref) {                      // dgrad with the stored activation derivative
    if (p.res || p.C2 || p.act != MSC_ACT_NONE || p.dact != MSC_ACT_MUL_REF || !(mask & 8)) return 0;
  } else if (p.C2) {                 // forward with a second output: fc1 (GELU + derivative)
    if (p.res || p.act != MSC_ACT_GELU_DSAVE || !(mask & 4)) return 0;
  } else if (p.act != MSC_ACT_NONE) {
    return 0;
  } else if (!(mask & (p.res ? 2 : 1))) {
    return 0;
  }
  // Tile height: 256 rows (128 FLOP per staged byte), 192 (110) or 160 (98): whichever needs least time for ceil(tiles / CUs) rounds of
  // tiles, a round costing its tile's rows (+ a fixed share for the per-tile epilogue / refill).  On the path's 19584-row linears that is
  // 160 everywhere (246 tiles for 512 output columns = one round on 256 CUs; measured 346 vs 386 (192) vs 431 us (256) per block);
  // large square contractions take 256 (8192 x 8192 x 4096: 1.36 PFLOP/s vs 1.11 / 1.04).  128 rows (85 FLOP per staged byte) only ever wins
  // where the taller tiles leave CUs without a tile: the 512-column linears of a 32-frame shard (9792 rows: 124 tiles of 160 rows on 256 CUs,
  // 154 of 128 rows — BASELINE.json configs[3]'s per-GPU geometry); at 64 frames the rule never picks it.  knob gemm_pp: 1 = this rule;
  // 256 / 192 / 160 / 128 = that height forced.  (128 rows for the two epilogue-heavy launches only — fc1 forward, fc2's dgrad; isolated 55 vs 59 us for
  // the latter — is +-0 inside the 64-frame step: profiles/r4_pp128_tile.txt.)
  const int forced = mscplan::g_knobs.gemm_pp;
  if (forced == 256 || forced == 192 || forced == 160 || forced == 128) return forced;
  const long ntn = p.N / 256;
  long best = 0, best_cost = 0;
  for (long bm : {256L, 192L, 160L, 128L}) {
    const long rounds = mscplan::cdivl(mscplan::cdivl(p.M, bm) * ntn, ncu);
    const long cost = rounds * (bm + 24);
    if (!best || cost < best_cost) { best = bm; best_cost = cost; }
  }
  return (int)best;
}

int msc_pp_launch(const GemmParams& p, int ncu, hipStream_t st) {
  const int bm = msc_pp_tile_rows(p, ncu);
  if (!bm) return MSC_ERR_UNSUPPORTED;
  if (p.dref) return launch_pp_epi<PP_DREF, MSC_ACT_MUL_REF>(p, bm, st, ncu);
  if (p.C2) return launch_pp_epi<PP_C2, MSC_ACT_GELU_DSAVE>(p, bm, st, ncu);
  if (p.res) return launch_pp_epi<PP_RES, MSC_ACT_NONE>(p, bm, st, ncu);
  return launch_pp_epi<0, MSC_ACT_NONE>(p, bm, st, ncu);
}
