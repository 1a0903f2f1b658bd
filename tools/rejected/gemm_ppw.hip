// REJECTED EXPERIMENT (round 4), kept out of the product build: the ping-pong kernel for the dense weight gradients (reduction-major operands,
// transposed LDS reads, split-K slabs).  Correct (16 parity cases incl. ragged reductions, accumulate, bias gradient) but SLOWER than the tiled
// 128x128 / 16-waves-per-CU weight-gradient kernel: 262 vs 220 us per fusion block, step 7.81 vs 7.53 ms (profiles/r4_ppw_wgrad_rejected.txt):
// 72 ds_read_b64_tr_b16 per K tile and wave make its read parts longer than the partner's MFMA cluster.  To try it again: copy next to gemm_pp.hip
// and re-apply the finish_wgrad hook (git history: "ppw").

// gemm_ppw.hip — the persistent ping-pong kernel (gemm_pp.hip) for the weight gradients of the dense layers, bf16:
//     C[n][k] (fp32) = sum over rows m of  dy[m][n] * x[m][k]          (both operands REDUCTION-major: [m][.] as they lie in HBM)
// Build-authored: the upstream reference has no code (SURVEY.md finding zero).  Role: FusionTransformer linear weight / bias gradients.
//
// Same 8-phase schedule, wave roles and counted waits as gemm_pp.hip (read that header first); what differs:
//   * a K tile is 64 REDUCTION rows m.  Its four LDS pieces are cut by k-step, not by operand half: X0 = x rows [0,32), X1 = dy rows [0,32),
//     X2 = x rows [32,64), X3 = dy rows [32,64), each [32 rows][512 B] exactly as the rows lie in HBM (one DMA instruction = two 512-byte
//     row segments: full-line coalescing; the dy tile is 160 or 192 columns wide, its unused 32-byte units are out-of-range lanes).
//     Phases: 1 reads X0 + the a = 0 fragments of X1, 2 the a = 1 fragments of X1, 3 reads X2 + X3(a = 0), 4 X3(a = 1); the MFMAs of a phase
//     cover ONE k-step of one row half against all four column fragments.  Piece lifetimes (X0: phase 1, X1: 1-2, X2: 3, X3: 3-4) give the
//     same staging order and the same "overwrite two phases after the last read" rule as gemm_pp.hip.
//   * fragments come from `ds_read_b64_tr_b16` (two per fragment): the 16-lane group g of a wave reads reduction rows 8g .. 8g+7 of 16
//     columns.  32-byte unit u of row m sits at unit u ^ ((m & 3) | ((m >> 3) & 1) << 2) of its 512-byte LDS row: the eight rows one
//     half-wave touches land on eight different 32-byte bank groups (tools/lds_bank_check.py).
//   * split-K over workgroups (work item = (K slice, tile)): fp32 partial tiles go to slabs [slice][N][K] (or straight to dw when there is one
//     slice), 16 bytes per lane (4 consecutive k of one n: the accumulator as it stands), summed in fixed order by gemm.hip's reduce kernel.
//     The bias gradient db[n] = sum_m dy[m][n] rides along in the column-0 tiles: one extra MFMA per dy fragment against an all-ones operand,
//     dealt over the four waves of a wave row.
#include "msc_common.h"
#include "gemm_params.h"
#include "gemm_plan.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 pw_bf16x8;
typedef __attribute__((ext_vector_type(8))) short pw_s16x8;
typedef __attribute__((ext_vector_type(4))) short pw_s16x4;
typedef __attribute__((ext_vector_type(4))) float pw_f32x4;

#define PW_OOB 0x80000000u
#define PW_LDS(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __forceinline__ void pw_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
template <int N> __device__ __forceinline__ void pw_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// one MFMA fragment (8 reduction rows x 16 columns, transposed on the way out of LDS)
__device__ __forceinline__ pw_bf16x8 pw_frag(const char* addr) {
  const pw_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) pw_s16x4*)(addr));
  const pw_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) pw_s16x4*)(addr + 4 * 512));
  const pw_s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(pw_bf16x8, v);
}

// p.A = dy [Mred][lda], p.B = x [Mred][ldb], p.M = output rows (dy columns), p.N = output columns (x columns, a multiple of 256),
// p.K = reduction rows, p.C = fp32 [nsplit][p.M][p.N] (ldc = p.N), p.tiles_per_split = K tiles (64 rows) per slice (even), p.rowsum = fp32 [nsplit][p.M] or null
template <int MF0, int MF1>
__global__ __launch_bounds__(512, 2) void gemm_ppw_kernel(const GemmParams p) {
  constexpr int MI = MF0 + MF1, BM = 32 * MI, BN = 256;
  constexpr int PIECE = 32 * 512;                       // 16 KB: 32 reduction rows x 512 B
  constexpr int OFF_X0 = 0, OFF_X1 = PIECE, OFF_X2 = 2 * PIECE, OFF_X3 = 3 * PIECE;
  constexpr int BUF_BYTES = 4 * PIECE;
  constexpr int SMEM_BYTES = 2 * BUF_BYTES;             // 128 KB
  constexpr int INFL = 8;                               // DMA instructions of four pieces per wave
  constexpr int NST = 4 * MI + 1;                       // epilogue stores per wave (accumulators + the bias-gradient row)
  static_assert(MI <= 6, "accumulator registers");
  __shared__ __attribute__((aligned(1024))) char smem[SMEM_BYTES];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int lr = lane & 15, lg = lane >> 4;
  const int ntn = p.N / BN;
  const int ntm = (p.M + BM - 1) / BM;
  const int ntiles = ntm * ntn;
  const int nitems = ntiles * p.nsplit;
  const int NK = p.tiles_per_split;                     // K tiles per item (even; rows beyond p.K are staged as zeros)
  const int gstep = (int)gridDim.x;

  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc((void*)p.C, 0, (int)0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc((void*)p.rowsum, 0, (int)0x7ffffff0, 0x00020000);
  const uint32_t lda2 = (uint32_t)p.lda * 2u, ldb2 = (uint32_t)p.ldb * 2u;

  // work item -> (K slice, tile); items of one slice and one row tile are neighbours (same XCD: they stream the same dy rows)
  auto item_ref = [&](int it, int& m0, int& n0, int& split) __attribute__((always_inline)) {
    const bool live = it < nitems;
    const int bid = mscplan::xcd_remap(live ? it : 0, nitems);
    const int sp = bid / ntiles, t = bid - sp * ntiles;
    const int tm = t / ntn;
    m0 = __builtin_amdgcn_readfirstlane(live ? tm * BM : p.M);
    n0 = __builtin_amdgcn_readfirstlane(live ? (t - tm * ntn) * BN : 0);
    split = __builtin_amdgcn_readfirstlane(live ? sp : p.nsplit);      // a dead item's rows start beyond p.K: every DMA lane out of range
  };

  // ---- DMA side.  Instruction q = wave + 8 i (i = 0, 1) of a piece writes LDS rows 2q, 2q+1 (512 B each): lane -> row 2q + (lane >> 5),
  // 16-byte chunk (lane & 31) of the row = physical 32-byte unit (lane & 31) >> 1, half lane & 1; it holds logical unit
  // u = phys ^ swz(row).  Source: row (k-step base + row) of the operand, columns 16 u + 8 half ... of the tile.
  // x: all 256 columns exist.  dy: columns beyond the tile's BM (or beyond p.M in the last row tile) are out-of-range lanes.
  uint32_t relA[2], relB[2];         // per instruction: (row in piece) * pitch + column bytes; dy: out-of-range marker folded in per item
  int colA[2];                       // first dy column (inside the tile) of the lane's chunk, for the per-item bound
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = 2 * (wave + 8 * i) + (lane >> 5);
    const int phys = (lane & 31) >> 1, half = lane & 1;
    const int u = phys ^ ((row & 3) | (((row >> 3) & 1) << 2));
    relB[i] = (uint32_t)row * ldb2 + (uint32_t)(u * 32 + half * 16);
    relA[i] = (uint32_t)row * lda2 + (uint32_t)(u * 32 + half * 16);
    colA[i] = u * 16 + half * 8;
  }
  uint32_t voffA[2];
  auto set_voffA = [&](int m0) __attribute__((always_inline)) {
    const int cols_left = min(p.M - m0, BM);
#pragma unroll
    for (int i = 0; i < 2; ++i) voffA[i] = colA[i] < cols_left ? relA[i] : PW_OOB;
  };

  int c_m0, c_n0, c_sp, n_m0, n_n0, n_sp;
  // piece `x` (0..3) of K tile `kts` (>= NK: of the next item) into buffer `buf`; rows beyond p.K: the row bound is checked in scalar code
  // per row pair is not possible (two rows per instruction), so it is per lane: row < rows_left
  auto stage = [&](int kts, int buf, auto x_c) __attribute__((always_inline)) {
    constexpr int X = decltype(x_c)::value;
    constexpr bool IS_A = (X & 1) != 0;
    const bool wrap = kts >= NK;
    const int sp = wrap ? n_sp : c_sp;
    const int kk = wrap ? kts - NK : kts;
    const long row0 = ((long)sp * NK + kk) * 64 + (X >> 1) * 32;           // first reduction row of the piece
    const int rows_left = (int)min((long)32, max((long)0, (long)p.K - row0));
    char* dst = smem + buf * BUF_BYTES + X * PIECE;
    const uint32_t base = (uint32_t)row0 * (IS_A ? lda2 : ldb2) + (uint32_t)(IS_A ? (wrap ? n_m0 : c_m0) : (wrap ? n_n0 : c_n0)) * 2u;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = 2 * (wave + 8 * i) + (lane >> 5);
      const uint32_t rel = IS_A ? voffA[i] : relB[i];
      __builtin_amdgcn_raw_ptr_buffer_load_lds(IS_A ? rsA : rsB, PW_LDS(dst + (wave + 8 * i) * 1024), 16, row < rows_left ? rel : PW_OOB, base, 0, 0);
    }
  };
  using X0 = std::integral_constant<int, 0>;
  using X1 = std::integral_constant<int, 1>;
  using X2 = std::integral_constant<int, 2>;
  using X3 = std::integral_constant<int, 3>;

  // ---- fragment read addresses.  Lane: group g = lane >> 4 reads rows 8 g + q (q = (lane >> 2) & 3), 8 bytes at column quad lane & 3 of a
  // 32-byte unit; unit u of row m at (u ^ swz(m)) * 32.  swz(8 g + q) = q | (g & 1) << 2, the same for the second read (rows + 4).
  const int fq = (lane >> 2) & 3, fp = lane & 3;
  const int frow = 8 * lg + fq;
  const int fsw = fq | ((lg & 1) << 2);
  int offB[4], offA[MI];
#pragma unroll
  for (int j = 0; j < 4; ++j) offB[j] = frow * 512 + (((wn * 4 + j) ^ fsw) << 5) + fp * 8;
#pragma unroll
  for (int f = 0; f < MI; ++f) offA[f] = frow * 512 + (((wm * MI + f) ^ fsw) << 5) + fp * 8;

  pw_f32x4 acc[4][MI], racc[2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < MI; ++j) acc[i][j] = pw_f32x4{0.f, 0.f, 0.f, 0.f};
  racc[0] = racc[1] = pw_f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int MFX = MF0 > MF1 ? MF0 : MF1;
  pw_bf16x8 fa[MFX], fb[4];
  pw_bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
  bool do_rowsum = false;

  auto read_b = [&](int buf, auto s_c) __attribute__((always_inline)) {
    constexpr int S = decltype(s_c)::value;
    const char* base = smem + buf * BUF_BYTES + (S ? OFF_X2 : OFF_X0);
#pragma unroll
    for (int j = 0; j < 4; ++j) fb[j] = pw_frag(base + offB[j]);
  };
  auto read_a = [&](int buf, auto s_c, auto a_c) __attribute__((always_inline)) {
    constexpr int S = decltype(s_c)::value, Ah = decltype(a_c)::value;
    constexpr int MFS = Ah ? MF1 : MF0, MB = Ah ? MF0 : 0;
    const char* base = smem + buf * BUF_BYTES + (S ? OFF_X3 : OFF_X1);
#pragma unroll
    for (int f = 0; f < MFS; ++f) fa[f] = pw_frag(base + offA[MB + f]);
  };
  // one k-step of row half a against the four column fragments (+ the bias-gradient MFMAs of this wave's share of the dy fragments)
  auto mma = [&](auto a_c) __attribute__((always_inline)) {
    constexpr int Ah = decltype(a_c)::value;
    constexpr int MFS = Ah ? MF1 : MF0, MB = Ah ? MF0 : 0;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int f = 0; f < MFS; ++f)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[j][MB + f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[f], acc[j][MB + f], 0, 0, 0);
    if (do_rowsum) {
#pragma unroll
      for (int f = 0; f < MFS; ++f) {
        if (((MB + f) & 3) == wn) racc[(MB + f) >> 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, fa[f], racc[(MB + f) >> 2], 0, 0, 0);
      }
    }
    __builtin_amdgcn_s_setprio(0);
  };

  auto wait_phase = [&](auto x_c) __attribute__((always_inline)) { pw_wait_vm<INFL + decltype(x_c)::value>(); };
  using W0 = std::integral_constant<int, 0>;
  using WS = std::integral_constant<int, NST>;

  // ---- prologue
  int it = (int)blockIdx.x;
  item_ref(it, c_m0, c_n0, c_sp);
  item_ref(it + gstep, n_m0, n_n0, n_sp);
  set_voffA(c_m0);
  stage(0, 0, X0{}); stage(0, 0, X1{}); stage(0, 0, X2{}); stage(0, 0, X3{});
  stage(1, 1, X0{}); stage(1, 1, X1{});
  wait_phase(W0{});
  if (wm == 1) pw_barrier();
  pw_barrier();

  // K tile t (buffer U):  1: read X0, X1(a0) | DMA X2(t+1) | k-step 0, a0      3: read X2, X3(a0) | DMA X0(t+2) | k-step 1, a0
  //                       2: read X1(a1)     | DMA X3(t+1) | k-step 0, a1      4: read X3(a1)     | DMA X1(t+2) | k-step 1, a1
  // after_epi: this is the first K tile after an epilogue (its stores are younger than the pieces that were in flight then)
  auto ktile = [&](int kt, bool after_epi, bool last_pair, auto u_c) __attribute__((always_inline)) {
    constexpr int U = decltype(u_c)::value;
    read_b(U, X0{});
    read_a(U, X0{}, X0{});
    stage(kt + 1, U ^ 1, X2{});
    if (after_epi) wait_phase(WS{}); else wait_phase(W0{});
    pw_barrier();
    mma(X0{});
    pw_barrier();
    read_a(U, X0{}, X1{});
    stage(kt + 1, U ^ 1, X3{});
    if (after_epi) wait_phase(WS{}); else wait_phase(W0{});
    pw_barrier();
    mma(X1{});
    pw_barrier();
    read_b(U, X1{});
    read_a(U, X1{}, X0{});
    stage(kt + 2, U, X0{});
    if (after_epi) wait_phase(WS{}); else wait_phase(W0{});
    pw_barrier();
    mma(X0{});
    pw_barrier();
    read_a(U, X1{}, X1{});
    if constexpr (U == 0) {
      if (last_pair) set_voffA(n_m0);      // X1(t+2) below is the next item's first dy piece: the dy stream moves on here, for good
    }
    stage(kt + 2, U, X1{});
    if (after_epi) wait_phase(WS{}); else wait_phase(W0{});
    pw_barrier();
    mma(X1{});
    pw_barrier();
  };

  const uint32_t ldc4 = (uint32_t)p.ldc * 4u;
  bool after_epi = false;
  for (; it < nitems; it += gstep) {
    do_rowsum = p.rowsum != nullptr && c_n0 == 0;
    const int npairs = NK / 2;
    for (int kp = 0; kp < npairs; ++kp) {
      ktile(2 * kp, kp == 0 && after_epi, kp == npairs - 1, X0{});
      ktile(2 * kp + 1, false, false, X1{});
    }
    // ---- epilogue of the item: fp32 partial tile, 16 bytes per lane; NST store instructions per wave, always (out-of-range lanes drop)
    {
      const int rows_left = p.M - c_m0;
      const uint32_t tile_off = (uint32_t)c_sp * (uint32_t)p.M * ldc4 + (uint32_t)c_m0 * ldc4 + (uint32_t)c_n0 * 4u;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int row = wm * (16 * MI) + mi * 16 + lr;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t off = row < rows_left ? tile_off + (uint32_t)row * ldc4 + (uint32_t)(wn * 64 + j * 16 + 4 * lg) * 4u : PW_OOB;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) uint32_t, acc[j][mi]), rsC, off, 0, 0);
          acc[j][mi] = pw_f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
      // bias-gradient partial of this slice: every register of racc[t] holds the column sum of dy column lr of fragment wn + 4 t.
      // ONE store instruction per wave: lanes 0-15 write fragment wn's 16 sums, lanes 16-31 fragment wn + 4's (when the tile has one)
      const int rmi = wn + 4 * lg;                         // fragment this lane group writes (lg = 0, 1)
      const int rr = wm * (16 * MI) + rmi * 16 + lr;
      const float rv = lg == 0 ? racc[0][0] : racc[1][0];
      const uint32_t roff = (do_rowsum && lg < 2 && rmi < MI && rr < rows_left) ? ((uint32_t)c_sp * (uint32_t)p.M + (uint32_t)(c_m0 + rr)) * 4u : PW_OOB;
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, rv), rsS, roff, 0, 0);
      racc[0] = racc[1] = pw_f32x4{0.f, 0.f, 0.f, 0.f};
    }
    after_epi = true;
    c_m0 = n_m0; c_n0 = n_n0; c_sp = n_sp;
    item_ref(it + 2 * gstep, n_m0, n_n0, n_sp);
  }
  if (wm == 0) pw_barrier();
  pw_wait_vm<0>();
}

}  // namespace

// Host side.  The launch replaces the tiled kernel inside gemm.hip's finish_wgrad: same slab layout ([slice][M][N] fp32 + [slice][M] bias
// rows behind it), same reduce kernel.  ns_max: slices the caller's workspace has room for.  Returns 0 when the shape is not taken.
int msc_ppw_plan(const GemmParams& p, int ns_max, int ncu, int& bm, int& ns, int& tps) {
  if (!mscplan::g_knobs.gemm_ppw) return 0;
  if (p.N % 256 || p.N < 256 || p.M < 128 || p.K < 4096 || p.lda % 8 || p.ldb % 8) return 0;
  if ((long)p.K * p.lda * 2 >= 0x7ffffff0L || (long)p.K * p.ldb * 2 >= 0x7ffffff0L) return 0;
  const long nkt = mscplan::cdivl(p.K, 64);
  long best_cost = 0;
  bm = 0;
  for (int b : {192, 160}) {
    const long tiles = mscplan::cdivl(p.M, b) * (p.N / 256);
    long s = ncu / tiles;                                  // one round of workgroups
    if (s < 1) s = 1;
    if (s > ns_max) s = ns_max;
    long t = mscplan::cdivl(nkt, s);
    t += t & 1;                                            // K tiles come in pairs
    if (t < 4) t = 4;
    s = mscplan::cdivl(nkt, t);
    if ((long)s * p.M * p.N * 4 >= 0x7ffffff0L) continue;
    const long rounds = mscplan::cdivl(tiles * s, ncu);
    const long cost = rounds * t * (b + 32);
    if (!bm || cost < best_cost) { bm = b; ns = (int)s; tps = (int)t; best_cost = cost; }
  }
  return bm;
}

int msc_ppw_launch(const GemmParams& p, int bm, int ncu, hipStream_t st) {
  const int nitems = cdiv(p.M, bm) * (p.N / 256) * p.nsplit;
  dim3 grid((unsigned)(nitems < ncu ? nitems : ncu), 1, 1);
  if (bm == 192) hipLaunchKernelGGL((gemm_ppw_kernel<3, 3>), grid, dim3(512), 0, st, p);
  else if (bm == 160) hipLaunchKernelGGL((gemm_ppw_kernel<3, 2>), grid, dim3(512), 0, st, p);
  else return MSC_ERR_UNSUPPORTED;
  return msc_launch_status();
}
