// gemm_skinny.hip — dense bf16 linear layers with at most 64 rows (gfx950).  Build-authored (the upstream reference has no code).
//
// The codec / head / state-encoder stretch of the step is ~20 linear layers over ONE ROW PER FRAME (64 rows at the benchmark batch): a few
// MFLOP each, issued one after the other on the critical chain.  The tiled kernel of gemm.hip runs them as 1..8 workgroups walking a
// staged K loop (LDS-DMA -> barrier -> MFMA, three stages deep): 7-14 us per launch for work that is one memory round trip long (kernel
// trace, round 4: profiles/r4_head_region.txt).  Here there is no staging at all:
//   * one workgroup per 16 output columns, its NW waves split the reduction (K) between them;
//   * a wave's operands go global -> registers in MFMA fragment layout directly (v_mfma_f32_16x16x32_bf16: lane l holds row / column
//     l & 15, k = 8 (l >> 4) .. + 7 — one 16-byte load per lane per fragment), every load of a chunk of k-steps issued before the
//     first MFMA of the chunk, the next chunk's loads in flight under it;
//   * the NW partial accumulators meet in LDS and are summed in wave order (fixed order: bit-reproducible), then the epilogue of the
//     tiled kernel in the same order of operations: + bias, activation, + residual, x act'(dref), bf16 store.
// Selected by launch_gemm (gemm.hip) for M <= 64, k-major operands, no split / statistics / second output; knob gemm_skinny = 0 turns
// it off.  fp32 launches stay on the tiled kernel.
#include "msc_common.h"
#include "gemm_params.h"
#include "gemm_plan.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 sk_bf16x8;
typedef __attribute__((ext_vector_type(4))) float sk_f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t sk_u32x4;

constexpr int SK_MT = 4;        // 16-row tiles of the output a workgroup covers (M <= 64)
constexpr int SK_CH = 4;        // k-steps (32 reduction elements each) per chunk of loads

// one chunk of fragments: SK_CH k-steps x (1 weight fragment + SK_MT activation fragments)
struct SkChunk {
  sk_u32x4 b[SK_CH];
  sk_u32x4 a[SK_CH][SK_MT];
};

template <int NW>
__global__ __launch_bounds__(64 * NW) void gemm_skinny_kernel(const GemmParams p) {
  __shared__ float part[NW][SK_MT][4][64];      // [wave][row tile][accumulator register][lane]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  const int n0 = blockIdx.x * 16;
  const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A);
  const bf16_t* B = reinterpret_cast<const bf16_t*>(p.B);

  // k-steps of this wave: a contiguous range of the K / 32 steps (the last wave may get fewer, or none)
  const int nks = (p.K + 31) >> 5;                         // K % 8 == 0; the lanes of a last partial step beyond K load zeros
  const int per = (nks + NW - 1) / NW;
  const int ks0 = wave * per, ks1 = min(nks, ks0 + per);

  // per-lane element offsets of the fragments at k-step 0
  const bool ncol_ok = n0 + lr < p.N;                       // N % 16 == 0 is required by the host; kept for safety
  const long boff = (long)(n0 + lr) * p.ldb + 8 * lg;
  long aoff[SK_MT];
  bool arow_ok[SK_MT];
#pragma unroll
  for (int t = 0; t < SK_MT; ++t) {
    arow_ok[t] = 16 * t + lr < p.M;
    aoff[t] = (long)(16 * t + lr) * p.lda + 8 * lg;
  }
  const int mt = (p.M + 15) >> 4;                            // row tiles that exist (wave-uniform): the others are never loaded or multiplied
  const sk_u32x4 zero = {0u, 0u, 0u, 0u};
  auto load_chunk = [&](SkChunk& c, int ks) __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < SK_CH; ++s) {
      const bool kin = ks + s < ks1 && (ks + s) * 32 + 8 * lg < p.K;
      c.b[s] = (kin && ncol_ok) ? *reinterpret_cast<const sk_u32x4*>(B + boff + (long)(ks + s) * 32) : zero;
#pragma unroll
      for (int t = 0; t < SK_MT; ++t)
        if (t < mt) c.a[s][t] = (kin && arow_ok[t]) ? *reinterpret_cast<const sk_u32x4*>(A + aoff[t] + (long)(ks + s) * 32) : zero;
    }
  };

  sk_f32x4 acc[SK_MT];
#pragma unroll
  for (int t = 0; t < SK_MT; ++t) acc[t] = sk_f32x4{0.f, 0.f, 0.f, 0.f};
  auto mma_chunk = [&](const SkChunk& c) __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < SK_CH; ++s)
#pragma unroll
      for (int t = 0; t < SK_MT; ++t)
        if (t < mt) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(sk_bf16x8, c.a[s][t]), __builtin_bit_cast(sk_bf16x8, c.b[s]), acc[t], 0, 0, 0);
  };

  // two chunks in flight: the loads of chunk i + 1 are issued before the MFMAs of chunk i
  SkChunk c0, c1;
  if (ks0 < ks1) load_chunk(c0, ks0);
  for (int ks = ks0; ks < ks1; ks += 2 * SK_CH) {
    if (ks + SK_CH < ks1) load_chunk(c1, ks + SK_CH);
    mma_chunk(c0);
    if (ks + SK_CH < ks1) {
      if (ks + 2 * SK_CH < ks1) load_chunk(c0, ks + 2 * SK_CH);
      mma_chunk(c1);
    }
  }

  // partial sums -> LDS; wave t finishes row tile t (all NW partials in wave order)
#pragma unroll
  for (int t = 0; t < SK_MT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) part[wave][t][r][lane] = acc[t][r];
  __syncthreads();
  for (int t = wave; t < SK_MT; t += NW) {
    if (16 * t >= p.M) break;
    const int n = n0 + lr;
    const float bias = (p.bias && ncol_ok) ? p.bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) v += part[w][t][r][lane];
      const int m = 16 * t + 4 * lg + r;                    // accumulator layout: column lane & 15, row 4 (lane >> 4) + register
      if (m >= p.M || !ncol_ok) continue;
      const long o = (long)m * p.ldc + n;
      v = p.alpha * v + bias;
      v = act_apply(v, p.act);
      if (p.res) v += bf16_to_f32(reinterpret_cast<const bf16_t*>(p.res)[o]);
      if (p.dref) v *= act_grad(bf16_to_f32(reinterpret_cast<const bf16_t*>(p.dref)[o]), p.dact);
      reinterpret_cast<bf16_t*>(p.C)[o] = f32_to_bf16(v);
    }
  }
}

}  // namespace

// Host side: does the skinny kernel take this dense bf16 launch?  (k-major operands are the caller's guarantee: launch_gemm asks only
// for OPM_DENSE_K x OPM_DENSE_K.)
bool msc_skinny_ok(const GemmParams& p) {
  if (!mscplan::g_knobs.gemm_skinny || mscplan::g_knobs.tile_bm) return false;     // (a forced tile means the tiled family)
  if (p.M > 16 * SK_MT || p.N % 16 || p.K % 8) return false;
  if (p.lda % 8 || p.ldb % 8) return false;
  if (p.nsplit != 1 || p.xcd_split || p.stats || p.colsum || p.bnpart || p.rowsum || p.out_f32 || p.C2) return false;
  if (p.act == MSC_ACT_GELU_DSAVE) return false;
  if (((uintptr_t)p.A | (uintptr_t)p.B) & 15) return false;
  return true;
}

int msc_skinny_launch(const GemmParams& p, hipStream_t st) {
  const unsigned nwg = (unsigned)(p.N / 16);
  // long reductions (recon.fc1's dgrad: 3136) over eight waves, the rest over four
  if (p.K >= 2048) hipLaunchKernelGGL(gemm_skinny_kernel<8>, dim3(nwg), dim3(512), 0, st, p);
  else hipLaunchKernelGGL(gemm_skinny_kernel<4>, dim3(nwg), dim3(256), 0, st, p);
  return msc_launch_status();
}
