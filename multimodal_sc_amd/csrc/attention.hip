// attention.hip — fused cross-modal attention of the FusionTransformer (flash style, bf16 storage, fp32 softmax),
// written for gfx950: head_dim 64, sequences of a few hundred tokens (T = 306 at the benchmark config).
//
// One workgroup per (batch, head).  The K and V (forward) / K,V (dQ pass) / Q,dO (dK,dV pass) panels of the head are
// staged ONCE into LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`, 8 rows of 128 B per wave-instruction, zero fill
// past T, XOR swizzle applied on the source side) and stay resident; each wave owns query tiles (or key tiles) of 32.
// All products run on v_mfma_f32_32x32x16_bf16 in the "transposed" orientation so that the softmax statistics are
// per-LANE:   S^T[key][q] = K Q^T   ->  P^T stays in the accumulator registers, is rounded to bf16 in place and used
// directly as the B operand of   O^T[d][q] += V^T[d][key] P^T[key][q]   (accumulator-as-operand, permuted k order),
// with V^T fragments produced by the hardware transpose read ds_read_b64_tr_b16 from the row-major V image.
// No score/probability matrix ever touches HBM; the backward recomputes P from the saved log-sum-exp.
//   msc_attention_fwd : qkv [B*T][3*H*64] -> o [B*T][H*64], lse [B*H][Tpad] (log2 domain)
//   msc_attention_bwd : + do -> dqkv [B*T][3*H*64]      (two kernels: dQ (+delta), then dK/dV)
// Deterministic: every output element is produced by exactly one wave, no atomics.
// Build-authored (the upstream reference has no code, SURVEY.md finding zero).  Role: FusionTransformer.
#include "msc_common.h"
#include "gemm_plan.h"     // mscplan::g_knobs (tuning knobs, msc_set_knob)
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bfx8;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
#define LDSP(p) ((__attribute__((address_space(3))) void*)(p))
#define ATT_OOB 0x80000000u
#define ATT_HD 64
// ATT_ABLATE (tools/attn_ablate.py only; results are WRONG by construction), forward kernel:
//   1 no K/V staging   2 no softmax arithmetic   3 no PV products / V reads   4 no QK products / K reads   5 no Q loads
#ifndef ATT_ABLATE
#define ATT_ABLATE 0
#endif
#ifndef ATT_SCHED_FENCE
#define ATT_SCHED_FENCE 1   // keep the compiler from hoisting one stage's LDS reads over the other stage (register pressure)
#endif
#if ATT_SCHED_FENCE
#define ATT_SB() __builtin_amdgcn_sched_barrier(0)
#else
#define ATT_SB() ((void)0)
#endif
#ifndef ATT_BWD_OCC
#define ATT_BWD_OCC 2      // waves per SIMD the backward kernels are compiled for (3 = 168 VGPRs spills both: 9 / 65 registers)
#endif

// ATT_STAMP: diagnostic build only (make alt ALT=attstamp ALT_SRC=attention ALT_FLAGS=-DATT_STAMP; tools/attn_stamps.py).  Wave 0 of every
// workgroup of the dK/dV pass sums shader-clock cycles per loop segment into a debug buffer of its own that nothing else reads:
// [0] statistics shuffle + S'/dP' (LDS row reads + 8 MFMAs, until their results can be read), [1] exponentials / dS / bf16 packing,
// [2] dV / dK (16 transposed LDS reads + 8 MFMAs, until results readable), [3] whole loop of the wave's first key tile, [4] prologue
// (staging), [5] iterations, [6] / [7] clock at kernel entry / after the wave's last store.
#ifdef ATT_STAMP
__device__ unsigned long long g_att_stamps[4096 * 8];
extern "C" int msc_debug_attn_stamps(unsigned long long* host_out, int nwg) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_att_stamps), (size_t)nwg * 8 * sizeof(unsigned long long)) == hipSuccess ? MSC_OK : MSC_ERR_LAUNCH;
}
#define ATT_T(var) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#define ATT_TOUCH(x) do { float t_; asm volatile("v_mov_b32 %0, %1" : "=v"(t_) : "v"(x)); asm volatile("" :: "v"(t_)); } while (0)
#endif

// Chunk swizzle of a 128-B-row image that is read BOTH by rows (ds_read_b128: the 8 even and 8 odd rows of a 16-lane
// group must spread one chunk index over 8 distinct values -> a bijection of t = (row>>1)&7) AND transposed
// (ds_read_b64_tr_b16: a 32-lane half reads 64 contiguous bytes of rows 4m..4m+3, so rows 4m and 4m+2 — same 256-B bank
// window — must land in different 64-B halves -> bit 2 of the swizzle must be t's LOW bit).  f(t) = (t0 << 2) | (t >> 1).
__device__ __forceinline__ int swz_row(int row) { return ((row & 2) << 1) | ((row >> 2) & 3); }
__device__ __forceinline__ int swz_tr(int row) { return ((row >> 1) & 1) << 2; }         // ds_read_b64_tr_b16 only

// Stage rows [0,Tpad) x 64 columns of a bf16 matrix (element (r,c) at elem0 + r*ld + c) into an LDS image of 128-B rows.
// DUAL = true: image read both by rows (b128) and transposed -> row swizzle; false: transposed reads only.
template <bool DUAL>
__device__ __forceinline__ void stage_image(char* img, __amdgpu_buffer_rsrc_t rs, uint32_t elem0, int ld, int T, int Tpad, int nwaves) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int j = wave; j < (Tpad >> 3); j += nwaves) {
    const int row = 8 * j + (lane >> 3), pos = lane & 7;
    const int c = pos ^ (DUAL ? swz_row(row) : swz_tr(row));
    const uint32_t v = row < T ? (elem0 + (uint32_t)row * (uint32_t)ld + (uint32_t)c * 8u) * 2u : ATT_OOB;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDSP(img + j * 1024), 16, v, 0, 0, 0);
  }
}

// A/B fragment of rows [row0, row0+32) x k-chunk c (8 bf16) from a row-read image: lane (r = lane&31) reads row row0+r
__device__ __forceinline__ bfx8 frag_rows(const char* img, int row, int c) {
  return *reinterpret_cast<const bfx8*>(img + row * 128 + ((c ^ swz_row(row)) << 4));
}

// A fragment (32 rows = image COLUMNS d0..d0+31, k = image rows) by hardware transpose, in the k order an accumulator
// tile has when it is reused as the B operand: element j of lane half h <-> k = k0 + 8*(j>>2) + 4*h + (j&3).
template <bool DUAL>
__device__ __forceinline__ bfx8 frag_tr_perm(const char* img, int k0, int d0) {
  const int lane = threadIdx.x & 63;
  const int tg = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
  const int col = d0 + 16 * (tg & 1) + 4 * tp;
  const int r_lo = k0 + 4 * (tg >> 1) + tq, r_hi = r_lo + 8;
  const int s_lo = DUAL ? swz_row(r_lo) : swz_tr(r_lo), s_hi = DUAL ? swz_row(r_hi) : swz_tr(r_hi);
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(img + r_lo * 128 + (((col >> 3) ^ s_lo) << 4) + (tp & 1) * 8));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(img + r_hi * 128 + (((col >> 3) ^ s_hi) << 4) + (tp & 1) * 8));
  s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bfx8, v);
}

// 8 bf16 of one matrix row straight from HBM (row-per-lane B fragments: Q / dO / K / V of the wave's own tile)
__device__ __forceinline__ bfx8 frag_global(const bf16_t* rowptr, int c, bool valid) {
  uint4 v = make_uint4(0, 0, 0, 0);
  if (valid) v = *reinterpret_cast<const uint4*>(rowptr + c * 8);
  return __builtin_bit_cast(bfx8, v);
}

__device__ __forceinline__ bfx8 pack8(const f32x16_t& a, int s) {
  bfx8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (__bf16)a[8 * s + j];
  return r;
}

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)

// row index (inside a 32-tile) that accumulator register r of lane-half lh holds
__device__ __forceinline__ int acc_row(int r, int lh) { return (r & 3) + 8 * (r >> 2) + 4 * lh; }

// store an O^T-style accumulator pair (d in registers, row on lanes) as 8-byte packets: out[row][col0 + d]
__device__ __forceinline__ void store_rowT(bf16_t* rowptr, const f32x16_t (&acc)[2], float mul, int lh) {
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      uint32_t w0 = (uint32_t)f32_to_bf16(acc[dt][4 * g + 0] * mul) | ((uint32_t)f32_to_bf16(acc[dt][4 * g + 1] * mul) << 16);
      uint32_t w1 = (uint32_t)f32_to_bf16(acc[dt][4 * g + 2] * mul) | ((uint32_t)f32_to_bf16(acc[dt][4 * g + 3] * mul) << 16);
      *reinterpret_cast<uint2*>(rowptr + dt * 32 + 8 * g + 4 * lh) = make_uint2(w0, w1);
    }
}

// ================================================================================================ forward
// grid = B*H, block = 64 * ceil(NT/2) threads; wave w owns query tiles 2w and 2w+1 and sweeps all key tiles.
__global__ __launch_bounds__(512) void attn_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ o, float* __restrict__ lse,
                                                        int T, int Tpad, int H, float c_log2) {
  MSC_BOOST();
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  char* Kimg = smem;
  char* Vimg = smem + Tpad * 128;
  const int D = H * ATT_HD, ld = 3 * D;
  const int bh = blockIdx.x, b = bh / H, h = bh % H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int NT = Tpad >> 5;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)qkv, 0, (int)0x7ffffff0, 0x00020000);
  const uint32_t row0 = (uint32_t)b * (uint32_t)T * (uint32_t)ld;
  if (ATT_ABLATE != 1) {
    stage_image<true>(Kimg, rs, row0 + D + h * ATT_HD, ld, T, Tpad, nwaves);
    stage_image<false>(Vimg, rs, row0 + 2 * D + h * ATT_HD, ld, T, Tpad, nwaves);
  }

  // this wave's two query tiles: B fragments (lane = query row) straight from HBM while the DMA is in flight
  bfx8 qf[2][4];
  int qrow[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    qrow[t] = (2 * wave + t) * 32 + lr;
    const bf16_t* qp = qkv + ((long)b * T + (qrow[t] < T ? qrow[t] : 0)) * ld + h * ATT_HD;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[t][ks] = frag_global(qp, 2 * ks + lh, ATT_ABLATE != 5 && qrow[t] < T);
  }
  f32x16_t oacc[2][2];
  float m[2], l[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    m[t] = -INFINITY;
    l[t] = 0.f;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[t][dt][r] = 0.f;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int kt = 0; kt < NT; ++kt) {
    f32x16_t s[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ATT_ABLATE == 4) break;
      const bfx8 kf = frag_rows(Kimg, kt * 32 + lr, 2 * ks + lh);
      s[0] = MFMA(kf, qf[0][ks], s[0]);
      s[1] = MFMA(kf, qf[1][ks], s[1]);
    }
    bfx8 pb[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (ATT_ABLATE == 2) { pb[t][0] = pack8(s[t], 0); pb[t][1] = pack8(s[t], 1); continue; }
      // VALU diet: the ragged-tail mask only on the last key tile (wave-uniform branch); the row maximum is taken on the
      // raw scores and scaled once (c_log2 > 0); scale and max-subtraction fold into one FMA per element below
      if (kt == NT - 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kt * 32 + acc_row(r, lh) >= T) s[t][r] = -INFINITY;
      }
      float mx = s[t][0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[t][r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * c_log2;
      if (__any(mx > m[t])) {            // exact online softmax; rescale only when some row's maximum moved
        const float mn = fmaxf(m[t], mx);
        const float alpha = __builtin_amdgcn_exp2f(m[t] - mn);
        m[t] = mn;
        l[t] *= alpha;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) oacc[t][dt][r] *= alpha;
      }
      float ps = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s[t][r] = __builtin_amdgcn_exp2f(fmaf(s[t][r], c_log2, -m[t]));
        ps += s[t][r];
      }
      l[t] += ps;
      pb[t][0] = pack8(s[t], 0);
      pb[t][1] = pack8(s[t], 1);
    }
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int sk = 0; sk < 2; ++sk) {
        if (ATT_ABLATE == 3) { oacc[0][dt][0] += (float)pb[0][sk][0]; oacc[1][dt][0] += (float)pb[1][sk][0]; continue; }
        const bfx8 vf = frag_tr_perm<false>(Vimg, kt * 32 + 16 * sk, dt * 32);
        oacc[0][dt] = MFMA(vf, pb[0][sk], oacc[0][dt]);
        oacc[1][dt] = MFMA(vf, pb[1][sk], oacc[1][dt]);
      }
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const float lt = l[t] + __shfl_xor(l[t], 32, 64);
    if (qrow[t] < T) {
      store_rowT(o + ((long)b * T + qrow[t]) * D + h * ATT_HD, oacc[t], 1.f / lt, lh);
      if (lh == 0) lse[(long)bh * Tpad + qrow[t]] = m[t] + __builtin_amdgcn_logf(lt);   // v_log_f32 = log2
    }
  }
}

// ================================================================================================ backward: dQ (+ delta)
// wave w owns query tiles w, w+nwaves, ...; K and V panels resident in LDS.  (One wave per tile was tried: at ~190 VGPRs
// per wave the extra waves only fit by spilling, and the kernel got slower.)
__global__ __launch_bounds__(512, ATT_BWD_OCC) void attn_bwd_dq_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ o,
                                                           const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                           float* __restrict__ delta, bf16_t* __restrict__ dqkv, int T, int Tpad,
                                                           int H, float c_log2, float scale) {
  MSC_BOOST();
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  char* Kimg = smem;                 // dual use: rows for S^T, transposed for dQ
  char* Vimg = smem + Tpad * 128;    // rows only (dP^T), staged with the row swizzle
  const int D = H * ATT_HD, ld = 3 * D;
  const int bh = blockIdx.x, b = bh / H, h = bh % H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int NT = Tpad >> 5;
  const long stat2 = (long)gridDim.x * Tpad;      // second half of the statistics scratch: -lse / c
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)qkv, 0, (int)0x7ffffff0, 0x00020000);
  const uint32_t row0 = (uint32_t)b * (uint32_t)T * (uint32_t)ld;
  if (ATT_ABLATE != 11) {
    stage_image<true>(Kimg, rs, row0 + D + h * ATT_HD, ld, T, Tpad, nwaves);
    stage_image<true>(Vimg, rs, row0 + 2 * D + h * ATT_HD, ld, T, Tpad, nwaves);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int qt = wave; qt < NT; qt += nwaves) {
    const int qrow = qt * 32 + lr;
    const bool qv = qrow < T;
    const long grow = (long)b * T + (qv ? qrow : 0);
    bfx8 qf[4], gf[4];
    float dl = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      qf[ks] = frag_global(qkv + grow * ld + h * ATT_HD, 2 * ks + lh, qv);
      gf[ks] = frag_global(dout + grow * D + h * ATT_HD, 2 * ks + lh, qv);
      const bfx8 of = frag_global(o + grow * D + h * ATT_HD, 2 * ks + lh, qv);
#pragma unroll
      for (int j = 0; j < 8; ++j) dl += (float)gf[ks][j] * (float)of[j];
    }
    dl += __shfl_xor(dl, 32, 64);                       // delta[q] = sum_d dO*O
    const float ls = qv ? lse[(long)bh * Tpad + qrow] : INFINITY;   // padded query rows: P = 0
    if (qv && lh == 0) {      // row constants of the dK/dV pass, stored the way that pass consumes them (accumulator initial values)
      delta[(long)bh * Tpad + qrow] = -dl;
      delta[stat2 + (long)bh * Tpad + qrow] = -ls / c_log2;
    }
    f32x16_t dq[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) dq[dt][r] = 0.f;
    for (int kt = 0; kt < NT; ++kt) {
      f32x16_t s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if (ATT_ABLATE == 14) break;
        s = MFMA(frag_rows(Kimg, kt * 32 + lr, 2 * ks + lh), qf[ks], s);
        dp = MFMA(frag_rows(Vimg, kt * 32 + lr, 2 * ks + lh), gf[ks], dp);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (ATT_ABLATE == 12) { s[r] += dp[r]; continue; }
        const float p = __builtin_amdgcn_exp2f(fmaf(s[r], c_log2, -ls));
        s[r] = p * (dp[r] - dl);                        // dS^T
      }
      if (kt == NT - 1) {                               // ragged tail: keys >= T carry no probability
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (kt * 32 + acc_row(r, lh) >= T) s[r] = 0.f;
      }
      const bfx8 d0 = pack8(s, 0), d1 = pack8(s, 1);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        if (ATT_ABLATE == 13) { dq[dt][0] += (float)d0[0] + (float)d1[0]; continue; }
        dq[dt] = MFMA(frag_tr_perm<true>(Kimg, kt * 32, dt * 32), d0, dq[dt]);
        dq[dt] = MFMA(frag_tr_perm<true>(Kimg, kt * 32 + 16, dt * 32), d1, dq[dt]);
      }
    }
    if (qv) store_rowT(dqkv + grow * ld + h * ATT_HD, dq, scale, lh);
  }
}

// ================================================================================================ backward: dK, dV
// wave w owns key tiles w, w+nwaves, ...; Q and dO panels resident in LDS (rows + transposed reads).
// (Rejected in round 3, removed from the source in round 4: K / V panels in LDS too with a second S' / dP' accumulator set and a software-
// pipelined loop over the query tiles — correct, 116-117 us per layer against 111: profiles/r3_attn_ablate.txt.)
__global__ __launch_bounds__(512, ATT_BWD_OCC) void attn_bwd_dkv_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dout,
                                                            const float* __restrict__ lse, const float* __restrict__ delta,
                                                            bf16_t* __restrict__ dqkv, int T, int Tpad, int H, float c_log2, float scale) {
  MSC_BOOST();
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  char* Qimg = smem;
  char* Gimg = smem + Tpad * 128;
  const int D = H * ATT_HD, ld = 3 * D;
  const int bh = blockIdx.x, b = bh / H, h = bh % H;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int NT = Tpad >> 5;
  __amdgpu_buffer_rsrc_t rsq = __builtin_amdgcn_make_buffer_rsrc((void*)qkv, 0, (int)0x7ffffff0, 0x00020000);
  __amdgpu_buffer_rsrc_t rsg = __builtin_amdgcn_make_buffer_rsrc((void*)dout, 0, (int)0x7ffffff0, 0x00020000);
#ifdef ATT_STAMP
  unsigned long long tp0, tp1;
  ATT_T(tp0);
#endif
  if (ATT_ABLATE != 11) {
    stage_image<true>(Qimg, rsq, (uint32_t)b * (uint32_t)T * (uint32_t)ld + h * ATT_HD, ld, T, Tpad, nwaves);
    stage_image<true>(Gimg, rsg, (uint32_t)b * (uint32_t)T * (uint32_t)D + h * ATT_HD, D, T, Tpad, nwaves);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#ifdef ATT_STAMP
  ATT_T(tp1);
#endif
  // row constants as the dQ pass left them: -delta and -lse / c.  S' = Q K^T - lse / c (so p = exp2(c S')) and dP' = dO V^T - delta
  // (so dS = p dP'): the constants enter the products as ONE MORE K-STEP of the MFMA chain (see aug() below)
  const float* nls_b = delta + (long)gridDim.x * Tpad + (long)bh * Tpad;
  const float* ndl_b = delta + (long)bh * Tpad;
  (void)lse;

  for (int kt = wave; kt < NT; kt += nwaves) {
    const int krow = kt * 32 + lr;
    const bool kvld = krow < T;
    const long grow = (long)b * T + (kvld ? krow : 0);
    bfx8 kf[4], vf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      kf[ks] = frag_global(qkv + grow * ld + D + h * ATT_HD, 2 * ks + lh, kvld);
      vf[ks] = frag_global(qkv + grow * ld + 2 * D + h * ATT_HD, 2 * ks + lh, kvld);
    }
    f32x16_t dk[2], dv[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) { dk[dt][r] = 0.f; dv[dt][r] = 0.f; }
    // The 32 row constants of a query tile travel ONE PER LANE (lane l and l+32 hold row l & 31): two 4-byte loads per tile, issued a
    // whole tile ahead.  A row constant is a per-ROW bias of a 32x32 product whose rows are this operand's rows: it is the product of
    // one extra k-step whose A fragment holds the constant — split into three bf16 terms (hi + mid + lo = 24 significant bits, every
    // product with 1.0 exact, summed in the fp32 accumulator) in k slots 0..2 of lane half 0, zeros elsewhere — and whose B fragment
    // is all ones.  Two more MFMAs per tile (16 -> 18) replace 32 ds_bpermute (the LDS pipe of the CU is the contended unit of this
    // loop: 8 waves x 16 KB of fragment reads per tile) + 32 VALU adds, and they depend on nothing the LDS reads deliver: the chain
    // starts while the row fragments are still in flight.  Rows >= T get -1e30 / 0 at the load: P = exp2(-huge) = 0, dP' finite
    // (their dO rows were staged as zeros), so the ragged tail needs no mask in the loop.
    auto stat_load = [&](int qt, float& a, float& bq) {
      const int q = qt * 32 + lr;
      const bool v = qt < NT && q < T;
      a = v ? nls_b[q] : -1e30f;
      bq = v ? ndl_b[q] : 0.f;
    };
    auto aug = [&](float x) -> bfx8 {
      const bf16_t h1 = f32_to_bf16(x);
      const float r1 = x - bf16_to_f32(h1);
      const bf16_t h2 = f32_to_bf16(r1);
      const bf16_t h3 = f32_to_bf16(r1 - bf16_to_f32(h2));
      u32x4 w = {lh ? 0u : ((uint32_t)h1 | ((uint32_t)h2 << 16)), lh ? 0u : (uint32_t)h3, 0u, 0u};
      return __builtin_bit_cast(bfx8, w);
    };
    const u32x4 ones_w = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
    const bfx8 ones = __builtin_bit_cast(bfx8, ones_w);
    // stage 1 of tile qt: S' and dP' (row-constant k-step first, then the four k-steps over the head dimension)
    auto stage1 = [&](int qt, float nls, float ndl, f32x16_t& s, f32x16_t& dp) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
      s = MFMA(aug(nls), ones, s);
      dp = MFMA(aug(ndl), ones, dp);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if (ATT_ABLATE == 14) break;
        s = MFMA(frag_rows(Qimg, qt * 32 + lr, 2 * ks + lh), kf[ks], s);     // S'[q][key] = S - lse / c
        dp = MFMA(frag_rows(Gimg, qt * 32 + lr, 2 * ks + lh), vf[ks], dp);   // dP'[q][key] = dP - delta
      }
    };
    // stage 2 of tile qt: P and dS (in place), then dV += dO^T P and dK += Q^T dS
    auto stage2 = [&](int qt, f32x16_t& s, f32x16_t& dp) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (ATT_ABLATE == 12) { dp[r] += s[r]; continue; }
        const float p = __builtin_amdgcn_exp2f(s[r] * c_log2);
        s[r] = p;
        dp[r] = p * dp[r];                                                    // dS, in place (padded KEYS are never stored)
      }
      const bfx8 p0 = pack8(s, 0), p1 = pack8(s, 1), e0 = pack8(dp, 0), e1 = pack8(dp, 1);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        if (ATT_ABLATE == 13) { dv[dt][0] += (float)p0[0] + (float)p1[0]; dk[dt][0] += (float)e0[0] + (float)e1[0]; continue; }
        const bfx8 g0 = frag_tr_perm<true>(Gimg, qt * 32, dt * 32), g1 = frag_tr_perm<true>(Gimg, qt * 32 + 16, dt * 32);
        dv[dt] = MFMA(g0, p0, dv[dt]);
        dv[dt] = MFMA(g1, p1, dv[dt]);
        const bfx8 a0 = frag_tr_perm<true>(Qimg, qt * 32, dt * 32), a1 = frag_tr_perm<true>(Qimg, qt * 32 + 16, dt * 32);
        dk[dt] = MFMA(a0, e0, dk[dt]);
        dk[dt] = MFMA(a1, e1, dk[dt]);
      }
    };
    float nls_c, ndl_c, nls_n, ndl_n;
    stat_load(0, nls_c, ndl_c);
#ifdef ATT_STAMP
    unsigned long long a0, a1, a2, a3, seg0 = 0, seg1 = 0, seg2 = 0, tl0, tl1, iters = 0;
    ATT_T(tl0);
#endif
    for (int qt = 0; qt < NT; ++qt) {
      stat_load(qt + 1, nls_n, ndl_n);
      f32x16_t s, dp;
#ifdef ATT_STAMP
      ATT_T(a0);
      stage1(qt, nls_c, ndl_c, s, dp);
      ATT_TOUCH(s[15]); ATT_TOUCH(dp[15]);
      ATT_T(a1);
      nls_c = nls_n;
      ndl_c = ndl_n;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(s[r] * c_log2);
        s[r] = p;
        dp[r] = p * dp[r];
      }
      const bfx8 p0 = pack8(s, 0), p1 = pack8(s, 1), e0 = pack8(dp, 0), e1 = pack8(dp, 1);
      asm volatile("" :: "v"(p0), "v"(p1), "v"(e0), "v"(e1));
      ATT_T(a2);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const bfx8 g0 = frag_tr_perm<true>(Gimg, qt * 32, dt * 32), g1 = frag_tr_perm<true>(Gimg, qt * 32 + 16, dt * 32);
        dv[dt] = MFMA(g0, p0, dv[dt]);
        dv[dt] = MFMA(g1, p1, dv[dt]);
        const bfx8 q0 = frag_tr_perm<true>(Qimg, qt * 32, dt * 32), q1 = frag_tr_perm<true>(Qimg, qt * 32 + 16, dt * 32);
        dk[dt] = MFMA(q0, e0, dk[dt]);
        dk[dt] = MFMA(q1, e1, dk[dt]);
      }
      ATT_TOUCH(dv[1][15]); ATT_TOUCH(dk[1][15]);
      ATT_T(a3);
      seg0 += a1 - a0; seg1 += a2 - a1; seg2 += a3 - a2; ++iters;
#else
      stage1(qt, nls_c, ndl_c, s, dp);
      nls_c = nls_n;
      ndl_c = ndl_n;
      stage2(qt, s, dp);
#endif
    }
#ifdef ATT_STAMP
    ATT_T(tl1);
    if (threadIdx.x == 0 && kt == 0 && blockIdx.x < 4096) {
      unsigned long long* o = g_att_stamps + (size_t)blockIdx.x * 8;
      o[0] = seg0; o[1] = seg1; o[2] = seg2; o[3] = tl1 - tl0; o[4] = tp1 - tp0; o[5] = iters; o[6] = tp0;
    }
#endif
    if (kvld) {
      store_rowT(dqkv + grow * ld + D + h * ATT_HD, dk, scale, lh);
      store_rowT(dqkv + grow * ld + 2 * D + h * ATT_HD, dv, 1.f, lh);
    }
  }
#ifdef ATT_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  ATT_T(tp1);
  if (threadIdx.x == 0 && blockIdx.x < 4096) g_att_stamps[(size_t)blockIdx.x * 8 + 7] = tp1;
#endif
}

// ================================================================================================ C-ABI
static inline int att_tpad(int T) { return (T + 31) / 32 * 32; }

extern "C" int64_t msc_attention_tpad(int T) { return att_tpad(T); }

extern "C" int msc_attention_fwd(const void* qkv, void* o, float* lse, int B, int T, int H, int hd, float scale, int dtype, void* stream) {
  MSC_CHECK_ARG(qkv && o && lse && B > 0 && T > 0 && H > 0 && scale > 0.f);
  if (dtype != MSC_BF16 || hd != ATT_HD) return MSC_ERR_UNSUPPORTED;
  const int Tpad = att_tpad(T);
  if (Tpad > 512 || (long)B * T * 3 * H * hd * 2 >= 0x7ffffff0L) return MSC_ERR_UNSUPPORTED;
  const int nw = (Tpad / 32 + 1) / 2;
  const size_t lds = (size_t)2 * Tpad * 128;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)attn_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)attn_bwd_dq_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(attn_fwd_kernel, dim3(B * H), dim3(64 * nw), lds, (hipStream_t)stream, (const bf16_t*)qkv, (bf16_t*)o, lse, T, Tpad, H,
                     scale * 1.4426950408889634f);
  return msc_launch_status();
}

// lse: from the forward; delta: fp32 scratch [2][B*H][Tpad] (written by the dQ kernel, read by the dK/dV kernel)
extern "C" int msc_attention_bwd(const void* qkv, const void* o, const void* dout, const float* lse, float* delta, void* dqkv, int B, int T,
                                 int H, int hd, float scale, int dtype, void* stream) {
  MSC_CHECK_ARG(qkv && o && dout && lse && delta && dqkv && B > 0 && T > 0 && H > 0 && scale > 0.f);
  if (dtype != MSC_BF16 || hd != ATT_HD) return MSC_ERR_UNSUPPORTED;
  const int Tpad = att_tpad(T);
  if (Tpad > 512 || (long)B * T * 3 * H * hd * 2 >= 0x7ffffff0L) return MSC_ERR_UNSUPPORTED;
  const int nw = (Tpad / 32 + 1) / 2;
  const size_t lds = (size_t)2 * Tpad * 128;
  hipStream_t st = (hipStream_t)stream;
  const float c = scale * 1.4426950408889634f;
  const int nwb_env = mscplan::g_knobs.att_bwd_waves;      // tuning knobs (msc_set_knob; 0 = default)
  // <= 8 waves: the kernels need ~190-256 VGPRs (two waves per SIMD); 8 measured best inside the step (5: -0.5 %)
  const int nt = Tpad / 32;
  const int nwb = nwb_env > 0 ? (nwb_env > 8 ? 8 : nwb_env) : (nt < 8 ? nt : 8);
  const int nwq_env = mscplan::g_knobs.att_dq_waves;
  const int nwq = nwq_env > 0 ? (nwq_env > 8 ? 8 : nwq_env) : nwb;
  hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3(B * H), dim3(64 * nwq), lds, st, (const bf16_t*)qkv, (const bf16_t*)o, (const bf16_t*)dout, lse,
                     delta, (bf16_t*)dqkv, T, Tpad, H, c, scale);
  hipLaunchKernelGGL(attn_bwd_dkv_kernel, dim3(B * H), dim3(64 * nwb), lds, st, (const bf16_t*)qkv, (const bf16_t*)dout, lse, delta,
                     (bf16_t*)dqkv, T, Tpad, H, c, scale);
  return msc_launch_status();
}
