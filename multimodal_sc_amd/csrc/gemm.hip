// gemm.hip — the one MFMA contraction kernel of the path, specialised by operand access mode:
//   C[z][m][n] = epilogue( sum_k A(m,k) * B(n,k) )
// Operand modes (template parameters, one per operand):
//   OPM_DENSE_K     operand stored [row][k], k contiguous                    (linear fwd/dgrad, QK^T ...)
//   OPM_DENSE_MN    operand stored [k][row], row contiguous                  (weight gradients, P^T dO ...)
//   OPM_CONV_FWD    A = NHWC activations gathered on the fly, rows=(b,ho,wo), k=(kh,kw,cin)
//   OPM_CONV_DGRAD  A = NHWC output-gradients gathered, rows=(b,h,w), k=(kh,kw,cout)
//   OPM_CONV_WGRAD  B = NHWC activations gathered, reduction k=(b,ho,wo), rows n=(kh,kw,cin)
// so convolution is an implicit GEMM: there is no im2col buffer in HBM.
//
// Tiles BM x BN in {64,128}^2 (4 waves, 2x2), 256x128 (8 waves, 4x2) and 256x256 (8 waves, 2x4), 128 BYTES of K per
// operand row per step (64 bf16 / 32 fp32 elements): every LDS-DMA instruction moves 8 rows x one full 128-byte cache line
// (half-line pieces cost the same address/tag work for half the bytes).  Waves own WM x WN as 32x32 MFMA tiles
// (v_mfma_f32_32x32x16_bf16; exact-fp32 v_mfma_f32_32x32x2_f32 in parity mode).
//
// Staging (gfx950): operands go HBM -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`, 1 KiB per wave-instruction, no
// VGPR round trip).  The per-lane SOURCE offset does the gather (implicit-GEMM taps, transposition-free weight
// gradients) and the XOR bank swizzle (the LDS image itself is lane-linear, as the hardware requires); lanes whose
// element is padding / out of range use an out-of-range buffer offset, which the hardware turns into zeros in LDS.
//   k-major operands : LDS rows of 128 B, chunk c of row r at c ^ ((r>>1)&7)     -> conflict-free ds_read_b128
//   mn-major operands: LDS image [k][row] as it lies in HBM, chunk ^ swizzle(k)  -> ds_read_b64_tr_b16 (hardware
//                      transpose) builds the MFMA fragment; no transposing pass anywhere (bf16).
// 2-3 LDS stages (by tile, see STAGES): the DMA of K-steps t+1.. is in flight while the MFMAs of step t run; one barrier
// per K step (16-32 MFMAs per wave).
// The exact-fp32 parity mode has no transposed LDS read, so its mn-major operands are staged through registers and
// transposed on the LDS write (correctness path, not tuned).
// Workgroup ids are remapped so that the tiles sharing an activation row-block run on one XCD (shared L2).
//
// Build-authored: the upstream reference has no code (SURVEY.md finding zero); roles per BASELINE.json:
// ImageEncoder/BEV conv stack, FusionTransformer, SemanticEncoder/Decoder, heads.
#include "msc_common.h"
#include <stdlib.h>
#include <type_traits>

#include "gemm_params.h"

typedef __attribute__((ext_vector_type(8))) __bf16 mfma_bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
// Transposed LDS read issued as inline assembly.  Through the builtin the compiler sees an LDS load it cannot prove independent of the
// LDS-DMA writes in flight and places s_waitcnt vmcnt(0) in front of the first one of a K step — i.e. behind the prefetch of the NEXT
// stage issued a few instructions earlier: DMA and MFMAs of the weight-gradient kernels ran one after the other (ISA of
// gemm_kernel<bf16,1,1,128,128,...>, end of round 4).  The staged buffer being read was waited for explicitly (counted vmcnt + barrier)
// at the top of the step, so no wait is needed here; the read's own completion is waited for by tr_wait().
__device__ __forceinline__ s16x4_t ds_read_tr_asm(const char* lds_generic_ptr) {
  const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)lds_generic_ptr;
  s16x4_t v;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define DMA_OOB 0x80000000u   // >= num_records of every descriptor built here: the DMA writes zeros
// K bytes per operand row per step (KB) and LDS ring depth are template parameters of the kernel; defaults per tile class
// (the -D overrides exist for A/B builds only):
//   4-wave tiles (<=128x128, 2-3 workgroups per CU): KB 128 (one full cache line per row and DMA piece), 2 stages
//   256x128 persistent (8 waves, one workgroup per CU): KB 128, 3 stages (96 KB in flight) + 16 KB epilogue slab = 160 KB
//   256x256 persistent (8 waves, one workgroup per CU): KB 64,  4 stages (96 KB in flight) + 32 KB epilogue slab = 160 KB
#ifndef GEMM_NP_KB
#define GEMM_NP_KB 128
#endif
#ifndef GEMM_NP_STAGES
#define GEMM_NP_STAGES 2
#endif
#ifndef GEMM_NP_W128
#define GEMM_NP_W128 8      // waves of the multi-workgroup 128x128 tile
#endif
#ifndef GEMM_WG_STAGES
#define GEMM_WG_STAGES 3    // ring depth of the gathered conv weight-gradient kernels (64-byte K steps)
#endif
#ifndef GEMM_FRAG_AHEAD
#define GEMM_FRAG_AHEAD 0   // 1 (A/B builds): all fragment reads of a K step issued before its first MFMA.  Measured (profiles/r3_gemm_frag_ahead.txt): the
                            // same time on every shape and in the step — the latency in front of each MFMA was already covered by the other waves of
                            // the SIMD; the K loop is bound by LDS THROUGHPUT.  Off: it costs 18 registers for nothing.
#endif
#ifndef GEMM_NT_STORE
#define GEMM_NT_STORE 1     // output tiles leave with non-temporal stores: written once, never re-read by this launch, so they
#endif                      // should not evict the operand panels other tiles of the launch still re-read from L2 (-3 % on the linears)
#ifndef GEMM_P128_KB
#define GEMM_P128_KB 128
#endif
#ifndef GEMM_P128_STAGES
#define GEMM_P128_STAGES 3
#endif
#ifndef GEMM_P256_KB
#define GEMM_P256_KB 64
#endif
#ifndef GEMM_P256_STAGES
#define GEMM_P256_STAGES 4
#endif
#define MSCPLAN_NP_KB GEMM_NP_KB
#include "gemm_plan.h"

// GEMM_STAMP: diagnostic build only (make stamp -> libmsc_hip_stamp.so, tools/gemm_stamps.py); never defined in the product
// build.  Wave 0 of every workgroup records shader-clock stamps (s_memtime) into a debug buffer of its own that no other
// code reads: [0] entry, [1] first data landed, [2] cycles spent in the per-step DMA wait + barrier (summed), [3] K-loop
// cycles (summed over tiles), [4] epilogue cycles (summed), [5] exit, [6] tiles, [7] K steps.
#ifdef GEMM_STAMP
static unsigned long long* g_stamp_buf = nullptr;
extern "C" int msc_debug_gemm_stamps(unsigned long long* host_out, int nwg) {
  if (!g_stamp_buf) return MSC_ERR_BAD_ARG;
  return hipMemcpy(host_out, g_stamp_buf, (size_t)nwg * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess ? MSC_OK : MSC_ERR_LAUNCH;
}
// epilogue detail (second record per workgroup, GEMM_STAMP_EP_OFF records further on): [0] wait at the barrier that ends the K loop,
// [1] BatchNorm-statistics part, [2] accumulators -> slab incl. its barrier (summed over passes), [3] slab -> registers -> global stores
// issued (summed), [4] barrier between passes (summed; __syncthreads waits for the pass's stores), [5] column-sum tail
#define GEMM_STAMP_EP_OFF (1 << 19)
extern "C" int msc_debug_gemm_stamps_ep(unsigned long long* host_out, int nwg) {
  if (!g_stamp_buf) return MSC_ERR_BAD_ARG;
  return hipMemcpy(host_out, g_stamp_buf + (size_t)GEMM_STAMP_EP_OFF * 8, (size_t)nwg * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess ? MSC_OK : MSC_ERR_LAUNCH;
}
#define STAMP_NOW() __builtin_amdgcn_s_memtime()
#define EP_STAMP(acc) do { const unsigned long long n_ = STAMP_NOW(); acc += n_ - st_e; st_e = n_; } while (0)
#else
#define EP_STAMP(acc) ((void)0)
#endif

template <int MODE> struct mode_is_kmajor { static constexpr bool value = (MODE == OPM_DENSE_K || MODE == OPM_CONV_FWD || MODE == OPM_CONV_DGRAD); };

// byte offset of 16-byte chunk c of row `row` in a k-major LDS image with KB-byte rows.  ds_read_b128 serves 16 lanes
// per LDS cycle from one 256-byte bank row = two image rows: with the XOR below the 16 rows of every lane group land on 16
// different 16-byte slots (rows r, r+1 share (r>>1) but sit in different halves of the bank row).
template <int KB> __device__ __forceinline__ int lds_chunk_off(int row, int c) {
  if constexpr (KB == 128) return row * 128 + ((c ^ ((row >> 1) & 7)) << 4);
  else return row * 64 + ((c ^ ((row >> 2) & 3)) << 4);     // 64-byte rows: four rows per bank row
}
// swizzle of the 16-byte chunk index inside an mn-major LDS row of ROWB bytes, keyed by the k row
template <int ROWB> __device__ __forceinline__ int mn_swz(int krow) {
  return ROWB >= 256 ? ((krow & 3) << 2) : (((krow >> 1) & 1) << 2);
}

// One K step of the schedule: kbase = first reduction index of the step; (kh, kw, co0) = its filter tap and first
// channel for the dgrad gather.  Wave-uniform.
struct KStep { int kbase, kh, kw, co0; };

// Strided dgrad orders its rows (input pixels) parity-class-major: class (h%S, w%S) first, then (b, h/S, w/S).  All
// rows of a class see the same subset of filter taps ((h+P-kh) % S == 0), so a tile inside one class iterates only
// those taps instead of multiplying 3/4 zeros (stride 2).  p.cls == 0: plain (b,h,w) order.
// Row / pixel indices are decoded with runtime divisors (map width 56, 28, 14, 7 ...): a 32-bit integer division is a ~40-instruction
// sequence, a row decode three of them, and every thread decodes several rows in the prologue and again in the epilogue of a tile — the
// strided dgrad's prologue was 9 k cycles against 3.7 k for the forward (clock stamps, round 4).  fdiv: floor(x / d) through the float
// reciprocal the host put next to d, exact for 0 <= x < 2^24 (x is exact in fp32, the estimate is off by at most one, one correction step
// each way); p.fastdiv = 0 (some row count >= 2^24): the integer division.
__device__ __forceinline__ int fdiv(const GemmParams& p, int x, int d, float rd) {
  if (!p.fastdiv) return x / d;
  int q = (int)((float)x * rd);
  const int r = x - q * d;
  q += (r >= d) - (r < 0);
  return q;
}
__device__ __forceinline__ void dgrad_row_to_pixel(const GemmParams& p, int row, int& b, int& h, int& w) {
  if (p.cls) {
    const int cid = fdiv(p, row, p.cMc, p.rMc), idx = row - cid * p.cMc;
    const int t = fdiv(p, idx, p.cW2, p.rW2), w2 = idx - t * p.cW2;
    b = fdiv(p, t, p.cH2, p.rH2);
    const int h2 = t - b * p.cH2;
    const int ch = cid >> p.cLogS, cw = cid - (ch << p.cLogS);
    h = h2 * p.cS + ch;
    w = w2 * p.cS + cw;
  } else {
    const int t = fdiv(p, row, p.cW, p.rW);
    w = row - t * p.cW;
    b = fdiv(p, t, p.cH, p.rH);
    h = t - b * p.cH;
  }
}

// ------------------------------------------------------------------------------------------------
// Operand loader.  DMA = true: LDS-DMA path (k-major any dtype; mn-major bf16).  DMA = false: registers + transposing
// LDS write (mn-major fp32 only).
template <typename T, int MODE, int ROWS, int NTHR, int KB> struct OperandLoader {
  static constexpr bool KM = mode_is_kmajor<MODE>::value;
  static constexpr bool DMA = KM || sizeof(T) == 2;
  static constexpr int VEC = 16 / sizeof(T);
  static constexpr int KT = KB / sizeof(T);
  static constexpr int NCH = ROWS * (KB / 16) / NTHR;          // 16-byte chunks per thread per K step
  static constexpr int ROWB = KM ? KB : ROWS * (int)sizeof(T); // LDS row bytes
  static_assert(NCH >= 1, "tile too small for the thread count");
  static constexpr int CPR = ROWB / 16;                        // chunks per LDS row (DMA) / per reduction row (REG)
  static constexpr int RPI = 1024 / ROWB;                      // LDS rows written by one DMA wave-instruction

  // k-major operands: the logical chunk c a lane moves depends on the lane and (KB = 128) on the PARITY of the DMA
  // instruction index only, so everything derived from c — the k offset, the filter tap of a conv gather — is kept once
  // per parity (NPAR values) instead of once per chunk; rows outside the matrix are marked in off / boff (DMA_OOB).
  static_assert(!KM || NCH == 1 || NCH % 2 == 0, "chunk parity = instruction parity needs an even chunk count");
  static constexpr int NPAR = KM ? (NCH > 1 ? 2 : 1) : NCH;
  __amdgpu_buffer_rsrc_t rs;
  const T* base;
  long ld;
  uint32_t off[NCH];   // dense modes: running byte offset (DENSE_K: DMA_OOB when the row is outside the matrix)
  int kk[NPAR];        // DENSE_K: k offset of the chunk inside a K step (per parity); mn-major: current k of the chunk
  bool ok[KM ? 1 : NCH];          // mn-major: column inside the matrix
  int a0[NCH], a1[NCH];           // fwd/dgrad: hi0, wi0 ; wgrad: ho, wo
  int a2[NPAR], a3[NPAR], a4[NPAR];   // fwd: kh, kw, cin (per parity) ; dgrad: a4 = channel offset ; wgrad: b / kh / kw per chunk
  uint32_t boff[NCH];             // batch base (elements, fwd/dgrad; DMA_OOB: row outside) or cin (wgrad)

  // (row-in-tile, logical chunk) this thread moves for slot i
  static __device__ __forceinline__ void slot(int i, int& r, int& c) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if constexpr (DMA) {
      const int j = wave * NCH + i;                 // DMA instruction index inside the tile
      if constexpr (KM && KB == 128) {
        r = 8 * j + (lane >> 3);                    // 8 rows x one 128-byte line per DMA instruction
        c = (lane & 7) ^ ((r >> 1) & 7);
      } else if constexpr (KM) {
        r = 16 * j + (lane >> 2);                   // 16 rows x half a line
        c = (lane & 3) ^ ((r >> 2) & 3);
      } else {
        r = RPI * j + lane / CPR;                   // k row inside the K step
        c = (lane % CPR) ^ mn_swz<ROWB>(r);        // logical chunk along the mn axis
      }
    } else {
      const int id = tid + i * NTHR;
      r = id / CPR;
      c = id % CPR;
    }
  }

  __device__ __forceinline__ void init(const GemmParams& p, const T* b, long ldm, int rows_total, int row0, int k0) {
    base = b;
    ld = ldm;
    rs = __builtin_amdgcn_make_buffer_rsrc((void*)b, 0, (int)0x7ffffff0, 0x00020000);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      int r, c;
      slot(i, r, c);
      if constexpr (MODE == OPM_DENSE_K) {
        // seek-based: the K step's base index comes from the schedule (KStep), so a strided dgrad can skip taps
        const int row = row0 + r;
        if (i < NPAR) kk[i] = c * VEC;
        off[i] = row < rows_total ? (uint32_t)(((long)row * ld + c * VEC) * (long)sizeof(T)) : DMA_OOB;
      } else if constexpr (MODE == OPM_DENSE_MN) {
        const int col = row0 + c * VEC;
        ok[i] = col < rows_total;
        kk[i] = k0 + r;
        off[i] = (uint32_t)(((long)kk[i] * ld + (ok[i] ? col : 0)) * (long)sizeof(T));
      } else if constexpr (MODE == OPM_CONV_FWD) {
        const int row = row0 + r;
        const bool inside = row < rows_total;
        const int rr = inside ? row : 0;
        const int t = fdiv(p, rr, p.cWo, p.rWo), wo = rr - t * p.cWo;
        const int bb = fdiv(p, t, p.cHo, p.rHo), ho = t - bb * p.cHo;
        a0[i] = ho * p.cS - p.cP;
        a1[i] = wo * p.cS - p.cP;
        boff[i] = inside ? (uint32_t)bb * (uint32_t)(p.cH * p.cW * p.cC) : DMA_OOB;
        if (i < NPAR) {
          const int k = k0 + c * VEC;
          const int tap = k >> p.cLogC;
          a4[i] = k & (p.cC - 1);
          a2[i] = tap / p.cKW;
          a3[i] = tap - a2[i] * p.cKW;
        }
      } else if constexpr (MODE == OPM_CONV_DGRAD) {
        // rows are INPUT pixels (b,h,w) of size cH x cW; gathered tensor is dy [cB,cHo,cWo,cC]
        const int row = row0 + r;
        const bool inside = row < rows_total;
        int h, w, bb;
        dgrad_row_to_pixel(p, inside ? row : 0, bb, h, w);
        a0[i] = h + p.cP;
        a1[i] = w + p.cP;
        boff[i] = inside ? (uint32_t)bb * (uint32_t)(p.cHo * p.cWo * p.cC) : DMA_OOB;
        if (i < NPAR) a4[i] = c * VEC;      // channel offset inside the K step; (kh, kw, co0) come from the schedule
      } else {  // OPM_CONV_WGRAD: columns n=(kh,kw,cin) fixed per thread, reduction over (b,ho,wo)
        const int col = row0 + c * VEC;
        ok[i] = col < rows_total;
        const int cl = ok[i] ? col : 0;
        const int tap = cl >> p.cLogC;
        boff[i] = (uint32_t)(cl & (p.cC - 1));
        a3[i] = tap / p.cKW;
        a4[i] = tap - a3[i] * p.cKW;
        const int pix = k0 + r;
        kk[i] = pix;
        const int t = fdiv(p, pix, p.cWo, p.rWo);
        a1[i] = pix - t * p.cWo;
        a2[i] = fdiv(p, t, p.cHo, p.rHo);
        a0[i] = t - a2[i] * p.cHo;
      }
    }
  }

  // byte offset of slot i's chunk for the current K step (DMA_OOB where the element is padding).  mn-major operands also
  // advance their per-chunk state here; k-major ones advance once per step in end_step().
  __device__ __forceinline__ uint32_t next_offset(const GemmParams& p, int klimit, int i, const KStep& ks) {
    uint32_t v = DMA_OOB;
    constexpr int PM = NPAR - 1;      // parity mask of the chunk index (k-major)
    if constexpr (MODE == OPM_DENSE_K) {
      if (off[i] != DMA_OOB && ks.kbase + kk[i & PM] < klimit) v = off[i] + (uint32_t)ks.kbase * (uint32_t)sizeof(T);
    } else if constexpr (MODE == OPM_DENSE_MN) {
      if (ok[i] && kk[i] < klimit) v = off[i];
      off[i] += (uint32_t)(KT * ld * (long)sizeof(T));
      kk[i] += KT;
    } else if constexpr (MODE == OPM_CONV_FWD) {
      const int q = i & PM;
      const int hi = a0[i] + a2[q], wi = a1[i] + a3[q];
      if (boff[i] != DMA_OOB && a2[q] < p.cKH && (unsigned)hi < (unsigned)p.cH && (unsigned)wi < (unsigned)p.cW)
        v = (boff[i] + ((uint32_t)(hi * p.cW + wi) << p.cLogC) + (uint32_t)a4[q]) * (uint32_t)sizeof(T);
    } else if constexpr (MODE == OPM_CONV_DGRAD) {
      const int th = a0[i] - ks.kh, tw = a1[i] - ks.kw;
      const int sm = p.cS - 1;
      const int ho = th >> p.cLogS, wo = tw >> p.cLogS;
      if (boff[i] != DMA_OOB && ks.kh < p.cKH && th >= 0 && tw >= 0 && !(th & sm) && !(tw & sm) && ho < p.cHo && wo < p.cWo)
        v = (boff[i] + ((uint32_t)(ho * p.cWo + wo) << p.cLogC) + (uint32_t)(ks.co0 + a4[i & PM])) * (uint32_t)sizeof(T);
    } else {
      const int hi = a0[i] * p.cS - p.cP + a3[i], wi = a1[i] * p.cS - p.cP + a4[i];
      if (ok[i] && kk[i] < klimit && (unsigned)hi < (unsigned)p.cH && (unsigned)wi < (unsigned)p.cW)
        v = ((((uint32_t)a2[i] * (uint32_t)p.cH + (uint32_t)hi) * (uint32_t)p.cW + (uint32_t)wi) << p.cLogC) * (uint32_t)sizeof(T) +
            boff[i] * (uint32_t)sizeof(T);
      kk[i] += KT;
      a1[i] += KT;
      while (a1[i] >= p.cWo) {
        a1[i] -= p.cWo;
        if (++a0[i] == p.cHo) { a0[i] = 0; ++a2[i]; }
      }
    }
    return v;
  }

  // once per K step after every chunk's offset was taken: advance the per-parity filter-tap state (conv forward gather)
  __device__ __forceinline__ void end_step(const GemmParams& p) {
    if constexpr (MODE == OPM_CONV_FWD) {
#pragma unroll
      for (int q = 0; q < NPAR; ++q) {
        const int cin = a4[q] + KT;
        a3[q] += cin >> p.cLogC;
        a4[q] = cin & (p.cC - 1);
        while (a3[q] >= p.cKW) { a3[q] -= p.cKW; ++a2[q]; }
      }
    }
  }

  // DMA path: enqueue chunk i of the current K step into `tile` (LDS)
  __device__ __forceinline__ void issue_one(const GemmParams& p, int klimit, char* tile, const KStep& ks, int i) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t v = next_offset(p, klimit, i, ks);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(tile + (wave * NCH + i) * 1024), 16, v, 0, 0, 0);
  }
  // ... all chunks of the step, then advance
  __device__ __forceinline__ void issue(const GemmParams& p, int klimit, char* tile, const KStep& ks) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) issue_one(p, klimit, tile, ks, i);
    end_step(p);
  }

  // register path (mn-major fp32): load now, transpose into LDS later
  __device__ __forceinline__ void load(const GemmParams& p, int klimit, uint4 (&regs)[NCH], const KStep& ks) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const uint32_t v = next_offset(p, klimit, i, ks);
      regs[i] = (v == DMA_OOB) ? make_uint4(0, 0, 0, 0)
                               : *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(base) + v);
    }
    end_step(p);
  }
  __device__ __forceinline__ void store(char* tile, const uint4 (&regs)[NCH]) const {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      int r, cc;
      slot(i, r, cc);
      const int kb = r * (int)sizeof(T);
      const uint32_t w[4] = {regs[i].x, regs[i].y, regs[i].z, regs[i].w};
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const int row = cc * VEC + j;
        *reinterpret_cast<uint32_t*>(tile + lds_chunk_off<KB>(row, kb >> 4) + (kb & 15)) = w[j];
      }
    }
  }
};

// ------------------------------------------------------------------------------------------------
// The kernel.  EPX: extended epilogue (fused activation backward + column sums) — compiled only into the kernels that use
// it so the common epilogue keeps its register budget.
// PERSIST (256-row tiles, bf16, both operands k-major, one 8-wave workgroup per CU): gridDim.x workgroups walk the tiles
// w = blockIdx.x, += gridDim.x.  The DMA issue side runs NSTAGE-1 K steps AHEAD of the MFMA side and simply rolls over
// into the next tile, so the first steps of tile t+1 are already landing in the ring while tile t's epilogue stores its
// output (the epilogue has its own LDS slab behind the ring and uses raw s_barrier + lgkmcnt only, so nothing drains the
// DMAs).  A lone workgroup per CU otherwise pays load latency + epilogue un-overlapped on every tile, which at this path's
// short K (8-72 steps) costs more than the larger tile saves.
// Wave roles in the persistent kernels: vmcnt counts a wave's loads AND stores in order, so a wave that both stores an
// output tile and waits for the next tile's DMA would have to see its stores complete first (HBM latency) — the K loop of
// tile t+1 would stall behind the store burst of tile t.  Therefore waves 0-3 are the LOADERS (they issue every LDS-DMA
// piece, interleaved between their MFMA sub-steps, and are the only ones that wait on vmcnt) and waves 4-7 are the STORERS
// (they run the slab -> global half of the epilogue; their stores drain in the background under the next K loop and are
// never waited for).  All eight waves run the MFMAs.
struct Sched { KStep ks; int nk, s_kw0, s_step; };

// NTHR: 512 (8 waves) for the 256-row tiles and for the 128x128 tile of the k-major / dense kernels (wave tile 32x64: four
// waves per SIMD at two workgroups per CU — measured 3 % faster over the path's shapes, 15-20 % on the weight gradients),
// 256 (4 waves, 2x2) otherwise.
// EPX: 0 plain epilogue, 1 + activation backward (dref) / column sums, 2 + BatchNorm-backward statistics (bnpart) — an instantiation of its
// own: its per-channel constants and second sum set cost ~45 registers the other extended epilogues (128-register budget) do not have
template <typename T, int AMODE, int BMODE, int BM, int BN, int EPX, int KB, int NSTAGE, bool PERSIST, int NTHR>
__global__ __launch_bounds__(NTHR, (NTHR == 512 && (BM == 128 || !PERSIST) ? 4 : 2)) void gemm_kernel(const GemmParams p) {
  constexpr int KT = KB / sizeof(T);
  constexpr int NWN = BN == 256 ? 4 : 2;        // waves along N: 256x256 -> 2x4 (wave tile 128x64), 256x128 -> 4x2, else 2x2
  constexpr int NWM = (NTHR / 64) / NWN;        // waves along M
  constexpr int WM = BM / NWM, WN = BN / NWN, MI = WM / 32, NI = WN / 32;
  constexpr int A_BYTES = BM * KB, B_BYTES = BN * KB;
  constexpr int NLOAD = PERSIST ? 256 : NTHR;   // threads that issue the staging loads (persistent: waves 0-3)
  constexpr int NSTORE = PERSIST ? 256 : NTHR;  // threads that move the slab to global memory (persistent: waves 4-7)
  using LA = OperandLoader<T, AMODE, BM, NLOAD, KB>;
  using LB = OperandLoader<T, BMODE, BN, NLOAD, KB>;
  static_assert(!PERSIST || (LA::DMA && LB::DMA), "the persistent pipeline is DMA-only");
  constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
  constexpr int STAGES = (LA::DMA && LB::DMA) ? NSTAGE : 2;
  constexpr int SLAB_ROWS = PERSIST ? 32 : 64;          // output rows per epilogue pass through the fp32 slab
  constexpr int EP_BYTES = SLAB_ROWS * BN * 4;
  constexpr int RING_BYTES = STAGES * STAGE_BYTES;
  // non-persistent: the slab aliases the ring (the K loop is over); persistent: it sits behind the ring (DMAs in flight)
  constexpr int SMEM_BYTES = PERSIST ? RING_BYTES + EP_BYTES : (RING_BYTES > EP_BYTES ? RING_BYTES : EP_BYTES);
  static_assert(SMEM_BYTES <= 163840, "LDS budget of a CU");
  __shared__ __attribute__((aligned(1024))) char smem[SMEM_BYTES];
  char* const ep_mem = PERSIST ? smem + RING_BYTES : smem;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / NWN, wn = wave % NWN;
  const bool is_loader = !PERSIST || __builtin_amdgcn_readfirstlane(wave) < 4;
  const bool is_storer = !PERSIST || !is_loader;
  const int stid = PERSIST ? tid - 256 : tid;     // index among the storing threads
  const int ntn = (p.N + BN - 1) / BN;
  const int ntiles = ((p.M + BM - 1) / BM) * ntn;       // persistent mode: work items (host guarantees z == 1, no K split)
  const int nk_total = (p.K + KT - 1) / KT;

  // ---- work item -> tile.  Workgroups are dealt to the 8 XCDs round-robin in launch order (x fastest, then z).  Plain
  // launches: tiles sharing an activation row block on one XCD (xcd_remap).  Split-K launches with nsplit % 8 == 0 (weight
  // gradients): ALL tiles of one K slice on one XCD — they stream the same rows of both operands at the same time, so the
  // slice is fetched into that L2 once instead of once per tile (PMC: weight-gradient traffic was ~4x the algorithmic bytes).
  // Persistent: item w runs on workgroup w % gridDim.x, and gridDim.x % 8 == 0 keeps w % 8 = the XCD group, so the same
  // remap applies to the item index.
  struct Geo { int tile_m, tile_n, m0, n0, split, zo, zi; };
  auto decode = [&](int w, Geo& g) {
    int bid, zz;
    if constexpr (PERSIST) {
      bid = mscplan::xcd_remap(w, ntiles);
      zz = 0;
      g.split = 0;
    } else {
      if (p.xcd_split) {
        mscplan::xcd_split_map(blockIdx.z * gridDim.x + blockIdx.x, (int)gridDim.x, g.split, bid);
        zz = 0;
      } else {
        bid = mscplan::xcd_remap(blockIdx.x, gridDim.x);
        zz = blockIdx.z / p.nsplit;
        g.split = blockIdx.z % p.nsplit;
      }
    }
    g.tile_m = bid / ntn; g.tile_n = bid % ntn;
    if constexpr (AMODE == OPM_CONV_DGRAD) {
      // strided dgrad, rows in parity-class-major order: the S*S classes of one pixel region gather the SAME rows of dy, so
      // their row tiles are made neighbours in launch order (same XCD, same time: dy comes from HBM once instead of S*S times)
      if (p.cls_interleave) {
        const int ncls = p.cS * p.cS, tpc = p.cMc / BM;
        const int region = bid / (ncls * ntn), rem = bid - region * (ncls * ntn);
        g.tile_m = (rem / ntn) * tpc + region;
        g.tile_n = rem % ntn;
      }
    }
    g.m0 = g.tile_m * BM; g.n0 = g.tile_n * BN;
    g.zo = zz / p.batch_inner; g.zi = zz % p.batch_inner;
  };

  // ---- K schedule of one tile (wave-uniform).  Default: steps kt_begin..kt_end-1, kbase = kt*KT.  Strided dgrad whose tile
  // lies inside one parity class: only the taps that class can see, (kh0, kh0+S, ..) x (kw0, kw0+S, ..).  Returns kt_begin.
  auto sched_setup = [&](int tm0, int tsplit, Sched& sc) -> int {
    const int kt_begin = tsplit * p.tiles_per_split;
    const int kt_end = min(nk_total, kt_begin + p.tiles_per_split);
    sc.ks = {kt_begin * KT, 0, 0, 0};
    sc.nk = kt_end - kt_begin;
    sc.s_kw0 = 0;
    sc.s_step = 1;
    if constexpr (AMODE == OPM_CONV_DGRAD) {
      int kh0 = 0;
      if (p.cls) {
        const int last = min(tm0 + BM, p.M) - 1;
        const int c0 = tm0 / p.cMc;
        if (c0 == last / p.cMc) {
          kh0 = (c0 / p.cS + p.cP) % p.cS;
          sc.s_kw0 = (c0 % p.cS + p.cP) % p.cS;
          sc.s_step = p.cS;
        }
      }
      const int cnt_h = kh0 < p.cKH ? (p.cKH - kh0 + sc.s_step - 1) / sc.s_step : 0;
      const int cnt_w = sc.s_kw0 < p.cKW ? (p.cKW - sc.s_kw0 + sc.s_step - 1) / sc.s_step : 0;
      sc.nk = cnt_h * cnt_w * (p.cC / KT);
      sc.ks.kh = kh0;
      sc.ks.kw = sc.s_kw0;
      sc.ks.kbase = (kh0 * p.cKW + sc.s_kw0) << p.cLogC;
      if (p.nsplit > 1) {   // split-K dgrad (the host only splits the plain (b,h,w) order): this slice starts mid-schedule
        const int e = kt_begin * KT, tap = e >> p.cLogC;
        sc.ks.co0 = e & (p.cC - 1);
        sc.ks.kh = tap / p.cKW;
        sc.ks.kw = tap - sc.ks.kh * p.cKW;
        sc.ks.kbase = e;
        sc.nk = kt_end - kt_begin;
      }
    }
    return kt_begin;
  };
  auto advance = [&](Sched& sc) {
    if constexpr (AMODE == OPM_CONV_DGRAD) {
      sc.ks.co0 += KT;
      if (sc.ks.co0 >= p.cC) {
        sc.ks.co0 = 0;
        sc.ks.kw += sc.s_step;
        if (sc.ks.kw >= p.cKW) { sc.ks.kw = sc.s_kw0; sc.ks.kh += sc.s_step; }
      }
      sc.ks.kbase = ((sc.ks.kh * p.cKW + sc.ks.kw) << p.cLogC) + sc.ks.co0;
    } else {
      sc.ks.kbase += KT;
    }
  };

  // ---- strided dgrad, tiles whose parity class sees NO filter tap (1x1 / stride-2 downsample convolutions: three of the four classes): their
  // output is the residual or zero.  As workgroups of their own they walked the whole prologue and slab epilogue (~15 k cycles by the clock
  // stamps) to store 16 KB; leaving early instead (first version of this path) did not help either: the launch is then bound by the
  // DISPATCH of 1568 workgroups (entry stamps spread over 18 us), with the 392 real tiles dealt in among them.  Now (p.cls_fill, set by the
  // launcher for 1x1 kernels) only the tap-carrying class is launched and each of its workgroups also copies / clears the rows of the
  // sibling classes of its pixel region after its own epilogue.  fill_rows: rows [fm0, fm0 + BM) x columns [fn0, fn0 + BN) <- residual or 0.
  auto fill_rows = [&](int fm0, int fn0) {
    constexpr int FV = 16 / sizeof(T), FCP = BN / FV;
    T* Cb = reinterpret_cast<T*>(p.C);
    const T* Rf = reinterpret_cast<const T*>(p.res);
    const bool vec_ok = (p.ldc % FV) == 0;
    for (int id = tid; id < BM * FCP; id += NTHR) {
      const int row = id / FCP, cc = id - row * FCP;
      const int m = fm0 + row, n = fn0 + cc * FV;
      if (m >= p.M || n >= p.N) continue;
      int bb, hh, ww;
      dgrad_row_to_pixel(p, m, bb, hh, ww);
      const long o = (((long)bb * p.cH + hh) * p.cW + ww) * p.ldc + n;
      if (vec_ok && n + FV <= p.N) {
        vecf<T> v;
        if (Rf) v.load(Rf + o);
        else
#pragma unroll
          for (int e = 0; e < FV; ++e) v.v[e] = 0.f;
        v.store(Cb + o);
      } else {
        for (int e = 0; e < FV && n + e < p.N; ++e) msc_traits<T>::st(Cb + o + e, Rf ? msc_traits<T>::ld(Rf + o + e) : 0.f);
      }
    }
  };
  if constexpr (AMODE == OPM_CONV_DGRAD && !PERSIST && EPX == 0) {
    // (launches that cannot take cls_fill — a tile height that does not divide the class — still skip the staging for a no-tap tile)
    if (p.cls && !p.cls_fill && p.nsplit == 1 && !p.out_f32 && !p.bias && !p.stats && p.act == MSC_ACT_NONE && !p.C2) {
      Geo g0;
      decode(0, g0);
      Sched s0;
      sched_setup(g0.m0, 0, s0);
      if (s0.nk == 0) {
        fill_rows(g0.m0, g0.n0);
        return;
      }
    }
  }

  // ---- issue side: (tile iw, step ik of isc.nk) and the loaders positioned there
  LA la;
  LB lb;
  Sched isc;
  int iw = PERSIST ? (int)blockIdx.x : 0, ik = 0;
  auto issue_begin_tile = [&](int w) {
    Geo g;
    decode(w, g);
    const int kt_begin = sched_setup(g.m0, g.split, isc);
    if (is_loader) {
      const T* Ab = reinterpret_cast<const T*>(p.A) + g.zo * p.sa_o + g.zi * p.sa_i;
      const T* Bb = reinterpret_cast<const T*>(p.B) + g.zo * p.sb_o + g.zi * p.sb_i;
      la.init(p, Ab, p.lda, p.M, g.m0, kt_begin * KT);
      lb.init(p, Bb, p.ldb, p.N, g.n0, kt_begin * KT);
    }
    ik = 0;
  };
  int islot = 0, cslot = 0, inflight = 0;   // ring slot of the next issue / next compute; K steps issued and not yet consumed
  // A K step goes into the ring in three parts so the loaders can spread the DMA pieces between their MFMA sub-steps:
  // issue_open (pick the step, rolling over into this workgroup's next tile in persistent mode; false: nothing left),
  // issue_pieces<lo,hi> (pieces lo..hi-1 of the NPS = A chunks then B chunks), issue_close (advance the schedule).
  constexpr int NPS = LA::NCH + LB::NCH;              // staging instructions per K step and loading wave
  auto issue_open = [&]() -> bool {
    while (ik == isc.nk) {
      if constexpr (!PERSIST) return false;
      iw += (int)gridDim.x;
      if (iw >= ntiles) { ik = isc.nk = 0; iw = ntiles; return false; }
      issue_begin_tile(iw);
    }
    return true;
  };
  auto issue_pieces = [&](auto lo_c, auto hi_c) {
    constexpr int LO = decltype(lo_c)::value, HI = decltype(hi_c)::value;
    if (!is_loader) return;
    char* buf = smem + islot * STAGE_BYTES;
    if constexpr (LA::DMA && LB::DMA) {
#pragma unroll
      for (int i = LO; i < HI; ++i) {
        if (i < LA::NCH) la.issue_one(p, p.K, buf, isc.ks, i);
        else lb.issue_one(p, p.K, buf + A_BYTES, isc.ks, i - LA::NCH);
      }
    } else if constexpr (LO == 0) {   // register path (fp32 mn-major operands, parity mode): everything in the first part
      if constexpr (LA::DMA) la.issue(p, p.K, buf, isc.ks);
      else { uint4 ra[LA::NCH]; la.load(p, p.K, ra, isc.ks); la.store(buf, ra); }
      if constexpr (LB::DMA) lb.issue(p, p.K, buf + A_BYTES, isc.ks);
      else { uint4 rb[LB::NCH]; lb.load(p, p.K, rb, isc.ks); lb.store(buf + A_BYTES, rb); }
    }
  };
  auto issue_close = [&]() {
    if constexpr (LA::DMA && LB::DMA) {
      if (is_loader) { la.end_step(p); lb.end_step(p); }
    }
    advance(isc);
    ++ik;
    ++inflight;
    islot = islot + 1 == STAGES ? 0 : islot + 1;
  };
  auto issue_step = [&]() {     // whole step at once (prologue)
    if (!issue_open()) return;
    issue_pieces(std::integral_constant<int, 0>{}, std::integral_constant<int, NPS>{});
    issue_close();
  };

  // Software pipeline: STAGES LDS buffers, STAGES-1 K steps of LDS-DMA in flight.  Every wave waits for its own DMA
  // instructions of the step it is about to read with a COUNTED vmcnt (the younger steps stay in flight across the
  // barrier), then a raw s_barrier makes every wave's part visible (cdna_hip_programming.md "Pipelining across barriers").
  constexpr int WAIT_N = NPS * (STAGES - 2);          // instructions of the younger in-flight steps
  // the OLDEST outstanding step must have landed; the inflight-1 younger ones may stay in flight when the pipe is full
  // (the epilogue's global stores count in vmcnt too: they only make this wait conservative, never early)
  auto wait_oldest = [&]() {
    if (is_loader) {
      if (STAGES > 2 && inflight - 1 == STAGES - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAIT_N) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // my LDS reads (and register-path writes) are done
    __builtin_amdgcn_s_barrier();                        // ... and every other wave's DMA part is visible
    asm volatile("" ::: "memory");                       // keep the next step's LDS traffic below the barrier
  };

#ifdef GEMM_STAMP
  unsigned long long st_entry = STAMP_NOW(), st_wait = 0, st_k = 0, st_ep = 0, st_first, st_tiles = 0, st_steps = 0;
#endif
  issue_begin_tile(iw);
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s) issue_step();
  wait_oldest();
#ifdef GEMM_STAMP
  st_first = STAMP_NOW();
#endif

  const int lr = lane & 31, lh = lane >> 5;
  for (int w = PERSIST ? (int)blockIdx.x : 0; PERSIST ? (w < ntiles) : (w == 0); w += PERSIST ? (int)gridDim.x : 1) {
    Sched csc;
    Geo cg;
    decode(w, cg);
    const int tile_m = cg.tile_m, m0 = cg.m0, n0 = cg.n0, split = cg.split, zo = cg.zo, zi = cg.zi;
    sched_setup(m0, split, csc);
    const int nk = csc.nk;

    f32x16_t acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // Bias gradient inside the weight-gradient GEMM: db[n] = sum_m dy[m][n] is the row sum of the A operand over the
    // reduction.  The A fragments are in registers anyway, so the workgroups of the FIRST column tile (wave column 0) run
    // one extra MFMA per fragment against an all-ones B fragment: every column of that product is the row sum.
    constexpr bool ROWSUM = (AMODE == OPM_DENSE_MN && BMODE == OPM_DENSE_MN);
    const bool do_rowsum = ROWSUM && p.rowsum != nullptr && cg.tile_n == 0 && wn == 0;      // wave-uniform
    f32x16_t racc[ROWSUM ? MI : 1];
    if constexpr (ROWSUM) {
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) racc[i][r] = 0.f;
    }

#ifdef GEMM_STAMP
    const unsigned long long st_k0 = STAMP_NOW();
    st_tiles += 1; st_steps += nk;
#endif
    for (int kt = 0; kt < nk; ++kt) {
      const bool more = issue_open();
      constexpr int NKS = sizeof(T) == 2 ? KT / 16 : 1;     // MFMA sub-steps the DMA pieces are spread over (bf16)
#ifndef GEMM_INTERLEAVE_NP
#define GEMM_INTERLEAVE_NP 0     // measured: spreading the pieces costs the multi-workgroup kernels 10-30 % (the DMA starts later)
#endif
      // DMA pieces spread between the MFMA sub-steps (bf16 DMA kernels) or issued up front (register path / fp32)
      constexpr bool SPREAD = sizeof(T) == 2 && LA::DMA && LB::DMA && (PERSIST || GEMM_INTERLEAVE_NP);
      if constexpr (!SPREAD) {
        if (more) issue_pieces(std::integral_constant<int, 0>{}, std::integral_constant<int, NPS>{});
      }
      const char* As = smem + cslot * STAGE_BYTES;
      const char* Bs = As + A_BYTES;
      // (experiment, off) k-major operands of the one-tile-per-workgroup kernels: ALL fragment reads of the K step issued before its first MFMA (the
      // compiler, left alone, keeps one read ahead of each MFMA — 12 fragment registers, an LDS latency in front of every product);
      // the MFMAs then wait with counted lgkmcnt for the fragments in the order they arrive
      constexpr bool FRAG_AHEAD = GEMM_FRAG_AHEAD && sizeof(T) == 2 && LA::KM && LB::KM && !PERSIST && !ROWSUM && (MI + NI) * (KT / 16) * 4 <= 64;
      if constexpr (FRAG_AHEAD) {
        constexpr int NKS2 = KT / 16;
        mfma_bf16x8_t fa[NKS2][MI], fb[NKS2][NI];
#pragma unroll
        for (int ks = 0; ks < NKS2; ++ks) {
#pragma unroll
          for (int i = 0; i < MI; ++i) fa[ks][i] = *reinterpret_cast<const mfma_bf16x8_t*>(As + lds_chunk_off<KB>(wm * WM + i * 32 + lr, 2 * ks + lh));
#pragma unroll
          for (int j = 0; j < NI; ++j) fb[ks][j] = *reinterpret_cast<const mfma_bf16x8_t*>(Bs + lds_chunk_off<KB>(wn * WN + j * 32 + lr, 2 * ks + lh));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < NKS2; ++ks)
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks][i], fb[ks][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      } else if constexpr (sizeof(T) == 2) {
        // lane geometry of the transposed read (mn-major images): 16-lane group g, row q / column quad pq inside it
        const int tg = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
#pragma unroll
        for (int ks = 0; ks < KT / 16; ++ks) {
          mfma_bf16x8_t fa[MI], fb[NI];
          s16x4_t alo[MI], ahi[MI], blo[NI], bhi[NI];      // halves of the transposed-read fragments (mn-major operands)
#pragma unroll
          for (int i = 0; i < MI; ++i) { alo[i] = ahi[i] = s16x4_t{0, 0, 0, 0}; }
#pragma unroll
          for (int j = 0; j < NI; ++j) { blo[j] = bhi[j] = s16x4_t{0, 0, 0, 0}; }
#pragma unroll
          for (int i = 0; i < MI; ++i) {
            if constexpr (LA::KM) {
              fa[i] = *reinterpret_cast<const mfma_bf16x8_t*>(As + lds_chunk_off<KB>(wm * WM + i * 32 + lr, 2 * ks + lh));
            } else {
              const int col = wm * WM + i * 32 + 16 * (tg & 1) + 4 * tp;
              const int k0r = 16 * ks + 8 * (tg >> 1) + tq;
              alo[i] = ds_read_tr_asm(As + k0r * LA::ROWB + (((col >> 3) ^ mn_swz<LA::ROWB>(k0r)) << 4) + (tp & 1) * 8);
              ahi[i] = ds_read_tr_asm(As + (k0r + 4) * LA::ROWB + (((col >> 3) ^ mn_swz<LA::ROWB>(k0r + 4)) << 4) + (tp & 1) * 8);
            }
          }
#pragma unroll
          for (int j = 0; j < NI; ++j) {
            if constexpr (LB::KM) {
              fb[j] = *reinterpret_cast<const mfma_bf16x8_t*>(Bs + lds_chunk_off<KB>(wn * WN + j * 32 + lr, 2 * ks + lh));
            } else {
              const int col = wn * WN + j * 32 + 16 * (tg & 1) + 4 * tp;
              const int k0r = 16 * ks + 8 * (tg >> 1) + tq;
              blo[j] = ds_read_tr_asm(Bs + k0r * LB::ROWB + (((col >> 3) ^ mn_swz<LB::ROWB>(k0r)) << 4) + (tp & 1) * 8);
              bhi[j] = ds_read_tr_asm(Bs + (k0r + 4) * LB::ROWB + (((col >> 3) ^ mn_swz<LB::ROWB>(k0r + 4)) << 4) + (tp & 1) * 8);
            }
          }
          // the assembly reads are invisible to the compiler's lgkmcnt bookkeeping: one wait for all of this sub-step's, tied to their
          // registers so that nothing consuming them moves above it
          typedef __attribute__((ext_vector_type(8))) short s16x8_t;
          if constexpr (!LA::KM) {
#pragma unroll
            for (int i = 0; i < MI; ++i) asm volatile("" : "+v"(alo[i]), "+v"(ahi[i]));
          }
          if constexpr (!LB::KM) {
#pragma unroll
            for (int j = 0; j < NI; ++j) asm volatile("" : "+v"(blo[j]), "+v"(bhi[j]));
          }
          if constexpr (!LA::KM || !LB::KM) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          if constexpr (!LA::KM) {
#pragma unroll
            for (int i = 0; i < MI; ++i) asm volatile("" : "+v"(alo[i]), "+v"(ahi[i]));
          }
          if constexpr (!LB::KM) {
#pragma unroll
            for (int j = 0; j < NI; ++j) asm volatile("" : "+v"(blo[j]), "+v"(bhi[j]));
          }
          if constexpr (!LA::KM) {
#pragma unroll
            for (int i = 0; i < MI; ++i) fa[i] = __builtin_bit_cast(mfma_bf16x8_t, (s16x8_t)__builtin_shufflevector(alo[i], ahi[i], 0, 1, 2, 3, 4, 5, 6, 7));
          }
          if constexpr (!LB::KM) {
#pragma unroll
            for (int j = 0; j < NI; ++j) fb[j] = __builtin_bit_cast(mfma_bf16x8_t, (s16x8_t)__builtin_shufflevector(blo[j], bhi[j], 0, 1, 2, 3, 4, 5, 6, 7));
          }
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
          if constexpr (ROWSUM) {
            if (do_rowsum) {
              mfma_bf16x8_t ones;
#pragma unroll
              for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
#pragma unroll
              for (int i = 0; i < MI; ++i) racc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], ones, racc[i], 0, 0, 0);
            }
          }
          if constexpr (SPREAD) {      // this sub-step's share of the next step's DMA pieces, behind its MFMAs
            if (more) {
              if (ks == 0) issue_pieces(std::integral_constant<int, 0>{}, std::integral_constant<int, NPS * 1 / NKS>{});
              if constexpr (NKS > 1) { if (ks == 1) issue_pieces(std::integral_constant<int, NPS * 1 / NKS>{}, std::integral_constant<int, NPS * 2 / NKS>{}); }
              if constexpr (NKS > 2) { if (ks == 2) issue_pieces(std::integral_constant<int, NPS * 2 / NKS>{}, std::integral_constant<int, NPS * 3 / NKS>{}); }
              if constexpr (NKS > 3) { if (ks == 3) issue_pieces(std::integral_constant<int, NPS * 3 / NKS>{}, std::integral_constant<int, NPS>{}); }
            }
          }
        }
      } else {
#pragma unroll
        for (int ks = 0; ks < KT / 2; ++ks) {
          const int k = 2 * ks + lh;  // float index inside the row
          float fa[MI], fb[NI];
#pragma unroll
          for (int i = 0; i < MI; ++i)
            fa[i] = *reinterpret_cast<const float*>(As + lds_chunk_off<KB>(wm * WM + i * 32 + lr, k >> 2) + (k & 3) * 4);
#pragma unroll
          for (int j = 0; j < NI; ++j)
            fb[j] = *reinterpret_cast<const float*>(Bs + lds_chunk_off<KB>(wn * WN + j * 32 + lr, k >> 2) + (k & 3) * 4);
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
          if constexpr (ROWSUM) {
            if (do_rowsum) {
#pragma unroll
              for (int i = 0; i < MI; ++i) racc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], 1.0f, racc[i], 0, 0, 0);
            }
          }
        }
      }
      if (more) issue_close();
      cslot = cslot + 1 == STAGES ? 0 : cslot + 1;
      --inflight;
#ifdef GEMM_STAMP
      const unsigned long long st_w0 = STAMP_NOW();
      wait_oldest();
      st_wait += STAMP_NOW() - st_w0;
#else
      wait_oldest();
#endif
    }
#ifdef GEMM_STAMP
    const unsigned long long st_k1 = STAMP_NOW();
    st_k += st_k1 - st_k0;
#endif
    if constexpr (!PERSIST) __syncthreads();   // the slab aliases the ring
#ifdef GEMM_STAMP
    unsigned long long st_e = st_k1, se0 = 0, se1 = 0, se2 = 0, se3 = 0, se4 = 0, se5 = 0;
#endif
    EP_STAMP(se0);

#define EP_SYNC()                                                                                   \
  do {                                                                                              \
    if constexpr (PERSIST) {                                                                        \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                            \
      __builtin_amdgcn_s_barrier();                                                                 \
      asm volatile("" ::: "memory");                                                                \
    } else {                                                                                        \
      __syncthreads();                                                                              \
    }                                                                                               \
  } while (0)

  // ---------------------------------------------------------------- epilogue
  // (0) bias-gradient partials of this K slice (weight-gradient launches): lanes 0 and 32 hold the 32 row sums of a tile
  if constexpr (ROWSUM) {
    if (do_rowsum && lr == 0) {
      float* rs = p.rowsum + (long)split * p.M;
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (row < p.M) rs[row] = racc[i][r];
        }
    }
  }
  // (1) BatchNorm partial statistics straight from the fp32 accumulators (conv forward only)
  if (p.stats) {
    float* red = reinterpret_cast<float*>(ep_mem);  // [NWM][2 stat][BN]
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float v = acc[i][j][r]; s1 += v; s2 += v * v; }
      // rows beyond M were staged as zeros and conv has no bias, so they add nothing to the sums
      s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 32, 64);
      if (lh == 0) {
        red[(wm * 2 + 0) * BN + wn * WN + j * 32 + lr] = s1;
        red[(wm * 2 + 1) * BN + wn * WN + j * 32 + lr] = s2;
      }
    }
    EP_SYNC();
    if (is_storer && stid < BN && n0 + stid < p.N) {
      float* st = p.stats + (long)tile_m * 2 * p.N;
      float a1 = 0.f, a2 = 0.f;
#pragma unroll
      for (int w = 0; w < NWM; ++w) { a1 += red[(w * 2 + 0) * BN + stid]; a2 += red[(w * 2 + 1) * BN + stid]; }
      st[n0 + stid] = a1;
      st[p.N + n0 + stid] = a2;
    }
    EP_SYNC();
  }

  EP_STAMP(se1);
  // (2) accumulators -> LDS slab (fp32, SLAB_ROWS rows per pass) -> 16-byte coalesced global stores with the fused
  //     bias / pre-activation copy / activation / residual applied on whole vectors
  float* slab = reinterpret_cast<float*>(ep_mem);
  const bool f32out = (p.nsplit > 1 || p.out_f32);
  char* Cbase = f32out ? reinterpret_cast<char*>(reinterpret_cast<float*>(p.C) + (p.nsplit > 1 ? (long)split * p.M * p.ldc : (zo * p.sc_o + zi * p.sc_i)))
                       : reinterpret_cast<char*>(reinterpret_cast<T*>(p.C) + zo * p.sc_o + zi * p.sc_i);
  constexpr int PASSES = BM > SLAB_ROWS ? BM / SLAB_ROWS : 1;   // SLAB_ROWS output rows per pass (BM = 64: the whole tile)
  float csum[16 / sizeof(T)];   // this thread's column chunk is the same in every iteration (NSTORE % (BN/VEC) == 0)
  float csum2[16 / sizeof(T)];
#pragma unroll
  for (int e = 0; e < (int)(16 / sizeof(T)); ++e) { csum[e] = 0.f; csum2[e] = 0.f; }
  // EPX 2: the four per-channel constants (mean, rstd, scale, shift) of the tile's BN columns wait in LDS behind the slab (the rest of the
  // dead operand ring) — as registers they would be 4 x VEC per thread on top of the two sum sets, past the 128-register budget
  float* const bnc = reinterpret_cast<float*>(ep_mem + EP_BYTES);     // [4][BN]
  if constexpr (EPX == 2) {
    static_assert(PERSIST || EP_BYTES + 4 * BN * 4 <= SMEM_BYTES, "no room behind the slab for the BatchNorm constants");
    if (tid < BN) {
      const int n = n0 + tid;
#pragma unroll
      for (int k = 0; k < 4; ++k) bnc[k * BN + tid] = n < p.N ? p.bnstat[(long)k * p.N + n] : 0.f;
    }
    // (visible to the storing threads after the barrier that follows the first slab write)
  }
#pragma unroll
  for (int ps = 0; ps < PASSES; ++ps) {
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int r0 = wm * WM + i * 32;            // first tile row of this 32-row accumulator block (wave-uniform)
      if (r0 / SLAB_ROWS != ps) continue;         // not in this pass's rows
      const int rbase = r0 - ps * SLAB_ROWS;
#pragma unroll
      for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          slab[(rbase + (r & 3) + 8 * (r >> 2) + 4 * lh) * BN + wn * WN + j * 32 + lr] = p.alpha * acc[i][j][r];
    }
    EP_SYNC();
    EP_STAMP(se2);
    if (!is_storer) {
    } else if (f32out) {
      constexpr int CPRW = BN / 4;
#pragma unroll
      for (int it = 0; it < SLAB_ROWS * CPRW / NSTORE; ++it) {
        const int id = stid + it * NSTORE;
        const int row = id / CPRW, cc = id % CPRW;
        const int m = m0 + ps * SLAB_ROWS + row, n = n0 + cc * 4;
        if (m >= p.M || n >= p.N) continue;
        const float4 v = *reinterpret_cast<const float4*>(slab + row * BN + cc * 4);
        float* dst = reinterpret_cast<float*>(Cbase) + (long)m * p.ldc + n;
        if (n + 4 <= p.N && (p.ldc & 3) == 0) {
          *reinterpret_cast<float4*>(dst) = v;
        } else {
          const float vv[4] = {v.x, v.y, v.z, v.w};
          for (int e = 0; e < 4 && n + e < p.N; ++e) dst[e] = vv[e];
        }
      }
    } else {
      constexpr int VEC = 16 / sizeof(T);
      constexpr int CPRW = BN / VEC;
      T* C2 = reinterpret_cast<T*>(p.C2);
      const T* Rs = reinterpret_cast<const T*>(p.res);
#pragma unroll
      for (int it = 0; it < SLAB_ROWS * CPRW / NSTORE; ++it) {
        const int id = stid + it * NSTORE;
        const int row = id / CPRW, cc = id % CPRW;
        const int m = m0 + ps * SLAB_ROWS + row, n = n0 + cc * VEC;
        if (m >= p.M || n >= p.N) continue;
        float v[VEC];
#pragma unroll
        for (int e = 0; e < VEC; e += 4) {
          const float4 t = *reinterpret_cast<const float4*>(slab + row * BN + cc * VEC + e);
          v[e] = t.x; v[e + 1] = t.y; v[e + 2] = t.z; v[e + 3] = t.w;
        }
        long orow = m;
        if constexpr (AMODE == OPM_CONV_DGRAD) {
          if (p.cls) {   // class-major GEMM row -> NHWC pixel row
            int bb, hh, ww;
            dgrad_row_to_pixel(p, m, bb, hh, ww);
            orow = ((long)bb * p.cH + hh) * p.cW + ww;
          }
        }
        const long o = orow * p.ldc + n;
        T* dst = reinterpret_cast<T*>(Cbase) + o;
        if (n + VEC <= p.N && (p.ldc % VEC) == 0) {
          if (p.bias) {
#pragma unroll
            for (int e = 0; e < VEC; e += 4) {
              const float4 b4 = *reinterpret_cast<const float4*>(p.bias + n + e);
              v[e] += b4.x; v[e + 1] += b4.y; v[e + 2] += b4.z; v[e + 3] += b4.w;
            }
          }
          vecf<T> vo;
          if (p.act == MSC_ACT_GELU_DSAVE) {      // y = gelu(u), side output = gelu'(u)
#pragma unroll
            for (int e = 0; e < VEC; ++e) gelu_both(v[e], v[e], vo.v[e]);
            if (C2) { if constexpr (GEMM_NT_STORE) vo.store_nt(C2 + o); else vo.store(C2 + o); }
          } else {
            if (C2) {
#pragma unroll
              for (int e = 0; e < VEC; ++e) vo.v[e] = v[e];
              if constexpr (GEMM_NT_STORE) vo.store_nt(C2 + o); else vo.store(C2 + o);
            }
            if (p.act != MSC_ACT_NONE) {
#pragma unroll
              for (int e = 0; e < VEC; ++e) v[e] = act_apply(v[e], p.act);
            }
          }
          if (Rs) {
            vecf<T> vr;
            vr.load(Rs + o);
#pragma unroll
            for (int e = 0; e < VEC; ++e) v[e] += vr.v[e];
          }
          if constexpr (EPX) {
            if (p.dref) {   // fused activation backward: gradient w.r.t. the pre-activation
              vecf<T> vr;
              vr.load(reinterpret_cast<const T*>(p.dref) + o);
#pragma unroll
              for (int e = 0; e < VEC; ++e) v[e] *= act_grad(vr.v[e], p.dact);
            }
            if (p.colsum) {
#pragma unroll
              for (int e = 0; e < VEC; ++e) csum[e] += v[e];
            }
            if constexpr (EPX == 2) {
              vecf<T> vx;
              vx.load(reinterpret_cast<const T*>(p.bnx) + o);
              const float* kc = bnc + cc * VEC;
#pragma unroll
              for (int e = 0; e < VEC; ++e) {
                const float dz = fmaf(vx.v[e], kc[2 * BN + e], kc[3 * BN + e]) > 0.f ? msc_traits<T>::rnd(v[e]) : 0.f;    // dy as the consumer reads it back
                csum[e] += dz;
                csum2[e] += dz * (vx.v[e] - kc[e]) * kc[BN + e];
              }
            }
          }
#pragma unroll
          for (int e = 0; e < VEC; ++e) vo.v[e] = v[e];
          if constexpr (GEMM_NT_STORE) vo.store_nt(dst); else vo.store(dst);
        } else {
          for (int e = 0; e < VEC && n + e < p.N; ++e) {
            float x = v[e] + (p.bias ? p.bias[n + e] : 0.f);
            if (p.act == MSC_ACT_GELU_DSAVE) {
              float d;
              gelu_both(x, x, d);
              if (C2) msc_traits<T>::st(C2 + o + e, d);
            } else {
              if (C2) msc_traits<T>::st(C2 + o + e, x);
              x = act_apply(x, p.act);
            }
            if (Rs) x += msc_traits<T>::ld(Rs + o + e);
            if constexpr (EPX) {
              if (p.dref) x *= act_grad(msc_traits<T>::ld(reinterpret_cast<const T*>(p.dref) + o + e), p.dact);
              if (p.colsum) csum[e] += x;
              if constexpr (EPX == 2) {
                const float xv = msc_traits<T>::ld(reinterpret_cast<const T*>(p.bnx) + o + e);
                const float* kc = bnc + cc * VEC + e;
                const float dz = fmaf(xv, kc[2 * BN], kc[3 * BN]) > 0.f ? msc_traits<T>::rnd(x) : 0.f;
                csum[e] += dz;
                csum2[e] += dz * (xv - kc[0]) * kc[BN];
              }
            }
            msc_traits<T>::st(dst + e, x);
          }
        }
      }
    }
    EP_STAMP(se3);
    if (ps + 1 < PASSES) EP_SYNC();
    EP_STAMP(se4);
  }
  if (EPX && (p.colsum || (EPX == 2 && p.bnpart)) && !f32out) {
    // fixed-order reduction of the per-thread column sums over the NTHR/(BN/VEC) threads that share a column chunk
    constexpr int VEC = 16 / sizeof(T);
    constexpr int CPRW = BN / VEC, NR = NSTORE / CPRW;
    static_assert(NR <= SLAB_ROWS, "column-sum scratch must fit the slab");
    const int cc = stid % CPRW, rr = stid / CPRW;
    auto reduce_cols = [&](const float (&cs)[VEC], float* dst_row) {
      EP_SYNC();
      if (is_storer) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) slab[rr * BN + cc * VEC + e] = cs[e];
      }
      EP_SYNC();
      if (is_storer && stid < BN && n0 + stid < p.N) {
        float a = 0.f;
        for (int r = 0; r < NR; ++r) a += slab[r * BN + stid];
        dst_row[n0 + stid] = a;
      }
    };
    if (EPX == 2 && p.bnpart) {
      reduce_cols(csum, p.bnpart + (long)tile_m * 2 * p.N);
      reduce_cols(csum2, p.bnpart + (long)tile_m * 2 * p.N + p.N);
    } else {
      reduce_cols(csum, p.colsum + (long)tile_m * p.N);
    }
  }
    if constexpr (AMODE == OPM_CONV_DGRAD && !PERSIST && EPX == 0) {
      if (p.cls_fill) {      // the sibling classes of this tile's pixel region (no taps there): residual or zero
        const int ncls = p.cS * p.cS;
        for (int c = 1; c < ncls; ++c) fill_rows(m0 + c * p.cMc, n0);
      }
    }
    if constexpr (PERSIST) EP_SYNC();   // the slab is free again before the next tile's epilogue touches it
#ifdef GEMM_STAMP
    EP_STAMP(se5);
    st_ep += STAMP_NOW() - st_k1;
    if (p.stamps && tid == 0) {
      unsigned long long* o = p.stamps + ((size_t)GEMM_STAMP_EP_OFF + (size_t)blockIdx.z * gridDim.x + blockIdx.x) * 8;
      o[0] = se0; o[1] = se1; o[2] = se2; o[3] = se3; o[4] = se4; o[5] = se5;
    }
#endif
  }   // tile loop
#ifdef GEMM_STAMP
  if (p.stamps && tid == 0) {
    unsigned long long* o = p.stamps + ((size_t)blockIdx.z * gridDim.x + blockIdx.x) * 8;
    o[0] = st_entry; o[1] = st_first; o[2] = st_wait; o[3] = st_k; o[4] = st_ep; o[5] = STAMP_NOW(); o[6] = st_tiles; o[7] = st_steps;
  }
#endif
#undef EP_SYNC
}

// dw[i] = (accumulate ? dw[i] : 0) + sum_s slab[s][i]   (fixed order: bitwise reproducible)
// Block = 64 float4 columns x RY slab groups: small outputs with many slabs (early conv layers: 36 K elements x ~100
// slabs) would otherwise walk the slabs serially from a handful of threads.  Group y sums slabs y, y+RY, ...; the groups
// are combined through LDS in the order 0..RY-1.
template <int RY>
__global__ __launch_bounds__(64 * RY) void splitk_reduce_kernel(const float* __restrict__ slab, float* __restrict__ out, long n4, int nsplit, int accumulate,
                                                                const float* __restrict__ slab2, float* __restrict__ out2, long n4b, unsigned blocks1) {
  __shared__ float4 part[RY][64];
  // blocks past blocks1 reduce the second (small) job of the launch: the bias-gradient row sums [nsplit][4*n4b] -> out2
  long i = (long)blockIdx.x * 64 + threadIdx.x;
  if (blockIdx.x >= blocks1) {
    slab = slab2; out = out2; n4 = n4b;
    i = (long)(blockIdx.x - blocks1) * 64 + threadIdx.x;
  }
  const int y = threadIdx.y;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (i < n4) {
    int k = y;
    for (; k + RY < nsplit; k += 2 * RY) {   // two independent loads in flight
      const float4 t0 = reinterpret_cast<const float4*>(slab)[(long)k * n4 + i];
      const float4 t1 = reinterpret_cast<const float4*>(slab)[(long)(k + RY) * n4 + i];
      s.x += t0.x; s.y += t0.y; s.z += t0.z; s.w += t0.w;
      s.x += t1.x; s.y += t1.y; s.z += t1.z; s.w += t1.w;
    }
    if (k < nsplit) {
      const float4 t0 = reinterpret_cast<const float4*>(slab)[(long)k * n4 + i];
      s.x += t0.x; s.y += t0.y; s.z += t0.z; s.w += t0.w;
    }
  }
  if (RY > 1) {
    part[y][threadIdx.x] = s;
    __syncthreads();
    if (y != 0) return;
    s = part[0][threadIdx.x];
#pragma unroll
    for (int r = 1; r < RY; ++r) { const float4 t = part[r][threadIdx.x]; s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w; }
  }
  if (i < n4) {
    if (accumulate) { const float4 o = reinterpret_cast<const float4*>(out)[i]; s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w; }
    reinterpret_cast<float4*>(out)[i] = s;
  }
}

// Finish of a split-K conv forward / dgrad: y[m][n] = T( sum_s slab[s][m][n] (+ res[m][n]) ), optionally with the BatchNorm
// partial statistics (sum, sum of squares per channel over each block of RB rows) the un-split epilogue would have written.
// Block = 64 columns x RB rows: thread (c4 = 4 columns, rl = one of 16 row lanes); fixed summation order everywhere.
template <typename T>
__global__ __launch_bounds__(256) void splitk_finish_kernel(const float* __restrict__ slab, int ns, long M, int N, T* __restrict__ y,
                                                            const T* __restrict__ res, float* __restrict__ stats, int RB,
                                                            const T* __restrict__ relu_ref, const T* __restrict__ bnx = nullptr,
                                                            const float* __restrict__ bnstat = nullptr) {
  __shared__ float red[2][16][64];
  const int c4 = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int n = blockIdx.x * 64 + c4 * 4;
  const long r0 = (long)blockIdx.y * RB;
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  if (n < N) {
    for (int r = rl; r < RB; r += 16) {
      const long m = r0 + r;
      if (m >= M) break;
      const long e = m * N + n;
      float4 a = *reinterpret_cast<const float4*>(slab + e);
      for (int k = 1; k < ns; ++k) {
        const float4 t = *reinterpret_cast<const float4*>(slab + (long)k * M * N + e);
        a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
      }
      float v[4] = {a.x, a.y, a.z, a.w};
      if (!bnx) {
        s1[0] += v[0]; s1[1] += v[1]; s1[2] += v[2]; s1[3] += v[3];
        s2[0] += v[0] * v[0]; s2[1] += v[1] * v[1]; s2[2] += v[2] * v[2]; s2[3] += v[3] * v[3];
      }
      if (res) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += msc_traits<T>::ld(res + e + j);
      }
      if (relu_ref) {     // dgrad: ReLU backward of the activation that produced this layer's input (relu_ref = that activation)
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = msc_traits<T>::ld(relu_ref + e + j) > 0.f ? v[j] : 0.f;
      }
      if (bnx) {          // dgrad: BatchNorm-backward sums of the producer of this layer's input (see GemmParams::bnpart)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float xv = msc_traits<T>::ld(bnx + e + j);
          const float dz = fmaf(xv, bnstat[2 * N + n + j], bnstat[3 * N + n + j]) > 0.f ? msc_traits<T>::rnd(v[j]) : 0.f;
          s1[j] += dz;
          s2[j] += dz * (xv - bnstat[n + j]) * bnstat[N + n + j];
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) msc_traits<T>::st(y + e + j, v[j]);
    }
  }
  if (!stats) return;
#pragma unroll
  for (int j = 0; j < 4; ++j) { red[0][rl][c4 * 4 + j] = s1[j]; red[1][rl][c4 * 4 + j] = s2[j]; }
  __syncthreads();
  if (threadIdx.x < 64 && blockIdx.x * 64 + threadIdx.x < N) {
    float a = 0.f, b = 0.f;
    for (int r = 0; r < 16; ++r) { a += red[0][r][threadIdx.x]; b += red[1][r][threadIdx.x]; }
    float* st = stats + (long)blockIdx.y * 2 * N;
    st[blockIdx.x * 64 + threadIdx.x] = a;
    st[N + blockIdx.x * 64 + threadIdx.x] = b;
  }
}

// gemm_pp.hip
int msc_pp_tile_rows(const GemmParams& p, int ncu);
int msc_pp_launch(const GemmParams& p, int ncu, hipStream_t st);
bool msc_conv_c64_ok(int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int dtype);      // conv_c64.hip: ResNet layer 1
int msc_conv_c64_launch(const void* x, const void* w, void* y, const void* res, const void* relu_ref, float* part, int part_rows, int B, int H,
                        int flip, int ncu, hipStream_t st);
bool msc_stem_fwd_halo_ok(int B, int Hs, int Ws, int Cout, int dtype);                           // conv_stem.hip
int msc_stem_fwd_halo_launch(const void* xs, const void* w, void* y, float* part, int part_rows, int B, int Hs, int ncu, hipStream_t st);
bool msc_stem_wgrad_halo_ok(int B, int Hs, int Ws, int Cout, int dtype, int64_t ws_bytes);
int msc_stem_wgrad_halo_launch(const void* xs, const void* dy, float* slab, int64_t slab_bytes, int B, int Hs, int ncu, hipStream_t st);
bool msc_conv_c64_wgrad_ok(int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int dtype, int64_t ws_bytes);      // conv_c64w.hip
int64_t msc_conv_c64_wgrad_workspace_bytes(int ncu);
int msc_conv_c64_wgrad_launch(const void* x, const void* dy, float* slab, int64_t slab_bytes, int B, int H, int ncu, hipStream_t st);
bool msc_skinny_ok(const GemmParams& p);                    // gemm_skinny.hip: dense bf16 linears of at most 64 rows
int msc_skinny_launch(const GemmParams& p, hipStream_t st);
// the most recent contraction launch of this process, for tests that assert which kernel variant a shape took (msc_debug_last_gemm):
// kind (1 one tile per workgroup, 2 persistent 256-row, 3 ping-pong), bm, bn, threads, K slices, xcd_split, cls_interleave, extended epilogue
static int g_last_gemm[8] = {0, 0, 0, 0, 0, 0, 0, 0};
static inline void note_gemm(int kind, int bm, int bn, int thr, int ns, int xs, int ci, int epx) {
  g_last_gemm[0] = kind; g_last_gemm[1] = bm; g_last_gemm[2] = bn; g_last_gemm[3] = thr;
  g_last_gemm[4] = ns; g_last_gemm[5] = xs; g_last_gemm[6] = ci; g_last_gemm[7] = epx;
}
extern "C" int msc_debug_last_gemm(int* host_info8) {
  if (!host_info8) return MSC_ERR_BAD_ARG;
  memcpy(host_info8, g_last_gemm, sizeof(g_last_gemm));
  return MSC_OK;
}
extern "C" int msc_set_knob(const char* name, int64_t value) { return mscplan::knob_set(name, (long)value) ? MSC_OK : MSC_ERR_BAD_ARG; }
extern "C" int64_t msc_get_knob(const char* name) {
  long v = 0;
  return mscplan::knob_get(name, v) ? (int64_t)v : (int64_t)INT64_MIN;
}

// ---- launch planning lives in gemm_plan.h (plain C++, fuzzed under ASan/UBSan by tests/host_san)
static int num_cus() {
  static int n = 0;
  if (!n) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    n = v;
  }
  return n;
}
// the persistent 256-row kernels exist for bf16 with both operands k-major (conv gathers count as k-major A)
static inline bool can_persist(int amode, int bmode, int dtype) {
  return dtype == MSC_BF16 && bmode == OPM_DENSE_K && (amode == OPM_DENSE_K || amode == OPM_CONV_FWD || amode == OPM_CONV_DGRAD);
}
// big = the launch may take the persistent 256x256 kernel: never for launches whose epilogue writes per-row-block side
// buffers sized by a query (BatchNorm partials, column sums), so those queries depend on (rows, cols) only
static inline mscplan::Tile tile_for(long M, long N, long zs, int amode, int bmode, int dtype, long K = 0, bool big = false, int bm_cap = 256) {
  return mscplan::choose_tile(M, N, zs, big && can_persist(amode, bmode, dtype) && zs == 1, num_cus(), K, bm_cap, bmode == OPM_CONV_WGRAD ? 500 : 0);
}

extern "C" int64_t msc_conv2d_splitk_workspace_bytes(int64_t rows, int cout, int64_t k_len, int dtype) {
  return mscplan::conv_split(rows, cout, k_len, dtype == MSC_F32).ws_bytes;
}

// rows of the BatchNorm partial-statistics buffer msc_conv2d_fwd writes for an output of `rows` pixels x `cout`
extern "C" int64_t msc_conv2d_fwd_stat_rows(int64_t rows, int cout, int dtype) {
  return cdiv(rows, tile_for(rows, cout, 1, OPM_CONV_FWD, OPM_DENSE_K, dtype).bm);
}

template <typename T, int AMODE, int BMODE, int EPX = 0>
static int launch_gemm(const GemmParams& p, int batch, hipStream_t st) {
  constexpr int DT = std::is_same<T, float>::value ? MSC_F32 : MSC_BF16;
  const long zs = (long)batch * p.nsplit;
#ifdef GEMM_STAMP
  if (!g_stamp_buf) (void)hipMalloc(&g_stamp_buf, (size_t)(1 << 20) * 8 * sizeof(unsigned long long));
  const_cast<GemmParams&>(p).stamps = g_stamp_buf;
#endif
  const bool big = p.nsplit == 1 && !p.xcd_split && !p.stats && !p.colsum && !p.bnpart;
  const mscplan::Tile t = tile_for(p.M, p.N, zs, AMODE, BMODE, DT, p.K, big, p.bm_cap ? p.bm_cap : 256);
  const int bm = t.bm, bn = t.bn;
  GemmParams q = p;
  q.cls_interleave = (AMODE == OPM_CONV_DGRAD && p.cls && mscplan::g_knobs.dgrad_cls_interleave && p.nsplit == 1 && p.cMc % bm == 0) ? 1 : 0;
  if (zs > 65535) return MSC_ERR_UNSUPPORTED;
  // dense bf16 linears (both operands k-major, one launch over the whole matrix): the persistent ping-pong kernel of gemm_pp.hip
  if constexpr (DT == MSC_BF16 && AMODE == OPM_DENSE_K && BMODE == OPM_DENSE_K && EPX != 2) {
    if (batch == 1 && msc_skinny_ok(p)) {     // one row per frame (codec, heads, state encoder): no staging, one memory round trip
      note_gemm(4, 64, 16, p.K >= 2048 ? 512 : 256, 1, 0, 0, EPX);
      return msc_skinny_launch(p, st);
    }
    if (batch == 1 && msc_pp_tile_rows(p, num_cus())) {
      note_gemm(3, msc_pp_tile_rows(p, num_cus()), 256, 512, 1, 0, 0, EPX);
      return msc_pp_launch(p, num_cus(), st);
    }
  }
  if constexpr (DT == MSC_BF16 && BMODE == OPM_DENSE_K && mode_is_kmajor<AMODE>::value && EPX != 2) {     // (EPX 2: never `big`)
    if (bm == 256) {     // persistent: one workgroup per CU walks the tiles
      const int ntiles = cdiv(p.M, 256) * cdiv(p.N, bn);
      const int ncu = num_cus();
      dim3 grid((unsigned)(ntiles < ncu ? ntiles : ncu), 1, 1);
      note_gemm(2, 256, bn, 512, 1, 0, q.cls_interleave, EPX);
      if (bn == 256) hipLaunchKernelGGL((gemm_kernel<T, AMODE, BMODE, 256, 256, EPX, GEMM_P256_KB, GEMM_P256_STAGES, true, 512>), grid, dim3(512), 0, st, q);
      else hipLaunchKernelGGL((gemm_kernel<T, AMODE, BMODE, 256, 128, EPX, GEMM_P128_KB, GEMM_P128_STAGES, true, 512>), grid, dim3(512), 0, st, q);
      return msc_launch_status();
    }
  }
  if (bm > 128 || bn > 128) return MSC_ERR_UNSUPPORTED;
  dim3 grid((unsigned)(cdiv(p.M, bm) * cdiv(p.N, bn)), 1, (unsigned)zs);
  if constexpr (AMODE == OPM_CONV_DGRAD && EPX == 0) {
    // 1x1 strided kernels (pad 0): only parity class 0 has a tap; launch its tiles alone, they fill the sibling classes' rows (kernel: fill_rows)
    if (p.cls && p.cKH == 1 && p.cKW == 1 && p.cP == 0 && p.nsplit == 1 && !p.out_f32 && !p.bias && !p.stats && p.act == MSC_ACT_NONE && !p.C2 &&
        p.cMc % bm == 0 && mscplan::g_knobs.dgrad_cls_fill) {
      q.cls_fill = 1;
      q.cls_interleave = 0;
      grid = dim3((unsigned)((p.cMc / bm) * cdiv(p.N, bn)), 1, (unsigned)zs);
    }
  }
  // K bytes per step / ring depth / waves of the multi-workgroup kernels (measured per operand mode, tools/gemm_sweep.py):
  // gathered weight gradients (conv wgrad) run best on 64-byte steps x 3 stages x 4 waves (three workgroups per CU);
  // everything else on 128-byte steps x 2 stages, the 128x128 tile with 8 waves
  constexpr bool WG = BMODE == OPM_CONV_WGRAD;
  constexpr int KBn = WG ? 64 : GEMM_NP_KB, S = WG ? GEMM_WG_STAGES : GEMM_NP_STAGES, W128 = WG ? 256 : 64 * GEMM_NP_W128;
  note_gemm(1, bm, bn, (bm == 128 && bn == 128) ? W128 : 256, p.nsplit, p.xcd_split, q.cls_interleave, EPX);
  if (bm == 128 && bn == 128) hipLaunchKernelGGL((gemm_kernel<T, AMODE, BMODE, 128, 128, EPX, KBn, S, false, W128>), grid, dim3(W128), 0, st, q);
  else if (bm == 128) hipLaunchKernelGGL((gemm_kernel<T, AMODE, BMODE, 128, 64, EPX, KBn, S, false, 256>), grid, dim3(256), 0, st, q);
  else if (bn == 128) hipLaunchKernelGGL((gemm_kernel<T, AMODE, BMODE, 64, 128, EPX, KBn, S, false, 256>), grid, dim3(256), 0, st, q);
  else hipLaunchKernelGGL((gemm_kernel<T, AMODE, BMODE, 64, 64, EPX, KBn, S + 1, false, 256>), grid, dim3(256), 0, st, q);
  return msc_launch_status();
}

template <int AMODE, int BMODE, int EPX = 0>
static int launch_gemm_dt(const GemmParams& p, int batch, int dtype, hipStream_t st) {
  if (dtype == MSC_F32) return launch_gemm<float, AMODE, BMODE, EPX>(p, batch, st);
  if (dtype == MSC_BF16) return launch_gemm<bf16_t, AMODE, BMODE, EPX>(p, batch, st);
  return MSC_ERR_BAD_ARG;
}

static GemmParams base_params() {
  GemmParams p;
  memset(&p, 0, sizeof(p));
  p.batch_inner = 1;
  p.nsplit = 1;
  p.tiles_per_split = 1 << 30;
  p.alpha = 1.f;
  return p;
}

static inline int vec_of(int dtype) { return dtype == MSC_F32 ? 4 : 8; }
static inline int kt_of(int dtype) { return mscplan::kt_of_dtype(dtype == MSC_F32); }
static inline int esize(int dtype) { return dtype == MSC_F32 ? 4 : 2; }
// every operand is addressed with 32-bit byte offsets below DMA_OOB
static inline bool fits32(long elems, int dtype) { return elems * esize(dtype) < 0x7ffffff0L; }

extern "C" int64_t msc_wgrad_workspace_bytes(int64_t rows_out, int64_t cols_out, int64_t reduce_len) {
  int64_t b = mscplan::wgrad_workspace_bytes(rows_out, cols_out, reduce_len);
  // the layer-1 kernel of conv_c64w.hip keeps one 64 x 576 partial per workgroup, one workgroup per CU (256 on MI355X; fewer slabs = fewer workgroups)
  if (rows_out == 64 && cols_out == 576) {
    const int64_t w = msc_conv_c64_wgrad_workspace_bytes(256);
    if (w > b) b = w;
  }
  return b;
}

// db (optional, linear layers): bias gradient [M] = row sums of the A operand, produced by the same launch (p.rowsum) and
// reduced over the K slices by extra blocks of the reduce launch
static int launch_splitk_reduce(const float* ws, float* dw, long n4, int ns, int accumulate, const float* ws_rows, float* db, long n4b, hipStream_t st);
static int finish_wgrad(GemmParams& p, float* dw, float* db, float* ws, int64_t ws_bytes, long reduce_len, int accumulate,
                        int dtype, hipStream_t st, int (*launch)(const GemmParams&, int, int, hipStream_t), int kstep_bytes) {
  const mscplan::WgradSplit w = mscplan::wgrad_split(p.M, p.N, reduce_len, kstep_bytes / esize(dtype), kstep_bytes == 64 /* gathered conv wgrad */);
  const int ns = w.ns;
  if ((int64_t)ns * p.M * (p.N + 1) * 4 > ws_bytes || !ws) return MSC_ERR_BAD_ARG;
  if (((long)p.M * p.N) % 4 || (db && p.M % 4)) return MSC_ERR_BAD_ARG;
  float* ws_rows = ws + (long)ns * p.M * p.N;        // [ns][M] behind the weight slabs
  p.rowsum = db ? ws_rows : nullptr;
  p.nsplit = ns;
  p.tiles_per_split = w.tps;
  p.xcd_split = w.xcd_split;
  p.C = ws;
  p.ldc = p.N;
  p.out_f32 = 1;
  if (ns == 1 && !accumulate) {   // no slab, no reduce pass: the GEMM epilogue writes dw (and db) itself
    p.C = dw;
    p.rowsum = db;
    return launch(p, 1, dtype, st);
  }
  int rc = launch(p, 1, dtype, st);
  if (rc != MSC_OK) return rc;
  return launch_splitk_reduce(ws, dw, (long)p.M * p.N / 4, ns, accumulate, ws_rows, db, db ? p.M / 4 : 0, st);
}

// fixed-order sum of ns fp32 slabs of n4 float4 each into dw (+ the [ns][n4b] row-sum slabs into db)
static int launch_splitk_reduce(const float* ws, float* dw, long n4, int ns, int accumulate, const float* ws_rows, float* db, long n4b, hipStream_t st) {
  const unsigned blocks = (unsigned)cdiv(n4, 64);
  const unsigned grid = blocks + (unsigned)cdiv(n4b, 64);
  // slab groups per block: as many as keep >= ~4 waves per CU busy, at most the slab count
  int ry = 1;
  while (ry < 16 && ry * 2 <= ns && (long)blocks * ry < 4096) ry *= 2;
  if (ry == 1) hipLaunchKernelGGL(splitk_reduce_kernel<1>, dim3(grid), dim3(64, 1), 0, st, ws, dw, n4, ns, accumulate, ws_rows, db, n4b, blocks);
  else if (ry == 2) hipLaunchKernelGGL(splitk_reduce_kernel<2>, dim3(grid), dim3(64, 2), 0, st, ws, dw, n4, ns, accumulate, ws_rows, db, n4b, blocks);
  else if (ry == 4) hipLaunchKernelGGL(splitk_reduce_kernel<4>, dim3(grid), dim3(64, 4), 0, st, ws, dw, n4, ns, accumulate, ws_rows, db, n4b, blocks);
  else if (ry == 8) hipLaunchKernelGGL(splitk_reduce_kernel<8>, dim3(grid), dim3(64, 8), 0, st, ws, dw, n4, ns, accumulate, ws_rows, db, n4b, blocks);
  else hipLaunchKernelGGL(splitk_reduce_kernel<16>, dim3(grid), dim3(64, 16), 0, st, ws, dw, n4, ns, accumulate, ws_rows, db, n4b, blocks);
  return msc_launch_status();
}

static int set_conv_geom(GemmParams& p, int B, int H, int W, int C, int Ho, int Wo, int KH, int KW, int s, int pad) {
  p.cB = B; p.cH = H; p.cW = W; p.cC = C; p.cHo = Ho; p.cWo = Wo; p.cKH = KH; p.cKW = KW; p.cS = s; p.cP = pad;
  p.cLogC = ilog2_exact(C);
  p.cLogS = ilog2_exact(s);
  p.rW = 1.f / (float)W; p.rH = 1.f / (float)H; p.rWo = 1.f / (float)Wo; p.rHo = 1.f / (float)Ho;
  p.fastdiv = ((long)B * H * W < (1L << 24) && (long)B * Ho * Wo < (1L << 24)) ? 1 : 0;     // every decoded index is a row / pixel index below these
  return (p.cLogC >= 0 && p.cLogS >= 0) ? MSC_OK : MSC_ERR_BAD_ARG;
}

// shared tail of msc_conv2d_fwd / msc_conv2d_dgrad: split-K launch + finish when it pays and the workspace allows
template <int AMODE>
static int launch_conv_maybe_split(GemmParams& p, void* out, const void* res, float* stats, float* ws, int64_t ws_bytes, int dtype,
                                   hipStream_t st, const void* relu_ref = nullptr, const void* bnx = nullptr, const float* bnstat = nullptr) {
  mscplan::ConvSplit cs = mscplan::conv_split(p.M, p.N, p.K, dtype == MSC_F32);
  if (AMODE == OPM_CONV_DGRAD && p.cls) cs.ns = 1;
  if (cs.ns <= 1 || !ws || ws_bytes < (int64_t)cs.ns * p.M * p.N * 4) {
    if constexpr (AMODE == OPM_CONV_DGRAD) {
      if (bnx) {           // un-split: the extended epilogue also sums dz, dz*xhat for the BatchNorm behind this layer's input
        p.dref = relu_ref;
        p.dact = MSC_ACT_RELU;
        p.bnx = bnx; p.bnstat = bnstat; p.bnpart = stats;
        return launch_gemm_dt<AMODE, OPM_DENSE_K, 2>(p, 1, dtype, st);
      }
      if (relu_ref) {      // un-split: the extended epilogue multiplies by relu'(relu_ref) after the residual add
        p.dref = relu_ref;
        p.dact = MSC_ACT_RELU;
        return launch_gemm_dt<AMODE, OPM_DENSE_K, 1>(p, 1, dtype, st);
      }
    }
    return launch_gemm_dt<AMODE, OPM_DENSE_K>(p, 1, dtype, st);
  }
  const int tps = cs.tps, ns = cs.ns;
  const int bm0 = tile_for(p.M, p.N, 1, AMODE, OPM_DENSE_K, dtype).bm;   // row granularity of the BatchNorm partials the caller sized its buffer for
  p.nsplit = ns;
  p.tiles_per_split = tps;
  p.C = ws; p.ldc = p.N; p.out_f32 = 1;
  p.stats = nullptr; p.res = nullptr;
  int rc = launch_gemm_dt<AMODE, OPM_DENSE_K>(p, 1, dtype, st);
  if (rc != MSC_OK) return rc;
  const dim3 grid((unsigned)cdiv(p.N, 64), (unsigned)cdiv(p.M, bm0));
  if (dtype == MSC_F32)
    hipLaunchKernelGGL(splitk_finish_kernel<float>, grid, dim3(256), 0, st, ws, ns, (long)p.M, p.N, (float*)out, (const float*)res, stats, bm0,
                       (const float*)relu_ref, (const float*)bnx, bnstat);
  else
    hipLaunchKernelGGL(splitk_finish_kernel<bf16_t>, grid, dim3(256), 0, st, ws, ns, (long)p.M, p.N, (bf16_t*)out, (const bf16_t*)res, stats, bm0,
                       (const bf16_t*)relu_ref, (const bf16_t*)bnx, bnstat);
  return msc_launch_status();
}

extern "C" int msc_conv2d_fwd(const void* x, const void* w, void* y, float* bn_partial, int B, int H, int W, int Cin,
                              int Cout, int KH, int KW, int stride, int pad, float* ws, int64_t ws_bytes, int dtype, void* stream) {
  MSC_CHECK_ARG(x && w && y && B > 0 && H > 0 && W > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0);
  MSC_CHECK_ARG(dtype == MSC_F32 || dtype == MSC_BF16);
  MSC_CHECK_ARG(Cin >= vec_of(dtype) && ilog2_exact(Cin) >= 0 && Cout > 0);
  const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
  MSC_CHECK_ARG(Ho > 0 && Wo > 0);
  if (!fits32((long)B * H * W * Cin, dtype) || !fits32((long)Cout * KH * KW * Cin, dtype)) return MSC_ERR_UNSUPPORTED;
  if (msc_conv_c64_ok(B, H, W, Cin, Cout, KH, KW, stride, pad, dtype)) {      // ResNet layer 1: halo resident in LDS, weights in registers
    note_gemm(5, 224, 64, 256, 1, 0, 0, 0);
    return msc_conv_c64_launch(x, w, y, nullptr, nullptr, bn_partial, bn_partial ? (int)msc_conv2d_fwd_stat_rows((int64_t)B * Ho * Wo, Cout, dtype) : 0, B, H,
                               0, num_cus(), (hipStream_t)stream);
  }
  GemmParams p = base_params();
  p.A = x; p.B = w; p.C = y; p.stats = bn_partial;
  p.M = B * Ho * Wo; p.N = Cout; p.K = KH * KW * Cin;
  p.ldb = p.K; p.ldc = Cout;
  if (set_conv_geom(p, B, H, W, Cin, Ho, Wo, KH, KW, stride, pad) != MSC_OK) return MSC_ERR_BAD_ARG;
  return launch_conv_maybe_split<OPM_CONV_FWD>(p, y, nullptr, bn_partial, ws, ws_bytes, dtype, (hipStream_t)stream);
}

// ---- ImageEncoder stem in space-to-depth form.  The 7x7 / stride-2 / pad-3 convolution over RGB padded to 8 channels spends
// 62 % of its MFMA work and staged bytes on zero channels (K = 7*7*8 = 392 for 147 real taps).  With the image re-blocked as
// [B][H/2][W/2][(dy,dx,c) = 12 -> 16] the same convolution is a 4x4 / stride-1 one over 16 channels (K = 256, 57 % real): output
// row ho reads re-blocked rows ho-2 .. ho+1 (pad 2 before, 1 after — the output size is passed explicitly, the gather's bounds
// check supplies the padding).  Weights in that layout come from msc_stem_weight_s2d, the weight gradient goes back through
// msc_stem_wgrad_unpack (csrc/elementwise.hip).
static int stem_geom(GemmParams& p, int B, int Hs, int Ws) { return set_conv_geom(p, B, Hs, Ws, 16, Hs, Ws, 4, 4, 1, 2); }

extern "C" int msc_stem_s2d_conv_fwd(const void* xs, const void* w_s2d, void* y, float* bn_partial, int B, int Hs, int Ws, int Cout,
                                     int dtype, void* stream) {
  MSC_CHECK_ARG(xs && w_s2d && y && B > 0 && Hs > 0 && Ws > 0 && Cout > 0);
  MSC_CHECK_ARG(dtype == MSC_F32 || dtype == MSC_BF16);
  if (!fits32((long)B * Hs * Ws * 16, dtype)) return MSC_ERR_UNSUPPORTED;
  if (msc_stem_fwd_halo_ok(B, Hs, Ws, Cout, dtype)) {      // halo staged once, weights in registers: conv_stem.hip
    note_gemm(8, 224, 64, 256, 1, 0, 0, 0);
    return msc_stem_fwd_halo_launch(xs, w_s2d, y, bn_partial, bn_partial ? (int)msc_conv2d_fwd_stat_rows((int64_t)B * Hs * Ws, Cout, dtype) : 0, B, Hs, num_cus(),
                                    (hipStream_t)stream);
  }
  GemmParams p = base_params();
  p.A = xs; p.B = w_s2d; p.C = y; p.stats = bn_partial;
  p.M = B * Hs * Ws; p.N = Cout; p.K = 256;
  p.ldb = p.K; p.ldc = Cout;
  if (stem_geom(p, B, Hs, Ws) != MSC_OK) return MSC_ERR_BAD_ARG;
  return launch_gemm_dt<OPM_CONV_FWD, OPM_DENSE_K>(p, 1, dtype, (hipStream_t)stream);
}

extern "C" int msc_stem_s2d_conv_wgrad(const void* xs, const void* dy, float* dw_s2d, float* ws, int64_t ws_bytes, int B, int Hs, int Ws,
                                       int Cout, int accumulate, int dtype, void* stream) {
  MSC_CHECK_ARG(xs && dy && dw_s2d && B > 0 && Hs > 0 && Ws > 0 && Cout % vec_of(dtype) == 0);
  MSC_CHECK_ARG(dtype == MSC_F32 || dtype == MSC_BF16);
  if (!fits32((long)B * Hs * Ws * 16, dtype) || !fits32((long)B * Hs * Ws * Cout, dtype)) return MSC_ERR_UNSUPPORTED;
  if (msc_stem_wgrad_halo_ok(B, Hs, Ws, Cout, dtype, ws_bytes) && ws) {     // both operands staged once per tile, accumulators resident: conv_stem.hip
    const int nwg = msc_stem_wgrad_halo_launch(xs, dy, ws, ws_bytes, B, Hs, num_cus(), (hipStream_t)stream);
    if (nwg < 0) return -nwg;
    note_gemm(6, 224, 256, 256, nwg, 0, 0, 0);
    return launch_splitk_reduce(ws, dw_s2d, (long)Cout * 256 / 4, nwg, accumulate, nullptr, nullptr, 0, (hipStream_t)stream);
  }
  GemmParams p = base_params();
  p.A = dy; p.B = xs;
  p.M = Cout; p.N = 256; p.K = B * Hs * Ws;
  p.lda = Cout;
  if (stem_geom(p, B, Hs, Ws) != MSC_OK) return MSC_ERR_BAD_ARG;
  return finish_wgrad(p, dw_s2d, nullptr, ws, ws_bytes, p.K, accumulate, dtype, (hipStream_t)stream,
                      launch_gemm_dt<OPM_DENSE_MN, OPM_CONV_WGRAD>, 64);
}

static int conv2d_dgrad_impl(const void* dy, const void* wt, const void* res, const void* relu_ref, void* dx, const void* bn_x, const float* bn_stats,
                            float* bn_partial, int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, float* ws,
                            int64_t ws_bytes, int dtype, void* stream) {
  MSC_CHECK_ARG(dy && wt && dx && B > 0 && H > 0 && W > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0);
  MSC_CHECK_ARG(dtype == MSC_F32 || dtype == MSC_BF16);
  MSC_CHECK_ARG(Cout >= vec_of(dtype) && ilog2_exact(Cout) >= 0 && Cin > 0 && ilog2_exact(stride) >= 0);
  const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
  MSC_CHECK_ARG(Ho > 0 && Wo > 0);
  if (!fits32((long)B * Ho * Wo * Cout, dtype) || !fits32((long)Cout * KH * KW * Cin, dtype)) return MSC_ERR_UNSUPPORTED;
  if (!bn_x && msc_conv_c64_ok(B, H, W, Cin, Cout, KH, KW, stride, pad, dtype)) {      // the same kernel on the transposed weights, taps negated
    note_gemm(5, 224, 64, 256, 1, 0, 0, relu_ref ? 1 : 0);
    return msc_conv_c64_launch(dy, wt, dx, res, relu_ref, nullptr, 0, B, H, 1, num_cus(), (hipStream_t)stream);
  }
  GemmParams p = base_params();
  p.A = dy; p.B = wt; p.C = dx; p.res = res;
  p.M = B * H * W; p.N = Cin; p.K = KH * KW * Cout;
  p.ldb = p.K; p.ldc = Cin;
  if (set_conv_geom(p, B, H, W, Cout, Ho, Wo, KH, KW, stride, pad) != MSC_OK) return MSC_ERR_BAD_ARG;
  MSC_CHECK_ARG(Cout % kt_of(dtype) == 0);   // a K step never straddles two filter taps
  if (stride > 1 && H % stride == 0 && W % stride == 0) {
    p.cls = 1;
    p.cH2 = H / stride;
    p.cW2 = W / stride;
    p.cMc = B * p.cH2 * p.cW2;
    p.rMc = 1.f / (float)p.cMc; p.rW2 = 1.f / (float)p.cW2; p.rH2 = 1.f / (float)p.cH2;
    // a tile that straddles two parity classes iterates all taps instead of the 1/S^2 its class sees: keep the row tiles
    // inside the classes when a 64-row tile does that and a 128-row one does not (layer 4 at 64 frames: 3136 rows per
    // class — measured 67 -> 32 us)
    if (p.cMc % 128 != 0 && p.cMc % 64 == 0) p.bm_cap = 64;
  }
  if (bn_x && p.cls) return MSC_ERR_UNSUPPORTED;      // the statistics rows follow the NHWC row tiles: stride-1 (and odd-size strided) launches
  return launch_conv_maybe_split<OPM_CONV_DGRAD>(p, dx, res, bn_partial, ws, ws_bytes, dtype, (hipStream_t)stream, relu_ref, bn_x, bn_stats);
}

extern "C" int msc_conv2d_dgrad(const void* dy, const void* wt, const void* res, const void* relu_ref, void* dx, int B, int H, int W,
                                int Cin, int Cout, int KH, int KW, int stride, int pad, float* ws, int64_t ws_bytes, int dtype, void* stream) {
  return conv2d_dgrad_impl(dy, wt, res, relu_ref, dx, nullptr, nullptr, nullptr, B, H, W, Cin, Cout, KH, KW, stride, pad, ws, ws_bytes, dtype, stream);
}

// rows of the BatchNorm-backward partial buffer msc_conv2d_dgrad_bnstat writes for a gradient of `rows` pixels x `cin`
extern "C" int64_t msc_conv2d_dgrad_stat_rows(int64_t rows, int cin, int dtype) {
  return cdiv(rows, tile_for(rows, cin, 1, OPM_CONV_DGRAD, OPM_DENSE_K, dtype).bm);
}

extern "C" int msc_conv2d_dgrad_bnstat(const void* dy, const void* wt, const void* res, void* dx, const void* bn_x, const float* bn_stats,
                                       float* bn_partial, int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                                       float* ws, int64_t ws_bytes, int dtype, void* stream) {
  MSC_CHECK_ARG(bn_x && bn_stats && bn_partial && Cin % 4 == 0);
  return conv2d_dgrad_impl(dy, wt, res, nullptr, dx, bn_x, bn_stats, bn_partial, B, H, W, Cin, Cout, KH, KW, stride, pad, ws, ws_bytes, dtype, stream);
}

extern "C" int msc_conv2d_wgrad(const void* x, const void* dy, float* dw, float* ws, int64_t ws_bytes, int B, int H,
                                int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int accumulate,
                                int dtype, void* stream) {
  MSC_CHECK_ARG(x && dy && dw && B > 0 && H > 0 && W > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0);
  MSC_CHECK_ARG(dtype == MSC_F32 || dtype == MSC_BF16);
  MSC_CHECK_ARG(Cin >= vec_of(dtype) && ilog2_exact(Cin) >= 0 && Cout % vec_of(dtype) == 0);
  const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
  MSC_CHECK_ARG(Ho > 0 && Wo > 0);
  if (!fits32((long)B * H * W * Cin, dtype) || !fits32((long)B * Ho * Wo * Cout, dtype)) return MSC_ERR_UNSUPPORTED;
  if (ws && msc_conv_c64_wgrad_ok(B, H, W, Cin, Cout, KH, KW, stride, pad, dtype, ws_bytes)) {     // ResNet layer 1: conv_c64w.hip
    const int nwg = msc_conv_c64_wgrad_launch(x, dy, ws, ws_bytes, B, H, num_cus(), (hipStream_t)stream);
    if (nwg < 0) return -nwg;
    note_gemm(7, 224, 576, 256, nwg, 0, 0, 0);
    return launch_splitk_reduce(ws, dw, (long)Cout * 576 / 4, nwg, accumulate, nullptr, nullptr, 0, (hipStream_t)stream);
  }
  GemmParams p = base_params();
  p.A = dy; p.B = x;
  p.M = Cout; p.N = KH * KW * Cin; p.K = B * Ho * Wo;
  p.lda = Cout;
  if (set_conv_geom(p, B, H, W, Cin, Ho, Wo, KH, KW, stride, pad) != MSC_OK) return MSC_ERR_BAD_ARG;
  return finish_wgrad(p, dw, nullptr, ws, ws_bytes, p.K, accumulate, dtype, (hipStream_t)stream,
                      launch_gemm_dt<OPM_DENSE_MN, OPM_CONV_WGRAD>, 64);
}

extern "C" int msc_linear_fwd(const void* x, const void* w, const float* bias, const void* res, void* y, void* pre,
                              int M, int N, int K, int act, int dtype, void* stream) {
  MSC_CHECK_ARG(x && w && y && M > 0 && N > 0 && K > 0);
  MSC_CHECK_ARG(dtype == MSC_F32 || dtype == MSC_BF16);
  MSC_CHECK_ARG(K % vec_of(dtype) == 0 && act >= MSC_ACT_NONE && act <= MSC_ACT_GELU_DSAVE);
  if (!fits32((long)M * K, dtype) || !fits32((long)N * K, dtype)) return MSC_ERR_UNSUPPORTED;
  GemmParams p = base_params();
  p.A = x; p.B = w; p.C = y; p.C2 = pre; p.res = res; p.bias = bias; p.act = act;
  p.M = M; p.N = N; p.K = K; p.lda = K; p.ldb = K; p.ldc = N;
  return launch_gemm_dt<OPM_DENSE_K, OPM_DENSE_K>(p, 1, dtype, (hipStream_t)stream);
}

// rows of the column-sum partial buffer msc_linear_dgrad writes for an [M][K] output
extern "C" int64_t msc_linear_dgrad_colsum_rows(int64_t M, int K, int dtype) {
  return cdiv(M, tile_for(M, K, 1, OPM_DENSE_K, OPM_DENSE_K, dtype).bm);
}

extern "C" int msc_linear_dgrad(const void* dy, const void* wt, const void* res, const void* dact_ref, int dact, float* colsum_partial, void* dx, int M,
                                int N, int K, int dtype, void* stream) {
  MSC_CHECK_ARG(dy && wt && dx && M > 0 && N > 0 && K > 0);
  MSC_CHECK_ARG(dtype == MSC_F32 || dtype == MSC_BF16);
  MSC_CHECK_ARG(N % vec_of(dtype) == 0 && dact != MSC_ACT_GELU_DSAVE);
  if (!fits32((long)M * N, dtype) || !fits32((long)N * K, dtype)) return MSC_ERR_UNSUPPORTED;
  GemmParams p = base_params();
  p.A = dy; p.B = wt; p.C = dx; p.res = res; p.dref = dact_ref; p.dact = dact; p.colsum = colsum_partial;
  p.M = M; p.N = K; p.K = N; p.lda = N; p.ldb = N; p.ldc = K;
  if (dact_ref || colsum_partial) return launch_gemm_dt<OPM_DENSE_K, OPM_DENSE_K, 1>(p, 1, dtype, (hipStream_t)stream);
  return launch_gemm_dt<OPM_DENSE_K, OPM_DENSE_K>(p, 1, dtype, (hipStream_t)stream);
}

extern "C" int msc_linear_wgrad(const void* x, const void* dy, float* dw, float* db, float* ws, int64_t ws_bytes, int M, int N,
                                int K, int accumulate, int dtype, void* stream) {
  MSC_CHECK_ARG(x && dy && dw && M > 0 && N > 0 && K > 0);
  MSC_CHECK_ARG(dtype == MSC_F32 || dtype == MSC_BF16);
  MSC_CHECK_ARG(N % vec_of(dtype) == 0 && K % vec_of(dtype) == 0);
  if (!fits32((long)M * N, dtype) || !fits32((long)M * K, dtype)) return MSC_ERR_UNSUPPORTED;
  GemmParams p = base_params();
  p.A = dy; p.B = x;
  p.M = N; p.N = K; p.K = M; p.lda = N; p.ldb = K;
  return finish_wgrad(p, dw, db, ws, ws_bytes, M, accumulate, dtype, (hipStream_t)stream,
                      launch_gemm_dt<OPM_DENSE_MN, OPM_DENSE_MN>, GEMM_NP_KB);
}

extern "C" int msc_gemm_batched(const void* A, const void* Bm, void* C, int M, int N, int K, int64_t lda, int64_t ldb,
                                int64_t ldc, int a_kmajor, int b_kmajor, int batch, int batch_inner, int64_t sa_o,
                                int64_t sa_i, int64_t sb_o, int64_t sb_i, int64_t sc_o, int64_t sc_i, float alpha,
                                int c_f32, int dtype, void* stream) {
  MSC_CHECK_ARG(A && Bm && C && M > 0 && N > 0 && K > 0 && batch > 0 && batch_inner > 0 && batch % batch_inner == 0);
  MSC_CHECK_ARG(dtype == MSC_F32 || dtype == MSC_BF16);
  const int V = vec_of(dtype);
  MSC_CHECK_ARG(lda % V == 0 && ldb % V == 0 && sa_o % V == 0 && sa_i % V == 0 && sb_o % V == 0 && sb_i % V == 0);
  // rows may end inside a 16-byte chunk as long as the padded tail (up to ld) is readable and finite/zero
  if (!fits32((long)(a_kmajor ? M : K) * lda, dtype) || !fits32((long)(b_kmajor ? N : K) * ldb, dtype)) return MSC_ERR_UNSUPPORTED;
  GemmParams p = base_params();
  p.A = A; p.B = Bm; p.C = C;
  p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.batch_inner = batch_inner;
  p.sa_o = sa_o; p.sa_i = sa_i; p.sb_o = sb_o; p.sb_i = sb_i; p.sc_o = sc_o; p.sc_i = sc_i;
  p.alpha = alpha; p.out_f32 = c_f32;
  hipStream_t st = (hipStream_t)stream;
  if (a_kmajor && b_kmajor) return launch_gemm_dt<OPM_DENSE_K, OPM_DENSE_K>(p, batch, dtype, st);
  if (a_kmajor && !b_kmajor) return launch_gemm_dt<OPM_DENSE_K, OPM_DENSE_MN>(p, batch, dtype, st);
  if (!a_kmajor && b_kmajor) return launch_gemm_dt<OPM_DENSE_MN, OPM_DENSE_K>(p, batch, dtype, st);
  return launch_gemm_dt<OPM_DENSE_MN, OPM_DENSE_MN>(p, batch, dtype, st);
}
