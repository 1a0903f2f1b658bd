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
// Tile 128 x BN (BN = 128 | 64) x 64 bytes of K per step; 256 threads = 4 waves (2x2), each wave owns 64 x BN/2 as
// 32x32 MFMA tiles (v_mfma_f32_32x32x16_bf16; exact-fp32 v_mfma_f32_32x32x2_f32 in parity mode).
//
// Staging (gfx950): operands go HBM -> LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`, 1 KiB per wave-instruction, no
// VGPR round trip).  The per-lane SOURCE offset does the gather (implicit-GEMM taps, transposition-free weight
// gradients) and the XOR bank swizzle (the LDS image itself is lane-linear, as the hardware requires); lanes whose
// element is padding / out of range use an out-of-range buffer offset, which the hardware turns into zeros in LDS.
//   k-major operands : LDS rows of 64 B, chunk c of row r at c ^ ((r>>2)&3)      -> conflict-free ds_read_b128
//   mn-major operands: LDS image [k][row] as it lies in HBM, chunk ^ swizzle(k)  -> ds_read_b64_tr_b16 (hardware
//                      transpose) builds the MFMA fragment; no transposing pass anywhere (bf16).
// Two LDS buffers: the DMA of K-step t+1 is in flight while the MFMAs of step t run; one barrier per K step.
// The exact-fp32 parity mode has no transposed LDS read, so its mn-major operands are staged through registers and
// transposed on the LDS write (correctness path, not tuned).
// Workgroup ids are remapped so that the tiles sharing an activation row-block run on one XCD (shared L2).
//
// Build-authored: the upstream reference has no code (SURVEY.md finding zero); roles per BASELINE.json:
// ImageEncoder/BEV conv stack, FusionTransformer, SemanticEncoder/Decoder, heads.
#include "msc_common.h"
#include <stdlib.h>

enum { OPM_DENSE_K = 0, OPM_DENSE_MN = 1, OPM_CONV_FWD = 2, OPM_CONV_DGRAD = 3, OPM_CONV_WGRAD = 4 };

struct GemmParams {
  const void* A; const void* B; void* C; void* C2; const void* res; const float* bias; float* stats;
  int M, N, K;
  long lda, ldb, ldc;
  int batch_inner; long sa_o, sa_i, sb_o, sb_i, sc_o, sc_i;
  int nsplit, tiles_per_split;
  float alpha; int act; int out_f32;
  // geometry of the gathered NHWC tensor (conv modes)
  int cB, cH, cW, cC, cHo, cWo, cKH, cKW, cS, cP, cLogC, cLogS;
  int cls, cMc, cH2, cW2;   // strided dgrad: parity-class-major row order (rows per class, H/S, W/S)
  int xcd_split;                // split-K launches: all tiles of one K slice on one XCD (see the kernel's index mapping)
  const void* dref; int dact;   // epilogue: multiply by act'(dref[m][n]) (fused activation backward)
  float* colsum;                // epilogue: per-row-block column sums of the stored output, [tile_m][N]
};

typedef __attribute__((ext_vector_type(8))) __bf16 mfma_bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define DMA_OOB 0x80000000u   // >= num_records of every descriptor built here: the DMA writes zeros
#ifndef GEMM_STAGES
#define GEMM_STAGES 3         // LDS buffers of the DMA pipeline (STAGES-1 K steps in flight)
#endif
// GEMM_ABLATE (tools/gemm_ablate.py only, never defined in the product build; results are WRONG by construction):
//   1 no DMA inside the K loop   2 no MFMA (fragments still read)   3 no fragment reads, no MFMA   4 no epilogue stores
#ifndef GEMM_ABLATE
#define GEMM_ABLATE 0
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define KEEP_LIVE(x) asm volatile("" ::"v"(x))
#else
#define KEEP_LIVE(x) (void)(x)
#endif

template <int MODE> struct mode_is_kmajor { static constexpr bool value = (MODE == OPM_DENSE_K || MODE == OPM_CONV_FWD || MODE == OPM_CONV_DGRAD); };

__device__ __forceinline__ int lds_chunk_off(int row, int c) { return row * 64 + ((c ^ ((row >> 2) & 3)) << 4); }
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
__device__ __forceinline__ void dgrad_row_to_pixel(const GemmParams& p, int row, int& b, int& h, int& w) {
  if (p.cls) {
    const int cid = row / p.cMc, idx = row - cid * p.cMc;
    const int w2 = idx % p.cW2, t = idx / p.cW2;
    const int h2 = t % p.cH2;
    b = t / p.cH2;
    h = h2 * p.cS + cid / p.cS;
    w = w2 * p.cS + cid % p.cS;
  } else {
    w = row % p.cW;
    const int t = row / p.cW;
    h = t % p.cH;
    b = t / p.cH;
  }
}

// ------------------------------------------------------------------------------------------------
// Operand loader.  DMA = true: LDS-DMA path (k-major any dtype; mn-major bf16).  DMA = false: registers + transposing
// LDS write (mn-major fp32 only).
template <typename T, int MODE, int ROWS, int NTHR> struct OperandLoader {
  static constexpr bool KM = mode_is_kmajor<MODE>::value;
  static constexpr bool DMA = KM || sizeof(T) == 2;
  static constexpr int VEC = 16 / sizeof(T);
  static constexpr int KT = 64 / sizeof(T);
  static constexpr int NCH = ROWS * 4 / NTHR;                  // 16-byte chunks per thread per K step
  static constexpr int ROWB = KM ? 64 : ROWS * (int)sizeof(T); // LDS row bytes
  static constexpr int CPR = ROWB / 16;                        // chunks per LDS row (DMA) / per reduction row (REG)
  static constexpr int RPI = 1024 / ROWB;                      // LDS rows written by one DMA wave-instruction

  __amdgpu_buffer_rsrc_t rs;
  const T* base;
  long ld;
  uint32_t off[NCH];   // dense modes: running byte offset
  int kk[NCH];         // current k (reduction index) of the chunk
  bool ok[NCH];        // row (k-major) or column (mn-major) inside the matrix
  int a0[NCH], a1[NCH], a2[NCH];  // fwd/dgrad: hi0, wi0, kh ; wgrad: ho, wo, b
  int a3[NCH], a4[NCH];           // fwd/dgrad: kw, cin     ; wgrad: kh, kw
  uint32_t boff[NCH];             // batch base (elements, fwd/dgrad) or cin (wgrad)

  // (row-in-tile, logical chunk) this thread moves for slot i
  static __device__ __forceinline__ void slot(int i, int& r, int& c) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if constexpr (DMA) {
      const int j = wave * NCH + i;                 // DMA instruction index inside the tile
      if constexpr (KM) {
        r = 16 * j + (lane >> 2);
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
        ok[i] = row < rows_total;
        kk[i] = c * VEC;
        off[i] = (uint32_t)(((long)(ok[i] ? row : 0) * ld + kk[i]) * (long)sizeof(T));
      } else if constexpr (MODE == OPM_DENSE_MN) {
        const int col = row0 + c * VEC;
        ok[i] = col < rows_total;
        kk[i] = k0 + r;
        off[i] = (uint32_t)(((long)kk[i] * ld + (ok[i] ? col : 0)) * (long)sizeof(T));
      } else if constexpr (MODE == OPM_CONV_FWD) {
        const int row = row0 + r;
        ok[i] = row < rows_total;
        const int rr = ok[i] ? row : 0;
        const int wo = rr % p.cWo, t = rr / p.cWo;
        const int ho = t % p.cHo, bb = t / p.cHo;
        a0[i] = ho * p.cS - p.cP;
        a1[i] = wo * p.cS - p.cP;
        boff[i] = (uint32_t)bb * (uint32_t)(p.cH * p.cW * p.cC);
        const int k = k0 + c * VEC;
        const int tap = k >> p.cLogC;
        a4[i] = k & (p.cC - 1);
        a2[i] = tap / p.cKW;
        a3[i] = tap - a2[i] * p.cKW;
      } else if constexpr (MODE == OPM_CONV_DGRAD) {
        // rows are INPUT pixels (b,h,w) of size cH x cW; gathered tensor is dy [cB,cHo,cWo,cC]
        const int row = row0 + r;
        ok[i] = row < rows_total;
        int h, w, bb;
        dgrad_row_to_pixel(p, ok[i] ? row : 0, bb, h, w);
        a0[i] = h + p.cP;
        a1[i] = w + p.cP;
        boff[i] = (uint32_t)bb * (uint32_t)(p.cHo * p.cWo * p.cC);
        a4[i] = c * VEC;      // channel offset inside the K step; (kh, kw, co0) come from the schedule
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
        a1[i] = pix % p.cWo;
        const int t = pix / p.cWo;
        a0[i] = t % p.cHo;
        a2[i] = t / p.cHo;
      }
    }
  }

  // byte offset of slot i's chunk for the current K step (DMA_OOB where the element is padding), then advance
  __device__ __forceinline__ uint32_t next_offset(const GemmParams& p, int klimit, int i, const KStep& ks) {
    uint32_t v = DMA_OOB;
    if constexpr (MODE == OPM_DENSE_K) {
      if (ok[i] && ks.kbase + kk[i] < klimit) v = off[i] + (uint32_t)ks.kbase * (uint32_t)sizeof(T);
    } else if constexpr (MODE == OPM_DENSE_MN) {
      if (ok[i] && kk[i] < klimit) v = off[i];
      off[i] += (uint32_t)(KT * ld * (long)sizeof(T));
      kk[i] += KT;
    } else if constexpr (MODE == OPM_CONV_FWD) {
      const int hi = a0[i] + a2[i], wi = a1[i] + a3[i];
      if (ok[i] && a2[i] < p.cKH && (unsigned)hi < (unsigned)p.cH && (unsigned)wi < (unsigned)p.cW)
        v = (boff[i] + ((uint32_t)(hi * p.cW + wi) << p.cLogC) + (uint32_t)a4[i]) * (uint32_t)sizeof(T);
      const int cin = a4[i] + KT;
      a3[i] += cin >> p.cLogC;
      a4[i] = cin & (p.cC - 1);
      while (a3[i] >= p.cKW) { a3[i] -= p.cKW; ++a2[i]; }
    } else if constexpr (MODE == OPM_CONV_DGRAD) {
      const int th = a0[i] - ks.kh, tw = a1[i] - ks.kw;
      const int sm = p.cS - 1;
      const int ho = th >> p.cLogS, wo = tw >> p.cLogS;
      if (ok[i] && ks.kh < p.cKH && th >= 0 && tw >= 0 && !(th & sm) && !(tw & sm) && ho < p.cHo && wo < p.cWo)
        v = (boff[i] + ((uint32_t)(ho * p.cWo + wo) << p.cLogC) + (uint32_t)(ks.co0 + a4[i])) * (uint32_t)sizeof(T);
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

  // DMA path: enqueue this thread's chunks of the current K step into `tile` (LDS) and advance
  __device__ __forceinline__ void issue(const GemmParams& p, int klimit, char* tile, const KStep& ks) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const uint32_t v = next_offset(p, klimit, i, ks);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_PTR(tile + (wave * NCH + i) * 1024), 16, v, 0, 0, 0);
    }
  }

  // register path (mn-major fp32): load now, transpose into LDS later
  __device__ __forceinline__ void load(const GemmParams& p, int klimit, uint4 (&regs)[NCH], const KStep& ks) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      const uint32_t v = next_offset(p, klimit, i, ks);
      regs[i] = (v == DMA_OOB) ? make_uint4(0, 0, 0, 0)
                               : *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(base) + v);
    }
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
        *reinterpret_cast<uint32_t*>(tile + lds_chunk_off(row, kb >> 4) + (kb & 15)) = w[j];
      }
    }
  }
};

// XCD-aware bijective remap of the linear workgroup id (8 XCDs, round-robin dispatch): ids that share an XCD get a
// contiguous range of tiles, so the n-tiles of one activation row-block hit one L2.  Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7, l = bid >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + l;
}

// ------------------------------------------------------------------------------------------------
// EPX: extended epilogue (fused activation backward + column sums) — compiled only into the kernels that use it so the
// common epilogue keeps its register budget.
template <typename T, int AMODE, int BMODE, int BM, int BN, bool EPX = false>
__global__ __launch_bounds__(BM == 256 ? 512 : 256, BM == 256 ? 4 : 2) void gemm_kernel(const GemmParams p) {
  constexpr int NTHR = BM == 256 ? 512 : 256;   // 8 waves (4x2) for the 256-row tile, 4 waves (2x2) otherwise
  constexpr int KT = 64 / sizeof(T);
  constexpr int NWM = BM == 256 ? 4 : 2;        // waves along M
  constexpr int WM = BM / NWM, WN = BN / 2, MI = WM / 32, NI = WN / 32;
  constexpr int A_BYTES = BM * 64, B_BYTES = BN * 64;
  constexpr int EP_BYTES = 64 * BN * 4;     // epilogue slab: 64 rows x BN fp32
  using LA = OperandLoader<T, AMODE, BM, NTHR>;
  using LB = OperandLoader<T, BMODE, BN, NTHR>;
  constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
  constexpr int STAGES = (LA::DMA && LB::DMA) ? GEMM_STAGES : 2;
  constexpr int SMEM_BYTES = STAGES * STAGE_BYTES > EP_BYTES ? STAGES * STAGE_BYTES : EP_BYTES;
  __shared__ __attribute__((aligned(1024))) char smem[SMEM_BYTES];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntn = (p.N + BN - 1) / BN;
  // Workgroups are dealt to the 8 XCDs round-robin in launch order (x fastest, then z).  Plain launches: tiles sharing an
  // activation row block on one XCD (xcd_remap).  Split-K launches with nsplit % 8 == 0 (weight gradients): ALL tiles of
  // one K slice on one XCD — they stream the same rows of both operands at the same time, so the slice is fetched into
  // that L2 once instead of once per tile (PMC: weight-gradient HBM/fabric traffic was ~4x the algorithmic bytes).
  int bid, zz, split;
  if (p.xcd_split) {
    const int L = blockIdx.z * gridDim.x + blockIdx.x;
    const int q = L >> 3;
    split = (L & 7) + 8 * (q / (int)gridDim.x);
    bid = q % (int)gridDim.x;
    zz = 0;
  } else {
    bid = xcd_remap(blockIdx.x, gridDim.x);
    zz = blockIdx.z / p.nsplit;
    split = blockIdx.z % p.nsplit;
  }
  const int tile_m = bid / ntn, tile_n = bid % ntn;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int zo = zz / p.batch_inner, zi = zz % p.batch_inner;

  const T* Ab = reinterpret_cast<const T*>(p.A) + zo * p.sa_o + zi * p.sa_i;
  const T* Bb = reinterpret_cast<const T*>(p.B) + zo * p.sb_o + zi * p.sb_i;

  const int nk_total = (p.K + KT - 1) / KT;
  const int kt_begin = split * p.tiles_per_split;
  const int kt_end = min(nk_total, kt_begin + p.tiles_per_split);

  // ---- K schedule (wave-uniform).  Default: steps kt_begin..kt_end-1, kbase = kt*KT.  Strided dgrad whose tile lies
  // inside one parity class: only the taps that class can see, (kh0, kh0+S, ..) x (kw0, kw0+S, ..).
  KStep ks = {kt_begin * KT, 0, 0, 0};
  int nk = kt_end - kt_begin;
  int s_kw0 = 0, s_step = 1;
  if constexpr (AMODE == OPM_CONV_DGRAD) {
    int kh0 = 0;
    if (p.cls) {
      const int last = min(m0 + BM, p.M) - 1;
      const int c0 = m0 / p.cMc;
      if (c0 == last / p.cMc) {
        kh0 = (c0 / p.cS + p.cP) % p.cS;
        s_kw0 = (c0 % p.cS + p.cP) % p.cS;
        s_step = p.cS;
      }
    }
    const int cnt_h = kh0 < p.cKH ? (p.cKH - kh0 + s_step - 1) / s_step : 0;
    const int cnt_w = s_kw0 < p.cKW ? (p.cKW - s_kw0 + s_step - 1) / s_step : 0;
    nk = cnt_h * cnt_w * (p.cC / KT);
    ks.kh = kh0;
    ks.kw = s_kw0;
    ks.kbase = (kh0 * p.cKW + s_kw0) << p.cLogC;
    if (p.nsplit > 1) {   // split-K dgrad (the host only splits the plain (b,h,w) order): this slice starts mid-schedule
      const int e = kt_begin * KT, tap = e >> p.cLogC;
      ks.co0 = e & (p.cC - 1);
      ks.kh = tap / p.cKW;
      ks.kw = tap - ks.kh * p.cKW;
      ks.kbase = e;
      nk = kt_end - kt_begin;
    }
  }
  auto advance = [&]() {
    if constexpr (AMODE == OPM_CONV_DGRAD) {
      ks.co0 += KT;
      if (ks.co0 >= p.cC) {
        ks.co0 = 0;
        ks.kw += s_step;
        if (ks.kw >= p.cKW) { ks.kw = s_kw0; ks.kh += s_step; }
      }
      ks.kbase = ((ks.kh * p.cKW + ks.kw) << p.cLogC) + ks.co0;
    } else {
      ks.kbase += KT;
    }
  };

  LA la;
  LB lb;
  la.init(p, Ab, p.lda, p.M, m0, kt_begin * KT);
  lb.init(p, Bb, p.ldb, p.N, n0, kt_begin * KT);

  f32x16_t acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  uint4 ra[LA::NCH], rb[LB::NCH];

  // Software pipeline: STAGES LDS buffers, STAGES-1 K steps of LDS-DMA in flight.  Every wave waits for its own DMA
  // instructions of the step it is about to read with a COUNTED vmcnt (the younger steps stay in flight across the
  // barrier), then a raw s_barrier makes every wave's part visible (cdna_hip_programming.md "Pipelining across barriers").
  constexpr int NPS = LA::NCH + LB::NCH;              // DMA wave-instructions per K step
  constexpr int WAIT_N = NPS * (STAGES - 2);          // instructions of the younger in-flight steps
  int issued = 0;
#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s) {
    if (issued < nk) {
      char* buf = smem + s * STAGE_BYTES;
      if constexpr (LA::DMA) la.issue(p, p.K, buf, ks); else { la.load(p, p.K, ra, ks); la.store(buf, ra); }
      if constexpr (LB::DMA) lb.issue(p, p.K, buf + A_BYTES, ks); else { lb.load(p, p.K, rb, ks); lb.store(buf + A_BYTES, rb); }
      advance();
      ++issued;
    }
  }
  if (issued == STAGES - 1 && STAGES > 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAIT_N) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  const int lr = lane & 31, lh = lane >> 5;
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    const bool more = issued < nk;
    int nb = cur + (STAGES - 1);
    if (nb >= STAGES) nb -= STAGES;
    char* nxt = smem + nb * STAGE_BYTES;
    if (more) {
      if constexpr (GEMM_ABLATE != 1) {
        if constexpr (LA::DMA) la.issue(p, p.K, nxt, ks); else la.load(p, p.K, ra, ks);
        if constexpr (LB::DMA) lb.issue(p, p.K, nxt + A_BYTES, ks); else lb.load(p, p.K, rb, ks);
      }
      advance();
      ++issued;
    }
    const char* As = smem + cur * STAGE_BYTES;
    const char* Bs = As + A_BYTES;
    if constexpr (GEMM_ABLATE == 3) {
    } else if constexpr (sizeof(T) == 2) {
      // lane geometry of the transposed read (mn-major images): 16-lane group g, row q / column quad pq inside it
      const int tg = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        mfma_bf16x8_t fa[MI], fb[NI];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          if constexpr (LA::KM) {
            fa[i] = *reinterpret_cast<const mfma_bf16x8_t*>(As + lds_chunk_off(wm * WM + i * 32 + lr, 2 * ks + lh));
          } else {
            const int col = wm * WM + i * 32 + 16 * (tg & 1) + 4 * tp;
            const int k0r = 16 * ks + 8 * (tg >> 1) + tq;
            s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(
                As + k0r * LA::ROWB + (((col >> 3) ^ mn_swz<LA::ROWB>(k0r)) << 4) + (tp & 1) * 8));
            s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(
                As + (k0r + 4) * LA::ROWB + (((col >> 3) ^ mn_swz<LA::ROWB>(k0r + 4)) << 4) + (tp & 1) * 8));
            typedef __attribute__((ext_vector_type(8))) short s16x8_t;
            s16x8_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            fa[i] = __builtin_bit_cast(mfma_bf16x8_t, v);
          }
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          if constexpr (LB::KM) {
            fb[j] = *reinterpret_cast<const mfma_bf16x8_t*>(Bs + lds_chunk_off(wn * WN + j * 32 + lr, 2 * ks + lh));
          } else {
            const int col = wn * WN + j * 32 + 16 * (tg & 1) + 4 * tp;
            const int k0r = 16 * ks + 8 * (tg >> 1) + tq;
            s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(
                Bs + k0r * LB::ROWB + (((col >> 3) ^ mn_swz<LB::ROWB>(k0r)) << 4) + (tp & 1) * 8));
            s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(
                Bs + (k0r + 4) * LB::ROWB + (((col >> 3) ^ mn_swz<LB::ROWB>(k0r + 4)) << 4) + (tp & 1) * 8));
            typedef __attribute__((ext_vector_type(8))) short s16x8_t;
            s16x8_t v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            fb[j] = __builtin_bit_cast(mfma_bf16x8_t, v);
          }
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j)
            if constexpr (GEMM_ABLATE == 2) { KEEP_LIVE(fa[i]); KEEP_LIVE(fb[j]); }
            else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const int k = 2 * ks + lh;  // float index inside the 16-float row
        float fa[MI], fb[NI];
#pragma unroll
        for (int i = 0; i < MI; ++i)
          fa[i] = *reinterpret_cast<const float*>(As + lds_chunk_off(wm * WM + i * 32 + lr, k >> 2) + (k & 3) * 4);
#pragma unroll
        for (int j = 0; j < NI; ++j)
          fb[j] = *reinterpret_cast<const float*>(Bs + lds_chunk_off(wn * WN + j * 32 + lr, k >> 2) + (k & 3) * 4);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
          for (int j = 0; j < NI; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fb[j], acc[i][j], 0, 0, 0);
      }
    }
    if (more) {
      if constexpr (!LA::DMA) la.store(nxt, ra);
      if constexpr (!LB::DMA) lb.store(nxt + A_BYTES, rb);
    }
    // step kt+1 must have landed (this wave's part); steps kt+2.. may stay in flight when the pipe is full
    if (STAGES > 2 && issued - (kt + 2) == STAGES - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAIT_N) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // my LDS reads of this buffer are done before it is refilled
    __builtin_amdgcn_s_barrier();                        // ... and every other wave's DMA part is visible
    asm volatile("" ::: "memory");                       // keep the next step's LDS traffic below the barrier
    cur = cur + 1 == STAGES ? 0 : cur + 1;
  }
  __syncthreads();
  if constexpr (GEMM_ABLATE == 4) {   // keep the accumulators live, store nothing
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) KEEP_LIVE(acc[i][j]);
    if (p.M > 0) return;
  }

  // ---------------------------------------------------------------- epilogue
  // (1) BatchNorm partial statistics straight from the fp32 accumulators (conv forward only)
  if (p.stats) {
    float* red = reinterpret_cast<float*>(smem);  // [NWM][2 stat][BN]
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
    __syncthreads();
    if (tid < BN && n0 + tid < p.N) {
      float* st = p.stats + (long)tile_m * 2 * p.N;
      float a1 = 0.f, a2 = 0.f;
#pragma unroll
      for (int w = 0; w < NWM; ++w) { a1 += red[(w * 2 + 0) * BN + tid]; a2 += red[(w * 2 + 1) * BN + tid]; }
      st[n0 + tid] = a1;
      st[p.N + n0 + tid] = a2;
    }
    __syncthreads();
  }

  // (2) accumulators -> LDS slab (fp32, 64 rows per pass) -> 16-byte coalesced global stores with the fused
  //     bias / pre-activation copy / activation / residual applied on whole vectors
  float* slab = reinterpret_cast<float*>(smem);
  const bool f32out = (p.nsplit > 1 || p.out_f32);
  char* Cbase = f32out ? reinterpret_cast<char*>(reinterpret_cast<float*>(p.C) + (p.nsplit > 1 ? (long)split * p.M * p.ldc : (zo * p.sc_o + zi * p.sc_i)))
                       : reinterpret_cast<char*>(reinterpret_cast<T*>(p.C) + zo * p.sc_o + zi * p.sc_i);
  constexpr int PASSES = BM / 64;
  float csum[16 / sizeof(T)];   // this thread's column chunk is the same in every iteration (NTHR % (BN/VEC) == 0)
#pragma unroll
  for (int e = 0; e < (int)(16 / sizeof(T)); ++e) csum[e] = 0.f;
#pragma unroll
  for (int ps = 0; ps < PASSES; ++ps) {
    const bool mine = (PASSES == 1) || (wm == ps);
    if (mine) {
      const int rbase = (PASSES == 1) ? wm * WM : 0;
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            slab[(rbase + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * BN + wn * WN + j * 32 + lr] = p.alpha * acc[i][j][r];
    }
    __syncthreads();
    if (f32out) {
      constexpr int CPRW = BN / 4;
#pragma unroll
      for (int it = 0; it < 64 * CPRW / NTHR; ++it) {
        const int id = tid + it * NTHR;
        const int row = id / CPRW, cc = id % CPRW;
        const int m = m0 + ps * 64 + row, n = n0 + cc * 4;
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
      for (int it = 0; it < 64 * CPRW / NTHR; ++it) {
        const int id = tid + it * NTHR;
        const int row = id / CPRW, cc = id % CPRW;
        const int m = m0 + ps * 64 + row, n = n0 + cc * VEC;
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
          if (C2) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) vo.v[e] = v[e];
            vo.store(C2 + o);
          }
          if (p.act != MSC_ACT_NONE) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) v[e] = act_apply(v[e], p.act);
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
          }
#pragma unroll
          for (int e = 0; e < VEC; ++e) vo.v[e] = v[e];
          vo.store(dst);
        } else {
          for (int e = 0; e < VEC && n + e < p.N; ++e) {
            float x = v[e] + (p.bias ? p.bias[n + e] : 0.f);
            if (C2) msc_traits<T>::st(C2 + o + e, x);
            x = act_apply(x, p.act);
            if (Rs) x += msc_traits<T>::ld(Rs + o + e);
            if constexpr (EPX) {
              if (p.dref) x *= act_grad(msc_traits<T>::ld(reinterpret_cast<const T*>(p.dref) + o + e), p.dact);
              if (p.colsum) csum[e] += x;
            }
            msc_traits<T>::st(dst + e, x);
          }
        }
      }
    }
    if (ps + 1 < PASSES) __syncthreads();
  }
  if (EPX && p.colsum && !f32out) {
    // fixed-order reduction of the per-thread column sums over the NTHR/(BN/VEC) threads that share a column chunk
    constexpr int VEC = 16 / sizeof(T);
    constexpr int CPRW = BN / VEC, NR = NTHR / CPRW;
    __syncthreads();
    const int cc = tid % CPRW, rr = tid / CPRW;
#pragma unroll
    for (int e = 0; e < VEC; ++e) slab[rr * BN + cc * VEC + e] = csum[e];
    __syncthreads();
    if (tid < BN && n0 + tid < p.N) {
      float a = 0.f;
      for (int r = 0; r < NR; ++r) a += slab[r * BN + tid];
      p.colsum[(long)tile_m * p.N + n0 + tid] = a;
    }
  }
}

// dw[i] = (accumulate ? dw[i] : 0) + sum_s slab[s][i]   (fixed order: bitwise reproducible)
// Block = 64 float4 columns x RY slab groups: small outputs with many slabs (early conv layers: 36 K elements x ~100
// slabs) would otherwise walk the slabs serially from a handful of threads.  Group y sums slabs y, y+RY, ...; the groups
// are combined through LDS in the order 0..RY-1.
template <int RY>
__global__ __launch_bounds__(64 * RY) void splitk_reduce_kernel(const float* __restrict__ slab, float* __restrict__ out, long n4, int nsplit, int accumulate) {
  __shared__ float4 part[RY][64];
  const long i = (long)blockIdx.x * 64 + threadIdx.x;
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
                                                            const T* __restrict__ res, float* __restrict__ stats, int RB) {
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
      s1[0] += v[0]; s1[1] += v[1]; s1[2] += v[2]; s1[3] += v[3];
      s2[0] += v[0] * v[0]; s2[1] += v[1] * v[1]; s2[2] += v[2] * v[2]; s2[3] += v[3] * v[3];
      if (res) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += msc_traits<T>::ld(res + e + j);
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

// Split-K for chip-underfilling conv forward / stride-1 dgrad launches (the deep ResNet layers: 64 frames x 7x7 or 14x14
// pixels give 100-200 tiles of 128x128 for 256 CUs, and 64-wide tiles run 4 MFMAs per barrier): 128x128 tiles over ns K
// slices + the finish kernel above.  Returns 1 when the launch fills the chip or K is too short to split.
static int plan_conv_split(long M, long N, long nk) {
  const char* e = getenv("MSC_CONV_SPLITK");      // A/B + test knob, read per call so a test can switch it inside one process
  const int on = e ? atoi(e) : 1;
  const long tiles = (long)cdiv(M, 128) * cdiv(N, 128);
  if (!on || tiles >= 200 || nk < 32 || N % 4) return 1;
  long ns = (400 + tiles / 2) / tiles;     // ~400 workgroups (measured: layer3 best at 2 slices, layer4 at 4)
  if (ns > 8) ns = 8;
  while (ns > 1 && nk / ns < 16) --ns;
  return (int)ns;
}

extern "C" int64_t msc_conv2d_splitk_workspace_bytes(int64_t rows, int cout, int64_t k_len, int dtype) {
  const int ns = plan_conv_split(rows, cout, cdiv(k_len, dtype == MSC_F32 ? 16 : 32));
  return ns > 1 ? (int64_t)ns * rows * cout * 4 : 0;
}

// ------------------------------------------------------------------------------------------------
// tile choice: narrow operands get 64-wide tiles; launches that would leave most of the 256 CUs idle trade MFMA work
// per LDS byte for more workgroups
static void choose_tile(long M, long N, long zs, int& bm, int& bn, bool allow256 = true) {
  bm = M <= 64 ? 64 : 128;
  bn = N <= 64 ? 64 : 128;
  // Measured (tools/gemm_ceiling.py): the kernel runs at the same rate whether its operands stream from HBM or sit in
  // cache — the bound is the per-CU LDS fill path (LDS-DMA, ~55 GB/s per CU), so FLOPs per staged byte is the lever:
  // 256x128 tiles (8 waves) raise it from 64 to 85 and are used when they still fill the chip.  A 256x256 tile (128
  // FLOP/B, one workgroup per CU) was built and measured: slower at this path's short K (16-72 steps), where the
  // un-overlapped prologue/epilogue of a lone workgroup costs more than the fill it saves.
  // 256-row tiles only for launches of >= 1000 of them: measured inside the three-stream step, 128x128 tiles co-schedule
  // better with the side-stream weight gradients for everything smaller (step 9.07 -> 8.99 ms), although the 256-row
  // tile wins some of those shapes when they run alone.
  if (allow256 && bn == 128 && M >= 1024 && (long)cdiv(M, 256) * cdiv(N, 128) * zs >= 1000) { bm = 256; return; }
  // (threshold measured inside the step: 200 tiles; 320+ shrinks tiles that were better left at 128)
  if (bn == 128 && (long)cdiv(M, bm) * cdiv(N, bn) * zs < 200) bn = 64;
  if (bm == 128 && (long)cdiv(M, bm) * cdiv(N, bn) * zs < 200) bm = 64;
}

// rows of the BatchNorm partial-statistics buffer msc_conv2d_fwd writes for an output of `rows` pixels x `cout`
extern "C" int64_t msc_conv2d_fwd_stat_rows(int64_t rows, int cout) {
  int bm, bn;
  choose_tile(rows, cout, 1, bm, bn);
  return cdiv(rows, bm);
}

template <typename T, int AMODE, int BMODE, bool EPX = false>
static int launch_gemm(const GemmParams& p, int batch, hipStream_t st) {
  int bm, bn;
  const long zs = (long)batch * p.nsplit;
  choose_tile(p.M, p.N, zs, bm, bn, mode_is_kmajor<AMODE>::value && !EPX);   // 256-row tiles: k-major A, plain epilogue
  dim3 grid((unsigned)(cdiv(p.M, bm) * cdiv(p.N, bn)), 1, (unsigned)zs);
  if (bm == 256) hipLaunchKernelGGL((gemm_kernel<T, AMODE, BMODE, 256, 128, EPX>), grid, dim3(512), 0, st, p);
  else if (bm == 128 && bn == 128) hipLaunchKernelGGL((gemm_kernel<T, AMODE, BMODE, 128, 128, EPX>), grid, dim3(256), 0, st, p);
  else if (bm == 128) hipLaunchKernelGGL((gemm_kernel<T, AMODE, BMODE, 128, 64, EPX>), grid, dim3(256), 0, st, p);
  else if (bn == 128) hipLaunchKernelGGL((gemm_kernel<T, AMODE, BMODE, 64, 128, EPX>), grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL((gemm_kernel<T, AMODE, BMODE, 64, 64, EPX>), grid, dim3(256), 0, st, p);
  return msc_launch_status();
}

template <int AMODE, int BMODE, bool EPX = false>
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
static inline int kt_of(int dtype) { return dtype == MSC_F32 ? 16 : 32; }
static inline int esize(int dtype) { return dtype == MSC_F32 ? 4 : 2; }
// every operand is addressed with 32-bit byte offsets below DMA_OOB
static inline bool fits32(long elems, int dtype) { return elems * esize(dtype) < 0x7ffffff0L; }

// split-K plan shared by the workspace query and the launchers (independent of dtype)
static int plan_split(long rows_out, long cols_out, long reduce_len) {
  const long tiles = (long)cdiv(rows_out, 128) * cdiv(cols_out, cols_out <= 64 ? 64 : 128);
  // weight gradients run on a side stream beside the dgrad chain, and share the chip with it; ~512 workgroups measured best
  // (fewer: underfilled when the main stream idles; more: fp32 slab traffic = nsplit x output)
  static long target = 0;
  if (!target) {
    const char* e = getenv("MSC_WGRAD_BLOCKS");   // tuning knob (A/B runs); default 512 (measured best of 128..1536)
    target = e ? atol(e) : 512;
    if (target < 1) target = 512;
  }
  long want = (target + tiles - 1) / (tiles > 0 ? tiles : 1);
  long cap = reduce_len / 1024;
  if (want > cap) want = cap;
  if (want < 1) want = 1;
  if (want > 128) want = 128;
  // multiples of 8 let the kernel pin each K slice to one XCD (xcd_split)
  static const int xs = getenv("MSC_WGRAD_XCD_SPLIT") ? atoi(getenv("MSC_WGRAD_XCD_SPLIT")) : 1;   // A/B knob
  if (xs && want >= 6) {
    long w8 = (want + 4) / 8 * 8;
    if (w8 > cap) w8 = cap / 8 * 8;
    if (w8 >= 8) want = w8;
  }
  return (int)want;
}

extern "C" int64_t msc_wgrad_workspace_bytes(int64_t rows_out, int64_t cols_out, int64_t reduce_len) {
  return (int64_t)plan_split(rows_out, cols_out, reduce_len) * rows_out * cols_out * 4;
}

static int finish_wgrad(GemmParams& p, float* dw, float* ws, int64_t ws_bytes, long reduce_len, int accumulate,
                        int dtype, hipStream_t st, int (*launch)(const GemmParams&, int, int, hipStream_t)) {
  const int KT = kt_of(dtype);
  const int nk = cdiv(reduce_len, KT);
  int ns = plan_split(p.M, p.N, reduce_len);
  int tps = cdiv(nk, ns);
  ns = cdiv(nk, tps);
  if ((int64_t)ns * p.M * p.N * 4 > ws_bytes || !ws) return MSC_ERR_BAD_ARG;
  if (((long)p.M * p.N) % 4) return MSC_ERR_BAD_ARG;
  p.nsplit = ns;
  p.tiles_per_split = tps;
  static const int xs = getenv("MSC_WGRAD_XCD_SPLIT") ? atoi(getenv("MSC_WGRAD_XCD_SPLIT")) : 1;
  p.xcd_split = (xs && ns >= 8 && ns % 8 == 0) ? 1 : 0;
  p.C = ws;
  p.ldc = p.N;
  p.out_f32 = 1;
  if (ns == 1 && !accumulate) {   // no slab, no reduce pass: the GEMM epilogue writes dw itself
    p.C = dw;
    return launch(p, 1, dtype, st);
  }
  int rc = launch(p, 1, dtype, st);
  if (rc != MSC_OK) return rc;
  const long n4 = (long)p.M * p.N / 4;
  const unsigned blocks = (unsigned)cdiv(n4, 64);
  // slab groups per block: as many as keep >= ~4 waves per CU busy, at most the slab count
  int ry = 1;
  while (ry < 16 && ry * 2 <= ns && (long)blocks * ry < 4096) ry *= 2;
  if (ry == 1) hipLaunchKernelGGL(splitk_reduce_kernel<1>, dim3(blocks), dim3(64, 1), 0, st, ws, dw, n4, ns, accumulate);
  else if (ry == 2) hipLaunchKernelGGL(splitk_reduce_kernel<2>, dim3(blocks), dim3(64, 2), 0, st, ws, dw, n4, ns, accumulate);
  else if (ry == 4) hipLaunchKernelGGL(splitk_reduce_kernel<4>, dim3(blocks), dim3(64, 4), 0, st, ws, dw, n4, ns, accumulate);
  else if (ry == 8) hipLaunchKernelGGL(splitk_reduce_kernel<8>, dim3(blocks), dim3(64, 8), 0, st, ws, dw, n4, ns, accumulate);
  else hipLaunchKernelGGL(splitk_reduce_kernel<16>, dim3(blocks), dim3(64, 16), 0, st, ws, dw, n4, ns, accumulate);
  return msc_launch_status();
}

static int set_conv_geom(GemmParams& p, int B, int H, int W, int C, int Ho, int Wo, int KH, int KW, int s, int pad) {
  p.cB = B; p.cH = H; p.cW = W; p.cC = C; p.cHo = Ho; p.cWo = Wo; p.cKH = KH; p.cKW = KW; p.cS = s; p.cP = pad;
  p.cLogC = ilog2_exact(C);
  p.cLogS = ilog2_exact(s);
  return (p.cLogC >= 0 && p.cLogS >= 0) ? MSC_OK : MSC_ERR_BAD_ARG;
}

// shared tail of msc_conv2d_fwd / msc_conv2d_dgrad: split-K launch + finish when it pays and the workspace allows
template <int AMODE>
static int launch_conv_maybe_split(GemmParams& p, void* out, const void* res, float* stats, float* ws, int64_t ws_bytes, int dtype,
                                   hipStream_t st) {
  const int nk = cdiv(p.K, kt_of(dtype));
  const int ns0 = (AMODE == OPM_CONV_DGRAD && p.cls) ? 1 : plan_conv_split(p.M, p.N, nk);
  if (ns0 <= 1 || !ws || ws_bytes < (int64_t)ns0 * p.M * p.N * 4) return launch_gemm_dt<AMODE, OPM_DENSE_K>(p, 1, dtype, st);
  const int tps = cdiv(nk, ns0), ns = cdiv(nk, tps);
  int bm0, bn0;
  choose_tile(p.M, p.N, 1, bm0, bn0);          // row granularity of the BatchNorm partials the caller sized its buffer for
  p.nsplit = ns;
  p.tiles_per_split = tps;
  p.C = ws; p.ldc = p.N; p.out_f32 = 1;
  p.stats = nullptr; p.res = nullptr;
  int rc = launch_gemm_dt<AMODE, OPM_DENSE_K>(p, 1, dtype, st);
  if (rc != MSC_OK) return rc;
  const dim3 grid((unsigned)cdiv(p.N, 64), (unsigned)cdiv(p.M, bm0));
  if (dtype == MSC_F32)
    hipLaunchKernelGGL(splitk_finish_kernel<float>, grid, dim3(256), 0, st, ws, ns, (long)p.M, p.N, (float*)out, (const float*)res, stats, bm0);
  else
    hipLaunchKernelGGL(splitk_finish_kernel<bf16_t>, grid, dim3(256), 0, st, ws, ns, (long)p.M, p.N, (bf16_t*)out, (const bf16_t*)res, stats, bm0);
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
  GemmParams p = base_params();
  p.A = x; p.B = w; p.C = y; p.stats = bn_partial;
  p.M = B * Ho * Wo; p.N = Cout; p.K = KH * KW * Cin;
  p.ldb = p.K; p.ldc = Cout;
  if (set_conv_geom(p, B, H, W, Cin, Ho, Wo, KH, KW, stride, pad) != MSC_OK) return MSC_ERR_BAD_ARG;
  return launch_conv_maybe_split<OPM_CONV_FWD>(p, y, nullptr, bn_partial, ws, ws_bytes, dtype, (hipStream_t)stream);
}

extern "C" int msc_conv2d_dgrad(const void* dy, const void* wt, const void* res, void* dx, int B, int H, int W, int Cin,
                                int Cout, int KH, int KW, int stride, int pad, float* ws, int64_t ws_bytes, int dtype, void* stream) {
  MSC_CHECK_ARG(dy && wt && dx && B > 0 && H > 0 && W > 0 && KH > 0 && KW > 0 && stride > 0 && pad >= 0);
  MSC_CHECK_ARG(dtype == MSC_F32 || dtype == MSC_BF16);
  MSC_CHECK_ARG(Cout >= vec_of(dtype) && ilog2_exact(Cout) >= 0 && Cin > 0 && ilog2_exact(stride) >= 0);
  const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
  MSC_CHECK_ARG(Ho > 0 && Wo > 0);
  if (!fits32((long)B * Ho * Wo * Cout, dtype) || !fits32((long)Cout * KH * KW * Cin, dtype)) return MSC_ERR_UNSUPPORTED;
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
  }
  return launch_conv_maybe_split<OPM_CONV_DGRAD>(p, dx, res, nullptr, ws, ws_bytes, dtype, (hipStream_t)stream);
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
  GemmParams p = base_params();
  p.A = dy; p.B = x;
  p.M = Cout; p.N = KH * KW * Cin; p.K = B * Ho * Wo;
  p.lda = Cout;
  if (set_conv_geom(p, B, H, W, Cin, Ho, Wo, KH, KW, stride, pad) != MSC_OK) return MSC_ERR_BAD_ARG;
  return finish_wgrad(p, dw, ws, ws_bytes, p.K, accumulate, dtype, (hipStream_t)stream,
                      launch_gemm_dt<OPM_DENSE_MN, OPM_CONV_WGRAD>);
}

extern "C" int msc_linear_fwd(const void* x, const void* w, const float* bias, const void* res, void* y, void* pre,
                              int M, int N, int K, int act, int dtype, void* stream) {
  MSC_CHECK_ARG(x && w && y && M > 0 && N > 0 && K > 0);
  MSC_CHECK_ARG(dtype == MSC_F32 || dtype == MSC_BF16);
  MSC_CHECK_ARG(K % vec_of(dtype) == 0);
  if (!fits32((long)M * K, dtype) || !fits32((long)N * K, dtype)) return MSC_ERR_UNSUPPORTED;
  GemmParams p = base_params();
  p.A = x; p.B = w; p.C = y; p.C2 = pre; p.res = res; p.bias = bias; p.act = act;
  p.M = M; p.N = N; p.K = K; p.lda = K; p.ldb = K; p.ldc = N;
  return launch_gemm_dt<OPM_DENSE_K, OPM_DENSE_K>(p, 1, dtype, (hipStream_t)stream);
}

// rows of the column-sum partial buffer msc_linear_dgrad writes for an [M][K] output
extern "C" int64_t msc_linear_dgrad_colsum_rows(int64_t M, int K) {
  int bm, bn;
  choose_tile(M, K, 1, bm, bn, false);   // the extended-epilogue kernels never use the 256-row tile
  return cdiv(M, bm);
}

extern "C" int msc_linear_dgrad(const void* dy, const void* wt, const void* dact_ref, int dact, float* colsum_partial, void* dx, int M,
                                int N, int K, int dtype, void* stream) {
  MSC_CHECK_ARG(dy && wt && dx && M > 0 && N > 0 && K > 0);
  MSC_CHECK_ARG(dtype == MSC_F32 || dtype == MSC_BF16);
  MSC_CHECK_ARG(N % vec_of(dtype) == 0);
  if (!fits32((long)M * N, dtype) || !fits32((long)N * K, dtype)) return MSC_ERR_UNSUPPORTED;
  GemmParams p = base_params();
  p.A = dy; p.B = wt; p.C = dx; p.dref = dact_ref; p.dact = dact; p.colsum = colsum_partial;
  p.M = M; p.N = K; p.K = N; p.lda = N; p.ldb = N; p.ldc = K;
  if (dact_ref || colsum_partial) return launch_gemm_dt<OPM_DENSE_K, OPM_DENSE_K, true>(p, 1, dtype, (hipStream_t)stream);
  return launch_gemm_dt<OPM_DENSE_K, OPM_DENSE_K>(p, 1, dtype, (hipStream_t)stream);
}

extern "C" int msc_linear_wgrad(const void* x, const void* dy, float* dw, float* ws, int64_t ws_bytes, int M, int N,
                                int K, int accumulate, int dtype, void* stream) {
  MSC_CHECK_ARG(x && dy && dw && M > 0 && N > 0 && K > 0);
  MSC_CHECK_ARG(dtype == MSC_F32 || dtype == MSC_BF16);
  MSC_CHECK_ARG(N % vec_of(dtype) == 0 && K % vec_of(dtype) == 0);
  if (!fits32((long)M * N, dtype) || !fits32((long)M * K, dtype)) return MSC_ERR_UNSUPPORTED;
  GemmParams p = base_params();
  p.A = dy; p.B = x;
  p.M = N; p.N = K; p.K = M; p.lda = N; p.ldb = K;
  return finish_wgrad(p, dw, ws, ws_bytes, M, accumulate, dtype, (hipStream_t)stream,
                      launch_gemm_dt<OPM_DENSE_MN, OPM_DENSE_MN>);
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
