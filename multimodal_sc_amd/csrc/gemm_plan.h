// gemm_plan.h — host-side launch planning of the contraction kernel (csrc/gemm.hip): tile choice, split-K plans, workspace
// sizes.  Plain C++ with no HIP dependency ON PURPOSE: tests/host_san/plan_fuzz.cpp builds this header with
// -fsanitize=address,undefined and fuzzes the size queries against the extents the launches index (SURVEY.md §5
// "race detection / sanitizers": host side under ASan/UBSan, device side by the bitwise-determinism tests).
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#if defined(__HIPCC__)
#define MSCPLAN_HD __host__ __device__ __forceinline__
#else
#define MSCPLAN_HD static inline
#endif

namespace mscplan {

// ---- workgroup-id mappings used INSIDE the kernel (speed only, never correctness) — here so the fuzzer can prove them
// bijective.  Workgroups are dealt to the 8 XCDs round-robin in launch order.
// xcd_remap: ids that share an XCD get a contiguous range of tiles (the n-tiles of one activation row block hit one L2).
MSCPLAN_HD int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7, l = bid >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + l;
}
// xcd_split_map: split-K launches with nsplit % 8 == 0 — ALL tiles of one K slice on one XCD.  L = linear id (z * gx + x).
MSCPLAN_HD void xcd_split_map(int L, int gx, int& split, int& bid) {
  const int q = L >> 3;
  split = (L & 7) + 8 * (q / gx);
  bid = q % gx;
}

static inline long cdivl(long a, long b) { return (a + b - 1) / b; }

// reduction elements per K step of the NON-persistent kernels (the only ones that run K-split launches): MSCPLAN_NP_KB bytes
// per operand row per step
#ifndef MSCPLAN_NP_KB
#define MSCPLAN_NP_KB 128
#endif
static inline int kt_of_dtype(int dtype_is_f32) { return MSCPLAN_NP_KB / (dtype_is_f32 ? 4 : 2); }

struct Tile { int bm, bn; };

// MSC_GEMM_TILE=BMxBN (tuning / A-B knob, read per call): force a tile where the operand modes allow it
static inline bool forced_tile(Tile& t) {
  const char* e = getenv("MSC_GEMM_TILE");
  if (!e || !*e) return false;
  int bm = 0, bn = 0;
  if (sscanf(e, "%dx%d", &bm, &bn) != 2) return false;
  if ((bm != 64 && bm != 128 && bm != 256) || (bn != 64 && bn != 128 && bn != 256)) return false;
  if (bn == 256 && bm != 256) return false;
  if (bm == 256 && bn == 64) return false;
  t.bm = bm; t.bn = bn;
  return true;
}

// Tile of one launch: M x N output, zs = batch * K-slices (grid z).  The kernel stages (BM + BN) x KB bytes per K step and
// the LDS fill path of a CU is the measured bound, so MFMA work per staged byte is the lever:
//   64x64: 32 FLOP/B   128x128: 64   256x128: 85   256x256: 128
// allow256: the launch may use the PERSISTENT 256-row kernels (bf16, both operands k-major, zs == 1): one 8-wave workgroup
// per CU walks the tiles with the next tile's first K steps in flight under the current epilogue.  They are taken when the
// tile count fills the CUs evenly enough: `waste` = idle share of the last round of ceil(tiles / CUs) rounds.
// min_wgs: workgroups (tiles x zs) below which the tile shrinks to 64-wide: 200 for the forward / dgrad launches (tuned in the
// step), 500 for the gathered conv weight gradients, whose K slices are capped by the reduction length (sweep: the best tile of
// every ResNet layer is the one that gives ~860 workgroups).
static inline Tile choose_tile(long M, long N, long zs, bool allow256, long ncu = 256, long K = 0, int bm_cap = 256, long min_wgs = 0) {
  Tile t;
  t.bm = M <= 64 ? 64 : 128;
  t.bn = N <= 64 ? 64 : 128;
  if (t.bm > bm_cap) t.bm = bm_cap;
  Tile f;
  if (forced_tile(f)) {
    if ((f.bm < 256 || allow256) && f.bm <= bm_cap) return f;
  }
  static const long T128env = getenv("MSC_GEMM_T128") ? atol(getenv("MSC_GEMM_T128")) : 0;   // A/B knob
  const long T128 = T128env ? T128env : (min_wgs ? min_wgs : 200);                            // below: shrink to 64-wide
  // The persistent 256x256 kernel pays only on LARGE contractions (measured, tools/gemm_sweep.py / gemm_big.py): at this
  // path's short K and 1-4 tiles per CU the per-CU store throughput (~8 B/clk) makes every epilogue as long as the K loop
  // and all workgroups hit it together; several smaller workgroups per CU overlap those phases better.
  if (allow256 && bm_cap >= 256 && K >= 2048 && N >= 512 && cdivl(M, 256) * cdivl(N, 256) >= 2 * ncu) { t.bm = 256; t.bn = 256; return t; }
  if (t.bn == 128 && cdivl(M, t.bm) * cdivl(N, t.bn) * zs < T128) t.bn = 64;
  if (t.bm == 128 && cdivl(M, t.bm) * cdivl(N, t.bn) * zs < T128) t.bm = 64;
  return t;
}

// Split-K for chip-underfilling conv forward / stride-1 dgrad launches (deep ResNet layers: 64 frames x 7x7 or 14x14
// pixels give 100-200 tiles of 128x128 for 256 CUs): 128x128 tiles over ns K slices + a fixed-order finish kernel.
// nk = K steps.  Returns 1 when the launch fills the chip or K is too short to split.  MSC_CONV_SPLITK=0 switches it off
// (read per call so a test can toggle it inside one process).
static inline int plan_conv_split(long M, long N, long nk) {
  const char* e = getenv("MSC_CONV_SPLITK");
  const int on = e ? atoi(e) : 1;
  const long tiles = cdivl(M, 128) * cdivl(N, 128);
  if (!on || tiles >= 200 || nk < 16 || N % 4) return 1;
  long ns = (400 + tiles / 2) / tiles;     // ~400 workgroups (measured: layer3 best at 2 slices, layer4 at 4)
  if (ns > 8) ns = 8;
  while (ns > 1 && nk / ns < 8) --ns;
  return (int)ns;
}

struct ConvSplit { int ns, tps; int64_t ws_bytes; };
// the slices a split conv launch really runs (every slice non-empty) and the fp32 slab bytes it writes
static inline ConvSplit conv_split(long M, long N, long k_len, int dtype_is_f32) {
  const long nk = cdivl(k_len, kt_of_dtype(dtype_is_f32));
  const int ns0 = plan_conv_split(M, N, nk);
  ConvSplit c = {1, (int)nk, 0};
  if (ns0 <= 1) return c;
  c.tps = (int)cdivl(nk, ns0);
  c.ns = (int)cdivl(nk, c.tps);
  c.ws_bytes = (int64_t)ns0 * M * N * 4;     // the query reserves for the planned count (>= the count that runs)
  return c;
}

// split-K of the weight-gradient GEMMs (output rows_out x cols_out, reduction over reduce_len pixels / tokens)
static inline int plan_split(long rows_out, long cols_out, long reduce_len) {
  const long tiles = cdivl(rows_out, 128) * cdivl(cols_out, cols_out <= 64 ? 64 : 128);
  // weight gradients run on a side stream beside the dgrad chain, and share the chip with it; ~640 workgroups measured best
  // (fewer: underfilled when the main stream idles; more: fp32 slab traffic = nsplit x output).  Same-box sweep of the step
  // at the end of round 2 (XCD pinning really applied, fusion as two chains): 384: +0.09 ms, 512: +0.05, 640: 0, 768: +0.07.
  static long target = 0;
  if (!target) {
    const char* e = getenv("MSC_WGRAD_BLOCKS");   // tuning knob (A/B runs)
    target = e ? atol(e) : 640;
    if (target < 1) target = 640;
  }
  long want = (target + tiles - 1) / (tiles > 0 ? tiles : 1);
  // A slice is a SERIAL K loop of one workgroup.  Outputs of one to four tiles (1x1 downsample convolutions: 16 workgroups x 25 steps ran
  // 50 us for 0.8 GFLOP; the stem: 256 workgroups x 98 steps, 150 us, the last kernel of the backward) are latency-bound on that
  // loop and their slabs are small: shorter slices (>= 256 reduction rows) and up to 512 of them.  MSC_WGRAD_SMALL=0: the old limits.
  static const int small_on = getenv("MSC_WGRAD_SMALL") ? atoi(getenv("MSC_WGRAD_SMALL")) : 1;     // A/B knob
  static const long small_max = getenv("MSC_WGRAD_SMALL_MAX") ? atol(getenv("MSC_WGRAD_SMALL_MAX")) : 512;   // A/B knob
  const bool small = small_on && tiles <= 4;
  long cap = reduce_len / (small_on && tiles <= 8 ? 256 : 1024);
  if (want > cap) want = cap;
  if (want < 1) want = 1;
  const long maxw = small ? small_max / tiles : 128;
  if (want > maxw) want = maxw;
  // multiples of 8 let the kernel pin each K slice to one XCD (xcd_split)
  static const int xs = getenv("MSC_WGRAD_XCD_SPLIT") ? atoi(getenv("MSC_WGRAD_XCD_SPLIT")) : 1;   // A/B knob
  if (xs && want >= 6) {
    long w8 = (want + 4) / 8 * 8;
    if (w8 > cap) w8 = cap / 8 * 8;
    if (w8 >= 8) want = w8;
  }
  return (int)want;
}

// fp32 workspace of a weight-gradient launch: `planned` slabs of rows_out x cols_out + `planned` rows of bias-gradient partials
static inline int64_t wgrad_workspace_bytes(long rows_out, long cols_out, long reduce_len) {
  return (int64_t)plan_split(rows_out, cols_out, reduce_len) * rows_out * (cols_out + 1) * 4;
}

struct WgradSplit { int ns, tps, xcd_split; int64_t ws_bytes; };
// kt = reduction elements per K step of the kernel variant that runs the launch
static inline WgradSplit wgrad_split(long rows_out, long cols_out, long reduce_len, int kt, bool gathered = false) {
  const long nk = cdivl(reduce_len, kt);
  const int planned = plan_split(rows_out, cols_out, reduce_len);
  WgradSplit w;
  w.tps = (int)cdivl(nk, planned);
  w.ns = (int)cdivl(nk, w.tps);
  static const int xs = getenv("MSC_WGRAD_XCD_SPLIT") ? atoi(getenv("MSC_WGRAD_XCD_SPLIT")) : 1;
  // the slice count that really runs is cdiv(nk, tps), often planned - 1: lengthen the slices a little (at most 25 %) until
  // it is a multiple of 8 again, or the XCD pinning below silently never applies (layer 1 at 64 frames: 104 planned, 103 ran)
  if (xs && planned >= 8 && w.ns % 8 != 0) {
    for (long t = w.tps + 1; t <= w.tps + w.tps / 4 + 1; ++t) {
      const long n = cdivl(nk, t);
      if (n >= 8 && n % 8 == 0) { w.tps = (int)t; w.ns = (int)n; break; }
    }
  }
  // measured (tools/gemm_bench.py, MSC_WGRAD_XCD_SPLIT=0/1): -8..-17 % where a slice has few tiles or many K steps (layer 1/2, BEV,
  // qkv), +15 % on layer 3's gathered conv weight gradient (36 tiles x 25 steps per slice) — short gathered slices of many tiles
  // stay on the plain mapping; the dense (linear) weight gradients are always pinned (in the step: same time, 2.6 GB less L2-miss traffic)
  const long tiles = cdivl(rows_out, 128) * cdivl(cols_out, cols_out <= 64 ? 64 : 128);
  w.xcd_split = (xs && w.ns >= 8 && w.ns % 8 == 0 && !(gathered && tiles > 32 && w.tps < 32)) ? 1 : 0;
  w.ws_bytes = (int64_t)planned * rows_out * (cols_out + 1) * 4;
  return w;
}

}  // namespace mscplan
