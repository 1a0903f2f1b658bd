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

// Tuning / A-B knobs of the launch planner.  Read ONCE, when the library is loaded (the environment variable of the same name in
// upper case with an MSC_ prefix gives the initial value); a process that wants to toggle one between launches (tests, sweeps) calls
// msc_set_knob(name, value).  No launch path reads the environment.
struct Knobs {
  int tile_bm = 0, tile_bn = 0;      // gemm_tile: force a tile (value = bm * 1000 + bn; 0 = planner's choice)          MSC_GEMM_TILE=BMxBN
  long t128 = 0;                     // gemm_t128: workgroup count below which a 128-wide tile shrinks to 64 (0 = default)
  int conv_splitk = 1;               // conv_splitk: split-K for chip-underfilling conv forward / stride-1 dgrad launches
  long wgrad_blocks = 512;           // wgrad_blocks: workgroup target of the weight-gradient K split
  int wgrad_small = 1;               // wgrad_small: short K slices for outputs of 1-8 tiles
  long wgrad_small_max = 512;        // wgrad_small_max
  int wgrad_xcd_split = 1;           // wgrad_xcd_split: pin every K slice of a weight gradient to one XCD
  int dgrad_cls_interleave = 1;      // dgrad_cls_interleave: parity classes of a strided dgrad interleaved in launch order
  int gemm_pp = 1;                   // gemm_pp: dense bf16 linears on the persistent ping-pong kernel (gemm_pp.hip); 256/192/160/128: tile rows forced
  int gemm_pp_k = 0;                 // gemm_pp_k: (bisecting aid) only contractions of this reduction length take the ping-pong kernel; 0 = any
  int c64_wgs = 0;                   // c64_wgs: workgroups of the layer-1 forward / data-gradient kernel (0: fewest that keep the round count of one per CU)
  int stem_fwd_halo = 1;             // stem_fwd_halo: the stem's forward convolution on the halo-resident kernel of conv_stem.hip (0: implicit GEMM)
  int c64w_wgs = 0;                  // c64w_wgs: workgroups of the layer-1 weight-gradient kernel (0: half the CUs; -1: the gathered split-K GEMM instead)
  int stem_wgrad_halo = 1;           // stem_wgrad_halo: the stem's weight gradient on the halo-resident kernel of conv_stem.hip (0: gathered split-K GEMM)
  int dgrad_cls_fill = 1;            // dgrad_cls_fill: 1x1 strided data gradients launch the tap-carrying parity class only (its workgroups fill the other classes' rows)
  int conv_c64 = 1;                  // conv_c64: 3x3 / stride 1 / 64 -> 64 channel convolutions on 56-wide maps (ResNet layer 1) on the halo-resident kernel of conv_c64.hip (0: implicit GEMM)
  int gemm_skinny = 1;               // gemm_skinny: dense bf16 linears of at most 64 rows on the staging-free kernel of gemm_skinny.hip (0: tiled family)
  int gemm_pp_mask = 15;             // gemm_pp_mask: which epilogue kinds may take it: 1 plain, 2 residual, 4 GELU + derivative, 8 dgrad x derivative
  int att_bwd_waves = 0, att_dq_waves = 0;   // attention backward wave counts (0 = default)
  int bn_un = 1;                     // bn_un: rows per thread in flight in the BatchNorm-backward reduction (1 or 4)
};
static inline long knob_env(const char* name, long dflt) {
  const char* e = getenv(name);
  return (e && *e) ? atol(e) : dflt;
}
static inline Knobs knobs_from_env() {
  Knobs k;
  if (const char* e = getenv("MSC_GEMM_TILE")) {
    int bm = 0, bn = 0;
    if (sscanf(e, "%dx%d", &bm, &bn) == 2) { k.tile_bm = bm; k.tile_bn = bn; }
  }
  k.t128 = knob_env("MSC_GEMM_T128", 0);
  k.conv_splitk = (int)knob_env("MSC_CONV_SPLITK", 1);
  k.wgrad_blocks = knob_env("MSC_WGRAD_BLOCKS", 512);
  if (k.wgrad_blocks < 1) k.wgrad_blocks = 512;
  k.wgrad_small = (int)knob_env("MSC_WGRAD_SMALL", 1);
  k.wgrad_small_max = knob_env("MSC_WGRAD_SMALL_MAX", 512);
  k.wgrad_xcd_split = (int)knob_env("MSC_WGRAD_XCD_SPLIT", 1);
  k.dgrad_cls_interleave = (int)knob_env("MSC_DGRAD_CLS_INTERLEAVE", 1);
  k.gemm_pp = (int)knob_env("MSC_GEMM_PP", 1);
  k.gemm_pp_mask = (int)knob_env("MSC_GEMM_PP_MASK", 15);
  k.gemm_skinny = (int)knob_env("MSC_GEMM_SKINNY", 1);
  k.conv_c64 = (int)knob_env("MSC_CONV_C64", 1);
  k.dgrad_cls_fill = (int)knob_env("MSC_DGRAD_CLS_FILL", 1);
  k.stem_wgrad_halo = (int)knob_env("MSC_STEM_WGRAD_HALO", 1);
  k.c64w_wgs = (int)knob_env("MSC_C64W_WGS", 0);
  k.stem_fwd_halo = (int)knob_env("MSC_STEM_FWD_HALO", 1);
  k.c64_wgs = (int)knob_env("MSC_C64_WGS", 0);
  k.gemm_pp_k = (int)knob_env("MSC_GEMM_PP_K", 0);
  k.att_bwd_waves = (int)knob_env("MSC_ATT_BWD_WAVES", 0);
  k.att_dq_waves = (int)knob_env("MSC_ATT_DQ_WAVES", 0);
  k.bn_un = (int)knob_env("MSC_BN_UN", 1);
  return k;
}
inline Knobs g_knobs = knobs_from_env();
// name -> field (msc_set_knob / msc_get_knob); nullptr: unknown name.  gemm_tile is handled by the callers (two fields).
static inline long* knob_slot(const char* name, int*& as_int) {
  as_int = nullptr;
  Knobs& k = g_knobs;
  if (!strcmp(name, "gemm_t128")) return &k.t128;
  if (!strcmp(name, "wgrad_blocks")) return &k.wgrad_blocks;
  if (!strcmp(name, "wgrad_small_max")) return &k.wgrad_small_max;
  if (!strcmp(name, "conv_splitk")) as_int = &k.conv_splitk;
  else if (!strcmp(name, "wgrad_small")) as_int = &k.wgrad_small;
  else if (!strcmp(name, "wgrad_xcd_split")) as_int = &k.wgrad_xcd_split;
  else if (!strcmp(name, "dgrad_cls_interleave")) as_int = &k.dgrad_cls_interleave;
  else if (!strcmp(name, "gemm_pp")) as_int = &k.gemm_pp;
  else if (!strcmp(name, "gemm_pp_mask")) as_int = &k.gemm_pp_mask;
  else if (!strcmp(name, "gemm_skinny")) as_int = &k.gemm_skinny;
  else if (!strcmp(name, "conv_c64")) as_int = &k.conv_c64;
  else if (!strcmp(name, "dgrad_cls_fill")) as_int = &k.dgrad_cls_fill;
  else if (!strcmp(name, "stem_wgrad_halo")) as_int = &k.stem_wgrad_halo;
  else if (!strcmp(name, "c64w_wgs")) as_int = &k.c64w_wgs;
  else if (!strcmp(name, "stem_fwd_halo")) as_int = &k.stem_fwd_halo;
  else if (!strcmp(name, "c64_wgs")) as_int = &k.c64_wgs;
  else if (!strcmp(name, "gemm_pp_k")) as_int = &k.gemm_pp_k;
  else if (!strcmp(name, "att_bwd_waves")) as_int = &k.att_bwd_waves;
  else if (!strcmp(name, "att_dq_waves")) as_int = &k.att_dq_waves;
  else if (!strcmp(name, "bn_un")) as_int = &k.bn_un;
  return nullptr;
}
static inline bool knob_set(const char* name, long v) {
  if (!name) return false;
  if (!strcmp(name, "gemm_tile")) { g_knobs.tile_bm = (int)(v / 1000); g_knobs.tile_bn = (int)(v % 1000); return true; }
  int* ip = nullptr;
  if (long* lp = knob_slot(name, ip)) { *lp = v; return true; }
  if (ip) { *ip = (int)v; return true; }
  return false;
}
static inline bool knob_get(const char* name, long& v) {
  if (!name) return false;
  if (!strcmp(name, "gemm_tile")) { v = (long)g_knobs.tile_bm * 1000 + g_knobs.tile_bn; return true; }
  int* ip = nullptr;
  if (long* lp = knob_slot(name, ip)) { v = *lp; return true; }
  if (ip) { v = *ip; return true; }
  return false;
}

// gemm_tile knob: force a tile where the operand modes allow it
static inline bool forced_tile(Tile& t) {
  const int bm = g_knobs.tile_bm, bn = g_knobs.tile_bn;
  if (!bm && !bn) return false;
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
  const long T128 = g_knobs.t128 ? g_knobs.t128 : (min_wgs ? min_wgs : 200);                  // below: shrink to 64-wide
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
// nk = K steps.  Returns 1 when the launch fills the chip or K is too short to split.  Knob conv_splitk = 0 switches it off.
static inline int plan_conv_split(long M, long N, long nk) {
  const int on = g_knobs.conv_splitk;
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
  // weight gradients run on a side stream beside the dgrad chain, and share the chip with it (fewer workgroups: underfilled when the
  // main stream idles; more: fp32 slab traffic = nsplit x output).  Same-box sweep of the step at the end of round 2 (XCD pinning really
  // applied, fusion as two chains): 384: +0.09 ms, 512: +0.05, 640: 0, 768: +0.07.  End of round 4 (fusion as one chain on the ping-pong
  // kernel, halo kernels for layer 1 and the stem): 512 beats 640 in every interleaved pair — 7.130 / 7.126 / 7.152 against 7.149 / 7.165 /
  // 7.182 ms at 64 frames, 448 and 576 in between or worse, 32 frames -0.01 ms (profiles/r4_wgrad_blocks_512.txt).
  const long target = g_knobs.wgrad_blocks > 0 ? g_knobs.wgrad_blocks : 512;
  long want = (target + tiles - 1) / (tiles > 0 ? tiles : 1);
  // A slice is a SERIAL K loop of one workgroup.  Outputs of one to four tiles (1x1 downsample convolutions: 16 workgroups x 25 steps ran
  // 50 us for 0.8 GFLOP; the stem: 256 workgroups x 98 steps, 150 us, the last kernel of the backward) are latency-bound on that
  // loop and their slabs are small: shorter slices (>= 256 reduction rows) and up to 512 of them.  Knob wgrad_small = 0: the old limits.
  const int small_on = g_knobs.wgrad_small;
  const long small_max = g_knobs.wgrad_small_max;
  const bool small = small_on && tiles <= 4;
  long cap = reduce_len / (small_on && tiles <= 8 ? 256 : 1024);
  if (want > cap) want = cap;
  if (want < 1) want = 1;
  const long maxw = small ? small_max / tiles : 128;
  if (want > maxw) want = maxw;
  // multiples of 8 let the kernel pin each K slice to one XCD (xcd_split)
  const int xs = g_knobs.wgrad_xcd_split;
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
  const int xs = g_knobs.wgrad_xcd_split;
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
