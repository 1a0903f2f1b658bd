// plan_fuzz.cpp — host-side sanitizer leg: builds csrc/gemm_plan.h (the launch planning of the contraction
// kernel: tile choice, split-K plans, workspace-size queries, workgroup-id mappings) with -fsanitize=address,undefined and
// fuzzes it against the extents the launches really index.  Every buffer the kernels would write is allocated at the size
// the QUERY reports and touched at every extreme index the LAUNCH plan produces, so an under-sized query is an ASan report,
// and any signed overflow / bad shift in the planning arithmetic is a UBSan report.  CPU only; no GPU code runs here.
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>
#include <set>
#include <utility>
#include "../../multimodal_sc_amd/csrc/gemm_plan.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }
static long pick(long lo, long hi) { return lo + (long)(rnd() % (uint64_t)(hi - lo + 1)); }
#define CHECK(c) do { if (!(c)) { fprintf(stderr, "CHECK failed: %s (line %d) M=%ld N=%ld K=%ld\n", #c, __LINE__, M, N, K); return 1; } } while (0)

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  long n_split = 0, n_xcd = 0, n256 = 0;
  for (int it = 0; it < iters; ++it) {
    // shapes around the path's real ones (rows up to ~1M, channels 8..2048, K up to 8192) plus degenerate small ones
    long M = (rnd() & 3) == 0 ? pick(1, 300) : pick(1, 1 << (int)pick(4, 20));
    long N = (rnd() & 1) ? (8L << pick(0, 8)) : pick(1, 2100);
    long K = (rnd() & 1) ? (8L << pick(0, 10)) : pick(1, 9000);
    const int f32 = (int)(rnd() & 1);
    const int KT = mscplan::kt_of_dtype(f32);
    CHECK(KT * (f32 ? 4 : 2) == MSCPLAN_NP_KB);
    // ---- tile choice for every operand-mode class and a few grid-z values
    for (int allow = 0; allow < 2; ++allow) {
      const long zs = allow ? 1 : pick(1, 64);
      const long ncu = (rnd() & 1) ? 256 : pick(8, 320);
      const int cap = (rnd() & 3) == 0 ? 64 : 256;
      mscplan::Tile t = mscplan::choose_tile(M, N, zs, allow, ncu, K, cap);
      CHECK(t.bm <= cap);
      CHECK(t.bm == 64 || t.bm == 128 || t.bm == 256);
      CHECK(t.bn == 64 || t.bn == 128 || t.bn == 256);
      CHECK(t.bm < 256 || allow);                    // 256-row tiles: persistent kernels only
      CHECK(t.bn < 256 || t.bm == 256);              // 256 columns: 256x256 only
      CHECK(mscplan::cdivl(M, t.bm) * mscplan::cdivl(N, t.bn) <= 0x7fffffffL);
      n256 += t.bm == 256;
    }
    // ---- conv forward: BatchNorm partial rows the caller allocates vs the rows the epilogue / finish kernel write
    {
      const mscplan::Tile t = mscplan::choose_tile(M, N, 1, false);
      const long stat_rows = mscplan::cdivl(M, t.bm);                 // msc_conv2d_fwd_stat_rows
      if (stat_rows * 2 * N < (1L << 22)) {
        std::vector<float> stats((size_t)(stat_rows * 2 * N));
        const long tiles_m = mscplan::cdivl(M, t.bm);                 // un-split epilogue: st = stats + tile_m*2N
        stats[(size_t)((tiles_m - 1) * 2 * N + N + (N - 1))] = 1.f;
        stats[0] = 1.f;
        const long fin_rows = mscplan::cdivl(M, t.bm);                // split finish: grid.y = cdiv(M, bm0), blockIdx.y*2N
        stats[(size_t)((fin_rows - 1) * 2 * N + N + (N - 1))] = 2.f;
      }
      const mscplan::ConvSplit c = mscplan::conv_split(M, N, K, f32);
      const long nk = mscplan::cdivl(K, KT);
      CHECK(c.ns >= 1);
      if (c.ns > 1) {
        ++n_split;
        CHECK((long)c.ns * c.tps >= nk);                 // the slices cover K
        CHECK((long)(c.ns - 1) * c.tps < nk);            // ... and the last one is not empty
        CHECK(c.ws_bytes >= (int64_t)c.ns * M * N * 4);  // the queried workspace holds every slab the launch writes
        CHECK(N % 4 == 0);                               // finish kernel reads float4
        if (c.ws_bytes < (1 << 24)) {
          std::vector<float> ws((size_t)(c.ws_bytes / 4));
          for (int sidx = 0; sidx < c.ns; ++sidx) ws[(size_t)((long)sidx * M * N + (M - 1) * N + (N - 1))] = 1.f;   // slab corner
        }
      } else {
        CHECK(c.ws_bytes == 0);
      }
    }
    // ---- weight gradients: output M2 x N2 (small), reduction over R rows
    {
      const long M2 = 8L << pick(0, 6), N2 = pick(1, 600) * 8, R = pick(1, 1 << 20);
      const int kt2 = (rnd() & 1) ? KT : KT / 2;
      const mscplan::WgradSplit w = mscplan::wgrad_split(M2, N2, R, kt2);
      const long nk = mscplan::cdivl(R, kt2);
      CHECK(w.ns >= 1 && w.ns <= 520);      // (128, or 512 / tiles for outputs of at most four tiles)
      CHECK((long)w.ns * w.tps >= nk && (long)(w.ns - 1) * w.tps < nk);
      CHECK(w.ws_bytes >= (int64_t)w.ns * M2 * (N2 + 1) * 4);                         // weight slabs + bias-gradient rows
      CHECK(mscplan::wgrad_workspace_bytes(M2, N2, R) == w.ws_bytes);                 // msc_wgrad_workspace_bytes
      if (w.ws_bytes < (1 << 24)) {            // every slab corner and the last bias-partial row the launch writes
        std::vector<float> ws((size_t)(w.ws_bytes / 4));
        for (int sidx = 0; sidx < w.ns; ++sidx) ws[(size_t)((long)sidx * M2 * N2 + (M2 - 1) * N2 + (N2 - 1))] = 1.f;
        ws[(size_t)((long)w.ns * M2 * N2 + (long)(w.ns - 1) * M2 + (M2 - 1))] = 2.f;
      }
      if (w.xcd_split) {
        ++n_xcd;
        CHECK(w.ns % 8 == 0);
        const mscplan::Tile t = mscplan::choose_tile(M2, N2, w.ns, false);
        const int gx = (int)(mscplan::cdivl(M2, t.bm) * mscplan::cdivl(N2, t.bn));
        if ((long)gx * w.ns < 50000) {       // (split, tile) must be hit exactly once by the kernel's id mapping
          std::set<std::pair<int, int>> seen;
          for (int L = 0; L < gx * w.ns; ++L) {
            int split, bid;
            mscplan::xcd_split_map(L, gx, split, bid);
            CHECK(split >= 0 && split < w.ns && bid >= 0 && bid < gx);
            CHECK(seen.insert({split, bid}).second);
          }
        }
      }
    }
    // ---- xcd_remap is a bijection of [0, nwg) for every grid size
    {
      const int nwg = (int)pick(1, 3000);
      std::vector<char> hit((size_t)nwg, 0);
      for (int b = 0; b < nwg; ++b) {
        const int v = mscplan::xcd_remap(b, nwg);
        CHECK(v >= 0 && v < nwg && !hit[(size_t)v]);
        hit[(size_t)v] = 1;
      }
    }
  }
  printf("plan_fuzz: %d iterations ok (split conv plans %ld, xcd-pinned wgrad plans %ld, 256-row tiles %ld)\n", iters, n_split, n_xcd, n256);
  return 0;
}
