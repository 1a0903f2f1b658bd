// conv_c64w.hip — weight gradient of the 3x3 / stride 1 / pad 1, 64 -> 64 channel convolutions on 56-wide maps (ResNet-18 layer 1), gfx950.
// Build-authored (the upstream reference has no code).
//
// dW[co][kh][kw][ci] = sum over pixels of dy[p][co] * x[p + (kh - 1, kw - 1)][ci]: a 64 x 576 output with a reduction over 200 704 pixels (64
// frames).  As a gathered split-K GEMM (gemm.hip) it is 52 us per launch, four launches per step on the weight-gradient stream.  Same scheme as
// the stem's weight gradient (conv_stem.hip), on the tile geometry of the layer's forward kernel (conv_c64.hip):
//   * persistent 4-wave workgroups; tile = 4 image rows x 56 pixels = 224 pixels = 7 k-steps of 32; per tile the dy rows (28 KB, contiguous) and
//     the 6 x 58-pixel halo of x (44 KB, zero padding by the DMA's bounds check) are staged ONCE by dense LDS-DMA, double-buffered;
//   * the reduction index is the pixel: both operands are read from the row-major images by ds_read_b64_tr_b16 (K-slot (lane group g, element e)
//     <-> pixel 16 (e >> 2) + 4 g + (e & 3) of the k-step); both images swap the four 32-byte channel groups of a pixel by
//     (pixel ^ (pixel >> 2)) & 3 (applied on the DMA's source side), which spreads the 8 pixel rows a 32-lane half reads over the banks;
//   * wave w owns input channels 16 w .. 16 w + 15: 9 taps x 4 output-channel fragments = 36 accumulators of 16 x 16 (144 registers) that stay in
//     registers across all the workgroup's tiles; per k-step 8 + 18 transposed reads feed 36 MFMAs;
//   * each workgroup stores one 64 x 576 fp32 partial at the end; the split-K reduce of gemm.hip sums them in workgroup order.
#include "msc_common.h"
#include "gemm_plan.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 cw_bf16x8;
typedef __attribute__((ext_vector_type(4))) float cw_f32x4;
typedef __attribute__((ext_vector_type(4))) short cw_s16x4;

#define CW_OOB 0x80000000u
#define CW_LDS(p) ((__attribute__((address_space(3))) void*)(p))

constexpr int CW_W = 56, CW_R = 4;
constexpr int CW_PIX = CW_W * CW_R;                       // 224 pixels = 7 k-steps of 32
constexpr int CW_KS = CW_PIX / 32;
constexpr int CW_HW = CW_W + 2, CW_HR = CW_R + 2;         // halo 58 x 6
constexpr int CW_HPIX = CW_HW * CW_HR;                    // 348
constexpr int CW_NH = (CW_HPIX * 128 + 1023) / 1024;      // 44 DMA instructions (8 halo pixels each)
constexpr int CW_ND = CW_PIX * 128 / 1024;                // 28
constexpr int CW_HALO_BYTES = CW_NH * 1024, CW_DY_BYTES = CW_ND * 1024;
constexpr int CW_BUF_BYTES = CW_HALO_BYTES + CW_DY_BYTES; // 73728

struct CWParams {
  const void* x; const void* dy; float* slab;             // slab: [gridDim.x][64][576] fp32
  int B, H;
};

__device__ __forceinline__ int cw_key(int q) { return (q ^ (q >> 2)) & 3; }

__global__ __launch_bounds__(256, 1) void conv_c64_wgrad_kernel(const CWParams p) {
  __shared__ __attribute__((aligned(1024))) char lds[2 * CW_BUF_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // = input-channel group
  const int g = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3;
  const int tiles_per_img = p.H / CW_R;
  const int ntiles = p.B * tiles_per_img;

  // ---- DMA side (x descriptor base = x - (1 row + 1 pixel): halo pixel (hr, hc) of the tile at image row r0 sits at ((b H + r0) W + hr W + hc) * 128)
  const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(p.x) - (CW_W + 1) * 128), 0, (int)0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, (int)0x7ffffff0, 0x00020000);
  constexpr int NHW = CW_NH / 4, NDW = CW_ND / 4;         // 11 and 7 instructions per wave
  uint32_t hrel[NHW], drel[NDW];
  uint32_t htop = 0, hbot = 0;                            // bit i: instruction i's pixel lies in the halo's first / last row
#pragma unroll
  for (int i = 0; i < NHW; ++i) {
    const int s = 64 * (wave + 4 * i) + lane;             // 16-byte slot: halo pixel s >> 3, position s & 7
    const int P = s >> 3, pos = s & 7, hr = P / CW_HW, hc = P - hr * CW_HW;
    const int chunk = ((((pos >> 1) ^ cw_key(P)) << 1) | (pos & 1));
    const bool ok = P < CW_HPIX && hc >= 1 && hc <= CW_W;
    hrel[i] = ok ? (uint32_t)((hr * CW_W + hc) * 128 + chunk * 16) : CW_OOB;
    if (hr == 0) htop |= 1u << i;
    if (hr == CW_HR - 1) hbot |= 1u << i;
  }
#pragma unroll
  for (int i = 0; i < NDW; ++i) {
    const int s = 64 * (wave + 4 * i) + lane;
    const int q = s >> 3, pos = s & 7;
    const int chunk = ((((pos >> 1) ^ cw_key(q)) << 1) | (pos & 1));
    drel[i] = (uint32_t)(q * 128 + chunk * 16);
  }
  auto stage = [&](int tile, int buf) __attribute__((always_inline)) {
    const int b = tile / tiles_per_img, r0 = (tile - b * tiles_per_img) * CW_R;
    const uint32_t pix0 = (uint32_t)((b * p.H + r0) * CW_W);
    const bool top = r0 == 0, bot = r0 + CW_R == p.H;
    char* base = lds + buf * CW_BUF_BYTES;
#pragma unroll
    for (int i = 0; i < NHW; ++i) {
      const bool dead = (top && ((htop >> i) & 1)) || (bot && ((hbot >> i) & 1));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, CW_LDS(base + (wave + 4 * i) * 1024), 16, dead ? CW_OOB : hrel[i], pix0 * 128u, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NDW; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsd, CW_LDS(base + CW_HALO_BYTES + (wave + 4 * i) * 1024), 16, drel[i], pix0 * 128u, 0, 0);
  };

  // ---- fragment addresses (tile-invariant): block blk of k-step ks covers tile pixels q = 32 ks + 16 blk + 4 g + qq; the lane supplies the address
  // of pixel row qq of its group's block, 8 bytes at channel 4 pp of the 16-channel group
  int qpix[CW_KS][2], hpix[CW_KS][2];
#pragma unroll
  for (int ks = 0; ks < CW_KS; ++ks)
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      const int q = 32 * ks + 16 * blk + 4 * g + qq;
      const int r = q / CW_W, c = q - r * CW_W;
      qpix[ks][blk] = q;
      hpix[ks][blk] = (r + 1) * CW_HW + c + 1;            // halo pixel of the centre tap
    }

  cw_f32x4 acc[4][9];                                     // [output-channel fragment j][tap]
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[j][t] = cw_f32x4{0.f, 0.f, 0.f, 0.f};

  int tile = (int)blockIdx.x;
  if (tile < ntiles) stage(tile, 0);
  int buf = 0;
  for (; tile < ntiles; tile += (int)gridDim.x, buf ^= 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this tile's images have landed
    __syncthreads();                                      // ... for every wave; and nobody still reads the other buffer
    if (tile + (int)gridDim.x < ntiles) stage(tile + (int)gridDim.x, buf ^ 1);
    const char* himg = lds + buf * CW_BUF_BYTES;
    const char* dimg = himg + CW_HALO_BYTES;
    // Transposed reads as inline assembly: through the builtin the compiler cannot prove the reads independent of the LDS-DMA writes of the
    // NEXT tile issued just above and waits for them (s_waitcnt vmcnt(0)) before the first read — staging and MFMAs of a tile then run one
    // after the other (the finding of csrc/gemm.hip ds_read_tr_asm).  The reads' own completion is waited for by hand: the operand of tap
    // t + 1 is in flight under the MFMAs of tap t (lgkmcnt(2) = all but the two youngest reads), registers tied to the wait.
    auto tr_issue = [&](cw_s16x4& lo, cw_s16x4& hi, const char* a0, const char* a1) __attribute__((always_inline)) {
      const uint32_t u0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)a0;
      const uint32_t u1 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)a1;
      asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(u0));
      asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"(u1));
    };
    auto join = [](const cw_s16x4& lo, const cw_s16x4& hi) __attribute__((always_inline)) {
      return __builtin_bit_cast(cw_bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
#pragma unroll
    for (int ks = 0; ks < CW_KS; ++ks) {
      cw_s16x4 alo[4], ahi[4], blo[2], bhi[2];
      const int q0 = qpix[ks][0], q1 = qpix[ks][1];
      auto issue_b = [&](int t) __attribute__((always_inline)) {
        const int toff = (t / 3 - 1) * CW_HW + (t % 3 - 1);
        const int P0 = hpix[ks][0] + toff, P1 = hpix[ks][1] + toff;
        tr_issue(blo[t & 1], bhi[t & 1], himg + P0 * 128 + ((wave ^ cw_key(P0)) << 5) + 8 * pp, himg + P1 * 128 + ((wave ^ cw_key(P1)) << 5) + 8 * pp);
      };
#pragma unroll
      for (int j = 0; j < 4; ++j)
        tr_issue(alo[j], ahi[j], dimg + q0 * 128 + ((j ^ cw_key(q0)) << 5) + 8 * pp, dimg + q1 * 128 + ((j ^ cw_key(q1)) << 5) + 8 * pp);
      issue_b(0);
      cw_bf16x8 af[4];
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        if (t < 8) {
          issue_b(t + 1);
          asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        asm volatile("" : "+v"(blo[t & 1]), "+v"(bhi[t & 1]));
        if (t == 0) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            asm volatile("" : "+v"(alo[j]), "+v"(ahi[j]));
            af[j] = join(alo[j], ahi[j]);
          }
        }
        const cw_bf16x8 bf = join(blo[t & 1], bhi[t & 1]);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[j], bf, acc[j][t], 0, 0, 0);
      }
    }
  }

  // ---- the workgroup's partial: accumulator [j][t] register r of lane l = dW[co = 16 j + 4 (l >> 4) + r][tap t][ci = 16 wave + (l & 15)]
  float* out = p.slab + (long)blockIdx.x * 64 * 576;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) out[(16 * j + 4 * g + r) * 576 + t * 64 + 16 * wave + (lane & 15)] = acc[j][t][r];
}

}  // namespace

bool msc_conv_c64_wgrad_ok(int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int dtype, int64_t ws_bytes) {
  if (!mscplan::g_knobs.conv_c64 || mscplan::g_knobs.c64w_wgs < 0 || mscplan::g_knobs.tile_bm) return false;
  if (dtype != MSC_BF16 || Cin != 64 || Cout != 64 || KH != 3 || KW != 3 || stride != 1 || pad != 1) return false;
  if (W != CW_W || H % CW_R || H < CW_R || B < 1) return false;
  if ((long)B * H * W * 128 >= 0x7ffffff0L - (CW_W + 1) * 128) return false;
  return ws_bytes >= (int64_t)64 * 576 * 4;               // room for at least one partial
}

// workspace that lets every CU hold a partial (the query of the gathered GEMM's split plan gives fewer slabs)
int64_t msc_conv_c64_wgrad_workspace_bytes(int ncu) { return (int64_t)ncu * 64 * 576 * 4; }

// launches the kernel with as many workgroups as CUs, tiles and slab room allow; returns that count (< 0: error code)
int msc_conv_c64_wgrad_launch(const void* x, const void* dy, float* slab, int64_t slab_bytes, int B, int H, int ncu, hipStream_t st) {
  if ((((uintptr_t)x | (uintptr_t)dy | (uintptr_t)slab) & 15) != 0) return -MSC_ERR_BAD_ARG;
  const int ntiles = B * (H / CW_R);
  // Workgroups: HALF the CUs by default.  The launch runs on the weight-gradient stream beside the critical chain; persistent one-per-CU
  // workgroups that own a CU for the whole launch (147 KB of LDS, all registers) leave nothing for that chain's kernels there.  On 128 CUs
  // the kernel alone is slower (43 vs 36 us at 64 frames) and the training step faster: 7.207 ms (128 or 160) against 7.28 (256), 7.24 (96),
  // 7.29 (64) and 7.30 for the gathered GEMM (profiles/r4_conv_c64_wgrad.txt); at 64 frames 128 workgroups also get exactly 7 tiles each.
  ncu = mscplan::g_knobs.c64w_wgs > 0 ? (mscplan::g_knobs.c64w_wgs < ncu ? mscplan::g_knobs.c64w_wgs : ncu) : (ncu + 1) / 2;
  long nwg = ntiles < ncu ? ntiles : ncu;
  const long room = slab_bytes / ((long)64 * 576 * 4);
  if (nwg > room) nwg = room;
  if (nwg > 0) { const long rounds = (ntiles + nwg - 1) / nwg; nwg = (ntiles + rounds - 1) / rounds; }     // fewest workgroups for that many rounds of tiles (32 frames: 112 x 4 instead of 128 x 3.5)
  CWParams p;
  p.x = x; p.dy = dy; p.slab = slab; p.B = B; p.H = H;
  hipLaunchKernelGGL(conv_c64_wgrad_kernel, dim3((unsigned)nwg), dim3(256), 0, st, p);
  return msc_launch_status() == MSC_OK ? (int)nwg : -MSC_ERR_LAUNCH;
}
