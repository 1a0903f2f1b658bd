// gemm_params.h — launch parameters shared by the contraction kernels (csrc/gemm.hip: the tiled one-tile-per-workgroup family;
// csrc/gemm_pp.hip: the persistent ping-pong kernel of the dense linears).  Build-authored (the upstream reference has no code).
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

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
  float rW, rH, rWo, rHo, rMc, rW2, rH2;   // reciprocals of the decode divisors (gemm.hip fdiv)
  int fastdiv;              // host: every row / pixel index of the launch is below 2^24 -> fdiv may use them
  int cls_interleave;       // set by the launcher when cMc is a multiple of the tile height it picked
  int cls_fill;             // set by the launcher for 1x1 strided dgrads: only class 0's tiles are launched, each fills its siblings' rows
  unsigned long long* stamps;   // GEMM_STAMP diagnostic build only (else null and unused)
  int bm_cap;                   // host only: largest tile height the launch may use (0 = no cap)
  float* rowsum;                // weight-gradient launches (both operands mn-major): fp32 [nsplit][M] sums over the reduction of the
                                // A operand's rows = the bias gradient (dy^T 1), produced by one extra MFMA against a ones fragment
  int xcd_split;                // split-K launches: all tiles of one K slice on one XCD (see the kernel's index mapping)
  const void* dref; int dact;   // epilogue: multiply by act'(dref[m][n]) (fused activation backward)
  float* colsum;                // epilogue: per-row-block column sums of the stored output, [tile_m][N]
  // epilogue (conv dgrad): the stored output is dy of the BatchNorm + ReLU that produced this layer's input.  bnx = that BatchNorm's
  // input (the previous conv's output, [M][N]), bnstat = its [4][N] mean, rstd, scale, shift, bnpart = [tile_m][2][N] per-row-block
  // sums of dz and dz*xhat with dz = dy * [x*scale+shift > 0] — the statistics pass of msc_bn_bwd, done where dy is produced
  const void* bnx; const float* bnstat; float* bnpart;
};

