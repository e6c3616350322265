// Parameter block shared by the fp32 and the split-fp16 implicit-GEMM convolution kernels.
#pragma once
#include "common.h"
#include <stdlib.h>

// see IgemmParams::walk: column-major (measured +0.6 % end to end, weight-gradient fetch of the 512^2 layers -25 %)
static inline int dc_tile_walk() { return 1; }

struct IgemmParams {
  const float* in;
  const float* wp;
  const float* bias;
  float* out;
  double* stats;  // per-(tile, column) BatchNorm partials (sum v, sum v^2) as doubles, see DcMoments (common.h)
  const float* scale;
  const float* shift;
  int N, Hin, Win, Cin;
  int Hout, Wout, Ncols;
  int tilesX, tilesY;
  // Order in which the workgroups walk the pixel tiles (speed only; the BatchNorm-partial index stays the row-major tile
  // id): 0 = x fastest; 1 = y fastest -- consecutive positions are VERTICALLY adjacent tiles, whose halo'd input patches
  // share KH - 1 full-width rows (2 of 10 / 18 rows for the 8- / 16-row tiles) against KW - 1 of TW + 2 columns (2 of 34)
  // for horizontal neighbours: what one XCD's workgroups fetch side by side / back to back overlaps more.  dc_tile_walk().
  int walk;
  int relu;
  int scatterCo;  // 0 = dense NHWC output; >0 = Conv2DTranspose scatter with Co = scatterCo
  int biasMod;    // bias index = n % biasMod
  long outLd;     // pixel stride of the output tensor in floats
  const float* inScale;  // f16x3 only: device scalar (power of two) applied to the input before the fp16 split
  // f16x3 data gradients, instead of inScale: the per-block max |dz| array dc_bn_bwd_apply wrote (inAbsmaxN entries); every
  // workgroup derives the power of two itself (dc_block_absmax_scale, target 1024), so no finalize launch sits between the
  // BatchNorm-backward apply pass and the data gradient
  const float* inAbsmax;
  int inAbsmaxN;
  // f16x3 only: per-input-channel BatchNorm affine of a NON-materialised activation.  `in` then holds the producer's
  // pre-BN tensor z and the operand relu(fmaf(z, inSc[c], inSh[c])) is formed while it is staged (padding stays 0).
  const float* inSc;
  const float* inSh;
  // f16x3 only: fp16 range guard (common.h).  inAbound: per-input-channel magnitude bound of the activation operand
  // (Cin floats, nullable -> scale 1).  The packed weights carry their own power-of-two scale in a 16-byte trailer
  // (dc_pack_weights_f16x3).  outAbsmax (nullable, biasMod floats): the epilogue folds max |output| per channel into it
  // (atomic max; the caller zeroes it) -- the bound the NEXT layer's guard reads in inference.
  const float* inAbound;
  long inAboundLd;   // 0: one slot (training bound); > 0: DC_ABOUND_SLOTS measured replicas, this many floats apart
  float* outAbsmax;
  long outAbsmaxLd;  // floats between the replicas of the output's array
  // role-split kernel only, data-gradient launches: the output IS `da` of a BatchNorm layer (dense, outLd == Ncols, no
  // dropout) -- the epilogue also reads that layer's pre-BN tensor bnZ (same layout) and emits its BatchNorm-backward
  // pass-1 partials bnPartial[tile][Ncols][2] = (sum dy, sum dy*xhat), dy = da * [fmaf(z, sc, sh) > 0] with (sc, sh) =
  // dc_bn_affine(mean, invstd, gamma, beta): what dc_bn_bwd_reduce would re-read da and z for.
  // role-split kernel only, inference: the output also feeds MaxPooling2D((2,2)) -- the epilogue writes the pooled tensor
  // [N][Hout/2][Wout/2][Ncols] (dense) next to the activation (same values, same maxima as dc_maxpool2x2_fwd on it)
  float* poolOut;
  const float* bnZ;
  const float* bnMean;
  const float* bnInvstd;
  const float* bnGamma;
  const float* bnBeta;
  float* bnPartial;
  float* bnAmax;        // nullable: [tile][Ncols] max |dy| per tile and column (dc_bn_bwd_finalize_dzin's amax_partial)
  // role-split kernel only, data-gradient launches, "dz on load": `in` holds da and `in2` the block's pre-BN tensor z (same
  // dense layout); the producer waves form dz = fmaf(A, [fmaf(z,sc,sh) > 0] * da, fmaf(D, z - mu, E)) from the table
  // dzCoef[DC_DZ_COEF_ROWS][Cin] (dc_bn_bwd_finalize_dzin; row 6 = the bound the fp16 range guard scales by)
  const float* in2;
  const float* dzCoef;
  // ... and (nullable) where the producers also write the dz they form: dense [N][Hin][Win][Cin] like `in`, every pixel once (the
  // interior of each tile's halo'd patch, by the workgroup that holds the tile's FIRST column block)
  float* dzOut;
  // 256-thread f16x3 kernel only, split-K (grid.y = S > 1: the narrow 16^2 / 8^2 layers of a small step put < 256 workgroups
  // on the chip with 16-64 serial chunks each): workgroup (x, s) contracts chunks [nch s / S, nch (s+1) / S) and writes its
  // scaled partial tile to the dense slab splitWs + s * splitSlab ([N][Hout][Wout][Ncols] floats, no bias / statistics);
  // splitk_combine_kernel adds the S slabs in index order (bit-reproducible), applies the bias, writes the output and forms
  // the BatchNorm partials of the same tile rows.
  float* splitWs;
  long splitSlab;
  // role-split kernel only, forward launches with BatchNorm partials: 1 = every (workgroup, consumer set) Chan-merges the
  // tiles it computes and writes ONE row stats[2 * blockIdx.x + set][Ncols][2] (dcunet.h dc_conv3x3_stats_rows) instead of one
  // row per pixel tile
  int statsPerWg;
};

