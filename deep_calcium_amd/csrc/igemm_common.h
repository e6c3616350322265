// Parameter block shared by the fp32 and the split-fp16 implicit-GEMM convolution kernels.
#pragma once
#include "common.h"

struct IgemmParams {
  const float* in;
  const float* wp;
  const float* bias;
  float* out;
  float* stats;
  const float* scale;
  const float* shift;
  int N, Hin, Win, Cin;
  int Hout, Wout, Ncols;
  int tilesX, tilesY;
  int relu;
  int scatterCo;  // 0 = dense NHWC output; >0 = Conv2DTranspose scatter with Co = scatterCo
  int biasMod;    // bias index = n % biasMod
  long outLd;     // pixel stride of the output tensor in floats
  const float* inScale;  // f16x3 only: device scalar (power of two) applied to the input before the fp16 split
  // f16x3 only: per-input-channel BatchNorm affine of a NON-materialised activation.  `in` then holds the producer's
  // pre-BN tensor z and the operand relu(fmaf(z, inSc[c], inSh[c])) is formed while it is staged (padding stays 0).
  const float* inSc;
  const float* inSh;
};

