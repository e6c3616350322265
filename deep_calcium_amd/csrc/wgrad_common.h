// Shared by the fp32 (wgrad.hip) and split-fp16 (wgrad_f16x3.hip) weight-gradient kernels.
#pragma once
#include "common.h"

struct WgradParams {
  const float* A;
  const float* B;
  float* slabs;
  int N, Ha, Wa, Cm, Hb, Wb, Cn;
  int tilesX, tilesY, tilesTotal, tilesPerSplit;
  int walk;   // 1: a workgroup's consecutive tiles are vertically adjacent (their halo'd x tiles share KH - 1 rows); 0: x fastest
};

// Epilogue shared by both kernels: the WK waves of a CTA that own the same (m,n) block first add their
// accumulators through LDS (free after the last barrier), then ONE slab per CTA goes to HBM.
// C/D map of the 32x32 MFMA: col = lane&31 -> n, row = (r&3) + 8*(r>>2) + 4*(lane>>5) -> m.
template <int TAPS, int WM, int WN, int WK>
__device__ __forceinline__ void wgrad_store(const WgradParams& p, f32x16 (&acc)[TAPS], char* smem, int split, int m0,
                                            int n0, int wm, int wn, int wk, int lane, float out_scale) {
  const int li = lane & 31, h = lane >> 5;
  if constexpr (WK > 1) {
    float* red = reinterpret_cast<float*>(smem);   // [wk-1][wm*WN+wn][16 regs][64 lanes]
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
      if (wk > 0) {
        float* dst = red + ((((wk - 1) * WM * WN + wm * WN + wn) * 16) * 64) + lane;
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[r * 64] = acc[tap][r];
      }
      __syncthreads();
      if (wk == 0) {
#pragma unroll
        for (int k = 0; k < WK - 1; ++k) {
          const float* src = red + (((k * WM * WN + wm * WN + wn) * 16) * 64) + lane;
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[tap][r] += src[r * 64];
        }
      }
      __syncthreads();
    }
    if (wk > 0) return;
  }
  const int n = n0 + wn * 32 + li;
  if (n < p.Cn) {
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
      float* dst = p.slabs + ((long)split * TAPS + tap) * p.Cm * p.Cn;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (m < p.Cm) dst[(long)m * p.Cn + n] = acc[tap][r] * out_scale;
      }
    }
  }
}

