// Weight-gradient contraction on the fp32 matrix cores, gfx950.
//
//   out[tap][m][n] = sum over pixels p of  A[S*p + tap - PAD][m] * B[p][n]
//
//   conv3x3 wgrad : A = layer input x (m = Cin), B = dz (n = Cout)  -> HWIO (3,3,Cin,Cout)
//   convT2x2 wgrad: A = dz (2H x 2W, m = Cout, S = 2), B = x (n = Cin) -> Keras (2,2,Cout,Cin)
// (the gradients Keras/TF compute for the Conv2D / Conv2DTranspose kernels created at
//  /root/reference/deepcalcium/models/neurons/unet_2d_summary.py:156-157,:164-165 during fit_generator :429).
//
// The contraction index is the pixel, so the MFMA "k" runs over pixels: one v_mfma_f32_32x32x2_f32
// consumes 2 neighbouring pixels (lane half h picks pixel 2s+h) for 32 m-channels x 32 n-channels.
// Both operand tiles sit in LDS in their natural NHWC order ([pixel][channel]), which makes the
// operand reads plain conflict-free ds_read_b32 (32 consecutive channels per half-wave).  The B
// operand is shared by all taps; each tap is a shifted window of the same A tile (halo staged once).
// A wave owns a 32x32 (m,n) block for ALL taps (taps*16 accumulator registers).  The pixel range is
// split over CTAs (and over the WK waves of a CTA, summed through LDS at the end); every CTA writes one
// partial slab and dc_reduce_partials sums the slabs in a fixed order => bit-reproducible, no atomics.
#include "wgrad_common.h"

template <int KH, int KW, int S, int PAD, int TW, int RW, int WM, int WN>
struct WgradCfg {
  static constexpr int TAPS = KH * KW;
  static constexpr int WK = 4 / (WM * WN);
  static constexpr int TH = WK * RW;
  static constexpr int CM = 32 * WM, CN = 32 * WN;
  static constexpr int THI = (TH - 1) * S + KH, TWI = (TW - 1) * S + KW;
  static constexpr int A_FLOATS = THI * TWI * CM, B_FLOATS = TH * TW * CN;
  static constexpr int LDS_BYTES = (A_FLOATS + B_FLOATS) * 4;
};

template <int KH, int KW, int S, int PAD, int TW, int RW, int WM, int WN>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(WgradParams p) {
  using Cfg = WgradCfg<KH, KW, S, PAD, TW, RW, WM, WN>;
  constexpr int TAPS = Cfg::TAPS, WK = Cfg::WK, TH = Cfg::TH, CM = Cfg::CM, CN = Cfg::CN;
  constexpr int THI = Cfg::THI, TWI = Cfg::TWI;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* ldsA = reinterpret_cast<float*>(smem);
  float* ldsB = ldsA + Cfg::A_FLOATS;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, h = lane >> 5;
  const int wm = wave % WM, wn = (wave / WM) % WN, wk = wave / (WM * WN);
  const int m0 = blockIdx.y * CM, n0 = blockIdx.z * CN;
  const int split = blockIdx.x;

  f32x16 acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int tile_beg = split * p.tilesPerSplit;
  const int tile_end = min(tile_beg + p.tilesPerSplit, p.tilesTotal);
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  for (int tile = tile_beg; tile < tile_end; ++tile) {
    int t = tile;
    const int tx = t % p.tilesX; t /= p.tilesX;
    const int ty = t % p.tilesY;
    const int img = t / p.tilesY;
    const int py0 = ty * TH, px0 = tx * TW;            // B-grid pixel origin
    const int ay0 = py0 * S - PAD, ax0 = px0 * S - PAD;  // A-grid origin

    // stage A (halo'd, zero-filled outside the image / beyond Cm)
    {
      constexpr int C4 = CM / 4, TOTAL = THI * TWI * C4;
      const f32x4* src = reinterpret_cast<const f32x4*>(p.A + (long)img * p.Ha * p.Wa * p.Cm);
      for (int idx = tid; idx < TOTAL; idx += 256) {
        const int pix = idx / C4, c4 = idx - pix * C4;
        const int r = pix / TWI, c = pix - r * TWI;
        const int y = ay0 + r, x = ax0 + c;
        const bool ok = y >= 0 && y < p.Ha && x >= 0 && x < p.Wa && (m0 + 4 * c4) < p.Cm;
        f32x4 v = ok ? src[((long)(y * p.Wa + x) * p.Cm + m0) / 4 + c4] : zero4;
        *reinterpret_cast<f32x4*>(ldsA + pix * CM + 4 * c4) = v;
      }
    }
    {
      constexpr int C4 = CN / 4, TOTAL = TH * TW * C4;
      const f32x4* src = reinterpret_cast<const f32x4*>(p.B + (long)img * p.Hb * p.Wb * p.Cn);
      for (int idx = tid; idx < TOTAL; idx += 256) {
        const int pix = idx / C4, c4 = idx - pix * C4;
        const int r = pix / TW, c = pix - r * TW;
        const int y = py0 + r, x = px0 + c;
        const bool ok = y < p.Hb && x < p.Wb && (n0 + 4 * c4) < p.Cn;
        f32x4 v = ok ? src[((long)(y * p.Wb + x) * p.Cn + n0) / 4 + c4] : zero4;
        *reinterpret_cast<f32x4*>(ldsB + pix * CN + 4 * c4) = v;
      }
    }
    __syncthreads();

#pragma unroll
    for (int rr = 0; rr < RW; ++rr) {
      const int prow = wk * RW + rr;
      const float* bptr = ldsB + (prow * TW + h) * CN + wn * 32 + li;
      const float* aptr = ldsA + ((prow * S) * TWI + h * S) * CM + wm * 32 + li;
#pragma unroll 4
      for (int s = 0; s < TW / 2; ++s) {
        const float b = bptr[2 * s * CN];
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
          const float a = aptr[((tap / KW) * TWI + 2 * s * S + (tap % KW)) * CM];
          acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[tap], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }

  wgrad_store<TAPS, WM, WN, WK>(p, acc, smem, split, m0, n0, wm, wn, wk, lane, 1.f);
}

// ---------------------------------------------------------------------------------------------------
struct WgradPlan {
  int splits, slabs, tilesX, tilesY, tilesTotal, tilesPerSplit;
};

template <int KH, int KW, int S, int PAD, int TW, int RW, int WM, int WN>
static WgradPlan wgrad_plan(int N, int Hb, int Wb, int Cm, int Cn) {
  using Cfg = WgradCfg<KH, KW, S, PAD, TW, RW, WM, WN>;
  WgradPlan pl;
  pl.tilesX = dc_cdiv(Wb, TW);
  pl.tilesY = dc_cdiv(Hb, Cfg::TH);
  pl.tilesTotal = N * pl.tilesX * pl.tilesY;
  const int blocks_mn = dc_cdiv(Cm, Cfg::CM) * dc_cdiv(Cn, Cfg::CN);
  int want = dc_cdiv(512, blocks_mn);  // 2 CTAs/CU x 256 CUs: one resident wave of CTAs, fewest slabs
  if (want > pl.tilesTotal) want = pl.tilesTotal;
  if (want < 1) want = 1;
  pl.tilesPerSplit = dc_cdiv(pl.tilesTotal, want);
  pl.splits = dc_cdiv(pl.tilesTotal, pl.tilesPerSplit);
  pl.slabs = pl.splits;  // the WK waves of a CTA are summed in LDS before the slab is written
  return pl;
}

template <int KH, int KW, int S, int PAD, int TW, int RW, int WM, int WN>
static long wgrad_ws(int N, int Hb, int Wb, int Cm, int Cn) {
  WgradPlan pl = wgrad_plan<KH, KW, S, PAD, TW, RW, WM, WN>(N, Hb, Wb, Cm, Cn);
  // slabs + the reduce kernel's second-stage scratch (32 * L)
  const long L = (long)KH * KW * Cm * Cn;
  return (long)pl.slabs * L + 32 * L;
}

template <int KH, int KW, int S, int PAD, int TW, int RW, int WM, int WN>
static int wgrad_launch(const float* A, const float* B, float* dw, float* ws, int N, int Ha, int Wa, int Hb, int Wb,
                        int Cm, int Cn, hipStream_t st, const char* name) {
  using Cfg = WgradCfg<KH, KW, S, PAD, TW, RW, WM, WN>;
  auto kern = wgrad_kernel<KH, KW, S, PAD, TW, RW, WM, WN>;
  static DcLdsAttr lds_attr;      // one per template instantiation; per-device inside
  if (int rc = dc_func_max_lds(lds_attr, reinterpret_cast<const void*>(kern), Cfg::LDS_BYTES, name)) return rc;
  WgradPlan pl = wgrad_plan<KH, KW, S, PAD, TW, RW, WM, WN>(N, Hb, Wb, Cm, Cn);
  WgradParams p{};
  p.A = A; p.B = B; p.slabs = ws;
  p.N = N; p.Ha = Ha; p.Wa = Wa; p.Cm = Cm; p.Hb = Hb; p.Wb = Wb; p.Cn = Cn;
  p.tilesX = pl.tilesX; p.tilesY = pl.tilesY; p.tilesTotal = pl.tilesTotal; p.tilesPerSplit = pl.tilesPerSplit;
  dim3 grid((unsigned)pl.splits, (unsigned)dc_cdiv(Cm, Cfg::CM), (unsigned)dc_cdiv(Cn, Cfg::CN));
  hipLaunchKernelGGL(kern, grid, dim3(256), Cfg::LDS_BYTES, st, p);
  DC_CHECK_LAUNCH(name);
  const long L = (long)KH * KW * Cm * Cn;
  return dc_reduce_partials(ws, pl.slabs, L, 1.0f, dw, ws + (long)pl.slabs * L, (dc_stream_t)st);
}

// conv3x3: pick the (m,n) wave arrangement from the channel counts, the pixel tile from the width.
#define CONV_WGRAD_DISPATCH(FN, ...)                                                   \
  if (W <= 8) return FN<3, 3, 1, 1, 8, 8, 2, 2>(__VA_ARGS__);                          \
  if (W <= 16) return FN<3, 3, 1, 1, 16, 4, 2, 2>(__VA_ARGS__);                        \
  if (Cin > 32 && Cout > 32) return FN<3, 3, 1, 1, 32, 2, 2, 2>(__VA_ARGS__);          \
  if (Cin > 32) return FN<3, 3, 1, 1, 32, 2, 2, 1>(__VA_ARGS__);                       \
  if (Cout > 32) return FN<3, 3, 1, 1, 32, 2, 1, 2>(__VA_ARGS__);                      \
  return FN<3, 3, 1, 1, 32, 2, 1, 1>(__VA_ARGS__);

#define CONVT_WGRAD_DISPATCH(FN, ...)                                                  \
  if (W <= 8) return FN<2, 2, 2, 0, 8, 4, 1, 2>(__VA_ARGS__);                          \
  if (W <= 16) return FN<2, 2, 2, 0, 16, 2, 1, 2>(__VA_ARGS__);                        \
  return FN<2, 2, 2, 0, 32, 1, 1, 2>(__VA_ARGS__);

static long conv_wgrad_ws_impl(int N, int H, int W, int Cin, int Cout) {
  CONV_WGRAD_DISPATCH(wgrad_ws, N, H, W, Cin, Cout)
}
static int conv_wgrad_impl(const float* x, const float* dz, float* dw, float* ws, int N, int H, int W, int Cin,
                           int Cout, hipStream_t st) {
  CONV_WGRAD_DISPATCH(wgrad_launch, x, dz, dw, ws, N, H, W, H, W, Cin, Cout, st, "conv3x3_wgrad")
}
static long convT_wgrad_ws_impl(int N, int H, int W, int Cin, int Cout) {
  CONVT_WGRAD_DISPATCH(wgrad_ws, N, H, W, Cout, Cin)
}
static int convT_wgrad_impl(const float* x, const float* dz, float* dw, float* ws, int N, int H, int W, int Cin,
                            int Cout, hipStream_t st) {
  CONVT_WGRAD_DISPATCH(wgrad_launch, dz, x, dw, ws, N, 2 * H, 2 * W, H, W, Cout, Cin, st, "convT2x2_wgrad")
}

static int check_wgrad(const char* fn, const void* a, const void* b, const void* c, const void* d, int N, int H, int W,
                       int Cin, int Cout) {
  DC_REQUIRE(a && b && c && d, DC_EINVAL, "%s: null pointer", fn);
  DC_REQUIRE(dc_aligned16(a) && dc_aligned16(b) && dc_aligned16(c) && dc_aligned16(d), DC_EINVAL,
             "%s: pointers must be 16-byte aligned", fn);
  DC_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, DC_EINVAL, "%s: non-positive dimension", fn);
  DC_REQUIRE(Cin % 4 == 0 && Cout % 4 == 0, DC_EUNSUP, "%s: channel counts must be multiples of 4 (Cin=%d Cout=%d)", fn,
             Cin, Cout);
  return DC_OK;
}

// Cin == 1 (first layer) has no matrix shape: conv_c1.hip
long dc_conv3x3_c1_wgrad_ws(int N, int H, int W, int Cout);
int dc_conv3x3_c1_wgrad(const float* x, const float* dz, float* dw, float* ws, int N, int H, int W, int Cout,
                        hipStream_t st);

// the split-fp16 kernels (wgrad_f16x3.hip) tile differently: one workspace size serves both
long dc_conv3x3_wgrad_f16x3_ws(int N, int H, int W, int Cin, int Cout);
long dc_convT2x2_wgrad_f16x3_ws(int N, int H, int W, int Cin, int Cout);

extern "C" long dc_conv3x3_wgrad_ws_floats(int N, int H, int W, int Cin, int Cout) {
  if (Cin == 1) return dc_conv3x3_c1_wgrad_ws(N, H, W, Cout);
  const long a = conv_wgrad_ws_impl(N, H, W, Cin, Cout), b = dc_conv3x3_wgrad_f16x3_ws(N, H, W, Cin, Cout);
  return a > b ? a : b;
}
extern "C" int dc_conv3x3_wgrad(const float* x, const float* dz, float* dw, float* ws, int N, int H, int W, int Cin,
                                int Cout, dc_stream_t stream) {
  if (Cin == 1) {
    DC_REQUIRE(x && dz && dw && ws, DC_EINVAL, "dc_conv3x3_wgrad: null pointer");
    return dc_conv3x3_c1_wgrad(x, dz, dw, ws, N, H, W, Cout, (hipStream_t)stream);
  }
  int rc = check_wgrad("dc_conv3x3_wgrad", x, dz, dw, ws, N, H, W, Cin, Cout);
  if (rc) return rc;
  return conv_wgrad_impl(x, dz, dw, ws, N, H, W, Cin, Cout, (hipStream_t)stream);
}
extern "C" long dc_convT2x2_wgrad_ws_floats(int N, int H, int W, int Cin, int Cout) {
  const long a = convT_wgrad_ws_impl(N, H, W, Cin, Cout), b = dc_convT2x2_wgrad_f16x3_ws(N, H, W, Cin, Cout);
  return a > b ? a : b;
}
extern "C" int dc_convT2x2_wgrad(const float* x, const float* dz, float* dw, float* ws, int N, int H, int W, int Cin,
                                 int Cout, dc_stream_t stream) {
  int rc = check_wgrad("dc_convT2x2_wgrad", x, dz, dw, ws, N, H, W, Cin, Cout);
  if (rc) return rc;
  return convT_wgrad_impl(x, dz, dw, ws, N, H, W, Cin, Cout, (hipStream_t)stream);
}

