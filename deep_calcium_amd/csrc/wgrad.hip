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
#include "common.h"

struct WgradParams {
  const float* A;
  const float* B;
  float* slabs;
  int N, Ha, Wa, Cm, Hb, Wb, Cn;
  int tilesX, tilesY, tilesTotal, tilesPerSplit;
};

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
// Split-fp16 variant (see igemm_f16x3.hip for the numerics): k = 16 pixels per v_mfma_f32_32x32x16_f16.
// The contraction index (pixel) is the ROW index of the NHWC tiles, so the MFMA fragments (8 consecutive k per
// lane) are fetched with the hardware-transposing ds_read_b64_tr_b16: each 16-lane group reads a 4-pixel x
// 16-channel block and every lane receives its channel's 4 pixels.  LDS holds fp16 hi and lo images of both
// tiles as 32-channel planes ([plane][pixel][32 ch] = 64-B rows): 4 consecutive pixel rows of one plane are 256
// contiguous bytes, so the transposed reads are bank-conflict-free and every tap / k-step offset folds into the
// instruction's immediate.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((__vector_size__(4 * sizeof(short)))) short tr_v4i16;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct WgradHParams {
  WgradParams g;
  const float* aScale;  // nullable device scalars (powers of two)
  const float* bScale;
};

__device__ __forceinline__ f16x8 tr_frag(const char* base, int off1, int off2) {
  typedef __attribute__((address_space(3))) tr_v4i16* lds_p;
  const tr_v4i16 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(base + off1));
  const tr_v4i16 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(base + off2));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  s16x8 v = {r0[0], r0[1], r0[2], r0[3], r1[0], r1[1], r1[2], r1[3]};
  return __builtin_bit_cast(f16x8, v);
}

__device__ __forceinline__ void split4_f16(const f32x4 v, float s, u32x2& hi, u32x2& lo) {
  f16x4 h, l;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float x = v[e] * s;
    const _Float16 hh = (_Float16)x;
    h[e] = hh;
    l[e] = (_Float16)(x - (float)hh);
  }
  hi = __builtin_bit_cast(u32x2, h);
  lo = __builtin_bit_cast(u32x2, l);
}

template <int KH, int KW, int S, int PAD, int TW, int RW, int WM, int WN>
__global__ __launch_bounds__(256, 2) void wgrad_f16x3_kernel(WgradHParams hp) {
  using Cfg = WgradCfg<KH, KW, S, PAD, TW, RW, WM, WN>;
  constexpr int TAPS = Cfg::TAPS, WK = Cfg::WK, TH = Cfg::TH, CM = Cfg::CM, CN = Cfg::CN;
  constexpr int THI = Cfg::THI, TWI = Cfg::TWI;
  constexpr int APIX = THI * TWI, BPIX = TH * TW;
  constexpr int A_PLANE = APIX * 64, B_PLANE = BPIX * 64;       // bytes of one 32-channel fp16 plane
  constexpr int A_IMG = WM * A_PLANE, B_IMG = WN * B_PLANE;     // one (hi or lo) image
  constexpr int KROWS = (TW >= 16) ? 1 : 16 / TW;               // pixel rows covered by one 16-pixel k-step
  constexpr int KX = (TW >= 16) ? TW / 16 : 1;                  // k-steps along a row
  static_assert(RW % KROWS == 0, "rows per wave must be a multiple of the k-step height");
  static_assert(2 * (A_IMG + B_IMG) == Cfg::LDS_BYTES, "fp16 hi+lo images occupy the fp32 tile bytes");
  const WgradParams& p = hp.g;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ldsAh = smem;
  char* ldsAl = smem + A_IMG;
  char* ldsBh = smem + 2 * A_IMG;
  char* ldsBl = smem + 2 * A_IMG + B_IMG;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, cb = (lane >> 4) & 1, c = lane & 15, q = c >> 2, pp = c & 3;
  const int wm = wave % WM, wn = (wave / WM) % WN, wk = wave / (WM * WN);
  const int m0 = blockIdx.y * CM, n0 = blockIdx.z * CN;
  const int split = blockIdx.x;
  const float a_scale = hp.aScale ? *hp.aScale : 1.f;
  const float b_scale = hp.bScale ? *hp.bScale : 1.f;

  // lane-constant byte offsets of the two transposed reads of a k-step whose first pixel is (row 0, x 0)
  int offA[2], offB[2];
#pragma unroll
  for (int r2 = 0; r2 < 2; ++r2) {
    const int kpix = 8 * h + q + 4 * r2;
    const int ky = (TW >= 16) ? 0 : kpix / TW, kx = (TW >= 16) ? kpix : kpix % TW;
    offA[r2] = wm * A_PLANE + ((ky * S) * TWI + kx * S) * 64 + cb * 32 + pp * 8;
    offB[r2] = wn * B_PLANE + (ky * TW + kx) * 64 + cb * 32 + pp * 8;
  }

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
    const int py0 = ty * TH, px0 = tx * TW;
    const int ay0 = py0 * S - PAD, ax0 = px0 * S - PAD;
    // Staging through buffer descriptors of this image's two tensors: 32-bit (24-bit multiply) offsets, rows above
    // / below the image fall outside the descriptor and read as zeros; only the left/right edge needs a compare.
    {
      constexpr int C4 = CM / 4, TOTAL = APIX * C4, PSTEP = 256 / C4;
      const __amdgpu_buffer_rsrc_t rs = dc_make_rsrc(p.A + (long)img * p.Ha * p.Wa * p.Cm, (unsigned)(p.Ha * p.Wa * p.Cm) * 4u);
      const int c4 = tid % C4;
      const bool ch_ok = (m0 + 4 * c4) < p.Cm;
      const int rowb = p.Wa * p.Cm * 4;                                   // bytes per image row
      const int base = (ay0 * p.Wa + ax0) * p.Cm * 4 + (m0 + 4 * c4) * 4;  // may be negative: wraps out of range
      const int lbase = (c4 >> 3) * A_PLANE + (c4 & 7) * 8;
#pragma unroll 3
      for (int k = 0; k < (TOTAL + 255) / 256; ++k) {
        const int pix = tid / C4 + k * PSTEP;
        const int r = __umul24(pix, (65536 + TWI - 1) / TWI) >> 16;       // pix / TWI for pix < 4096
        const int cc = pix - __umul24(r, TWI);
        const bool ok = ch_ok && pix < APIX && (unsigned)(ax0 + cc) < (unsigned)p.Wa;
        const unsigned off = ok ? (unsigned)(base + __mul24(r, rowb) + __mul24(cc, p.Cm * 4)) : 0x80000000u;
        const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
        u32x2 hi, lo;
        split4_f16(v, a_scale, hi, lo);
        if (pix < APIX) {
          *reinterpret_cast<u32x2*>(ldsAh + lbase + pix * 64) = hi;
          *reinterpret_cast<u32x2*>(ldsAl + lbase + pix * 64) = lo;
        }
      }
    }
    {
      constexpr int C4 = CN / 4, TOTAL = BPIX * C4, PSTEP = 256 / C4;
      const __amdgpu_buffer_rsrc_t rs = dc_make_rsrc(p.B + (long)img * p.Hb * p.Wb * p.Cn, (unsigned)(p.Hb * p.Wb * p.Cn) * 4u);
      const int c4 = tid % C4;
      const bool ch_ok = (n0 + 4 * c4) < p.Cn;
      const int rowb = p.Wb * p.Cn * 4;
      const int base = (py0 * p.Wb + px0) * p.Cn * 4 + (n0 + 4 * c4) * 4;
      const int lbase = (c4 >> 3) * B_PLANE + (c4 & 7) * 8;
#pragma unroll 4
      for (int k = 0; k < (TOTAL + 255) / 256; ++k) {
        const int pix = tid / C4 + k * PSTEP;
        const int r = pix / TW, cc = pix % TW;                           // TW is a power of two
        const bool ok = ch_ok && pix < BPIX && (px0 + cc) < p.Wb;
        const unsigned off = ok ? (unsigned)(base + __mul24(r, rowb) + __mul24(cc, p.Cn * 4)) : 0x80000000u;
        const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
        u32x2 hi, lo;
        split4_f16(v, b_scale, hi, lo);
        if (pix < BPIX) {
          *reinterpret_cast<u32x2*>(ldsBh + lbase + pix * 64) = hi;
          *reinterpret_cast<u32x2*>(ldsBl + lbase + pix * 64) = lo;
        }
      }
    }
    __syncthreads();

#pragma unroll
    for (int rr = 0; rr < RW; rr += KROWS) {
      const int prow = wk * RW + rr;
#pragma unroll
      for (int xs = 0; xs < KX; ++xs) {
        const int kb = (prow * TW + xs * 16) * 64;                       // B-tile byte offset of the k-step
        const int ka = ((prow * S) * TWI + xs * 16 * S) * 64;            // A-tile byte offset (tap 0,0)
        const f16x8 bh = tr_frag(ldsBh, offB[0] + kb, offB[1] + kb);
        const f16x8 bl = tr_frag(ldsBl, offB[0] + kb, offB[1] + kb);
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
          const int toff = ((tap / KW) * TWI + (tap % KW)) * 64;
          const f16x8 ah = tr_frag(ldsAh, offA[0] + ka + toff, offA[1] + ka + toff);
          const f16x8 al = tr_frag(ldsAl, offA[0] + ka + toff, offA[1] + ka + toff);
          acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[tap], 0, 0, 0);
          acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[tap], 0, 0, 0);
          acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[tap], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }

  wgrad_store<TAPS, WM, WN, WK>(p, acc, smem, split, m0, n0, wm, wn, wk, lane, 1.f / (a_scale * b_scale));
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
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
    DC_REQUIRE(e == hipSuccess, DC_EHIP, "%s: hipFuncSetAttribute: %s", name, hipGetErrorString(e));
    attr_set = true;
  }
  WgradPlan pl = wgrad_plan<KH, KW, S, PAD, TW, RW, WM, WN>(N, Hb, Wb, Cm, Cn);
  WgradParams p;
  p.A = A; p.B = B; p.slabs = ws;
  p.N = N; p.Ha = Ha; p.Wa = Wa; p.Cm = Cm; p.Hb = Hb; p.Wb = Wb; p.Cn = Cn;
  p.tilesX = pl.tilesX; p.tilesY = pl.tilesY; p.tilesTotal = pl.tilesTotal; p.tilesPerSplit = pl.tilesPerSplit;
  dim3 grid((unsigned)pl.splits, (unsigned)dc_cdiv(Cm, Cfg::CM), (unsigned)dc_cdiv(Cn, Cfg::CN));
  hipLaunchKernelGGL(kern, grid, dim3(256), Cfg::LDS_BYTES, st, p);
  DC_CHECK_LAUNCH(name);
  const long L = (long)KH * KW * Cm * Cn;
  return dc_reduce_partials(ws, pl.slabs, L, 1.0f, dw, ws + (long)pl.slabs * L, (dc_stream_t)st);
}

template <int KH, int KW, int S, int PAD, int TW, int RW, int WM, int WN>
static int wgrad_h_launch(const float* A, const float* B, float* dw, float* ws, const float* aScale, const float* bScale,
                          int N, int Ha, int Wa, int Hb, int Wb, int Cm, int Cn, hipStream_t st, const char* name) {
  using Cfg = WgradCfg<KH, KW, S, PAD, TW, RW, WM, WN>;
  auto kern = wgrad_f16x3_kernel<KH, KW, S, PAD, TW, RW, WM, WN>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
    DC_REQUIRE(e == hipSuccess, DC_EHIP, "%s: hipFuncSetAttribute: %s", name, hipGetErrorString(e));
    attr_set = true;
  }
  WgradPlan pl = wgrad_plan<KH, KW, S, PAD, TW, RW, WM, WN>(N, Hb, Wb, Cm, Cn);
  WgradHParams hp;
  WgradParams& p = hp.g;
  p.A = A; p.B = B; p.slabs = ws;
  p.N = N; p.Ha = Ha; p.Wa = Wa; p.Cm = Cm; p.Hb = Hb; p.Wb = Wb; p.Cn = Cn;
  p.tilesX = pl.tilesX; p.tilesY = pl.tilesY; p.tilesTotal = pl.tilesTotal; p.tilesPerSplit = pl.tilesPerSplit;
  hp.aScale = aScale; hp.bScale = bScale;
  dim3 grid((unsigned)pl.splits, (unsigned)dc_cdiv(Cm, Cfg::CM), (unsigned)dc_cdiv(Cn, Cfg::CN));
  hipLaunchKernelGGL(kern, grid, dim3(256), Cfg::LDS_BYTES, st, hp);
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
static int conv_wgrad_h_impl(const float* x, const float* dz, float* dw, float* ws, const float* dzScale, int N, int H,
                             int W, int Cin, int Cout, hipStream_t st) {
  const float* none = nullptr;
  CONV_WGRAD_DISPATCH(wgrad_h_launch, x, dz, dw, ws, none, dzScale, N, H, W, H, W, Cin, Cout, st, "conv3x3_wgrad_f16x3")
}
static int convT_wgrad_h_impl(const float* x, const float* dz, float* dw, float* ws, const float* dzScale, int N, int H,
                              int W, int Cin, int Cout, hipStream_t st) {
  const float* none = nullptr;
  CONVT_WGRAD_DISPATCH(wgrad_h_launch, dz, x, dw, ws, dzScale, none, N, 2 * H, 2 * W, H, W, Cout, Cin, st, "convT2x2_wgrad_f16x3")
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

extern "C" long dc_conv3x3_wgrad_ws_floats(int N, int H, int W, int Cin, int Cout) {
  if (Cin == 1) return dc_conv3x3_c1_wgrad_ws(N, H, W, Cout);
  return conv_wgrad_ws_impl(N, H, W, Cin, Cout);
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
  return convT_wgrad_ws_impl(N, H, W, Cin, Cout);
}
extern "C" int dc_convT2x2_wgrad(const float* x, const float* dz, float* dw, float* ws, int N, int H, int W, int Cin,
                                 int Cout, dc_stream_t stream) {
  int rc = check_wgrad("dc_convT2x2_wgrad", x, dz, dw, ws, N, H, W, Cin, Cout);
  if (rc) return rc;
  return convT_wgrad_impl(x, dz, dw, ws, N, H, W, Cin, Cout, (hipStream_t)stream);
}

// split-fp16 variants: same workspace sizes (dc_*_wgrad_ws_floats) and slab reduction; dz_scale = device scalar
// from dc_pow2_scale_from_absmax (nullable).
extern "C" int dc_conv3x3_wgrad_f16x3(const float* x, const float* dz, float* dw, float* ws, const float* dz_scale,
                                      int N, int H, int W, int Cin, int Cout, dc_stream_t stream) {
  if (Cin == 1) {
    DC_REQUIRE(x && dz && dw && ws, DC_EINVAL, "dc_conv3x3_wgrad_f16x3: null pointer");
    return dc_conv3x3_c1_wgrad(x, dz, dw, ws, N, H, W, Cout, (hipStream_t)stream);
  }
  int rc = check_wgrad("dc_conv3x3_wgrad_f16x3", x, dz, dw, ws, N, H, W, Cin, Cout);
  if (rc) return rc;
  return conv_wgrad_h_impl(x, dz, dw, ws, dz_scale, N, H, W, Cin, Cout, (hipStream_t)stream);
}
extern "C" int dc_convT2x2_wgrad_f16x3(const float* x, const float* dz, float* dw, float* ws, const float* dz_scale,
                                       int N, int H, int W, int Cin, int Cout, dc_stream_t stream) {
  int rc = check_wgrad("dc_convT2x2_wgrad_f16x3", x, dz, dw, ws, N, H, W, Cin, Cout);
  if (rc) return rc;
  return convT_wgrad_h_impl(x, dz, dw, ws, dz_scale, N, H, W, Cin, Cout, (hipStream_t)stream);
}
