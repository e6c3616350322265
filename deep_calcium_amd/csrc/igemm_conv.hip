// Implicit-GEMM convolution on the fp32 matrix cores (v_mfma_f32_32x32x2_f32), gfx950.
//
// One kernel template serves every "pixels x K -> pixels x Ncols" contraction of the UNet2DS path
// (reference call sites: Conv2D / Conv2DTranspose layers built at
//  /root/reference/deepcalcium/models/neurons/unet_2d_summary.py:156-157,:164-165):
//   conv3x3 forward   (KH=KW=3, S=1, PAD=1)   K = 9*Cin,  Ncols = Cout
//   conv3x3 dgrad     (same, flipped taps, transposed weights)   K = 9*Cout, Ncols = Cin
//   convT2x2 forward  (KH=KW=1, S=1, PAD=0)   K = Cin,  Ncols = 4*Cout, scatter epilogue
//   convT2x2 dgrad    (KH=KW=2, S=2, PAD=0)   K = 4*Cout, Ncols = Cin
//
// Design (im2col-free):
//   * a CTA (256 threads = 4 waves, one per SIMD) owns a TH x TW patch of output pixels and BN columns;
//   * per CK-channel chunk the input patch WITH its halo is staged once into LDS as
//     [channel-group g][pixel][4 channels] (one 16-B slot per pixel and group), the weight slab as
//     [tap][g][column][4 channels]; both are read back with conflict-free ds_read_b128
//     (consecutive lanes -> consecutive 16-B slots);
//   * the 9 taps are 9 shifted windows of the SAME LDS patch: tap offsets fold into the ds_read
//     immediate, nothing is re-staged and no im2col matrix ever exists;
//   * each lane's float4 feeds 4 MFMAs (k-pairs {4g+e, 4(g+1)+e}); MFMA M = 32 pixels
//     (32/TW rows x TW cols), N = 32 columns; a wave holds MB x NB accumulator tiles;
//   * next chunk's global loads are issued into registers before the MFMA block (latency hidden
//     behind ~37k MFMA cycles), written to LDS after the barrier; 2 CTAs/CU give a second layer of overlap;
//   * epilogue: channel sits on the lane (C/D col = lane&31), so bias / folded-BN affine are per-lane
//     scalars and the BatchNorm (sum, sumsq) partials are an in-register reduction + one xor-32 shuffle.
#include "igemm_common.h"

template <int KH, int KW, int S, int PAD, int TW, int WAVES_M, int MB, int NB, int CK>
struct IgemmCfg {
  static constexpr int TAPS = KH * KW;
  static constexpr int WAVES_N = 4 / WAVES_M;
  static constexpr int ROWS_PER_MBLK = 32 / TW;
  static constexpr int TH = WAVES_M * MB * ROWS_PER_MBLK;
  static constexpr int BN = WAVES_N * NB * 32;
  static constexpr int THI = (TH - 1) * S + KH;
  static constexpr int TWI = (TW - 1) * S + KW;
  static constexpr int NPIXH = THI * TWI;
  static constexpr int PS = ((NPIXH + 5) / 8) * 8 + 2;  // plane stride == 2 (mod 8): ds_write_b128 conflict-free
  static constexpr int G = CK / 4;
  static constexpr int NA = (NPIXH * G + 255) / 256;
  static constexpr int NBV = (TAPS * G * BN + 255) / 256;
  static constexpr int LDS_BYTES = (G * PS + TAPS * G * BN) * 16;
  static_assert(G % 2 == 0, "CK must be a multiple of 8");
  static_assert(256 % G == 0 && 256 % BN == 0, "staging assumes G and BN divide the block size");
  static_assert(32 % TW == 0, "TW must divide 32");
};

template <int KH, int KW, int S, int PAD, int TW, int WAVES_M, int MB, int NB, int CK>
__global__ __launch_bounds__(256, 2) void igemm_kernel(IgemmParams p) {
  using Cfg = IgemmCfg<KH, KW, S, PAD, TW, WAVES_M, MB, NB, CK>;
  constexpr int TAPS = Cfg::TAPS, TH = Cfg::TH, BN = Cfg::BN, TWI = Cfg::TWI, NPIXH = Cfg::NPIXH;
  constexpr int PS = Cfg::PS, G = Cfg::G, NA = Cfg::NA, NBV = Cfg::NBV, RPM = Cfg::ROWS_PER_MBLK;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x4* ldsA = reinterpret_cast<f32x4*>(smem);
  f32x4* ldsB = ldsA + G * PS;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform
  const int li = lane & 31, h = lane >> 5;
  const int wave_m = wave % WAVES_M, wave_n = wave / WAVES_M;

  // XCD-aware rasterisation: workgroups are dealt round-robin over the 8 XCDs (ids b and b+8 share an L2), so
  // each XCD walks a contiguous range of (pixel tile, column block) pairs with the column block fastest: the CTAs
  // that re-read one input patch for different output columns run back to back on ONE L2 (speed only; any
  // placement computes the same result).  Bijective for every grid size.
  const int nblk = (p.Ncols + BN - 1) / BN, total = (int)gridDim.x;
  const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3, qq = total >> 3, rr = total & 7;
  const int work = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + seq;
  const int tile_id = work / nblk;
  int t = tile_id;
  const int tx = t % p.tilesX; t /= p.tilesX;
  const int ty = t % p.tilesY;
  const int img = t / p.tilesY;
  const int n0 = (work - tile_id * nblk) * BN;
  const int oy0 = ty * TH, ox0 = tx * TW;
  const int iy0 = oy0 * S - PAD, ix0 = ox0 * S - PAD;
  const int Cin4 = p.Cin >> 2;

  const f32x4* in4 = reinterpret_cast<const f32x4*>(p.in) + (long)img * p.Hin * p.Win * Cin4;
  const f32x4* wp4 = reinterpret_cast<const f32x4*>(p.wp);

  // ---- chunk-invariant staging coordinates --------------------------------------------------
  // idx = tid + it*256 and G, BN divide 256, so a thread's channel group / column are the same for
  // every `it`; only the pixel (A) / slab row (B) advance, by a compile-time step.
  constexpr int A_STEP = 256 / G;   // pixels per it
  constexpr int B_STEP = 256 / BN;  // slab rows per it
  const int a_g = tid % G, a_pix0 = tid / G;
  const int b_j = tid % BN, b_row0 = tid / BN;
  const bool b_col_ok = n0 + b_j < p.Ncols;
  int a_goff[NA];  // float4 offset inside the image (channel group 0 of the chunk), -1 = zero-fill
#pragma unroll
  for (int it = 0; it < NA; ++it) {
    const int pix = a_pix0 + it * A_STEP;
    const int r = pix / TWI, c = pix - r * TWI;
    const int y = iy0 + r, x = ix0 + c;
    const bool in_img = pix < NPIXH && y >= 0 && y < p.Hin && x >= 0 && x < p.Win;
    a_goff[it] = in_img ? ((y * p.Win + x) * Cin4 + a_g) : -1;
  }

  f32x4 ra[NA], rb[NBV];
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  auto load_chunk = [&](int c0) {
    const int cg0 = c0 >> 2;
    const bool a_ch_ok = (cg0 + a_g) < Cin4;
#pragma unroll
    for (int it = 0; it < NA; ++it) {
      const bool ok = a_goff[it] >= 0 && a_ch_ok;
      ra[it] = ok ? in4[a_goff[it] + cg0] : zero4;
    }
#pragma unroll
    for (int it = 0; it < NBV; ++it) {
      const int row = b_row0 + it * B_STEP;
      const int tap = row / G, g = row - tap * G;
      const bool ok = b_col_ok && row < TAPS * G && (cg0 + g) < Cin4;
      rb[it] = ok ? wp4[(long)(tap * Cin4 + cg0 + g) * p.Ncols + n0 + b_j] : zero4;
    }
  };

  // ---- per-lane MFMA operand bases ------------------------------------------------------------
  int a_base[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int mblk = wave_m * MB + mb;
    const int row = mblk * RPM + li / TW, col = li % TW;
    a_base[mb] = h * PS + (row * S) * TWI + col * S;
  }
  const int b_base = h * BN + wave_n * NB * 32 + li;

  f32x16 acc[MB][NB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;

  load_chunk(0);
  for (int c0 = 0; c0 < p.Cin; c0 += CK) {
#pragma unroll
    for (int it = 0; it < NA; ++it) {
      const int pix = a_pix0 + it * A_STEP;
      if (pix < NPIXH) ldsA[a_g * PS + pix] = ra[it];
    }
#pragma unroll
    for (int it = 0; it < NBV; ++it) {
      const int row = b_row0 + it * B_STEP;
      if (row < TAPS * G) ldsB[row * BN + b_j] = rb[it];
    }
    __syncthreads();
    if (c0 + CK < p.Cin) load_chunk(c0 + CK);  // in flight behind the MFMA block below

#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
      const int toff = (tap / KW) * TWI + (tap % KW);
#pragma unroll
      for (int g2 = 0; g2 < G / 2; ++g2) {
        f32x4 a[MB], b[NB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) a[mb] = ldsA[a_base[mb] + 2 * g2 * PS + toff];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) b[nb] = ldsB[b_base + (tap * G + 2 * g2) * BN + nb * 32];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
              acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mb][e], b[nb][e], acc[mb][nb], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  // ---- epilogue ---------------------------------------------------------------------------------
  // C/D map of the 32x32 MFMA: col = lane&31 (column n), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (pixel m).
  DcMoments* red = reinterpret_cast<DcMoments*>(smem);  // [4 waves][NB][32], LDS is free after the last barrier
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int n = n0 + (wave_n * NB + nb) * 32 + li;
    const bool n_ok = n < p.Ncols;
    const float bv = (p.bias && n_ok) ? p.bias[n % p.biasMod] : 0.f;
    const float sc = (p.scale && n_ok) ? p.scale[n % p.biasMod] : 1.f;
    const float sh = (p.shift && n_ok) ? p.shift[n % p.biasMod] : 0.f;
    const float K = acc[0][nb][0] + bv;      // shifted sums around the lane's first value (common.h DcMoments)
    float s1 = 0.f, s2 = 0.f, cnt = 0.f;
    long obase;
    long ostride_y, ostride_x;  // in floats
    if (p.scatterCo > 0) {
      const int ab = n / p.scatterCo, o = n - ab * p.scatterCo;
      const int W2 = 2 * p.Wout;
      obase = (((long)img * 2 * p.Hout + (ab >> 1)) * W2 + (ab & 1)) * p.outLd + o;
      ostride_y = 2 * W2 * p.outLd;
      ostride_x = 2 * p.outLd;
    } else {
      obase = (long)img * p.Hout * p.Wout * p.outLd + n;
      ostride_y = p.Wout * p.outLd;
      ostride_x = p.outLd;
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const int mblk = wave_m * MB + mb;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * h;
        const int oy = oy0 + mblk * RPM + m / TW, ox = ox0 + m % TW;
        const bool ok = n_ok && oy < p.Hout && ox < p.Wout;
        float v = acc[mb][nb][r] + bv;
        if (ok) {
          const float d = v - K;
          s1 += d;
          s2 += d * d;
          cnt += 1.f;
          if (p.scale) v = v * sc + sh;
          if (p.relu) v = fmaxf(v, 0.f);
          p.out[obase + oy * ostride_y + ox * ostride_x] = v;
        }
      }
    }
    if (p.stats) {
      DcMoments m = dc_moments_from_shifted(cnt, K, s1, s2);
      DcMoments o;
      o.n = __shfl_xor(m.n, 32); o.mean = __shfl_xor(m.mean, 32); o.m2 = __shfl_xor(m.m2, 32);
      m = dc_moments_merge(m, o);
      if (h == 0) red[(wave * NB + nb) * 32 + li] = m;
    }
  }
  if (p.stats) {
    __syncthreads();
    // one thread per (wave_n, nb, li): sum over the WAVES_M waves that share those columns
    if (tid < Cfg::WAVES_N * NB * 32) {
      const int wn = tid / (NB * 32), rem = tid % (NB * 32);
      const int nb = rem / 32, l = rem % 32;
      DcMoments m = red[((wn * WAVES_M) * NB + nb) * 32 + l];
#pragma unroll
      for (int wm = 1; wm < WAVES_M; ++wm) m = dc_moments_merge(m, red[((wn * WAVES_M + wm) * NB + nb) * 32 + l]);
      const int n = n0 + (wn * NB + nb) * 32 + l;
      if (n < p.Ncols) dc_moments_store(p.stats + ((long)tile_id * p.Ncols + n) * 2, m);
    }
  }
}

// ---------------------------------------------------------------------------------------------------
template <int KH, int KW, int S, int PAD, int TW, int WAVES_M, int MB, int NB, int CK>
static int igemm_launch(IgemmParams p, hipStream_t st, const char* name) {
  using Cfg = IgemmCfg<KH, KW, S, PAD, TW, WAVES_M, MB, NB, CK>;
  auto kern = igemm_kernel<KH, KW, S, PAD, TW, WAVES_M, MB, NB, CK>;
  static DcLdsAttr lds_attr;      // one per template instantiation; per-device inside
  if (int rc = dc_func_max_lds(lds_attr, reinterpret_cast<const void*>(kern), Cfg::LDS_BYTES, name)) return rc;
  p.tilesX = dc_cdiv(p.Wout, TW);
  p.tilesY = dc_cdiv(p.Hout, Cfg::TH);
  dim3 grid((unsigned)(p.N * p.tilesX * p.tilesY * dc_cdiv(p.Ncols, Cfg::BN)));
  hipLaunchKernelGGL(kern, grid, dim3(256), Cfg::LDS_BYTES, st, p);
  DC_CHECK_LAUNCH(name);
  return DC_OK;
}

template <int KH, int KW, int S, int PAD, int TW, int WAVES_M, int MB, int NB, int CK>
static int igemm_tiles(int N, int Hout, int Wout) {
  using Cfg = IgemmCfg<KH, KW, S, PAD, TW, WAVES_M, MB, NB, CK>;
  return N * dc_cdiv(Wout, TW) * dc_cdiv(Hout, Cfg::TH);
}

// Tile-shape choice.  cfg A: 256 px x 64 cols, cfg B: 512 px x 32 cols (Cout <= 32), cfg C: 64 px x 128 cols
// (8x8 patches for the small feature maps of 96^2/128^2 training windows).
enum { CFG_A32, CFG_B32, CFG_A16, CFG_C8 };
static int pick_cfg(int Wout, int Ncols) {
  if (Wout > 16) return (Ncols <= 32) ? CFG_B32 : CFG_A32;
  if (Wout > 8) return CFG_A16;
  return CFG_C8;
}

#define IGEMM_DISPATCH(KH, KW, S, PAD, CK, CKB, CKC, FN, ...)                                 \
  switch (pick_cfg(Wout_, Ncols_)) {                                                          \
    case CFG_A32: return FN<KH, KW, S, PAD, 32, 4, 2, 2, CK>(__VA_ARGS__);                    \
    case CFG_B32: return FN<KH, KW, S, PAD, 32, 4, 4, 1, CKB>(__VA_ARGS__);                   \
    case CFG_A16: return FN<KH, KW, S, PAD, 16, 4, 2, 2, CK>(__VA_ARGS__);                    \
    default: return FN<KH, KW, S, PAD, 8, 2, 1, 2, CKC>(__VA_ARGS__);                         \
  }

static int conv3x3_tiles_impl(int N, int H, int W, int Cout) {
  const int Wout_ = W, Ncols_ = Cout;
  IGEMM_DISPATCH(3, 3, 1, 1, 16, 16, 8, igemm_tiles, N, H, W)
}
static int conv3x3_launch_impl(IgemmParams p, hipStream_t st) {
  const int Wout_ = p.Wout, Ncols_ = p.Ncols;
  IGEMM_DISPATCH(3, 3, 1, 1, 16, 16, 8, igemm_launch, p, st, "conv3x3")
}
static int convT_tiles_impl(int N, int H, int W, int Ncols) {
  const int Wout_ = W, Ncols_ = Ncols;
  IGEMM_DISPATCH(1, 1, 1, 0, 16, 16, 16, igemm_tiles, N, H, W)
}
static int convT_fwd_launch_impl(IgemmParams p, hipStream_t st) {
  const int Wout_ = p.Wout, Ncols_ = p.Ncols;
  IGEMM_DISPATCH(1, 1, 1, 0, 16, 16, 16, igemm_launch, p, st, "convT2x2_fwd")
}
static int convT_dgrad_launch_impl(IgemmParams p, hipStream_t st) {
  const int Wout_ = p.Wout, Ncols_ = p.Ncols;
  IGEMM_DISPATCH(2, 2, 2, 0, 8, 8, 8, igemm_launch, p, st, "convT2x2_dgrad")
}

// ---------------------------------------------------------------------------------------------------
__global__ void pack_weights_kernel(const float* __restrict__ src, float* __restrict__ dst, int taps, int K, int Ncols,
                                    long s_tap, long s_k, long s_n, int flip, long total) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int e = (int)(i & 3);
    long r = i >> 2;
    const int n = (int)(r % Ncols); r /= Ncols;
    const int kg = (int)(r % (K / 4));
    const int tap = (int)(r / (K / 4));
    const int k = kg * 4 + e;
    const int ts = flip ? (taps - 1 - tap) : tap;
    dst[i] = src[ts * s_tap + k * s_k + n * s_n];
  }
}

extern "C" int dc_pack_weights(const float* src, float* dst, int taps, int K, int Ncols, long s_tap, long s_k,
                               long s_n, int flip, dc_stream_t stream) {
  DC_REQUIRE(src && dst, DC_EINVAL, "dc_pack_weights: null pointer");
  DC_REQUIRE(taps > 0 && K > 0 && Ncols > 0 && K % 4 == 0, DC_EINVAL, "dc_pack_weights: K=%d must be a positive multiple of 4", K);
  const long total = (long)taps * K * Ncols;
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(pack_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, dst, taps, K, Ncols,
                     s_tap, s_k, s_n, flip, total);
  DC_CHECK_LAUNCH("dc_pack_weights");
  return DC_OK;
}

static int check_conv_args(const char* fn, const void* a, const void* b, const void* c, int N, int H, int W, int Cin,
                           int Cout) {
  DC_REQUIRE(a && b && c, DC_EINVAL, "%s: null pointer", fn);
  DC_REQUIRE(dc_aligned16(a) && dc_aligned16(b) && dc_aligned16(c), DC_EINVAL, "%s: pointers must be 16-byte aligned", fn);
  DC_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, DC_EINVAL, "%s: non-positive dimension", fn);
  DC_REQUIRE(Cin % 4 == 0, DC_EUNSUP, "%s: Cin=%d must be a multiple of 4", fn, Cin);
  return DC_OK;
}

extern "C" int dc_conv3x3_tiles(int N, int H, int W, int Cout) { return conv3x3_tiles_impl(N, H, W, Cout); }

extern "C" int dc_conv3x3_fwd(const float* x, const float* wp, const float* bias, float* z, long z_ld, double* stats,
                              const float* scale, const float* shift, int relu, int N, int H, int W, int Cin, int Cout,
                              dc_stream_t stream) {
  int rc = check_conv_args("dc_conv3x3_fwd", x, wp, z, N, H, W, Cin, Cout);
  if (rc) return rc;
  DC_REQUIRE((scale == nullptr) == (shift == nullptr), DC_EINVAL, "dc_conv3x3_fwd: scale and shift go together");
  IgemmParams p{};
  p.in = x; p.wp = wp; p.bias = bias; p.out = z; p.stats = stats; p.scale = scale; p.shift = shift;
  p.N = N; p.Hin = H; p.Win = W; p.Cin = Cin; p.Hout = H; p.Wout = W; p.Ncols = Cout;
  p.relu = relu; p.scatterCo = 0; p.biasMod = Cout; p.outLd = z_ld;
  DC_REQUIRE(z_ld >= Cout, DC_EINVAL, "dc_conv3x3_fwd: z_ld < Cout");
  return conv3x3_launch_impl(p, (hipStream_t)stream);
}

extern "C" int dc_conv3x3_dgrad(const float* dz, const float* wp, float* dx, int N, int H, int W, int Cin, int Cout,
                                dc_stream_t stream) {
  int rc = check_conv_args("dc_conv3x3_dgrad", dz, wp, dx, N, H, W, Cout, Cin);
  if (rc) return rc;
  IgemmParams p{};
  p.in = dz; p.wp = wp; p.out = dx;
  p.N = N; p.Hin = H; p.Win = W; p.Cin = Cout; p.Hout = H; p.Wout = W; p.Ncols = Cin;
  p.biasMod = Cin; p.outLd = Cin;
  return conv3x3_launch_impl(p, (hipStream_t)stream);
}

extern "C" int dc_convT2x2_tiles(int N, int H, int W, int Cout) { return convT_tiles_impl(N, H, W, 4 * Cout); }

extern "C" int dc_convT2x2_fwd(const float* x, const float* wp, const float* bias, float* z, long z_ld, double* stats,
                               const float* scale, const float* shift, int relu, int N, int H, int W, int Cin,
                               int Cout, dc_stream_t stream) {
  int rc = check_conv_args("dc_convT2x2_fwd", x, wp, z, N, H, W, Cin, Cout);
  if (rc) return rc;
  DC_REQUIRE((scale == nullptr) == (shift == nullptr), DC_EINVAL, "dc_convT2x2_fwd: scale and shift go together");
  IgemmParams p{};
  p.in = x; p.wp = wp; p.bias = bias; p.out = z; p.stats = stats; p.scale = scale; p.shift = shift;
  p.N = N; p.Hin = H; p.Win = W; p.Cin = Cin; p.Hout = H; p.Wout = W; p.Ncols = 4 * Cout;
  p.relu = relu; p.scatterCo = Cout; p.biasMod = Cout; p.outLd = z_ld;
  DC_REQUIRE(z_ld >= Cout, DC_EINVAL, "dc_convT2x2_fwd: z_ld < Cout");
  return convT_fwd_launch_impl(p, (hipStream_t)stream);
}

extern "C" int dc_convT2x2_dgrad(const float* dz, const float* wp, float* dx, int N, int H, int W, int Cin, int Cout,
                                 dc_stream_t stream) {
  int rc = check_conv_args("dc_convT2x2_dgrad", dz, wp, dx, N, H, W, Cout, Cin);
  if (rc) return rc;
  IgemmParams p{};
  p.in = dz; p.wp = wp; p.out = dx;
  p.N = N; p.Hin = 2 * H; p.Win = 2 * W; p.Cin = Cout; p.Hout = H; p.Wout = W; p.Ncols = Cin;
  p.biasMod = Cin; p.outLd = Cin;
  return convT_dgrad_launch_impl(p, (hipStream_t)stream);
}
