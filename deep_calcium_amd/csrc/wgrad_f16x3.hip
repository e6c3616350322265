// Weight-gradient contraction with fp32-grade accuracy on the fp16 matrix cores (numerics: igemm_f16x3.hip).
//
//   out[tap][m][n] = sum over pixels p of  A[S*p + tap - PAD][m] * B[p][n]
//   conv3x3 wgrad : A = layer input x (m = Cin), B = dz (n = Cout, pow2-scaled)  -> HWIO (3,3,Cin,Cout)
//   convT2x2 wgrad: A = dz (2H x 2W, m = Cout, S = 2, pow2-scaled), B = x (n = Cin) -> Keras (2,2,Cout,Cin)
//
// k = 16 pixels per v_mfma_f32_32x32x16_f16.  The contraction index (pixel) is the ROW index of the NHWC tiles, so
// the MFMA fragments (8 consecutive k per lane) come from the hardware-transposing ds_read_b64_tr_b16: each 16-lane
// group reads a 4-pixel x 16-channel block and every lane receives its channel's 4 pixels.  LDS holds fp16 hi and
// lo images of both tiles as 32-channel planes ([plane][pixel][32 ch] = 64-B rows): 4 consecutive pixel rows of a
// plane are 256 contiguous bytes => conflict-free transposed reads, every tap / k-step offset is an immediate.
//
// Pipeline (ONE 512-thread CTA per CU = two waves per SIMD with fixed roles):
//   * waves 0-3 (consumers) own the matrix pipe: a wave owns 32 m-channels x (32*NBW) n-channels for ALL taps, so
//     every A fragment pair feeds 3*NBW MFMAs; fragments of group g+1 are requested from LDS before the MFMAs of
//     group g issue.  With nothing else in their instruction stream they run at the MFMA-only rate (measured by
//     ablation: 440-480 TF/s-equivalent; the earlier single-role version, whose waves also staged, ran at the SUM
//     of its MFMA-only and staging-only times, 210-300).
//   * waves 4-7 (producers, the SIMD partners of 0-3) request the raw fp32 rows of tile t+2 (bounds-checked buffer
//     loads into a second register set) and then split tile t+1 into fp16 hi/lo and write it to the other LDS
//     image set: their VALU / VMEM / LDS-write work fills the issue slots the consumers' MFMAs leave free
//     (an MFMA holds the issue port 8 of its 32 cycles).
//   * two LDS image sets, one barrier per tile; the WK consumer waves that share an (m,n) block are summed through
//     LDS at the end, one slab per CTA goes to HBM and dc_reduce_partials adds the slabs in a fixed order
//     (bit-reproducible, no atomics).
#include "wgrad_common.h"
#include <stdlib.h>

// Experiment switches (scripts/wgrad_variants.py builds copies of the library with them; the shipped build uses the defaults).
// DC_WG_ABL (ablation builds only, results are garbage): bit 0 producers stage only the first two tiles (no split / LDS
// writes in the loop), bit 1 producers request only the first two tiles (no global loads in the loop), bit 2 consumers
// issue no fragment reads / MFMAs (producers alone), bit 3 no slab store, bit 4 the producers write the raw fp32 bits instead of
// the fp16 split (what a PRE-SPLIT operand would cost: the LDS writes without the VALU).  DC_WG_CLOCK: wave 0 of workgroup 0 stamps
// s_memtime / s_memrealtime around its loop (dc_debug_wgrad_stamps): the clock the chip holds in this kernel.
#ifndef DC_WG_ABL
#define DC_WG_ABL 0
#endif
#ifndef DC_WG_RW
#define DC_WG_RW 4          // pixel rows per 16-wide tile of the Cin, Cout > 32 instantiation
#endif
#ifndef DC_WG_PRIO
#define DC_WG_PRIO 0        // s_setprio of the consumer waves
#endif
#ifndef DC_WG_DEPTH
#define DC_WG_DEPTH 1       // (k-step, tap) groups the A fragments are requested ahead of their MFMAs (DC_WG_XTILE=0 only)
#endif
#ifndef DC_WG_XTILE
// 1: the consumers' fragments run ahead ACROSS tiles (barrier two groups before the end of a tile, the next tile's first fragments
// requested behind it).  Built, bit-identical, measured in round 6 and NOT adopted: the loop takes MORE cycles (265 000 against
// 258 000 per 64 tiles) and the step 17.20-17.27 against 17.10-17.18 ms -- the producers are co-critical in this kernel (their own
// loop is 85 % of the full one), so a consumer that reaches the barrier earlier only waits there (profiles/r06_wgrad_ablation.txt).
#define DC_WG_XTILE 0
#endif
#ifdef DC_WG_CLOCK
__device__ unsigned long long g_wg_stamps[4];
// per workgroup: s_memrealtime (the chip-wide 100 MHz counter) at kernel entry, first tile ready, tile loop done, slab stored
__device__ unsigned long long g_wg_timeline[1024 * 4];
extern "C" int dc_debug_wgrad_stamps(unsigned long long* out4) {
  return hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_wg_stamps), sizeof(g_wg_stamps)) == hipSuccess ? 0 : -2;
}
extern "C" int dc_debug_wgrad_timeline(unsigned long long* out4096) {
  return hipMemcpyFromSymbol(out4096, HIP_SYMBOL(g_wg_timeline), sizeof(g_wg_timeline)) == hipSuccess ? 0 : -2;
}
#define WG_TL(i) do { if (tid == 0 && blockIdx.x < 1024) g_wg_timeline[blockIdx.x * 4 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define WG_TL(i) do {} while (0)
#endif

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((__vector_size__(4 * sizeof(short)))) short tr_v4i16;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

struct WgradHParams {
  WgradParams g;
  const float* aScale;  // nullable device scalars (powers of two)
  const float* bScale;
  // BatchNorm + ReLU on load for the operand that is a NON-materialised activation (conv3x3: A = x; convT2x2: B = x):
  // that operand pointer then holds the producer's pre-BN tensor z and relu(fmaf(z, sc[c], sh[c])) is formed by the
  // producer waves while they split it (zero padding stays zero).  Nullable, per channel of that operand.
  const float* aSc; const float* aSh;
  const float* bSc; const float* bSh;
  // fp16 range guard of the ACTIVATION operand (common.h dc_block_guard_scale): its per-channel magnitude bound
  // (xChannels floats, nullable).  The gradient operand brings its scale in aScale / bScale.
  const float* xAbound; int xChannels;
  // "dz on load" (conv3x3 only, DZIN instantiations): B holds da, bZ the block's pre-BN tensor z (same layout) and the
  // producer waves form dz = fmaf(A, [fmaf(z,sc,sh) > 0] * da, fmaf(D, z - mu, E)) from dzCoef[DC_DZ_COEF_ROWS][Cn]
  // (dc_bn_bwd_finalize_dzin); row 6 is the magnitude bound the operand's power-of-two scale comes from
  const float* bZ;
  const float* dzCoef;
};

// WM x WNW waves tile the CTA's (m, n) block, each wave covering 32 m x (32*NBW) n; the remaining
// WK = 4/(WM*WNW) waves split the pixel rows of a tile.
template <int KH, int KW, int S, int PAD, int TW, int RW, int WM, int WNW, int NBW>
struct WgradHCfg {
  static constexpr int TAPS = KH * KW;
  static constexpr int WN = WNW * NBW;              // 32-column blocks of the CTA
  static constexpr int WK = 4 / (WM * WNW);
  static constexpr int TH = WK * RW;
  static constexpr int CM = 32 * WM, CN = 32 * WN;
  static constexpr int THI = (TH - 1) * S + KH, TWI = (TW - 1) * S + KW;
  static constexpr int APIX = THI * TWI, BPIX = TH * TW;
  static constexpr int A_PLANE = APIX * 64, B_PLANE = BPIX * 64;
  static constexpr int A_IMG = WM * A_PLANE, B_IMG = WN * B_PLANE;
  static constexpr int SET = 2 * (A_IMG + B_IMG);
  static constexpr int LDS_BYTES = 2 * SET;
  static_assert(LDS_BYTES <= 160 * 1024, "two image sets must fit the 160 KiB LDS");
};

__device__ __forceinline__ f16x8 tr_frag(const char* base, int off1, int off2) {
  typedef __attribute__((address_space(3))) tr_v4i16* lds_p;
  const tr_v4i16 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(base + off1));
  const tr_v4i16 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(base + off2));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  s16x8 v = {r0[0], r0[1], r0[2], r0[3], r1[0], r1[1], r1[2], r1[3]};
  return __builtin_bit_cast(f16x8, v);
}

// hi = fp16(x*s), lo = fp16(x*s - hi): two v_fma_mix per element (see split_f16 in igemm_f16x3.hip)
template <bool SCALED>
__device__ __forceinline__ void split4_f16(const f32x4 v, float s, u32x2& hi, u32x2& lo) {
  if (DC_WG_ABL & 16) {      // ablation: an operand stored PRE-SPLIT in HBM -- the same LDS writes, none of the split VALU
    hi = u32x2{__builtin_bit_cast(unsigned, v[0]), __builtin_bit_cast(unsigned, v[1])};
    lo = u32x2{__builtin_bit_cast(unsigned, v[2]), __builtin_bit_cast(unsigned, v[3])};
    return;
  }
  unsigned h01, h23, l01, l23;
  asm("v_fma_mixlo_f16 %0, %4, %8, 0\n\t"
      "v_fma_mixlo_f16 %1, %6, %8, 0\n\t"
      "v_fma_mixhi_f16 %0, %5, %8, 0\n\t"
      "v_fma_mixhi_f16 %1, %7, %8, 0\n\t"
      "v_fma_mixlo_f16 %2, %4, %8, -%0 op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixlo_f16 %3, %6, %8, -%1 op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %2, %5, %8, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
      "v_fma_mixhi_f16 %3, %7, %8, -%1 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
      : "=&v"(h01), "=&v"(h23), "=&v"(l01), "=&v"(l23)
      : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(s));
  hi = u32x2{h01, h23};
  lo = u32x2{l01, l23};
}

template <int KH, int KW, int S, int PAD, int TW, int RW, int WM, int WNW, int NBW, bool A_SCALED, bool DZIN = false>
__global__ __launch_bounds__(512, 1) void wgrad_f16x3_kernel(WgradHParams hp) {
  static_assert(!(DZIN && A_SCALED), "dz on load: conv3x3 (B = dz) only");
  using Cfg = WgradHCfg<KH, KW, S, PAD, TW, RW, WM, WNW, NBW>;
  constexpr int TAPS = Cfg::TAPS, WK = Cfg::WK, TH = Cfg::TH, CM = Cfg::CM, CN = Cfg::CN;
  constexpr int TWI = Cfg::TWI, APIX = Cfg::APIX, BPIX = Cfg::BPIX;
  constexpr int A_PLANE = Cfg::A_PLANE, B_PLANE = Cfg::B_PLANE, A_IMG = Cfg::A_IMG, B_IMG = Cfg::B_IMG, SET = Cfg::SET;
  constexpr int KROWS = (TW >= 16) ? 1 : 16 / TW;               // pixel rows covered by one 16-pixel k-step
  constexpr int KX = (TW >= 16) ? TW / 16 : 1;                  // k-steps along a row
  constexpr int AC4 = CM / 4, BC4 = CN / 4;
  constexpr int NA = (APIX * AC4 + 255) / 256, NB = (BPIX * BC4 + 255) / 256;   // float4 loads per producer thread
  constexpr int KSTEPS = (RW / KROWS) * KX;
  constexpr int GROUPS = KSTEPS * TAPS;                         // (k-step, tap) MFMA groups per tile
  static_assert(RW % KROWS == 0, "rows per wave must be a multiple of the k-step height");
  const WgradParams& p = hp.g;

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);    // 0..7, provably wave-uniform
  WG_TL(0);
  // operand scales (powers of two, undone in the epilogue): the gradient operand's device scalar, the activation
  // operand's range guard -- every wave needs them (producers to split, consumers to un-scale)
  const float x_scale = dc_block_guard_scale(hp.xAbound, hp.xChannels, reinterpret_cast<float*>(smem));
  const float a_scale = A_SCALED ? (hp.aScale ? *hp.aScale : 1.f) : x_scale;
  const float b_scale = DZIN ? dc_block_guard_scale(hp.dzCoef + 6 * p.Cn, p.Cn, reinterpret_cast<float*>(smem))
                             : (A_SCALED ? x_scale : (hp.bScale ? *hp.bScale : 1.f));
  // XCD-aware rasterisation (speed only): ids b and b+8 share an L2, so each XCD walks a contiguous range of
  // (pixel split, channel block) pairs with the channel block fastest -- the CTAs that stream the same pixel range
  // for different (m,n) blocks run side by side on one L2.
  const int nbm = (p.Cm + CM - 1) / CM, nbn = (p.Cn + CN - 1) / CN, total = (int)gridDim.x;
  const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3, qq = total >> 3, rr8 = total & 7;
  const int work = (xcd < rr8 ? xcd * (qq + 1) : rr8 * (qq + 1) + (xcd - rr8) * qq) + seq;
  const int split = work / (nbm * nbn), blk = work - split * (nbm * nbn);
  const int m0 = (blk / nbn) * CM, n0 = (blk % nbn) * CN;
  const int tile_beg = split * p.tilesPerSplit;
  const int nt = min(tile_beg + p.tilesPerSplit, p.tilesTotal) - tile_beg;   // tiles of this CTA (>= 1)

  if (wave >= 4) {
    // ================= producer waves: HBM -> registers -> fp16 hi/lo images in LDS =========================
    const int st = tid & 255;
    const int a_c4 = st % AC4, b_c4 = st % BC4;
    const bool a_ch_ok = (m0 + 4 * a_c4) < p.Cm, b_ch_ok = (n0 + 4 * b_c4) < p.Cn;
    const int a_rowb = p.Wa * p.Cm * 4, b_rowb = p.Wb * p.Cn * 4;
    const int a_lbase = (a_c4 >> 3) * A_PLANE + (a_c4 & 7) * 8;
    const int b_lbase = 2 * A_IMG + (b_c4 >> 3) * B_PLANE + (b_c4 & 7) * 8;
    const bool a_bn = hp.aSc != nullptr, b_bn = hp.bSc != nullptr;
    f32x4 a_sc = {1.f, 1.f, 1.f, 1.f}, a_sh = {0.f, 0.f, 0.f, 0.f}, b_sc = a_sc, b_sh = a_sh;
    if (a_bn && a_ch_ok) {
      a_sc = *reinterpret_cast<const f32x4*>(hp.aSc + m0 + 4 * a_c4);
      a_sh = *reinterpret_cast<const f32x4*>(hp.aSh + m0 + 4 * a_c4);
    }
    if (b_bn && b_ch_ok) {
      b_sc = *reinterpret_cast<const f32x4*>(hp.bSc + n0 + 4 * b_c4);
      b_sh = *reinterpret_cast<const f32x4*>(hp.bSh + n0 + 4 * b_c4);
    }
    f32x4 d_mu = b_sh, d_A = b_sh, d_D = b_sh, d_E = b_sh;     // DZIN: this thread's channel quad of the dz table
    if (DZIN && b_ch_ok) {
      const float* t = hp.dzCoef + n0 + 4 * b_c4;
      b_sc = *reinterpret_cast<const f32x4*>(t);
      b_sh = *reinterpret_cast<const f32x4*>(t + p.Cn);
      d_mu = *reinterpret_cast<const f32x4*>(t + 2 * p.Cn);
      d_A = *reinterpret_cast<const f32x4*>(t + 3 * p.Cn);
      d_D = *reinterpret_cast<const f32x4*>(t + 4 * p.Cn);
      d_E = *reinterpret_cast<const f32x4*>(t + 5 * p.Cn);
    }

    // request tile `tile`'s raw rows (zeros outside the image: rows via the descriptor bounds, columns by compare)
    // ma / mb: per-load 'inside the image' bits, needed only when BN + ReLU / dz is formed on load (0 must stay 0)
    constexpr int NZ = DZIN ? NB : 1;
    struct TilePos { int img, py0, px0; };
    auto tile_pos = [&](int tile) {
      TilePos t;
      int tx, ty;
      if (p.walk) { ty = tile % p.tilesY; const int q = tile / p.tilesY; tx = q % p.tilesX; t.img = q / p.tilesX; }
      else { tx = tile % p.tilesX; const int q = tile / p.tilesX; ty = q % p.tilesY; t.img = q / p.tilesY; }
      t.py0 = ty * TH; t.px0 = tx * TW;
      return t;
    };
    auto request_a = [&](int tile, f32x4 (&ra)[NA], unsigned& ma) {
      ma = 0u;
      const TilePos t = tile_pos(tile);
      const int ay0 = t.py0 * S - PAD, ax0 = t.px0 * S - PAD;
      const __amdgpu_buffer_rsrc_t rsA = dc_make_rsrc(p.A + (long)t.img * p.Ha * p.Wa * p.Cm, (unsigned)(p.Ha * p.Wa * p.Cm) * 4u);
      const int abase = (ay0 * p.Wa + ax0) * p.Cm * 4 + (m0 + 4 * a_c4) * 4;   // may be negative: wraps out of range
#pragma unroll
      for (int k = 0; k < NA; ++k) {
        const int pix = st / AC4 + k * (256 / AC4);
        const int r = __umul24(pix, (65536 + TWI - 1) / TWI) >> 16;          // pix / TWI for pix < 4096
        const int cc = pix - __umul24(r, TWI);
        const bool ok = a_ch_ok && pix < APIX && (unsigned)(ax0 + cc) < (unsigned)p.Wa;
        const unsigned off = ok ? (unsigned)(abase + __mul24(r, a_rowb) + __mul24(cc, p.Cm * 4)) : 0x80000000u;
        ra[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, off, 0, 0));
        if (a_bn && ok && (unsigned)(ay0 + r) < (unsigned)p.Ha) ma |= 1u << k;
      }
    };
    auto request_b = [&](int tile, f32x4 (&rb)[NB], f32x4 (&rz)[NZ], unsigned& mb) {
      mb = 0u;
      const TilePos t = tile_pos(tile);
      const __amdgpu_buffer_rsrc_t rsB = dc_make_rsrc(p.B + (long)t.img * p.Hb * p.Wb * p.Cn, (unsigned)(p.Hb * p.Wb * p.Cn) * 4u);
      const __amdgpu_buffer_rsrc_t rsZ = DZIN ? dc_make_rsrc(hp.bZ + (long)t.img * p.Hb * p.Wb * p.Cn, (unsigned)(p.Hb * p.Wb * p.Cn) * 4u) : rsB;
      const int bbase = (t.py0 * p.Wb + t.px0) * p.Cn * 4 + (n0 + 4 * b_c4) * 4;
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        const int pix = st / BC4 + k * (256 / BC4);
        const int r = pix / TW, cc = pix % TW;
        const bool ok = b_ch_ok && pix < BPIX && (t.px0 + cc) < p.Wb;
        const unsigned off = ok ? (unsigned)(bbase + __mul24(r, b_rowb) + __mul24(cc, p.Cn * 4)) : 0x80000000u;
        rb[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsB, off, 0, 0));
        if constexpr (DZIN) rz[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsZ, off, 0, 0));
        if ((b_bn || DZIN) && ok && (t.py0 + r) < p.Hb) mb |= 1u << k;
      }
    };
    // split the requested rows into fp16 hi/lo and write them into image set `set`
    auto bn_relu = [&](f32x4 v, const f32x4& sc, const f32x4& sh, bool live) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = live ? fmaxf(__builtin_fmaf(v[e], sc[e], sh[e]), 0.f) : 0.f;
      return v;
    };
    auto dz_on_load = [&](const f32x4 da, const f32x4 zz, bool live) {
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float y = __builtin_fmaf(zz[e], b_sc[e], b_sh[e]);           // the forward's own expression: identical ReLU gate
        const float dy = y > 0.f ? da[e] : 0.f;
        const float dzv = __builtin_fmaf(d_A[e], dy, __builtin_fmaf(d_D[e], zz[e] - d_mu[e], d_E[e]));
        v[e] = live ? dzv : 0.f;
      }
      return v;
    };
    auto stage_a = [&](const f32x4 (&ra)[NA], unsigned ma, char* set) {
#pragma unroll
      for (int j = 0; j < NA; ++j) {
        const int pix = st / AC4 + j * (256 / AC4);
        u32x2 hi, lo;
        split4_f16<true>(a_bn ? bn_relu(ra[j], a_sc, a_sh, (ma >> j) & 1u) : ra[j], a_scale, hi, lo);
        if (pix < APIX) {
          *reinterpret_cast<u32x2*>(set + a_lbase + pix * 64) = hi;
          *reinterpret_cast<u32x2*>(set + A_IMG + a_lbase + pix * 64) = lo;
        }
      }
    };
    auto stage_b = [&](const f32x4 (&rb)[NB], const f32x4 (&rz)[NZ], unsigned mb, char* set) {
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        const int pix = st / BC4 + k * (256 / BC4);
        u32x2 hi, lo;
        f32x4 bv = rb[k];
        if constexpr (DZIN) bv = dz_on_load(rb[k], rz[k], (mb >> k) & 1u);
        else if (b_bn) bv = bn_relu(rb[k], b_sc, b_sh, (mb >> k) & 1u);
        split4_f16<true>(bv, b_scale, hi, lo);
        if (pix < BPIX) {
          *reinterpret_cast<u32x2*>(set + b_lbase + pix * 64) = hi;
          *reinterpret_cast<u32x2*>(set + B_IMG + b_lbase + pix * 64) = lo;
        }
      }
    };

    // Two register sets: while tile i+1 is split into LDS, tile i+2's rows are already in flight (a full tile of
    // MFMA time plus the split covers the HBM latency even when a bandwidth-bound kernel shares the chip).
    // B1 (the DZIN instantiations with 32-wide tiles: 2 x (A + da + z) would need > 256 VGPRs): ONE (da, z) set, requested
    // right after the previous tile's has been split -- one tile of MFMA time ahead instead of two.
    constexpr bool B1 = DZIN && 8 * (NA + NB + NZ) > 160;
    f32x4 ra0[NA], rb0[NB], rz0[NZ], ra1[NA], rb1[B1 ? 1 : NB], rz1[B1 ? 1 : NZ];
    unsigned ma0, mb0, ma1 = 0u, mb1 = 0u;
    if constexpr (B1) {
      request_a(tile_beg, ra0, ma0);
      request_b(tile_beg, rb0, rz0, mb0);
      if (nt > 1) request_a(tile_beg + 1, ra1, ma1);
      stage_a(ra0, ma0, smem);
      stage_b(rb0, rz0, mb0, smem);
      if (nt > 1) request_b(tile_beg + 1, rb0, rz0, mb0);
      __syncthreads();
      for (int i = 0; i < nt; i += 2) {
        if (i + 2 < nt) request_a(tile_beg + i + 2, ra0, ma0);
        if (i + 1 < nt) { stage_a(ra1, ma1, smem + SET); stage_b(rb0, rz0, mb0, smem + SET); }
        if (i + 2 < nt) request_b(tile_beg + i + 2, rb0, rz0, mb0);
        __syncthreads();
        if (i + 1 < nt) {
          if (i + 3 < nt) request_a(tile_beg + i + 3, ra1, ma1);
          if (i + 2 < nt) { stage_a(ra0, ma0, smem); stage_b(rb0, rz0, mb0, smem); }
          if (i + 3 < nt) request_b(tile_beg + i + 3, rb0, rz0, mb0);
          __syncthreads();
        }
      }
    } else {
      auto request = [&](int tile, f32x4 (&ra)[NA], f32x4 (&rb)[NB], f32x4 (&rz)[NZ], unsigned& ma, unsigned& mb) {
        request_a(tile, ra, ma);
        request_b(tile, rb, rz, mb);
      };
      auto stage = [&](const f32x4 (&ra)[NA], const f32x4 (&rb)[NB], const f32x4 (&rz)[NZ], unsigned ma, unsigned mb, char* set) {
        stage_a(ra, ma, set);
        stage_b(rb, rz, mb, set);
      };
      f32x4 (&rb1r)[NB] = reinterpret_cast<f32x4 (&)[NB]>(rb1);
      f32x4 (&rz1r)[NZ] = reinterpret_cast<f32x4 (&)[NZ]>(rz1);
      request(tile_beg, ra0, rb0, rz0, ma0, mb0);
      if (nt > 1) request(tile_beg + 1, ra1, rb1r, rz1r, ma1, mb1);
      stage(ra0, rb0, rz0, ma0, mb0, smem);
      __syncthreads();
      for (int i = 0; i < nt; i += 2) {
        if (i + 2 < nt && !(DC_WG_ABL & 2)) request(tile_beg + i + 2, ra0, rb0, rz0, ma0, mb0);
        if (i + 1 < nt && !((DC_WG_ABL & 1) && i > 0)) stage(ra1, rb1r, rz1r, ma1, mb1, smem + SET);
        __syncthreads();
        if (i + 1 < nt) {
          if (i + 3 < nt && !(DC_WG_ABL & 2)) request(tile_beg + i + 3, ra1, rb1r, rz1r, ma1, mb1);
          if (i + 2 < nt && !(DC_WG_ABL & 1)) stage(ra0, rb0, rz0, ma0, mb0, smem);
          __syncthreads();
        }
      }
    }
    if constexpr (WK > 1) {   // keep in step with the barriers of the consumers' cross-wave reduction
#pragma unroll 1
      for (int k = 0; k < 2 * TAPS * NBW; ++k) __syncthreads();
    }
    return;
  }

  // ===================== consumer waves: transposed LDS fragments -> MFMA ====================================
  const int h = lane >> 5, cb = (lane >> 4) & 1, c = lane & 15, q = c >> 2, pp = c & 3;
  const int wm = wave % WM, wnw = (wave / WM) % WNW, wk = wave / (WM * WNW);

  // lane-constant byte offsets of the two transposed reads of a k-step whose first pixel is (row 0, x 0)
  int offA[2], offB[2];
#pragma unroll
  for (int r2 = 0; r2 < 2; ++r2) {
    const int kpix = 8 * h + q + 4 * r2;
    const int ky = (TW >= 16) ? 0 : kpix / TW, kx = (TW >= 16) ? kpix : kpix % TW;
    offA[r2] = wm * A_PLANE + ((ky * S) * TWI + kx * S) * 64 + cb * 32 + pp * 8;
    offB[r2] = 2 * A_IMG + (ky * TW + kx) * 64 + cb * 32 + pp * 8;
  }

  f32x16 acc[NBW][TAPS];
#pragma unroll
  for (int w = 0; w < NBW; ++w)
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[w][t][r] = 0.f;

  // byte offset of group g's A fragment block / of k-step ks's B fragment block
  auto a_off = [&](int g) {
    const int ks = g / TAPS, tap = g % TAPS;
    const int prow = wk * RW + (ks / KX) * KROWS, xs = ks % KX;
    return ((prow * S) * TWI + xs * 16 * S) * 64 + ((tap / KW) * TWI + (tap % KW)) * 64;
  };
  auto b_off = [&](int ks, int w) {
    const int prow = wk * RW + (ks / KX) * KROWS, xs = ks % KX;
    return (wnw * NBW + w) * B_PLANE + (prow * TW + xs * 16) * 64;
  };

  __syncthreads();   // image set 0 is ready
  WG_TL(1);
  if (DC_WG_PRIO) __builtin_amdgcn_s_setprio(DC_WG_PRIO);
#ifdef DC_WG_CLOCK
  unsigned long long t0c = 0, t0r = 0;
  if (blockIdx.x == 0 && tid == 0) { t0c = __builtin_amdgcn_s_memtime(); t0r = __builtin_amdgcn_s_memrealtime(); }
#endif
#if DC_WG_XTILE
  // Fragments run AHEAD ACROSS TILES.  A tile's last reads are issued RING - 1 groups before its end; the tile barrier sits right
  // there -- "my reads of this image set have landed, the producers have finished the other set" -- and the groups that remain
  // (their fragments already in registers) cover the LDS latency of the NEXT tile's first fragments, which are requested from the
  // other set straight behind the barrier.  The earlier loop took the barrier at the very end of a tile and started the next one
  // with a burst of reads nothing covered: ~350 of a tile's ~3 800 cycles (profiles/r06_wgrad_ablation.txt: consumers alone 90.7 %).
  // Same MFMAs on the same fragments in the same order: bit-identical results.
  constexpr int RING = (GROUPS % 3 == 0) ? 3 : ((GROUPS % 4 == 0) ? 4 : 2), DEPTH = RING - 1;
  static_assert(GROUPS % RING == 0 && GROUPS > 2 * DEPTH, "the fragment ring must come round once per tile");
  static_assert(KSTEPS % 2 == 0 || KSTEPS == 1, "the B fragment double buffer must come round once per tile");
  static_assert(DEPTH <= TAPS, "the next tile's B fragments go to buffer 0: its last reader (k-step KSTEPS - 2) must be done");
  f16x8 ah[RING], al[RING], bh[2][NBW], bl[2][NBW];
  if (!(DC_WG_ABL & 4)) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      ah[d] = tr_frag(smem, offA[0] + a_off(d), offA[1] + a_off(d));
      al[d] = tr_frag(smem + A_IMG, offA[0] + a_off(d), offA[1] + a_off(d));
    }
#pragma unroll
    for (int w = 0; w < NBW; ++w) {
      bh[0][w] = tr_frag(smem, offB[0] + b_off(0, w), offB[1] + b_off(0, w));
      bl[0][w] = tr_frag(smem + B_IMG, offB[0] + b_off(0, w), offB[1] + b_off(0, w));
    }
  }
  for (int i = 0; i < nt; ++i) {
    const char* cur = smem + (i & 1) * SET;
    const char* nxt = smem + ((i + 1) & 1) * SET;
    const bool more = i + 1 < nt;
    if (DC_WG_ABL & 4) { __syncthreads(); continue; }
#pragma unroll
    for (int g = 0; g < GROUPS; ++g) {
      const int ca = g % RING, ks = g / TAPS, tap = g % TAPS, cbuf = (KSTEPS == 1) ? 0 : (ks & 1);
      if (g == GROUPS - DEPTH) __syncthreads();     // every read of this set has been issued (and is waited for here)
      const int gn = g + DEPTH;
      if (gn < GROUPS) {
        ah[gn % RING] = tr_frag(cur, offA[0] + a_off(gn), offA[1] + a_off(gn));
        al[gn % RING] = tr_frag(cur + A_IMG, offA[0] + a_off(gn), offA[1] + a_off(gn));
      } else if (more) {                            // the next tile's first groups, from the other image set
        ah[gn % RING] = tr_frag(nxt, offA[0] + a_off(gn - GROUPS), offA[1] + a_off(gn - GROUPS));
        al[gn % RING] = tr_frag(nxt + A_IMG, offA[0] + a_off(gn - GROUPS), offA[1] + a_off(gn - GROUPS));
      }
      if (KSTEPS > 1 && tap == 0 && ks + 1 < KSTEPS) {           // the next k-step's B fragments: the other buffer is free
#pragma unroll
        for (int w = 0; w < NBW; ++w) {
          bh[cbuf ^ 1][w] = tr_frag(cur, offB[0] + b_off(ks + 1, w), offB[1] + b_off(ks + 1, w));
          bl[cbuf ^ 1][w] = tr_frag(cur + B_IMG, offB[0] + b_off(ks + 1, w), offB[1] + b_off(ks + 1, w));
        }
      }
      if (KSTEPS > 1 && g == GROUPS - DEPTH && more) {           // the next tile's k-step 0 (buffer 0: k-step KSTEPS - 2 is long done)
#pragma unroll
        for (int w = 0; w < NBW; ++w) {
          bh[0][w] = tr_frag(nxt, offB[0] + b_off(0, w), offB[1] + b_off(0, w));
          bl[0][w] = tr_frag(nxt + B_IMG, offB[0] + b_off(0, w), offB[1] + b_off(0, w));
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int w = 0; w < NBW; ++w) {
        acc[w][tap] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ca], bh[cbuf][w], acc[w][tap], 0, 0, 0);
        acc[w][tap] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ca], bl[cbuf][w], acc[w][tap], 0, 0, 0);
        acc[w][tap] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ca], bh[cbuf][w], acc[w][tap], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (KSTEPS == 1 && more) {                      // one k-step per tile: its only B buffer is free only now
#pragma unroll
      for (int w = 0; w < NBW; ++w) {
        bh[0][w] = tr_frag(nxt, offB[0] + b_off(0, w), offB[1] + b_off(0, w));
        bl[0][w] = tr_frag(nxt + B_IMG, offB[0] + b_off(0, w), offB[1] + b_off(0, w));
      }
    }
  }
#else
  constexpr int DEPTH = DC_WG_DEPTH, RING = DEPTH + 1;
  for (int i = 0; i < nt; ++i) {
    char* cur = smem + (i & 1) * SET;
    if (!(DC_WG_ABL & 4)) {
    f16x8 ah[RING], al[RING], bh[2][NBW], bl[2][NBW];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      if (d < GROUPS) {
        ah[d] = tr_frag(cur, offA[0] + a_off(d), offA[1] + a_off(d));
        al[d] = tr_frag(cur + A_IMG, offA[0] + a_off(d), offA[1] + a_off(d));
      }
    }
#pragma unroll
    for (int w = 0; w < NBW; ++w) {
      bh[0][w] = tr_frag(cur, offB[0] + b_off(0, w), offB[1] + b_off(0, w));
      bl[0][w] = tr_frag(cur + B_IMG, offB[0] + b_off(0, w), offB[1] + b_off(0, w));
    }
#pragma unroll
    for (int g = 0; g < GROUPS; ++g) {
      const int ca = g % RING, ks = g / TAPS, tap = g % TAPS, cbuf = ks & 1;
      if (g + DEPTH < GROUPS) {  // fragments of group g + DEPTH are requested before the MFMAs of group g issue
        ah[(g + DEPTH) % RING] = tr_frag(cur, offA[0] + a_off(g + DEPTH), offA[1] + a_off(g + DEPTH));
        al[(g + DEPTH) % RING] = tr_frag(cur + A_IMG, offA[0] + a_off(g + DEPTH), offA[1] + a_off(g + DEPTH));
      }
      // the B fragments of the next k-step: one group ahead (DEPTH 1) or at the head of this k-step (the other buffer is free)
      if ((DEPTH == 1 ? (g + 1) % TAPS == 0 : tap == 0) && (ks + 1) * TAPS < GROUPS) {
#pragma unroll
        for (int w = 0; w < NBW; ++w) {
          bh[cbuf ^ 1][w] = tr_frag(cur, offB[0] + b_off(ks + 1, w), offB[1] + b_off(ks + 1, w));
          bl[cbuf ^ 1][w] = tr_frag(cur + B_IMG, offB[0] + b_off(ks + 1, w), offB[1] + b_off(ks + 1, w));
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int w = 0; w < NBW; ++w) {
        acc[w][tap] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ca], bh[cbuf][w], acc[w][tap], 0, 0, 0);
        acc[w][tap] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ca], bl[cbuf][w], acc[w][tap], 0, 0, 0);
        acc[w][tap] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ca], bh[cbuf][w], acc[w][tap], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    }
    __syncthreads();   // this set may be overwritten, the other one is complete
  }
#endif
  if (DC_WG_PRIO) __builtin_amdgcn_s_setprio(0);
  WG_TL(2);
#ifdef DC_WG_CLOCK
  if (blockIdx.x == 0 && tid == 0) {
    g_wg_stamps[0] = __builtin_amdgcn_s_memtime() - t0c;
    g_wg_stamps[1] = __builtin_amdgcn_s_memrealtime() - t0r;
    g_wg_stamps[2] = (unsigned long long)nt;
    g_wg_stamps[3] = (unsigned long long)GROUPS;
  }
#endif

  const float out_scale = 1.f / (a_scale * b_scale);
#pragma unroll
  for (int w = 0; w < NBW; ++w)   // one 32x32 block at a time through the shared cross-wave reduction + store
    if (WK > 1 || !(DC_WG_ABL & 8) || p.tilesTotal < 0) wgrad_store<TAPS, WM, WNW, WK>(p, acc[w], smem, split, m0, n0 + 32 * (wnw * NBW + w) - 32 * wnw, wm, wnw, wk, lane,
                                   out_scale);
#ifdef DC_WG_CLOCK
  __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0): the slab stores have left
  WG_TL(3);
#endif
}

// ---------------------------------------------------------------------------------------------------
#ifndef DC_WGRAD_CTAS
#define DC_WGRAD_CTAS 256   // one workgroup per CU, ONE round: half the slab traffic of 512 and -0.33 ms/step end to end
#endif
struct WgradHPlan {
  int splits, tilesX, tilesY, tilesTotal, tilesPerSplit;
};

template <int KH, int KW, int S, int PAD, int TW, int RW, int WM, int WNW, int NBW>
static WgradHPlan wgrad_h_plan(int N, int Hb, int Wb, int Cm, int Cn) {
  using Cfg = WgradHCfg<KH, KW, S, PAD, TW, RW, WM, WNW, NBW>;
  WgradHPlan pl;
  pl.tilesX = dc_cdiv(Wb, TW);
  pl.tilesY = dc_cdiv(Hb, Cfg::TH);
  pl.tilesTotal = N * pl.tilesX * pl.tilesY;
  const int blocks_mn = dc_cdiv(Cm, Cfg::CM) * dc_cdiv(Cn, Cfg::CN);
  int want = dc_cdiv(DC_WGRAD_CTAS, blocks_mn);   // one CTA per CU
  if (want > pl.tilesTotal) want = pl.tilesTotal;
  if (want < 1) want = 1;
  pl.tilesPerSplit = dc_cdiv(pl.tilesTotal, want);
  pl.splits = dc_cdiv(pl.tilesTotal, pl.tilesPerSplit);
  return pl;
}

template <int KH, int KW, int S, int PAD, int TW, int RW, int WM, int WNW, int NBW>
static long wgrad_h_ws(int N, int Hb, int Wb, int Cm, int Cn) {
  WgradHPlan pl = wgrad_h_plan<KH, KW, S, PAD, TW, RW, WM, WNW, NBW>(N, Hb, Wb, Cm, Cn);
  const long L = (long)KH * KW * Cm * Cn;
  return (long)pl.splits * L + 32 * L;
}

template <int KH, int KW, int S, int PAD, int TW, int RW, int WM, int WNW, int NBW, bool A_SCALED, bool DZIN = false>
static int wgrad_h_launch(const float* A, const float* B, float* dw, float* ws, const float* aScale, const float* bScale,
                          const float* xSc, const float* xSh, const float* xAbound, int N, int Ha, int Wa, int Hb, int Wb, int Cm, int Cn,
                          hipStream_t st, const char* name, const float* bZ = nullptr, const float* dzCoef = nullptr) {
  using Cfg = WgradHCfg<KH, KW, S, PAD, TW, RW, WM, WNW, NBW>;
  auto kern = wgrad_f16x3_kernel<KH, KW, S, PAD, TW, RW, WM, WNW, NBW, A_SCALED, DZIN>;
  static DcLdsAttr lds_attr;      // one per template instantiation; per-device inside
  if (int rc = dc_func_max_lds(lds_attr, reinterpret_cast<const void*>(kern), Cfg::LDS_BYTES, name)) return rc;
  WgradHPlan pl = wgrad_h_plan<KH, KW, S, PAD, TW, RW, WM, WNW, NBW>(N, Hb, Wb, Cm, Cn);
  WgradHParams hp;
  WgradParams& p = hp.g;
  p.A = A; p.B = B; p.slabs = ws;
  p.N = N; p.Ha = Ha; p.Wa = Wa; p.Cm = Cm; p.Hb = Hb; p.Wb = Wb; p.Cn = Cn;
  p.tilesX = pl.tilesX; p.tilesY = pl.tilesY; p.tilesTotal = pl.tilesTotal; p.tilesPerSplit = pl.tilesPerSplit;
  p.walk = 1;     // a workgroup's consecutive tiles are vertical neighbours (see IgemmParams::walk)
  hp.aScale = aScale; hp.bScale = bScale;
  // the activation operand is the UNscaled one: A for conv3x3 (A_SCALED = false), B for convT2x2
  hp.aSc = A_SCALED ? nullptr : xSc; hp.aSh = A_SCALED ? nullptr : xSh;
  hp.bSc = A_SCALED ? xSc : nullptr; hp.bSh = A_SCALED ? xSh : nullptr;
  hp.xAbound = xAbound; hp.xChannels = A_SCALED ? Cn : Cm;
  hp.bZ = bZ; hp.dzCoef = dzCoef;
  dim3 grid((unsigned)(pl.splits * dc_cdiv(Cm, Cfg::CM) * dc_cdiv(Cn, Cfg::CN)));
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  const bool bracket = dc_take_bracket(&ev0, &ev1);
  if (bracket && ev0) (void)hipEventRecord(ev0, st);
  hipLaunchKernelGGL(kern, grid, dim3(512), Cfg::LDS_BYTES, st, hp);
  if (bracket && ev1) (void)hipEventRecord(ev1, st);
  DC_CHECK_LAUNCH(name);
  const long L = (long)KH * KW * Cm * Cn;
  return dc_reduce_partials(ws, pl.splits, L, 1.0f, dw, ws + (long)pl.splits * L, (dc_stream_t)st);
}

// conv3x3 (A = x unscaled, B = dz scaled).  <KH,KW,S,PAD, TW, RW, WM, WNW, NBW>: CTA block 32*WM x 32*WNW*NBW.
// Cin, Cout > 32 (one wave per 32x32 block, WK = 1): 16-wide x 4-row tiles -- the halo'd A tile is 6x18 pixels for 64
// outputs (1.69x) instead of 4x34 (2.13x) with 32x2 tiles: less L2->LDS traffic, +3.5..5 % measured.
#define CONV_H_DISPATCH(FN, ...)                                                       \
  if (W <= 8) return FN<3, 3, 1, 1, 8, 8, 2, 2, 1 __VA_ARGS__;                         \
  if (W <= 16) return FN<3, 3, 1, 1, 16, 4, 2, 2, 1 __VA_ARGS__;                       \
  if (Cin > 32 && Cout > 32) return FN<3, 3, 1, 1, 16, DC_WG_RW, 2, 2, 1 __VA_ARGS__;  \
  if (Cin > 32) return FN<3, 3, 1, 1, 32, 2, 2, 1, 1 __VA_ARGS__;                      \
  if (Cout > 32) return FN<3, 3, 1, 1, 32, 2, 1, 2, 1 __VA_ARGS__;                     \
  return FN<3, 3, 1, 1, 32, 2, 1, 1, 1 __VA_ARGS__;

// convT2x2 (A = dz scaled, m = Cout; B = x, n = Cin >= 64); 16-wide tiles: the stride-2 A image is 4x the B tile
#define CONVT_H_DISPATCH(FN, ...)                                                      \
  if (W <= 8) return FN<2, 2, 2, 0, 8, 4, 1, 2, 1 __VA_ARGS__;                         \
  if (Cin >= 128) return FN<2, 2, 2, 0, 16, 2, 1, 2, 2 __VA_ARGS__;  /* 128 n-columns per workgroup: the 4x larger dz tile is staged for twice the MFMAs (116 -> 88 us) */ \
  return FN<2, 2, 2, 0, 16, 2, 1, 2, 1 __VA_ARGS__;

long dc_conv3x3_wgrad_f16x3_ws(int N, int H, int W, int Cin, int Cout) {
  CONV_H_DISPATCH(wgrad_h_ws, >(N, H, W, Cin, Cout))
}
long dc_convT2x2_wgrad_f16x3_ws(int N, int H, int W, int Cin, int Cout) {
  CONVT_H_DISPATCH(wgrad_h_ws, >(N, H, W, Cout, Cin))
}

int dc_conv3x3_c1_wgrad(const float* x, const float* dz, float* dw, float* ws, int N, int H, int W, int Cout,
                        hipStream_t st);

// the symbol of the instantiation CONV_H_DISPATCH picks, spelled as rocprofv3 prints it
template <int KH, int KW, int S, int PAD, int TW, int RW, int WM, int WNW, int NBW>
static const char* wgrad_h_name(int dzin) {
  static thread_local char buf[2][96];
  snprintf(buf[dzin ? 1 : 0], sizeof(buf[0]), "wgrad_f16x3_kernel<%d,%d,%d,%d,%d,%d,%d,%d,%d,false,%s>", KH, KW, S, PAD, TW, RW, WM, WNW, NBW,
           dzin ? "true" : "false");
  return buf[dzin ? 1 : 0];
}
extern "C" const char* dc_conv3x3_wgrad_kernel_name(int N, int H, int W, int Cin, int Cout, int dzin) {
  (void)N; (void)H;
  if (Cin == 1) return dzin ? "conv_c1_wgrad4_kernel<true>" : "conv_c1_wgrad4_kernel<false>";
  CONV_H_DISPATCH(wgrad_h_name, >(dzin))
}

static int check_h(const char* fn, const void* a, const void* b, const void* c, const void* d, int N, int H, int W,
                   int Cin, int Cout) {
  DC_REQUIRE(a && b && c && d, DC_EINVAL, "%s: null pointer", fn);
  DC_REQUIRE(dc_aligned16(a) && dc_aligned16(b) && dc_aligned16(c) && dc_aligned16(d), DC_EINVAL,
             "%s: pointers must be 16-byte aligned", fn);
  DC_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, DC_EINVAL, "%s: non-positive dimension", fn);
  DC_REQUIRE(Cin % 4 == 0 && Cout % 4 == 0, DC_EUNSUP, "%s: channel counts must be multiples of 4 (Cin=%d Cout=%d)", fn,
             Cin, Cout);
  return DC_OK;
}

static int conv_h_impl(const float* x, const float* xSc, const float* xSh, const float* xAb, const float* dz, float* dw,
                       float* ws, const float* dzScale, int N, int H, int W, int Cin, int Cout, hipStream_t st) {
  const float* none = nullptr;
  CONV_H_DISPATCH(wgrad_h_launch, , false>(x, dz, dw, ws, none, dzScale, xSc, xSh, xAb, N, H, W, H, W, Cin, Cout, st, "conv3x3_wgrad_f16x3"))
}
static int conv_h_dzin_impl(const float* x, const float* xSc, const float* xSh, const float* xAb, const float* da, const float* z,
                            const float* dzCoef, float* dw, float* ws, int N, int H, int W, int Cin, int Cout, hipStream_t st) {
  const float* none = nullptr;
  CONV_H_DISPATCH(wgrad_h_launch, , false, true>(x, da, dw, ws, none, none, xSc, xSh, xAb, N, H, W, H, W, Cin, Cout, st, "conv3x3_wgrad_dzin_f16x3", z, dzCoef))
}
static int convT_h_impl(const float* x, const float* xSc, const float* xSh, const float* xAb, const float* dz, float* dw,
                        float* ws, const float* dzScale, int N, int H, int W, int Cin, int Cout, hipStream_t st) {
  const float* none = nullptr;
  CONVT_H_DISPATCH(wgrad_h_launch, , true>(dz, x, dw, ws, dzScale, none, xSc, xSh, xAb, N, 2 * H, 2 * W, H, W, Cout, Cin, st, "convT2x2_wgrad_f16x3"))
}

// same workspace (dc_*_wgrad_ws_floats) as the fp32 entry points; dz_scale = device scalar from
// dc_pow2_scale_from_absmax (nullable).
extern "C" int dc_conv3x3_wgrad_f16x3(const float* x, const float* dz, float* dw, float* ws, const float* dz_scale,
                                      const float* x_abound, int N, int H, int W, int Cin, int Cout, dc_stream_t stream) {
  if (Cin == 1) {
    DC_REQUIRE(x && dz && dw && ws, DC_EINVAL, "dc_conv3x3_wgrad_f16x3: null pointer");
    return dc_conv3x3_c1_wgrad(x, dz, dw, ws, N, H, W, Cout, (hipStream_t)stream);
  }
  int rc = check_h("dc_conv3x3_wgrad_f16x3", x, dz, dw, ws, N, H, W, Cin, Cout);
  if (rc) return rc;
  return conv_h_impl(x, nullptr, nullptr, x_abound, dz, dw, ws, dz_scale, N, H, W, Cin, Cout, (hipStream_t)stream);
}
extern "C" int dc_convT2x2_wgrad_f16x3(const float* x, const float* dz, float* dw, float* ws, const float* dz_scale,
                                       const float* x_abound, int N, int H, int W, int Cin, int Cout, dc_stream_t stream) {
  int rc = check_h("dc_convT2x2_wgrad_f16x3", x, dz, dw, ws, N, H, W, Cin, Cout);
  if (rc) return rc;
  return convT_h_impl(x, nullptr, nullptr, x_abound, dz, dw, ws, dz_scale, N, H, W, Cin, Cout, (hipStream_t)stream);
}

// BN + ReLU on load: z_in is the producer's pre-BN tensor, (in_scale_c, in_shift_c) its per-channel training-mode affine
// (dc_bn_stats_finalize_affine); the layer input relu(fmaf(z, sc, sh)) is never materialised.
extern "C" int dc_conv3x3_wgrad_bnin_f16x3(const float* z_in, const float* in_sc, const float* in_sh,
                                           const float* in_abound, const float* dz, float* dw, float* ws,
                                           const float* dz_scale, int N, int H, int W, int Cin, int Cout,
                                           dc_stream_t stream) {
  int rc = check_h("dc_conv3x3_wgrad_bnin_f16x3", z_in, dz, dw, ws, N, H, W, Cin, Cout);
  if (rc) return rc;
  DC_REQUIRE(in_sc && in_sh && dc_aligned16(in_sc) && dc_aligned16(in_sh), DC_EINVAL,
             "dc_conv3x3_wgrad_bnin_f16x3: scale/shift must be non-null and 16-byte aligned");
  return conv_h_impl(z_in, in_sc, in_sh, in_abound, dz, dw, ws, dz_scale, N, H, W, Cin, Cout, (hipStream_t)stream);
}
extern "C" int dc_convT2x2_wgrad_bnin_f16x3(const float* z_in, const float* in_sc, const float* in_sh,
                                            const float* in_abound, const float* dz, float* dw, float* ws,
                                            const float* dz_scale, int N, int H, int W, int Cin, int Cout,
                                            dc_stream_t stream) {
  int rc = check_h("dc_convT2x2_wgrad_bnin_f16x3", z_in, dz, dw, ws, N, H, W, Cin, Cout);
  if (rc) return rc;
  DC_REQUIRE(in_sc && in_sh && dc_aligned16(in_sc) && dc_aligned16(in_sh), DC_EINVAL,
             "dc_convT2x2_wgrad_bnin_f16x3: scale/shift must be non-null and 16-byte aligned");
  return convT_h_impl(z_in, in_sc, in_sh, in_abound, dz, dw, ws, dz_scale, N, H, W, Cin, Cout, (hipStream_t)stream);
}

int dc_conv3x3_c1_wgrad_dzin(const float* x, const float* da, const float* z, const float* dz_coef, float* dw, float* ws,
                             int N, int H, int W, int Cout, hipStream_t st);     // conv_c1.hip

// "dz on load" weight gradient (dcunet.h): dz is formed from (da, z, dz_coef) by the producer waves.
extern "C" int dc_conv3x3_wgrad_dzin_f16x3(const float* x, const float* in_sc, const float* in_sh, const float* x_abound,
                                           const float* da, const float* z, const float* dz_coef, float* dw, float* ws,
                                           int N, int H, int W, int Cin, int Cout, dc_stream_t stream) {
  DC_REQUIRE(z && dz_coef && dc_aligned16(z) && dc_aligned16(dz_coef), DC_EINVAL,
             "dc_conv3x3_wgrad_dzin_f16x3: z / dz_coef must be non-null and 16-byte aligned");
  DC_REQUIRE((in_sc == nullptr) == (in_sh == nullptr), DC_EINVAL, "dc_conv3x3_wgrad_dzin_f16x3: in_scale and in_shift go together");
  if (Cin == 1) {
    DC_REQUIRE(x && da && dw && ws && !in_sc, DC_EINVAL, "dc_conv3x3_wgrad_dzin_f16x3: bad first-layer arguments");
    return dc_conv3x3_c1_wgrad_dzin(x, da, z, dz_coef, dw, ws, N, H, W, Cout, (hipStream_t)stream);
  }
  int rc = check_h("dc_conv3x3_wgrad_dzin_f16x3", x, da, dw, ws, N, H, W, Cin, Cout);
  if (rc) return rc;
  DC_REQUIRE(!in_sc || (dc_aligned16(in_sc) && dc_aligned16(in_sh)), DC_EINVAL,
             "dc_conv3x3_wgrad_dzin_f16x3: scale/shift must be 16-byte aligned");
  return conv_h_dzin_impl(x, in_sc, in_sh, x_abound, da, z, dz_coef, dw, ws, N, H, W, Cin, Cout, (hipStream_t)stream);
}
