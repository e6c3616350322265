// Implicit-GEMM convolution with fp32-grade accuracy on the fp16 matrix cores (v_mfma_f32_32x32x16_f16), gfx950.
//
// The fp32 MFMA (v_mfma_f32_32x32x2_f32) runs at the vector rate, 157 TF/s -- the roof that bounds the fp32
// path.  The fp16 MFMA is 16x faster per instruction-cycle.  This kernel feeds it an exact
// two-term split of every fp32 operand,
//        x = hi + lo,   hi = fp16(x),  lo = fp16(x - hi)        (22-bit significand, |err| <= 2^-22 |x|)
// and accumulates   a_hi*b_hi + a_hi*b_lo + a_lo*b_hi   in the fp32 accumulator (the dropped lo*lo term is
// 2^-22 relative): 3 MFMAs replace 8, i.e. 16/3 = 5.3x the fp32 matrix rate at ~4x fp32's rounding noise,
// well inside the 1e-4 parity budget (tests/test_hip_ops.py, tests/test_engine_gpu.py run both paths).
// fp16's narrow exponent range is handled with per-tensor POWER-OF-TWO scales (exact in fp32), undone exactly in the
// epilogue: gradients bring theirs as a device scalar (dc_bn_bwd_apply_finalize: their 1/(N*H*W) factor would otherwise
// underflow), forward activations get one from the per-channel magnitude bound of the producing layer (common.h
// dc_block_guard_scale: no input can reach inf), and the packed weights carry the scale that brought max|w| into
// [2^10, 2^11) in a 16-byte trailer (dc_pack_weights_f16x3): small-magnitude kernels keep their 22-bit split.
//
// Structure = the fp32 kernel's (igemm_conv.hip): halo'd input patch staged once per 16-channel chunk, taps are
// shifted ds_read_b128 windows of the same LDS patch, channel on the lane in the epilogue.  Differences:
//   * LDS slots hold 8 fp16 channels (16 B): [hi|lo][8-channel group][pixel]; one b128 read = one MFMA operand
//     (lane half h supplies k = 8h..8h+7, the 32x32x16 operand map);
//   * activations are split while being written to LDS (v_cvt_pk_f16_f32 x2 + 2 sub per pair);
//   * weights arrive PRE-split from dc_pack_weights_f16x3 ([tap][K/8][hi|lo][col][8]) and are copied verbatim.
#include "igemm_common.h"
#include <stdlib.h>
#include <type_traits>

// Phase timestamps of sampled workgroups (scripts/igemm_phases.py builds the library with -DDC_IGEMM_TRACE and reads
// them back): wave 0 / lane 0 of every 37th workgroup stores s_memtime at each DC_TRACE point.  Compiled out otherwise.
#ifdef DC_IGEMM_TRACE
__device__ unsigned long long* g_dc_trace = nullptr;
extern "C" int dc_debug_set_trace(unsigned long long* p) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_dc_trace), &p, sizeof(p)) == hipSuccess ? 0 : -2;
}
#define DC_TRACE_INIT()                                                                                     \
  unsigned long long* trc_ = (g_dc_trace && blockIdx.x % 37 == 0 && threadIdx.x == 0)                       \
                                 ? g_dc_trace + (blockIdx.x / 37) * 64 : nullptr;                           \
  int trn_ = 0
#define DC_TRACE() do { if (trc_ && trn_ < 63) trc_[++trn_] = __builtin_readcyclecounter(); if (trc_) trc_[0] = trn_; } while (0)
#else
#define DC_TRACE_INIT() do {} while (0)
#define DC_TRACE() do {} while (0)
#endif

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

template <int KH, int KW, int S, int PAD, int TW, int WAVES_M, int MB, int NB, int CK_ = 16>
struct IgemmH {
  static constexpr int CK = CK_;
  static constexpr int TAPS = KH * KW;
  static constexpr int WAVES_N = 4 / WAVES_M;
  static constexpr int RPM = 32 / TW;
  static constexpr int TH = WAVES_M * MB * RPM;
  static constexpr int BN = WAVES_N * NB * 32;
  static constexpr int THI = (TH - 1) * S + KH;
  static constexpr int TWI = (TW - 1) * S + KW;
  static constexpr int NPIXH = THI * TWI;
  static constexpr int PS = ((NPIXH + 5) / 8) * 8 + 2;
  static constexpr int G4 = CK / 4;   // float4 groups per pixel in HBM
  static constexpr int G8 = CK / 8;   // 8-channel fp16 slots per pixel in LDS
  static constexpr int NA = (NPIXH * G4 + 255) / 256;
  static constexpr int BROWS = TAPS * G8 * 2;  // (tap, g8, hi|lo) rows of BN slots
  static constexpr int NBV = (BROWS * BN + 255) / 256;
  static constexpr int A_SLOTS = 2 * G8 * PS;
  static constexpr int LDS_BYTES = (A_SLOTS + BROWS * BN) * 16;
  static_assert(256 % BN == 0, "staging assumes BN divides the block size");
  static_assert(CK % 16 == 0 && 256 % G4 == 0, "CK must be a multiple of the MFMA k (16)");
};

// hi = fp16(x*s), lo = fp16(x*s - hi) for 4 elements in EIGHT vector instructions: v_fma_mix{lo,hi}_f16 multiplies in
// fp32, adds an fp16 (or zero) and rounds once to fp16 into one half of the destination.  (s is a power of two and
// x*s - hi is exactly representable, so the bits equal the two-step form (half)(x*s - (float)hi).)  hipcc's own lowering
// of the C expression spent 13 instructions per 4 elements, part of them packed-fp32 ops that issue at half rate next to
// MFMAs -- and this split runs in the matrix waves' own instruction stream, once per staged element.
__device__ __forceinline__ void split_f16(const f32x4 v, float s, u32x2& hi, u32x2& lo) {
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

// BNRED (its own instantiations, conv-transpose data gradients only): the output IS `da` of the BatchNorm layer in front (dense,
// no Dropout); the epilogue also reads that layer's pre-BN tensor bnZ (same layout) and emits its pass-1 partials
// bnPartial[tile][Ncols][2] = (sum dy, sum dy*xhat) and bnAmax[tile][Ncols] = max |dy| -- as igemm_pp.hip's BNRED variant does.
template <int KH, int KW, int S, int PAD, int TW, int WAVES_M, int MB, int NB, int CK_, bool BNRED = false>
__global__ __launch_bounds__(256, 2) void igemm_f16x3_kernel(IgemmParams p) {
  using Cfg = IgemmH<KH, KW, S, PAD, TW, WAVES_M, MB, NB, CK_>;
  constexpr int TAPS = Cfg::TAPS, TH = Cfg::TH, BN = Cfg::BN, TWI = Cfg::TWI, NPIXH = Cfg::NPIXH, CK = Cfg::CK;
  constexpr int PS = Cfg::PS, G4 = Cfg::G4, G8 = Cfg::G8, NA = Cfg::NA, NBV = Cfg::NBV, RPM = Cfg::RPM;
  constexpr int BROWS = Cfg::BROWS;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  u32x4* ldsA = reinterpret_cast<u32x4*>(smem);  // [hl][g8][PS] slots of 8 halfs
  u32x4* ldsB = ldsA + Cfg::A_SLOTS;             // [tap][g8][hl][BN]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: tile math goes to the SALU
  DC_TRACE_INIT();
  const int li = lane & 31, h = lane >> 5;
  const int wave_m = wave % WAVES_M, wave_n = wave / WAVES_M;

  // XCD-aware rasterisation: workgroups are dealt round-robin over the 8 XCDs (ids b and b+8 share an L2), so
  // each XCD walks a contiguous range of (pixel tile, column block) pairs with the column block fastest: the CTAs
  // that re-read one input patch for different output columns run back to back on ONE L2 (speed only; any
  // placement computes the same result).  Bijective for every grid size.
  const int nblk = (p.Ncols + BN - 1) / BN, total = (int)gridDim.x;
  const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3, qq = total >> 3, rr = total & 7;
  const int work = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + seq;
  const int wpos = work / nblk;                                // position in the walk over the pixel tiles
  int tx, ty, img;
  if (p.walk) { ty = wpos % p.tilesY; const int t = wpos / p.tilesY; tx = t % p.tilesX; img = t / p.tilesX; }
  else { tx = wpos % p.tilesX; const int t = wpos / p.tilesX; ty = t % p.tilesY; img = t / p.tilesY; }
  const int tile_id = (img * p.tilesY + ty) * p.tilesX + tx;     // index of the BatchNorm partials: row-major whatever the walk
  const int n0 = (work - wpos * nblk) * BN;
  const int oy0 = ty * TH, ox0 = tx * TW;
  const int iy0 = oy0 * S - PAD, ix0 = ox0 * S - PAD;
  const int Cin8 = (p.Cin + 7) >> 3;
  const float w_scale = p.wp[(long)TAPS * Cin8 * 2 * p.Ncols * 4];       // trailer of the packed weights
  // BatchNorm + ReLU on load (non-materialised input activation): the per-channel (scale, shift) pairs sit behind the
  // operand images in LDS; the staging pass reads its 4 channels' pairs once per chunk.
  const bool bnin = p.inSc != nullptr;
  float* lds_sc = reinterpret_cast<float*>(smem + Cfg::LDS_BYTES);
  const int Cinp = (p.Cin + 3) & ~3;

  // Staging addresses: one buffer descriptor per operand (this image's activations / the packed weight slabs) and
  // ONE 32-bit byte offset per load, computed once.  Pixels outside the image, columns beyond Ncols and padding
  // rows carry an out-of-range offset: the hardware bounds check returns zeros, so the loop has no branches, no
  // selects and no 64-bit address arithmetic; a chunk only adds a wave-uniform constant to the offsets.
  const __amdgpu_buffer_rsrc_t rsrcA =
      dc_make_rsrc(p.in + (long)img * p.Hin * p.Win * p.Cin, (unsigned)(p.Hin * p.Win * p.Cin) * 4u);
  const __amdgpu_buffer_rsrc_t rsrcB = dc_make_rsrc(p.wp, (unsigned)(TAPS * Cin8 * 2 * p.Ncols) * 16u);
  constexpr unsigned OOB = 0x80000000u;

  constexpr int A_STEP = 256 / G4, B_STEP = 256 / BN;
  const int a_g = tid % G4, a_pix0 = tid / G4;
  const int b_j = tid % BN, b_row0 = tid / BN;
  unsigned a_voff[NA], b_voff[NBV];
#pragma unroll
  for (int it = 0; it < NA; ++it) {
    const int pix = a_pix0 + it * A_STEP;
    const int r = pix / TWI, c = pix - r * TWI;
    const int y = iy0 + r, x = ix0 + c;
    const bool in_img = pix < NPIXH && y >= 0 && y < p.Hin && x >= 0 && x < p.Win;
    a_voff[it] = in_img ? (unsigned)(((y * p.Win + x) * p.Cin + 4 * a_g) * 4) : OOB;
  }
#pragma unroll
  for (int it = 0; it < NBV; ++it) {
    const int row = b_row0 + it * B_STEP;       // (tap, g8, hl)
    const int tap = row / (2 * G8), g8 = (row >> 1) % G8, hl = row & 1;
    const bool ok = row < BROWS && (n0 + b_j) < p.Ncols;
    b_voff[it] = ok ? (unsigned)((((tap * Cin8 + g8) * 2 + hl) * p.Ncols + n0 + b_j) * 16) : OOB;
  }

  f32x4 ra[NA];
  u32x4 rb[NBV];
  auto load_chunk = [&](int c0) {
    const unsigned a_add = (unsigned)c0 * 4u, b_add = (unsigned)(c0 >> 3) * 2u * (unsigned)p.Ncols * 16u;
    const bool partial = c0 + CK > p.Cin;     // wave-uniform; only the ragged last chunk of Cin % 16 != 0
#pragma unroll
    // the chunk's byte offset rides in the scalar soffset operand: no per-load VALU add (an out-of-range voffset
    // stays out of range: the marker is 2 GiB)
    for (int it = 0; it < NA; ++it) {
      unsigned off = a_voff[it];
      if (partial && c0 + 4 * a_g >= p.Cin) off = OOB;
      ra[it] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrcA, off, (int)a_add, 0));
    }
#pragma unroll
    for (int it = 0; it < NBV; ++it) {
      unsigned off = b_voff[it];
      if (partial) {
        const int row = b_row0 + it * B_STEP;
        if ((c0 >> 3) + ((row >> 1) % G8) >= Cin8) off = OOB;
      }
      rb[it] = __builtin_amdgcn_raw_buffer_load_b128(rsrcB, off, (int)b_add, 0);
    }
  };

  int a_base[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int mblk = wave_m * MB + mb;
    const int row = mblk * RPM + li / TW, col = li % TW;
    a_base[mb] = h * PS + (row * S) * TWI + col * S;      // g8 = 2*ks + h for k-step ks (16 channels each)
  }
  const int b_base = (h * 2) * BN + wave_n * NB * 32 + li;  // row (tap 0, g8 = h, hl = 0)

  f32x16 acc[MB][NB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;

  // split-K (IgemmParams::splitWs): this workgroup's chunk range, and a plain dense store of the partial tile into its slab
  int c_begin = 0, c_end = p.Cin;
  if constexpr (TW <= 16 && !BNRED)               // (only the narrow-tile instantiations are ever launched with grid.y > 1)
  if (gridDim.y > 1) {
    const int nch = (p.Cin + CK - 1) / CK, ks = (int)blockIdx.y, nsl = (int)gridDim.y;
    c_begin = (nch * ks / nsl) * CK;
    c_end = min(p.Cin, (nch * (ks + 1) / nsl) * CK);
    p.out = p.splitWs + (long)ks * p.splitSlab;
    p.outLd = p.Ncols;
    p.bias = nullptr; p.stats = nullptr; p.scale = nullptr; p.shift = nullptr; p.relu = 0; p.outAbsmax = nullptr;
  }
  DC_TRACE();            // 1: entry + address setup
  load_chunk(c_begin);
  // Per-channel tables and the operand scale AFTER the first chunk's loads are in flight (their latency is the same
  // L2 round trip): the BN-on-load (scale, shift) pairs, and the powers of two -- device scalar of a gradient tensor x
  // range guard of an activation tensor (common.h dc_block_guard_scale; its barriers also publish the table).
  if (bnin)
    for (int i = tid; i < p.Cin; i += 256) { lds_sc[i] = p.inSc[i]; lds_sc[Cinp + i] = p.inSh[i]; }
  const float in_scale = (p.inAbsmax ? dc_block_absmax_scale(p.inAbsmax, p.inAbsmaxN, 1024.f, reinterpret_cast<float*>(smem))
                                     : (p.inScale ? *p.inScale : 1.f)) *
                         dc_block_guard_scale(p.inAbound, p.Cin, reinterpret_cast<float*>(smem), p.inAboundLd);
  if (bnin && p.inAbound == nullptr) __syncthreads();
  DC_TRACE();            // 2: first loads issued, tables / guard done
  for (int c0 = c_begin; c0 < c_end; c0 += CK) {
    DC_TRACE();          // per chunk a: top
#ifdef DC_IGEMM_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    DC_TRACE();          // a2: this chunk's global loads have landed (trace build only: separates waiting from splitting)
#endif
    f32x4 csc = {1.f, 1.f, 1.f, 1.f}, csh = {0.f, 0.f, 0.f, 0.f};
    const bool ch_ok = c0 + 4 * a_g < p.Cin;
    if (bnin && ch_ok) {
      csc = *reinterpret_cast<const f32x4*>(lds_sc + c0 + 4 * a_g);
      csh = *reinterpret_cast<const f32x4*>(lds_sc + Cinp + c0 + 4 * a_g);
    }
#pragma unroll
    for (int it = 0; it < NA; ++it) {
      const int pix = a_pix0 + it * A_STEP;
      if (pix < NPIXH) {
        u32x2 hi, lo;
        if (bnin) {
          const bool live = ch_ok && !(a_voff[it] >> 31);     // zero padding stays zero
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float y = fmaxf(__builtin_fmaf(ra[it][e], csc[e], csh[e]), 0.f);
            ra[it][e] = live ? y : 0.f;
          }
        }
        split_f16(ra[it], in_scale, hi, lo);
        char* base = smem + ((a_g >> 1) * PS + pix) * 16 + (a_g & 1) * 8;
        *reinterpret_cast<u32x2*>(base) = hi;
        *reinterpret_cast<u32x2*>(base + G8 * PS * 16) = lo;
      }
    }
#pragma unroll
    for (int it = 0; it < NBV; ++it) {
      const int row = b_row0 + it * B_STEP;
      if (row < BROWS) ldsB[row * BN + b_j] = rb[it];
    }
    DC_TRACE();          // b: operands split and written to LDS (includes the wait for this chunk's global loads)
    __syncthreads();
    DC_TRACE();          // c: barrier passed
    if (c0 + CK < c_end) load_chunk(c0 + CK);

    // Software-pipelined operand fetch: the fragments of step tk+1 are requested from LDS before the 3*MB*NB
    // MFMAs of step tk issue, so the ~100-cycle ds_read latency hides behind 12 x 32 MFMA cycles instead of
    // stalling every group (hipcc otherwise sinks each ds_read next to its first use).
    constexpr int NT = TAPS * (CK / 16);
    f16x8 ah[2][MB], al[2][MB], bh[2][NB], bl[2][NB];
    auto fetch = [&](int tk, int buf) {
      const int tap = tk / (CK / 16), ks = tk % (CK / 16);
      const int toff = (tap / KW) * TWI + (tap % KW) + 2 * ks * PS;
      const int boff = (tap * G8 + 2 * ks) * 2 * BN;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        ah[buf][mb] = __builtin_bit_cast(f16x8, ldsA[a_base[mb] + toff]);
        al[buf][mb] = __builtin_bit_cast(f16x8, ldsA[a_base[mb] + toff + G8 * PS]);
      }
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        bh[buf][nb] = __builtin_bit_cast(f16x8, ldsB[b_base + boff + nb * 32]);
        bl[buf][nb] = __builtin_bit_cast(f16x8, ldsB[b_base + boff + BN + nb * 32]);
      }
    };
    fetch(0, 0);
#pragma unroll
    for (int tk = 0; tk < NT; ++tk) {
      const int cur = tk & 1;
      if (tk + 1 < NT) fetch(tk + 1, cur ^ 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[cur][mb], bh[cur][nb], acc[mb][nb], 0, 0, 0);
          acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cur][mb], bl[cur][nb], acc[mb][nb], 0, 0, 0);
          acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cur][mb], bh[cur][nb], acc[mb][nb], 0, 0, 0);
        }
      __builtin_amdgcn_sched_barrier(0);
    }
    DC_TRACE();          // d: MFMA block done
    __syncthreads();
  }
  DC_TRACE();            // e: main loop left

  // ---- epilogue: C/D col = lane&31 -> output column n, row m = (r&3) + 8*(r>>2) + 4*(lane>>5) -> pixel --------
  // Stores go through a buffer descriptor of this image's output; pixels outside the image / columns beyond
  // Ncols get an out-of-range offset and are dropped by the bounds check (no branches, 32-bit offsets).
  const float out_scale = 1.f / (in_scale * w_scale);
  DcMoments* red = reinterpret_cast<DcMoments*>(smem);
  const bool scatter = p.scatterCo > 0;
  const long out_img_floats = scatter ? 4L * p.Hout * p.Wout * p.outLd : (long)p.Hout * p.Wout * p.outLd;
  const __amdgpu_buffer_rsrc_t rsrcO = dc_make_rsrc(p.out + (long)img * out_img_floats, (unsigned)(out_img_floats * 4));
  const int ld = (int)p.outLd;
  const int sy = scatter ? 4 * p.Wout * ld : p.Wout * ld;   // floats per output-pixel row step
  const int sx = scatter ? 2 * ld : ld;                     // floats per output-pixel column step
  // Interior items (every pixel and column of the tile exists: all but the ragged border) take a lean epilogue: no
  // per-element validity selects, and the per-element address term rides in the scalar soffset operand.
  const bool interior = (oy0 + TH <= p.Hout) && (ox0 + TW <= p.Wout) && (n0 + BN <= p.Ncols);   // wave-uniform
  // (A variant with the MFMA operands swapped -- 4 consecutive channels per lane, 16-byte stores -- was measured and
  // dropped: its stores touch 32 cache lines of 32 bytes each per instruction and the epilogue took 1.7x LONGER.)
  auto epilogue = [&](auto interior_tag, auto mode_tag) {
    constexpr bool INT = decltype(interior_tag)::value;
    constexpr int MODE = decltype(mode_tag)::value;        // 0 plain (data gradients), 1 BatchNorm partials, 2 inference
    constexpr bool STATS = MODE == 1;
    constexpr bool TRACK = MODE == 2;                      // fold max |output| per channel into outAbsmax
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int n = n0 + (wave_n * NB + nb) * 32 + li;
      const bool n_ok = INT || n < p.Ncols;
      const float bv = (p.bias && n_ok) ? p.bias[n % p.biasMod] : 0.f;
      const float sc = (p.scale && n_ok) ? p.scale[n % p.biasMod] : 1.f;
      const float sh = (p.shift && n_ok) ? p.shift[n % p.biasMod] : 0.f;
      // BatchNorm partials as shifted sums around this lane's first value (common.h DcMoments)
      const float K = __builtin_fmaf(acc[0][nb][0], out_scale, bv);
      float s1 = 0.f, s2 = 0.f, cnt = 0.f, amax = 0.f;
      int colterm = n;
      if (scatter) {
        const int ab = n / p.scatterCo, o = n - ab * p.scatterCo;
        colterm = ((ab >> 1) * 2 * p.Wout + (ab & 1)) * ld + o;
      }
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const int mblk = wave_m * MB + mb;                       // wave-uniform
        const int oyb = oy0 + mblk * RPM, oxb = ox0 + 4 * h;
        const unsigned base = (unsigned)((oyb * sy + oxb * sx + colterm) * 4);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int mr = (r & 3) + 8 * (r >> 2);                 // compile-time; + 4h stays inside one tile row
          const int rowc = mr / TW, colc = mr % TW;
          float v = __builtin_fmaf(acc[mb][nb][r], out_scale, bv);
          const float d = v - K;
          if constexpr (INT) {
            if constexpr (STATS) {
              s1 += d;
              s2 = __builtin_fmaf(d, d, s2);
            }
            if (p.scale) v = v * sc + sh;
            if (p.relu) v = fmaxf(v, 0.f);
            if constexpr (TRACK) amax = fmaxf(amax, fabsf(v));
#if defined(DC_F_ABL) && (DC_F_ABL & 1)
            // ablation (garbage results): no output stores -- one conditional store keeps the values alive
            if (v == 12345.678f) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrcO, base, 0, 0);
#else
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrcO, base,
                                                  (rowc * sy + colc * sx) * 4, 0);      // scalar addend
#endif
          } else {
            const bool ok = n_ok && (oyb + rowc) < p.Hout && (oxb + colc) < p.Wout;
            if constexpr (STATS) {
              s1 += ok ? d : 0.f;
              s2 += ok ? d * d : 0.f;
              cnt += ok ? 1.f : 0.f;
            }
            if (p.scale) v = v * sc + sh;
            if (p.relu) v = fmaxf(v, 0.f);
            if constexpr (TRACK) amax = fmaxf(amax, ok ? fabsf(v) : 0.f);
            const unsigned off = ok ? base + (unsigned)((rowc * sy + colc * sx) * 4) : OOB;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrcO, off, 0, 0);
          }
        }
      }
      if constexpr (STATS) {
        DcMoments m;
        if constexpr (INT) {      // every lane holds 16*MB values: no divisions
          constexpr float NL = (float)(16 * MB);
          const float ms = s1 * (1.f / NL);
          m.n = NL; m.mean = K + ms; m.m2 = fmaxf(__builtin_fmaf(-s1, ms, s2), 0.f);
          const float om = __shfl_xor(m.mean, 32), o2 = __shfl_xor(m.m2, 32), d = om - m.mean;
          m.m2 = m.m2 + o2 + d * d * (0.5f * NL);
          m.mean = __builtin_fmaf(d, 0.5f, m.mean);
          m.n = 2.f * NL;
        } else {
          m = dc_moments_from_shifted(cnt, K, s1, s2);
          DcMoments o;
          o.n = __shfl_xor(m.n, 32); o.mean = __shfl_xor(m.mean, 32); o.m2 = __shfl_xor(m.m2, 32);
          m = dc_moments_merge(m, o);
        }
        if (h == 0) red[(wave * NB + nb) * 32 + li] = m;
      }
      if constexpr (TRACK) {      // lanes h = 0 / 1 -> one value per (wave, column); the waves are folded below
        amax = fmaxf(amax, __shfl_xor(amax, 32));
        if (h == 0) reinterpret_cast<float*>(smem + 4096)[(wave * NB + nb) * 32 + li] = n_ok ? amax : 0.f;   // behind `red`
      }
    }
  };
  using Plain = std::integral_constant<int, 0>;
  using Stats = std::integral_constant<int, 1>;
  using Track = std::integral_constant<int, 2>;
  if constexpr (BNRED) {
    const __amdgpu_buffer_rsrc_t rsrcZ = dc_make_rsrc(p.bnZ + (long)img * out_img_floats, (unsigned)(out_img_floats * 4));
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int n = n0 + (wave_n * NB + nb) * 32 + li;
      const bool n_ok = n < p.Ncols;
      const int nl = n_ok ? n : 0;
      const float mu = p.bnMean[nl], is = p.bnInvstd[nl];
      float gsc, gsh;
      dc_bn_affine(mu, is, p.bnGamma[nl], p.bnBeta[nl], gsc, gsh);
      float s1 = 0.f, s2 = 0.f, amax = 0.f;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const int mblk = wave_m * MB + mb;
        const int oyb = oy0 + mblk * RPM, oxb = ox0 + 4 * h;
        unsigned offs[16];
        float zr[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {            // the z values are requested first, the stores of da go out while they fly
          const int mr = (r & 3) + 8 * (r >> 2);
          const int rowc = mr / TW, colc = mr % TW;
          const bool ok = n_ok && (oyb + rowc) < p.Hout && (oxb + colc) < p.Wout;
          offs[r] = ok ? (unsigned)((((oyb + rowc) * p.Wout + oxb + colc) * ld + n) * 4) : OOB;
          zr[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrcZ, offs[r], 0, 0));
        }
#pragma unroll
        for (int r = 0; r < 16; ++r)
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, acc[mb][nb][r] * out_scale), rsrcO, offs[r], 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = acc[mb][nb][r] * out_scale;
          const float y = __builtin_fmaf(zr[r], gsc, gsh);        // the forward's own expression: identical ReLU gate
          const float dy = (!(offs[r] >> 31) && y > 0.f) ? v : 0.f;
          s1 += dy;
          s2 = __builtin_fmaf(dy, (zr[r] - mu) * is, s2);
          amax = fmaxf(amax, fabsf(dy));
        }
      }
      DcMoments m;                                 // container: (max |dy|, sum dy, sum dy*xhat)
      m.n = fmaxf(amax, __shfl_xor(amax, 32));
      m.mean = s1 + __shfl_xor(s1, 32);
      m.m2 = s2 + __shfl_xor(s2, 32);
      if (h == 0) red[(wave * NB + nb) * 32 + li] = m;
    }
    __syncthreads();
    if (tid < Cfg::WAVES_N * NB * 32) {
      const int wn = tid / (NB * 32), rem = tid % (NB * 32);
      const int nb = rem / 32, l = rem % 32;
      float s1 = 0.f, s2 = 0.f, am = 0.f;
#pragma unroll
      for (int wm = 0; wm < WAVES_M; ++wm) {
        const DcMoments m = red[((wn * WAVES_M + wm) * NB + nb) * 32 + l];
        s1 += m.mean; s2 += m.m2; am = fmaxf(am, m.n);
      }
      const int n = n0 + (wn * NB + nb) * 32 + l;
      if (n < p.Ncols) {
        float* dst = p.bnPartial + ((long)tile_id * p.Ncols + n) * 2;
        dst[0] = s1; dst[1] = s2;
        if (p.bnAmax) p.bnAmax[(long)tile_id * p.Ncols + n] = am;
      }
    }
    return;
  }
  if (p.outAbsmax) {
    if (interior) epilogue(std::true_type{}, Track{}); else epilogue(std::false_type{}, Track{});
    __syncthreads();
    if (p.outAbsmaxLd < 0) {                  // optimistic inference: only flag an output beyond fp16's range
      if (tid < Cfg::WAVES_N * NB * 32) {
        const int wn = tid / (NB * 32), rem = tid % (NB * 32);
        const float* fm = reinterpret_cast<const float*>(smem + 4096);
        float m = fm[(wn * WAVES_M) * NB * 32 + rem];
#pragma unroll
        for (int wm = 1; wm < WAVES_M; ++wm) m = fmaxf(m, fm[(wn * WAVES_M + wm) * NB * 32 + rem]);
        if (!(m <= DC_F16_SAFE_MAX)) p.outAbsmax[0] = 1.f;        // (also catches NaN)
      }
    } else if (tid < Cfg::WAVES_N * NB * 32) {       // one atomic per output column per workgroup, into this workgroup's slot
      const int wn = tid / (NB * 32), rem = tid % (NB * 32);
      const float* fm = reinterpret_cast<const float*>(smem + 4096);
      float m = fm[(wn * WAVES_M) * NB * 32 + rem];
#pragma unroll
      for (int wm = 1; wm < WAVES_M; ++wm) m = fmaxf(m, fm[(wn * WAVES_M + wm) * NB * 32 + rem]);
      const int n = n0 + wn * NB * 32 + rem;
      if (n < p.Ncols) dc_atomic_absmax(p.outAbsmax + dc_absmax_slot(p.outAbsmaxLd) + n % p.biasMod, m);
    }
  } else if (p.stats) {
    if (interior) epilogue(std::true_type{}, Stats{}); else epilogue(std::false_type{}, Stats{});
  } else {
    if (interior) epilogue(std::true_type{}, Plain{}); else epilogue(std::false_type{}, Plain{});
  }
  DC_TRACE();            // f: epilogue stores issued
  if (p.stats) {
    __syncthreads();
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
// Second half of a split-K launch: out = sum_s slab_s (index order) + bias, and the BatchNorm partials of the SAME pixel-tile
// rows the un-split kernel writes (tile_id row-major, one row per TW x TH tile).  One workgroup per (pixel tile, 16 columns)
// -- 320 / 640 workgroups for the 16^2 / 8^2 layers of a 128^2 x 20 step --, thread = (column quad, one of 64 pixel lanes):
// a wave reads 16 pixels x 64 bytes per slab.
__global__ __launch_bounds__(256) void splitk_combine_kernel(const float* __restrict__ ws, int S, long slab,
                                                            const float* __restrict__ bias, float* __restrict__ out, long outLd,
                                                            double* __restrict__ stats, int H, int W, int Ncols, int TW, int TH,
                                                            int tilesX, int tilesY) {
  __shared__ DcMoments red[4][4][4];                           // [wave][column quad][element]
  const int tid = threadIdx.x, cq = tid & 3, pl = tid >> 2, lane = tid & 63, wave = tid >> 6;
  const int col = blockIdx.y * 16 + 4 * cq;
  const int tile = blockIdx.x, tx = tile % tilesX, t2 = tile / tilesX, ty = t2 % tilesY, img = t2 / tilesY;
  const int ncol = Ncols - col;                                // columns of this quad that exist (Ncols % 4 may be != 0)
  f32x4 bv = {0.f, 0.f, 0.f, 0.f};
  if (bias)
    for (int e = 0; e < 4; ++e) bv[e] = e < ncol ? bias[col + e] : 0.f;
  float n = 0.f;
  f32x4 K = {0.f, 0.f, 0.f, 0.f}, s1 = K, s2 = K;
  if (ncol > 0)
    for (int pi = pl; pi < TW * TH; pi += 64) {
      const int y = ty * TH + pi / TW, x = tx * TW + pi % TW;
      if (y < H && x < W) {
        const long pix = ((long)img * H + y) * W + x;
        const float* src = ws + pix * Ncols + col;
        f32x4 v;
        if (ncol >= 4 && (Ncols & 3) == 0) {
          v = *reinterpret_cast<const f32x4*>(src);
          for (int s = 1; s < S; ++s) v += *reinterpret_cast<const f32x4*>(src + s * slab);
        } else {
          for (int e = 0; e < 4; ++e) {
            float a = 0.f;
            if (e < ncol) { a = src[e]; for (int s = 1; s < S; ++s) a += src[s * slab + e]; }
            v[e] = a;
          }
        }
        v += bv;
        float* dst = out + pix * outLd + col;
        if (ncol >= 4 && (outLd & 3) == 0) *reinterpret_cast<f32x4*>(dst) = v;
        else for (int e = 0; e < 4 && e < ncol; ++e) dst[e] = v[e];
        if (n == 0.f) K = v;
        const f32x4 d = v - K;
        s1 += d;
        s2 += d * d;
        n += 1.f;
      }
    }
  if (stats == nullptr) return;
  // moments of this thread's 4 columns -> merge over the 16 pixel lanes of the wave (lane bits 2-5), then over the 4 waves
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    DcMoments m = dc_moments_from_shifted(n, K[e], s1[e], s2[e]);
#pragma unroll
    for (int o = 4; o < 64; o <<= 1) {
      DcMoments q;
      q.n = __shfl_xor(m.n, o); q.mean = __shfl_xor(m.mean, o); q.m2 = __shfl_xor(m.m2, o);
      m = dc_moments_merge(m, q);
    }
    if (lane < 4) red[wave][cq][e] = m;
  }
  __syncthreads();
  if (tid < 16) {
    const int q = tid >> 2, e = tid & 3;
    DcMoments m = red[0][q][e];
#pragma unroll
    for (int w = 1; w < 4; ++w) m = dc_moments_merge(m, red[w][q][e]);
    const int c = blockIdx.y * 16 + 4 * q + e;
    if (c < Ncols) dc_moments_store(stats + ((long)tile * Ncols + c) * 2, m);
  }
}

template <int KH, int KW, int S, int PAD, int TW, int WAVES_M, int MB, int NB, int CK_ = 16, bool BNRED = false>
static int igemm_h_launch(IgemmParams p, hipStream_t st, const char* name, long* query = nullptr) {
  // query != NULL: no launch, no HIP call -- *query = floats of split-K scratch this launch would use (0: it never splits)
  using Cfg = IgemmH<KH, KW, S, PAD, TW, WAVES_M, MB, NB, CK_>;
  auto kern = igemm_f16x3_kernel<KH, KW, S, PAD, TW, WAVES_M, MB, NB, CK_, BNRED>;
  static DcLdsAttr lds_attr;      // one per template instantiation; per-device inside
  if (!query)
    if (int rc = dc_func_max_lds(lds_attr, reinterpret_cast<const void*>(kern), Cfg::LDS_BYTES + 8 * 1024, name)) return rc;
  p.tilesX = dc_cdiv(p.Wout, TW);
  p.tilesY = dc_cdiv(p.Hout, Cfg::TH);
  p.walk = dc_tile_walk();
  dim3 grid((unsigned)(p.N * p.tilesX * p.tilesY * dc_cdiv(p.Ncols, Cfg::BN)));
  const int lds = Cfg::LDS_BYTES + (p.inSc ? 8 * ((p.Cin + 3) & ~3) : 0);
  DC_REQUIRE(lds <= Cfg::LDS_BYTES + 8 * 1024, DC_EUNSUP, "%s: Cin=%d too large for the BN-on-load table", name, p.Cin);
  // Split-K for the narrow layers (images of at most 16 x 16 pixels: 256 / 512 channels at the reference's own training
  // windows): training forwards (bias + BatchNorm partials) and plain data gradients whose grid leaves most of the chip idle.
  // 2-4 slabs, at least 2 chunks per slab; the S partial tiles are added in index order by splitk_combine_kernel.
  int split = 1;
  const int nch = dc_cdiv(p.Cin, Cfg::CK);
  if (!BNRED && TW <= 16 && p.scatterCo == 0 && p.outAbsmax == nullptr && p.scale == nullptr && p.relu == 0 && p.poolOut == nullptr &&
      p.bnPartial == nullptr && grid.x < 256 && nch >= 4) {
    // (same-box sweep, 128^2 x 20 step: no split 3.10-3.12 ms, <= 2 slabs 3.07-3.09, <= 4 slabs aiming at 512 workgroups
    // 3.01-3.04, <= 8 / 1 024 3.07; the 96^2 x 32 step is flat)
    split = (int)(512 / grid.x);
    split = split > 4 ? 4 : split;
    split = split > nch / 2 ? nch / 2 : split;
  }
  if (query) {
    *query = split > 1 ? (long)p.N * p.Hout * p.Wout * p.Ncols * split : 0;
    return DC_OK;
  }
  if (split > 1 && p.splitWs != nullptr) {        // the caller's scratch (dc_*_splitk_ws_floats()); none: one un-split launch
    const long slab = (long)p.N * p.Hout * p.Wout * p.Ncols;
    DC_REQUIRE(dc_aligned16(p.splitWs), DC_EINVAL, "%s: splitk_ws must be 16-byte aligned", name);
    IgemmParams q = p;
    q.splitSlab = slab;
    grid.y = (unsigned)split;
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, q);
    DC_CHECK_LAUNCH(name);
    hipLaunchKernelGGL(splitk_combine_kernel, dim3((unsigned)(p.N * p.tilesX * p.tilesY), (unsigned)dc_cdiv(p.Ncols, 16)), dim3(256), 0, st,
                       q.splitWs, split, slab, p.bias, p.out, p.outLd, p.stats, p.Hout, p.Wout, p.Ncols, TW, Cfg::TH, p.tilesX,
                       p.tilesY);
    DC_CHECK_LAUNCH("splitk_combine");
    return DC_OK;
  }
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, p);
  DC_CHECK_LAUNCH(name);
  return DC_OK;
}

// Same tile-shape choice (and therefore the same `tiles` count for the BN partials) as the fp32 kernel.
bool dc_igemm_pp_serves(const IgemmParams& p);                         // igemm_pp.hip: persistent role-split variant
int dc_igemm_pp_stats_rows(const IgemmParams& p);
int dc_igemm_pp_launch(IgemmParams p, hipStream_t st, const char* name);
static int conv3x3_h_launch(IgemmParams p, hipStream_t st, long* query = nullptr) {
  if (query) *query = 0;
  if (dc_igemm_pp_serves(p)) return query ? DC_OK : dc_igemm_pp_launch(p, st, "conv3x3_f16x3_pp");
  if (p.Wout > 16) {
    if (query) return DC_OK;                       // (the wide-tile instantiations carry no split-K code)
    if (p.Ncols <= 32) return igemm_h_launch<3, 3, 1, 1, 32, 4, 4, 1>(p, st, "conv3x3_f16x3");
    return igemm_h_launch<3, 3, 1, 1, 32, 4, 2, 2>(p, st, "conv3x3_f16x3");
  }
  // The narrow layers at small batch x window (the reference's 128^2 x 20 training configuration: 16^2 and 8^2 images at
  // 256 / 512 channels) give 80 workgroups of 64 / 128 columns: half-width column blocks double the workgroups -- same
  // pixel tiles, so the BatchNorm-partial tile count does not change.
  constexpr bool narrow = true;
  if (p.Wout > 8) {
    const long wgs = (long)p.N * dc_cdiv(p.Wout, 16) * dc_cdiv(p.Hout, 16) * dc_cdiv(p.Ncols, 64);
    if (narrow && wgs < 256 && p.Ncols > 32) return igemm_h_launch<3, 3, 1, 1, 16, 4, 2, 1>(p, st, "conv3x3_f16x3", query);
    return igemm_h_launch<3, 3, 1, 1, 16, 4, 2, 2>(p, st, "conv3x3_f16x3", query);
  }
  const long wgs = (long)p.N * dc_cdiv(p.Wout, 8) * dc_cdiv(p.Hout, 8) * dc_cdiv(p.Ncols, 128);
  if (narrow && wgs < 256 && p.Ncols > 64) return igemm_h_launch<3, 3, 1, 1, 8, 2, 1, 1>(p, st, "conv3x3_f16x3", query);
  return igemm_h_launch<3, 3, 1, 1, 8, 2, 1, 2>(p, st, "conv3x3_f16x3", query);
}

// Conv2DTranspose forward = a 1x1 contraction into 4*Cout columns with the scatter epilogue.  The tile shapes
// (and so the BN-partial `tiles` count) are those of the fp32 kernel (dc_convT2x2_tiles).
static int convT_fwd_h_launch(IgemmParams p, hipStream_t st) {
  if (p.Wout > 16) {
    if (p.Ncols <= 32) return igemm_h_launch<1, 1, 1, 0, 32, 4, 4, 1, 32>(p, st, "convT2x2_fwd_f16x3");
    // With ONE tap a staged input tile feeds only 24 MFMAs per wave and chunk: wider column blocks (all four waves side by side on
    // 2 x 32 pixels x 256 columns, or 2 x 2 on 4 x 32 x 128) split a quarter / half of the input elements per MFMA -- the staging
    // VALU, not the matrix pipe, bounds these launches.  Measured (scripts/one_convT.py, u0..u3): 200 / 143 / 97 / 82 -> 179 / 123 /
    // 82 / 72 us (round 3).  dc_convT2x2_f16x3_tiles() reports the resulting BatchNorm-partial row count.
    // Round 5: 16-channel chunks instead of 32 (half the LDS per workgroup: these one-tap launches are occupancy / latency-bound -- matrix
    // pipe 21 % busy, HBM 45 % --, not staging-bound alone), and the 2 x 2 wave layout for 256 columns as well.  scripts/one_convT.py,
    // u0..u3 at batch 16: 173 / 128 / 89 / 71 -> 158 / 112 / 73 / 74 us (64-channel chunks: 202 / 190 / 136 / 104).
    if (p.Ncols % 512 == 0) return igemm_h_launch<1, 1, 1, 0, 32, 1, 2, 2, 16>(p, st, "convT2x2_fwd_f16x3");
    if (p.Ncols % 128 == 0) return igemm_h_launch<1, 1, 1, 0, 32, 2, 2, 2, 16>(p, st, "convT2x2_fwd_f16x3");
    return igemm_h_launch<1, 1, 1, 0, 32, 4, 2, 2, 32>(p, st, "convT2x2_fwd_f16x3");
  }
  // (16-channel chunks change nothing for the 16^2 / 8^2 inputs of the 128^2 windows: 20-21 us either way)
  if (p.Wout > 8) return igemm_h_launch<1, 1, 1, 0, 16, 4, 2, 2, 32>(p, st, "convT2x2_fwd_f16x3");
  return igemm_h_launch<1, 1, 1, 0, 8, 2, 1, 2, 32>(p, st, "convT2x2_fwd_f16x3");
}
extern "C" int dc_convT2x2_f16x3_tiles(int N, int H, int W, int Cout) {
  if (N <= 0 || H <= 0 || W <= 0 || Cout <= 0) return 0;
  const int Ncols = 4 * Cout;
  int tw, th;                                       // the tile shapes of convT_fwd_h_launch above
  if (W > 16) { tw = 32; th = Ncols <= 32 ? 16 : (Ncols % 512 == 0 ? 2 : (Ncols % 128 == 0 ? 4 : 8)); }
  else if (W > 8) { tw = 16; th = 16; }
  else { tw = 8; th = 8; }
  return N * dc_cdiv(W, tw) * dc_cdiv(H, th);
}
// Conv2DTranspose dgrad: 2x2 taps over the stride-2 gradient image (no BN partials => free choice of tile).
// ... with the pass-1 sums of the layer whose `da` it writes (W > 16 only: dc_convT2x2_dgrad_bnred_blocks())
static int convT_dgrad_bnred_h_launch(IgemmParams p, hipStream_t st) {
  if (p.Ncols >= 128) return igemm_h_launch<2, 2, 2, 0, 32, 4, 1, 4, 16, true>(p, st, "convT2x2_dgrad_bnred_f16x3");
  return igemm_h_launch<2, 2, 2, 0, 32, 4, 1, 2, 16, true>(p, st, "convT2x2_dgrad_bnred_f16x3");
}
static int convT_dgrad_h_launch(IgemmParams p, hipStream_t st, long* query = nullptr) {
  if (query) *query = 0;
  // >= 128 columns: 128-column workgroups (each staged dz chunk feeds twice the MFMAs: 82 -> 70 us on the deep layers; the
  // same widening of the conv-transpose FORWARD, 8 accumulator blocks per wave, ran twice as slow)
  if (p.Wout > 16 && query) return DC_OK;
  if (p.Wout > 16 && p.Ncols >= 128) return igemm_h_launch<2, 2, 2, 0, 32, 4, 1, 4>(p, st, "convT2x2_dgrad_f16x3");
  if (p.Wout > 16) return igemm_h_launch<2, 2, 2, 0, 32, 4, 1, 2>(p, st, "convT2x2_dgrad_f16x3");
  if (p.Wout > 8) return igemm_h_launch<2, 2, 2, 0, 16, 4, 1, 2>(p, st, "convT2x2_dgrad_f16x3", query);
  return igemm_h_launch<2, 2, 2, 0, 8, 2, 1, 2>(p, st, "convT2x2_dgrad_f16x3", query);
}

// dst (fp16 pairs, same byte size as the fp32 source) + a 16-byte trailer {w_scale, max|src| bits, 0, 0}:
//   [tap][K/8][hi|lo][n][8 halfs]  with  value(tap,k,n) = w_scale * src[(flip ? taps-1-tap : tap)*s_tap + k*s_k + n*s_n]
// w_scale = the power of two that brings max|src| into [2^10, 2^11) (1 for an all-zero kernel): the fp16 hi/lo split is
// relative only inside fp16's NORMAL range, so unscaled he-normal weights of ~0.02 (lo in the subnormals) or trained
// kernels of 1e-4 would lose bits; the contraction kernels undo the scale exactly in their epilogue.
// One job = DC_PACK_JOB_LONGS longs { src pointer, dst pointer, taps, K, Ncols, s_tap, s_k, s_n, flip, first block };
// a trailing sentinel entry carries the grid size in its last field.  Three launches: zero the trailers' max slots,
// max|src| per job (order-independent atomic max on the float bits; consecutive jobs over one source share the sweep), scale + split.
#define DC_PACK_JOB_LONGS 10
struct PackJob {
  const float* src; _Float16* dst; int taps, K, Ncols, flip, b0, nb; long s_tap, s_k, s_n, total; float* trailer;
  const long* next;   // batch: the job entries behind this one; `ndup` of them pack the SAME source (forward + data-gradient image of
  int ndup, dup;      // one layer): their max slots are written here; dup: this job's source is the previous job's (max taken there)
};
__device__ __forceinline__ float* pack_trailer_of(const long* jb) {
  return reinterpret_cast<float*>(reinterpret_cast<_Float16*>(jb[1]) + 2 * (jb[2] * ((jb[3] + 7) >> 3) * 8 * jb[4]));
}
__device__ __forceinline__ PackJob pack_job(const long* __restrict__ jobs, int njobs) {
  // which job is this block's: one table entry per thread (a serial walk over the 44 entries of the U-Net's table costs every block
  // ~44 dependent loads: measured, the walk -- not the sweep -- was what these launches took)
  __shared__ int sj;
  for (int t = threadIdx.x; t < njobs; t += blockDim.x)
    if (jobs[t * DC_PACK_JOB_LONGS + 9] <= (long)blockIdx.x && (long)blockIdx.x < jobs[(t + 1) * DC_PACK_JOB_LONGS + 9]) sj = t;
  __syncthreads();
  const int j = sj;
  const long* jb = jobs + (long)j * DC_PACK_JOB_LONGS;
  PackJob q;
  q.src = reinterpret_cast<const float*>(jb[0]);
  q.dst = reinterpret_cast<_Float16*>(jb[1]);
  q.taps = (int)jb[2]; q.K = (int)jb[3]; q.Ncols = (int)jb[4]; q.flip = (int)jb[8];
  q.s_tap = jb[5]; q.s_k = jb[6]; q.s_n = jb[7];
  q.b0 = (int)jb[9]; q.nb = (int)jb[DC_PACK_JOB_LONGS + 9] - q.b0;
  q.total = (long)q.taps * ((q.K + 7) >> 3) * 8 * q.Ncols;
  q.trailer = reinterpret_cast<float*>(q.dst + 2 * q.total);
  q.dup = j > 0 && jb[-DC_PACK_JOB_LONGS] == jb[0];
  q.next = jb + DC_PACK_JOB_LONGS;
  q.ndup = 0;
  while (j + 1 + q.ndup < njobs && q.next[(long)q.ndup * DC_PACK_JOB_LONGS] == jb[0]) ++q.ndup;
  return q;
}
// One 16-byte slot = 8 consecutive k of one (tap, n): one hi and one lo store.  32-bit index arithmetic (a job has at most
// taps * K/8 * Ncols = 9 * 128 * 1024 slots).  A thread takes DC_PACK_UNROLL slots per round, all loads first: these launches are
// short (15 M weights), what they cost is the dependent-load latency per block, not bandwidth.
#define DC_PACK_UNROLL 4
struct PackSlot { const float* base; _Float16* d; int kleft; };
__device__ __forceinline__ PackSlot pack_slot_of(const PackJob& q, unsigned sidx) {
  const unsigned K8 = (unsigned)(q.K + 7) >> 3, nc = (unsigned)q.Ncols;
  const unsigned n = sidx % nc, r = sidx / nc;
  const unsigned k8 = r % K8, tap = r / K8;
  const unsigned ts = q.flip ? ((unsigned)q.taps - 1u - tap) : tap;
  PackSlot s;
  s.base = q.src + (long)ts * q.s_tap + (long)(k8 * 8u) * q.s_k + (long)n * q.s_n;
  s.d = q.dst + ((long)(tap * K8 + k8) * 2 * nc + n) * 8;
  s.kleft = q.K - (int)(k8 * 8u);
  return s;
}
__global__ void pack_zero_kernel(const long* __restrict__ jobs, int njobs) {
  for (int j = threadIdx.x; j < njobs; j += blockDim.x) {
    const long* jb = jobs + (long)j * DC_PACK_JOB_LONGS;
    const long total = jb[2] * ((jb[3] + 7) >> 3) * 8 * jb[4];
    reinterpret_cast<unsigned*>(reinterpret_cast<_Float16*>(jb[1]) + 2 * total)[1] = 0u;
  }
}
__device__ __forceinline__ void pack_absmax_body(const PackJob& q) {
  // the source tensor is contiguous (taps * K * Ncols floats, K a multiple of 4): a linear float4 sweep, one atomic per block
  __shared__ float wmax[4];
  const long n4 = (long)q.taps * q.K * q.Ncols / 4;
  const f32x4* s4 = reinterpret_cast<const f32x4*>(q.src);
  const long stride = (long)q.nb * blockDim.x;
  float m = 0.f;
  for (long i = (blockIdx.x - q.b0) * (long)blockDim.x + threadIdx.x; i < n4; i += DC_PACK_UNROLL * stride) {
    f32x4 v[DC_PACK_UNROLL];
#pragma unroll
    for (int u = 0; u < DC_PACK_UNROLL; ++u) v[u] = (i + u * stride < n4) ? s4[i + u * stride] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < DC_PACK_UNROLL; ++u)
      m = fmaxf(fmaxf(m, fmaxf(fabsf(v[u][0]), fabsf(v[u][1]))), fmaxf(fabsf(v[u][2]), fabsf(v[u][3])));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    dc_atomic_absmax(q.trailer + 1, m);
    for (int d = 0; d < q.ndup; ++d) dc_atomic_absmax(pack_trailer_of(q.next + (long)d * DC_PACK_JOB_LONGS) + 1, m);
  }
}
__device__ __forceinline__ void pack_split_body(const PackJob& q) {
  const float amax = q.trailer[1];
  const unsigned ex = (__builtin_bit_cast(unsigned, amax) >> 23) & 0xffu;
  // 2^(10 - floor(log2 amax)); all-zero / denormal / non-finite kernels stay unscaled
  const float ws = (amax > 0.f && ex >= 14u && ex != 0xffu) ? __builtin_bit_cast(float, (264u - ex) << 23) : 1.f;
  if (blockIdx.x == (unsigned)q.b0 && threadIdx.x == 0) q.trailer[0] = ws;
  const unsigned slots = (unsigned)(q.total >> 3), stride = (unsigned)q.nb * blockDim.x;
  for (unsigned i = (blockIdx.x - q.b0) * blockDim.x + threadIdx.x; i < slots; i += DC_PACK_UNROLL * stride) {
    PackSlot sl[DC_PACK_UNROLL];
    float x[DC_PACK_UNROLL][8];
#pragma unroll
    for (int u = 0; u < DC_PACK_UNROLL; ++u) {
      const bool on = i + u * stride < slots;
      sl[u] = pack_slot_of(q, on ? i + u * stride : i);
      if (!on) sl[u].kleft = 0;
#pragma unroll
      for (int e = 0; e < 8; ++e) x[u][e] = (e < sl[u].kleft) ? sl[u].base[e * q.s_k] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < DC_PACK_UNROLL; ++u) {
      if (i + u * stride >= slots) break;
      f16x8 hi, lo;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xs = x[u][e] * ws;
        const _Float16 h = (_Float16)xs;
        hi[e] = h;
        lo[e] = (_Float16)(xs - (float)h);
      }
      *reinterpret_cast<f16x8*>(sl[u].d) = hi;
      *reinterpret_cast<f16x8*>(sl[u].d + (long)q.Ncols * 8) = lo;
    }
  }
}
__global__ __launch_bounds__(256) void pack_absmax_kernel(const long* __restrict__ jobs, int njobs) {
  const PackJob q = pack_job(jobs, njobs);
  if (!q.dup) pack_absmax_body(q);
}
__global__ __launch_bounds__(256) void pack_weights_f16x3_batch_kernel(const long* __restrict__ jobs, int njobs) {
  pack_split_body(pack_job(jobs, njobs));
}
__global__ __launch_bounds__(256) void pack1_absmax_kernel(PackJob q) { pack_absmax_body(q); }
__global__ __launch_bounds__(256) void pack1_split_kernel(PackJob q) { pack_split_body(q); }

extern "C" int dc_pack_weights_f16x3(const float* src, void* dst, int taps, int K, int Ncols, long s_tap, long s_k,
                                     long s_n, int flip, dc_stream_t stream) {
  DC_REQUIRE(src && dst, DC_EINVAL, "dc_pack_weights_f16x3: null pointer");
  DC_REQUIRE(taps > 0 && K > 0 && Ncols > 0 && K % 4 == 0, DC_EINVAL, "dc_pack_weights_f16x3: K=%d must be a positive multiple of 4", K);
  PackJob q;
  q.src = src; q.dst = reinterpret_cast<_Float16*>(dst); q.taps = taps; q.K = K; q.Ncols = Ncols; q.flip = flip;
  q.s_tap = s_tap; q.s_k = s_k; q.s_n = s_n;
  q.total = (long)taps * ((K + 7) / 8) * 8 * Ncols;
  q.trailer = reinterpret_cast<float*>(q.dst + 2 * q.total);
  q.next = nullptr; q.ndup = 0; q.dup = 0;
  q.b0 = 0;
  q.nb = (int)((q.total + 255) / 256 > 4096 ? 4096 : (q.total + 255) / 256);
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(q.trailer, 0, 16, st);
  DC_REQUIRE(e == hipSuccess, DC_EHIP, "dc_pack_weights_f16x3: hipMemsetAsync: %s", hipGetErrorString(e));
  hipLaunchKernelGGL(pack1_absmax_kernel, dim3(q.nb), dim3(256), 0, st, q);
  hipLaunchKernelGGL(pack1_split_kernel, dim3(q.nb), dim3(256), 0, st, q);
  DC_CHECK_LAUNCH("dc_pack_weights_f16x3");
  return DC_OK;
}

extern "C" int dc_pack_weights_f16x3_batch(const long* jobs_dev, int njobs, int total_blocks, dc_stream_t stream) {
  DC_REQUIRE(jobs_dev && njobs > 0 && total_blocks > 0, DC_EINVAL, "dc_pack_weights_f16x3_batch: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(pack_zero_kernel, dim3(1), dim3(64), 0, st, jobs_dev, njobs);
  hipLaunchKernelGGL(pack_absmax_kernel, dim3(total_blocks), dim3(256), 0, st, jobs_dev, njobs);
  hipLaunchKernelGGL(pack_weights_f16x3_batch_kernel, dim3(total_blocks), dim3(256), 0, st, jobs_dev, njobs);
  DC_CHECK_LAUNCH("dc_pack_weights_f16x3_batch");
  return DC_OK;
}

extern "C" long dc_pack_weights_f16x3_floats(int taps, int K, int Ncols) {
  return (long)taps * ((K + 7) / 8) * 8 * Ncols + 4;   // 2 halfs per element = one float's worth of bytes; + trailer
}

static int check_h(const char* fn, const void* a, const void* b, const void* c, int N, int H, int W, int Cin, int Cout) {
  DC_REQUIRE(a && b && c, DC_EINVAL, "%s: null pointer", fn);
  DC_REQUIRE(dc_aligned16(a) && dc_aligned16(b) && dc_aligned16(c), DC_EINVAL, "%s: pointers must be 16-byte aligned", fn);
  DC_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, DC_EINVAL, "%s: non-positive dimension", fn);
  DC_REQUIRE(Cin % 4 == 0, DC_EUNSUP, "%s: Cin=%d must be a multiple of 4", fn, Cin);
  return DC_OK;
}

// BatchNorm-partial rows of a forward launch with stats_rows = this value (dcunet.h): one row per (workgroup, consumer set) where
// the role-split kernel serves the launch, one per pixel tile elsewhere.
static int stats_rows_of(int N, int H, int W, int Cin, int Cout) {
  double dstat = 0.0;
  IgemmParams p{};
  p.N = N; p.Hin = H; p.Win = W; p.Hout = H; p.Wout = W; p.Cin = Cin; p.Ncols = Cout; p.stats = &dstat;
  p.biasMod = Cout; p.outLd = Cout;
  const int tiles = dc_conv3x3_tiles(N, H, W, Cout);
  if (!dc_igemm_pp_serves(p)) return tiles;
  const int rows = dc_igemm_pp_stats_rows(p);
  return rows > 0 && rows < tiles ? rows : tiles;
}
extern "C" int dc_conv3x3_stats_rows(int N, int H, int W, int Cin, int Cout) {
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return 0;
  return stats_rows_of(N, H, W, Cin, Cout);
}
// -> IgemmParams::statsPerWg for a caller-given stats_rows (0 / tiles: per tile), or an error
static int stats_mode(const char* fn, const double* stats, int stats_rows, int N, int H, int W, int Cin, int Cout, int* per_wg) {
  *per_wg = 0;
  if (stats == nullptr || stats_rows == 0) return DC_OK;
  const int tiles = dc_conv3x3_tiles(N, H, W, Cout);
  if (stats_rows == tiles) return DC_OK;
  DC_REQUIRE(stats_rows == stats_rows_of(N, H, W, Cin, Cout), DC_EINVAL,
             "%s: stats_rows=%d is neither dc_conv3x3_tiles()=%d nor dc_conv3x3_stats_rows()", fn, stats_rows, tiles);
  *per_wg = 1;
  return DC_OK;
}

extern "C" int dc_conv3x3_fwd_f16x3(const float* x, const void* wp16, const float* bias, float* z, long z_ld,
                                    double* stats, int stats_rows, const float* scale, const float* shift, int relu,
                                    const float* in_abound, long in_abound_ld, float* out_absmax, long out_absmax_ld,
                                    float* splitk_ws, int N, int H, int W, int Cin, int Cout,
                                    dc_stream_t stream) {
  int rc = check_h("dc_conv3x3_fwd_f16x3", x, wp16, z, N, H, W, Cin, Cout);
  if (rc) return rc;
  DC_REQUIRE((scale == nullptr) == (shift == nullptr), DC_EINVAL, "dc_conv3x3_fwd_f16x3: scale and shift go together");
  DC_REQUIRE(!(stats && out_absmax), DC_EINVAL, "dc_conv3x3_fwd_f16x3: stats (training) and out_absmax (inference) are exclusive");
  DC_REQUIRE(z_ld >= Cout, DC_EINVAL, "dc_conv3x3_fwd_f16x3: z_ld < Cout");
  IgemmParams p{};
  p.in = x; p.wp = reinterpret_cast<const float*>(wp16); p.bias = bias; p.out = z; p.stats = stats;
  p.scale = scale; p.shift = shift; p.inAbound = in_abound; p.inAboundLd = in_abound_ld;
  p.outAbsmax = out_absmax; p.outAbsmaxLd = out_absmax_ld;
  p.N = N; p.Hin = H; p.Win = W; p.Cin = Cin; p.Hout = H; p.Wout = W; p.Ncols = Cout;
  p.relu = relu; p.scatterCo = 0; p.biasMod = Cout; p.outLd = z_ld; p.splitWs = splitk_ws;
  if ((rc = stats_mode("dc_conv3x3_fwd_f16x3", stats, stats_rows, N, H, W, Cin, Cout, &p.statsPerWg))) return rc;
  return conv3x3_h_launch(p, (hipStream_t)stream);
}

// Split-K scratch of a conv3x3 launch (dcunet.h): the routing and split decision of conv3x3_h_launch / igemm_h_launch for the
// only launch forms that split -- the training forward (bias + BatchNorm partials, no fused epilogue) and the plain data gradient.
extern "C" long dc_conv3x3_splitk_ws_floats(int N, int H, int W, int Cin, int Cout, int dgrad) {
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return 0;
  float dummy = 1.f;
  double dstat = 0.0;
  IgemmParams p{};
  p.N = N; p.Hin = H; p.Win = W; p.Hout = H; p.Wout = W;
  if (dgrad) { p.Cin = Cout; p.Ncols = Cin; p.inScale = &dummy; }
  else { p.Cin = Cin; p.Ncols = Cout; p.stats = &dstat; }
  p.biasMod = p.Ncols; p.outLd = p.Ncols;
  long need = 0;
  if (conv3x3_h_launch(p, nullptr, &need) != DC_OK) return 0;
  return need;
}
extern "C" long dc_convT2x2_dgrad_splitk_ws_floats(int N, int H, int W, int Cin, int Cout) {
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return 0;
  float dummy = 1.f;
  IgemmParams p{};
  p.inScale = &dummy;
  p.N = N; p.Hin = 2 * H; p.Win = 2 * W; p.Cin = Cout; p.Hout = H; p.Wout = W; p.Ncols = Cin;
  p.biasMod = Cin; p.outLd = Cin;
  long need = 0;
  if (convT_dgrad_h_launch(p, nullptr, &need) != DC_OK) return 0;
  return need;
}

extern "C" int dc_conv3x3_fwd_bnin_f16x3(const float* z_in, const float* in_sc, const float* in_sh, const float* in_abound,
    const void* wp16, const float* bias, float* z, long z_ld, double* stats, int stats_rows, const float* scale,
                                         const float* shift, int relu, float* splitk_ws, int N, int H, int W, int Cin,
                                         int Cout, dc_stream_t stream) {
  int rc = check_h("dc_conv3x3_fwd_bnin_f16x3", z_in, wp16, z, N, H, W, Cin, Cout);
  if (rc) return rc;
  DC_REQUIRE(in_sc && in_sh, DC_EINVAL, "dc_conv3x3_fwd_bnin_f16x3: null input scale/shift");
  DC_REQUIRE((scale == nullptr) == (shift == nullptr), DC_EINVAL, "dc_conv3x3_fwd_bnin_f16x3: scale and shift go together");
  DC_REQUIRE(z_ld >= Cout, DC_EINVAL, "dc_conv3x3_fwd_bnin_f16x3: z_ld < Cout");
  IgemmParams p{};
  p.in = z_in; p.wp = reinterpret_cast<const float*>(wp16); p.bias = bias; p.out = z; p.stats = stats;
  p.scale = scale; p.shift = shift; p.inSc = in_sc; p.inSh = in_sh; p.inAbound = in_abound;
  p.N = N; p.Hin = H; p.Win = W; p.Cin = Cin; p.Hout = H; p.Wout = W; p.Ncols = Cout;
  p.relu = relu; p.scatterCo = 0; p.biasMod = Cout; p.outLd = z_ld; p.splitWs = splitk_ws;
  if ((rc = stats_mode("dc_conv3x3_fwd_bnin_f16x3", stats, stats_rows, N, H, W, Cin, Cout, &p.statsPerWg))) return rc;
  return conv3x3_h_launch(p, (hipStream_t)stream);
}

extern "C" int dc_convT2x2_fwd_bnin_f16x3(const float* z_in, const float* in_sc, const float* in_sh, const float* in_abound,
    const void* wp16, const float* bias, float* z, long z_ld, double* stats, const float* scale,
                                          const float* shift, int relu, int N, int H, int W, int Cin, int Cout,
                                          dc_stream_t stream) {
  int rc = check_h("dc_convT2x2_fwd_bnin_f16x3", z_in, wp16, z, N, H, W, Cin, Cout);
  if (rc) return rc;
  DC_REQUIRE(in_sc && in_sh, DC_EINVAL, "dc_convT2x2_fwd_bnin_f16x3: null input scale/shift");
  DC_REQUIRE((scale == nullptr) == (shift == nullptr), DC_EINVAL, "dc_convT2x2_fwd_bnin_f16x3: scale and shift go together");
  DC_REQUIRE(z_ld >= Cout, DC_EINVAL, "dc_convT2x2_fwd_bnin_f16x3: z_ld < Cout");
  IgemmParams p{};
  p.in = z_in; p.wp = reinterpret_cast<const float*>(wp16); p.bias = bias; p.out = z; p.stats = stats;
  p.scale = scale; p.shift = shift; p.inSc = in_sc; p.inSh = in_sh; p.inAbound = in_abound;
  p.N = N; p.Hin = H; p.Win = W; p.Cin = Cin; p.Hout = H; p.Wout = W; p.Ncols = 4 * Cout;
  p.relu = relu; p.scatterCo = Cout; p.biasMod = Cout; p.outLd = z_ld;
  return convT_fwd_h_launch(p, (hipStream_t)stream);
}

static int check_absmax(const char* fn, const float* in_scale, const float* in_absmax, int n) {
  DC_REQUIRE(!(in_scale && in_absmax), DC_EINVAL, "%s: pass dz_scale OR dz_absmax, not both", fn);
  DC_REQUIRE(in_absmax == nullptr || (n > 0 && n <= 65536), DC_EINVAL, "%s: dz_absmax_n=%d out of range", fn, n);
  return DC_OK;
}
extern "C" int dc_conv3x3_dgrad_f16x3(const float* dz, const void* wp16, float* dx, const float* in_scale,
                                      const float* in_absmax, int in_absmax_n, float* splitk_ws, int N, int H,
                                      int W, int Cin, int Cout, dc_stream_t stream) {
  int rc = check_h("dc_conv3x3_dgrad_f16x3", dz, wp16, dx, N, H, W, Cout, Cin);
  if (rc) return rc;
  if ((rc = check_absmax("dc_conv3x3_dgrad_f16x3", in_scale, in_absmax, in_absmax_n))) return rc;
  IgemmParams p{};
  p.in = dz; p.wp = reinterpret_cast<const float*>(wp16); p.out = dx; p.inScale = in_scale;
  p.inAbsmax = in_absmax; p.inAbsmaxN = in_absmax_n;
  p.N = N; p.Hin = H; p.Win = W; p.Cin = Cout; p.Hout = H; p.Wout = W; p.Ncols = Cin;
  p.biasMod = Cin; p.outLd = Cin; p.splitWs = splitk_ws;
  return conv3x3_h_launch(p, (hipStream_t)stream);
}

// Which kernel a conv3x3 launch of this shape runs: > 0 (the pixel-tile count) when dc_conv3x3_fwd*_f16x3 (dgrad == 0;
// with_stats: BatchNorm partials requested) / dc_conv3x3_dgrad*_f16x3 (dgrad != 0) is served by the persistent role-split
// kernel of igemm_pp.hip.  Depends on the shape and DC_IGEMM_PP only -- not on the fusion knobs -- so the backward schedule
// and bench.py's per-symbol timer can ask it.
extern "C" int dc_conv3x3_pp_blocks(int N, int H, int W, int Cin, int Cout, int dgrad, int with_stats) {
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return 0;
  float dummy = 1.f;
  double dstat = 0.0;
  IgemmParams p{};
  p.N = N; p.Hin = H; p.Win = W; p.Hout = H; p.Wout = W;
  if (dgrad) { p.Cin = Cout; p.Ncols = Cin; p.inScale = &dummy; }
  else { p.Cin = Cin; p.Ncols = Cout; p.stats = with_stats ? &dstat : nullptr; }
  p.biasMod = p.Ncols; p.outLd = p.Ncols;
  if (!dc_igemm_pp_serves(p)) return 0;
  return N * dc_cdiv(W, 32) * dc_cdiv(H, p.Ncols <= 32 ? 16 : 8);
}

// Inference convolution whose output also feeds MaxPooling2D((2,2)): role-split kernel only
// (dc_conv3x3_fwd_pool_blocks() == 0: use dc_conv3x3_fwd_f16x3 + dc_maxpool2x2_fwd).
static IgemmParams fwd_pool_params(const float* x, const void* wp16, const float* bias, float* z, long z_ld, const float* scale,
                                   const float* shift, int relu, float* out_flag, float* pool, int N, int H, int W, int Cin,
                                   int Cout) {
  IgemmParams p{};
  p.in = x; p.wp = reinterpret_cast<const float*>(wp16); p.bias = bias; p.out = z;
  p.scale = scale; p.shift = shift; p.outAbsmax = out_flag; p.outAbsmaxLd = -1;
  p.N = N; p.Hin = H; p.Win = W; p.Cin = Cin; p.Hout = H; p.Wout = W; p.Ncols = Cout;
  p.relu = relu; p.biasMod = Cout; p.outLd = z_ld; p.poolOut = pool;
  return p;
}
extern "C" int dc_conv3x3_fwd_pool_blocks(int N, int H, int W, int Cin, int Cout) {
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || (H & 1) || (W & 1)) return 0;
  constexpr bool off = false;
  IgemmParams p = fwd_pool_params(nullptr, nullptr, nullptr, nullptr, Cout, nullptr, nullptr, 1, nullptr, nullptr, N, H, W, Cin, Cout);
  if (off || !dc_igemm_pp_serves(p)) return 0;
  return N * dc_cdiv(W, 32) * dc_cdiv(H, Cout <= 32 ? 16 : 8);
}
extern "C" int dc_conv3x3_fwd_pool_f16x3(const float* x, const void* wp16, const float* bias, float* z, long z_ld,
                                         const float* scale, const float* shift, int relu, float* out_flag, float* pool,
                                         int N, int H, int W, int Cin, int Cout, dc_stream_t stream) {
  int rc = check_h("dc_conv3x3_fwd_pool_f16x3", x, wp16, z, N, H, W, Cin, Cout);
  if (rc) return rc;
  DC_REQUIRE(pool && z_ld >= Cout && (scale == nullptr) == (shift == nullptr), DC_EINVAL, "dc_conv3x3_fwd_pool_f16x3: bad arguments");
  DC_REQUIRE(dc_conv3x3_fwd_pool_blocks(N, H, W, Cin, Cout) > 0, DC_EUNSUP,
             "dc_conv3x3_fwd_pool_f16x3: shape not served (dc_conv3x3_fwd_pool_blocks() == 0): use the two-kernel path");
  IgemmParams p = fwd_pool_params(x, wp16, bias, z, z_ld, scale, shift, relu, out_flag, pool, N, H, W, Cin, Cout);
  return dc_igemm_pp_launch(p, (hipStream_t)stream, "conv3x3_fwd_pool_f16x3_pp");
}

// Data gradient that also emits the BatchNorm-backward pass-1 sums of the layer whose `da` it writes (role-split kernel
// only: dc_conv3x3_dgrad_bnred_blocks() == 0 tells the caller to use dc_conv3x3_dgrad_f16x3 + dc_bn_bwd_reduce instead).
static IgemmParams dgrad_bnred_params(const float* dz, const void* wp16, float* dx, const float* in_scale, int N, int H, int W,
                                      int Cin, int Cout) {
  IgemmParams p{};
  p.in = dz; p.wp = reinterpret_cast<const float*>(wp16); p.out = dx; p.inScale = in_scale;
  p.N = N; p.Hin = H; p.Win = W; p.Cin = Cout; p.Hout = H; p.Wout = W; p.Ncols = Cin;
  p.biasMod = Cin; p.outLd = Cin;
  return p;
}
extern "C" int dc_conv3x3_dgrad_bnred_blocks(int N, int H, int W, int Cin, int Cout) {
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return 0;
  constexpr bool off = false;
  float dummy = 1.f;      // (a non-null inScale marks a data-gradient launch for the DC_IGEMM_PP=2 setting)
  IgemmParams p = dgrad_bnred_params(nullptr, nullptr, nullptr, &dummy, N, H, W, Cin, Cout);
  if (off || !dc_igemm_pp_serves(p)) return 0;
  return N * dc_cdiv(W, 32) * dc_cdiv(H, Cin <= 32 ? 16 : 8);
}
extern "C" int dc_conv3x3_dgrad_bnred_f16x3(const float* dz, const void* wp16, float* dx, const float* in_scale,
                                            const float* in_absmax, int in_absmax_n, const float* z, const float* mean, const float* invstd, const float* gamma,
                                            const float* beta, float* bn_partial, float* amax_partial, int N, int H, int W,
                                            int Cin, int Cout, dc_stream_t stream) {
  int rc = check_h("dc_conv3x3_dgrad_bnred_f16x3", dz, wp16, dx, N, H, W, Cout, Cin);
  if (rc) return rc;
  DC_REQUIRE(z && mean && invstd && gamma && beta && bn_partial, DC_EINVAL, "dc_conv3x3_dgrad_bnred_f16x3: null pointer");
  if ((rc = check_absmax("dc_conv3x3_dgrad_bnred_f16x3", in_scale, in_absmax, in_absmax_n))) return rc;
  IgemmParams p = dgrad_bnred_params(dz, wp16, dx, in_scale, N, H, W, Cin, Cout);
  p.inAbsmax = in_absmax; p.inAbsmaxN = in_absmax_n;
  p.bnZ = z; p.bnMean = mean; p.bnInvstd = invstd; p.bnGamma = gamma; p.bnBeta = beta; p.bnPartial = bn_partial;
  p.bnAmax = amax_partial;
  DC_REQUIRE(dc_conv3x3_dgrad_bnred_blocks(N, H, W, Cin, Cout) > 0, DC_EUNSUP,
             "dc_conv3x3_dgrad_bnred_f16x3: shape not served (dc_conv3x3_dgrad_bnred_blocks() == 0): use the two-pass path");
  return dc_igemm_pp_launch(p, (hipStream_t)stream, "conv3x3_dgrad_bnred_f16x3_pp");
}

// "dz on load" data gradient (dcunet.h): role-split kernel only.
extern "C" int dc_conv3x3_dgrad_dzin_blocks(int N, int H, int W, int Cin, int Cout) {
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return 0;
  float dummy = 1.f;
  IgemmParams p = dgrad_bnred_params(nullptr, nullptr, nullptr, nullptr, N, H, W, Cin, Cout);
  p.dzCoef = &dummy;
  if (!dc_igemm_pp_serves(p)) return 0;
  return N * dc_cdiv(W, 32) * dc_cdiv(H, Cin <= 32 ? 16 : 8);
}
extern "C" int dc_conv3x3_dgrad_dzin_f16x3(const float* da, const float* z, const float* dz_coef, const void* wp16, float* dx,
                                           float* dz_out, const float* red_z, const float* red_mean, const float* red_invstd,
                                           const float* red_gamma, const float* red_beta, float* bn_partial,
                                           float* amax_partial, int N, int H, int W, int Cin, int Cout,
                                           dc_stream_t stream) {
  int rc = check_h("dc_conv3x3_dgrad_dzin_f16x3", da, wp16, dx, N, H, W, Cout, Cin);
  if (rc) return rc;
  DC_REQUIRE(z && dz_coef && dc_aligned16(z) && dc_aligned16(dz_coef), DC_EINVAL,
             "dc_conv3x3_dgrad_dzin_f16x3: z / dz_coef must be non-null and 16-byte aligned");
  DC_REQUIRE(Cout % 4 == 0, DC_EUNSUP, "dc_conv3x3_dgrad_dzin_f16x3: Cout=%d must be a multiple of 4", Cout);
  DC_REQUIRE(red_z == nullptr || (red_mean && red_invstd && red_gamma && red_beta && bn_partial), DC_EINVAL,
             "dc_conv3x3_dgrad_dzin_f16x3: red_z needs red_mean / red_invstd / red_gamma / red_beta / bn_partial");
  DC_REQUIRE(dc_conv3x3_dgrad_dzin_blocks(N, H, W, Cin, Cout) > 0, DC_EUNSUP,
             "dc_conv3x3_dgrad_dzin_f16x3: shape not served (dc_conv3x3_dgrad_dzin_blocks() == 0): use dc_bn_bwd_apply + dc_conv3x3_dgrad_f16x3");
  IgemmParams p = dgrad_bnred_params(da, wp16, dx, nullptr, N, H, W, Cin, Cout);
  p.in2 = z; p.dzCoef = dz_coef; p.dzOut = dz_out;
  DC_REQUIRE(dz_out == nullptr || dc_aligned16(dz_out), DC_EINVAL, "dc_conv3x3_dgrad_dzin_f16x3: dz_out must be 16-byte aligned");
  if (red_z) {
    p.bnZ = red_z; p.bnMean = red_mean; p.bnInvstd = red_invstd; p.bnGamma = red_gamma; p.bnBeta = red_beta;
    p.bnPartial = bn_partial; p.bnAmax = amax_partial;
  }
  return dc_igemm_pp_launch(p, (hipStream_t)stream, "conv3x3_dgrad_dzin_f16x3_pp");
}

extern "C" int dc_convT2x2_fwd_f16x3(const float* x, const void* wp16, const float* bias, float* z, long z_ld,
                                     double* stats, const float* scale, const float* shift, int relu,
                                     const float* in_abound, long in_abound_ld, float* out_absmax, long out_absmax_ld,
                                    int N, int H, int W, int Cin, int Cout,
                                     dc_stream_t stream) {
  int rc = check_h("dc_convT2x2_fwd_f16x3", x, wp16, z, N, H, W, Cin, Cout);
  if (rc) return rc;
  DC_REQUIRE((scale == nullptr) == (shift == nullptr), DC_EINVAL, "dc_convT2x2_fwd_f16x3: scale and shift go together");
  DC_REQUIRE(!(stats && out_absmax), DC_EINVAL, "dc_convT2x2_fwd_f16x3: stats (training) and out_absmax (inference) are exclusive");
  DC_REQUIRE(z_ld >= Cout, DC_EINVAL, "dc_convT2x2_fwd_f16x3: z_ld < Cout");
  IgemmParams p{};
  p.in = x; p.wp = reinterpret_cast<const float*>(wp16); p.bias = bias; p.out = z; p.stats = stats;
  p.scale = scale; p.shift = shift; p.inAbound = in_abound; p.inAboundLd = in_abound_ld;
  p.outAbsmax = out_absmax; p.outAbsmaxLd = out_absmax_ld;
  p.N = N; p.Hin = H; p.Win = W; p.Cin = Cin; p.Hout = H; p.Wout = W; p.Ncols = 4 * Cout;
  p.relu = relu; p.scatterCo = Cout; p.biasMod = Cout; p.outLd = z_ld;
  return convT_fwd_h_launch(p, (hipStream_t)stream);
}

extern "C" int dc_convT2x2_dgrad_f16x3(const float* dz, const void* wp16, float* dx, const float* in_scale,
                                       const float* in_absmax, int in_absmax_n, float* splitk_ws, int N, int H,
                                       int W, int Cin, int Cout, dc_stream_t stream) {
  int rc = check_h("dc_convT2x2_dgrad_f16x3", dz, wp16, dx, N, H, W, Cout, Cin);
  if (rc) return rc;
  if ((rc = check_absmax("dc_convT2x2_dgrad_f16x3", in_scale, in_absmax, in_absmax_n))) return rc;
  IgemmParams p{};
  p.in = dz; p.wp = reinterpret_cast<const float*>(wp16); p.out = dx; p.inScale = in_scale;
  p.inAbsmax = in_absmax; p.inAbsmaxN = in_absmax_n;
  p.N = N; p.Hin = 2 * H; p.Win = 2 * W; p.Cin = Cout; p.Hout = H; p.Wout = W; p.Ncols = Cin;
  p.biasMod = Cin; p.outLd = Cin; p.splitWs = splitk_ws;
  return convT_dgrad_h_launch(p, (hipStream_t)stream);
}

// Conv2DTranspose data gradient that also emits the BatchNorm-backward pass-1 sums (and max |dy|) of the layer whose `da` it
// writes (the block in front of the up-convolution: dense [N,H,W,Cin], no Dropout, pre-BN tensor z of the same shape):
// rows = dc_convT2x2_dgrad_bnred_blocks() partial rows; 0 -> shape not served (W <= 16), use dc_convT2x2_dgrad_f16x3 +
// dc_bn_bwd_reduce.
extern "C" int dc_convT2x2_dgrad_bnred_blocks(int N, int H, int W, int Cin, int Cout) {
  if (N <= 0 || H <= 0 || W <= 16 || Cin < 4 || Cin % 4 || Cout <= 0) return 0;
  return N * dc_cdiv(W, 32) * dc_cdiv(H, 4);            // 4 x 32-pixel tiles of both instantiations
}
extern "C" int dc_convT2x2_dgrad_bnred_f16x3(const float* dz, const void* wp16, float* dx, const float* in_scale,
                                             const float* in_absmax, int in_absmax_n, const float* z, const float* mean,
                                             const float* invstd, const float* gamma, const float* beta, float* bn_partial,
                                             float* amax_partial, int N, int H, int W, int Cin, int Cout, dc_stream_t stream) {
  int rc = check_h("dc_convT2x2_dgrad_bnred_f16x3", dz, wp16, dx, N, H, W, Cout, Cin);
  if (rc) return rc;
  if ((rc = check_absmax("dc_convT2x2_dgrad_bnred_f16x3", in_scale, in_absmax, in_absmax_n))) return rc;
  DC_REQUIRE(z && mean && invstd && gamma && beta && bn_partial, DC_EINVAL, "dc_convT2x2_dgrad_bnred_f16x3: null pointer");
  DC_REQUIRE(dc_convT2x2_dgrad_bnred_blocks(N, H, W, Cin, Cout) > 0, DC_EUNSUP,
             "dc_convT2x2_dgrad_bnred_f16x3: shape not served (dc_convT2x2_dgrad_bnred_blocks() == 0): use the two-pass path");
  IgemmParams p{};
  p.in = dz; p.wp = reinterpret_cast<const float*>(wp16); p.out = dx; p.inScale = in_scale;
  p.inAbsmax = in_absmax; p.inAbsmaxN = in_absmax_n;
  p.N = N; p.Hin = 2 * H; p.Win = 2 * W; p.Cin = Cout; p.Hout = H; p.Wout = W; p.Ncols = Cin;
  p.biasMod = Cin; p.outLd = Cin;
  p.bnZ = z; p.bnMean = mean; p.bnInvstd = invstd; p.bnGamma = gamma; p.bnBeta = beta; p.bnPartial = bn_partial; p.bnAmax = amax_partial;
  return convT_dgrad_bnred_h_launch(p, (hipStream_t)stream);
}
