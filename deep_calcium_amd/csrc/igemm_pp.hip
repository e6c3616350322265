// Persistent, role-split conv3x3 implicit GEMM on the fp16 matrix cores (split-fp16 operands, numerics and data layouts of
// igemm_f16x3.hip: same packed weights, same LDS operand images, same accumulation order => bit-identical results).
//
// Why: scripts/igemm_phases.py shows a workgroup of the 256-thread kernel spending ~1.9x as long
// OUTSIDE its MFMA blocks (fp16 split + LDS writes, barriers, tile decode, epilogue) as inside them, so two such
// workgroups per CU keep the matrix pipe busy only ~53 %.  Here ONE 768-thread workgroup per CU runs for the whole launch:
//   waves 8-11  producers : request chunk k+2's global rows, split chunk k+1 into the OTHER LDS stage (BN + ReLU on load,
//                           range-guard scale) -- always one step ahead, across tile boundaries (no prologue bubble);
//   waves 0-3   consumers A, waves 4-7 consumers B: alternate output tiles.  While one set runs the 108 MFMAs of a step on
//                           the stage that is ready, the other runs one SLICE of the epilogue of its previous tile (bias,
//                           BatchNorm partials, stores): the matrix pipe of every SIMD always has exactly one wave feeding it.
// One s_barrier per step (a 16-channel chunk).  Every SIMD hosts one wave of each role.
// Served launches: conv3x3 forward (training, BN-on-load, inference) and data gradient with W > 16, > 32 output columns,
// Cin a multiple of 16 and >= 64 (>= 4 steps per tile for the sliced epilogue); everything else stays on igemm_f16x3.hip.
#include "igemm_common.h"
#include <stdlib.h>
#include <type_traits>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// Step timestamps of sampled workgroups (scripts/igemm_pp_phases.py; compiled out unless -DDC_IGEMM_TRACE): per role the
// first wave stores s_memtime when its work of a step is done and again when the step's barrier has released it.
// DC_PP_ABL (debug builds only, scripts/igemm_pp_ablate.py): bit 0 no epilogue work (the compiler then drops the MFMAs
// too: unusable), bit 1 producers do not split / write LDS, bit 2 consumers read their fragments once per step, bit 3
// producers do not load, bit 4 two more stamps per consumer step (tile setup done, first fragments landed), bit 6 interior tiles issue
// no output stores (scripts/level0_store_ablate.py), bit 5 the producers
// write the raw fp32 bits instead of the fp16 split (the LDS traffic of a PRE-SPLIT operand, none of its VALU).  Results are
// garbage for bits 0-3 and 5.
#ifndef DC_PP_ABL
#define DC_PP_ABL 0
#endif
// DC_PP_PF = 1 (round 6, built, bit-identical, measured, NOT adopted): the consumers take a step's barrier BEFORE its last tap (every
// fragment of the stage is in registers by then) and, when the next step belongs to the same tile, request that step's first four
// fragments from the other stage under the last tap's MFMAs -- meant to cover the ~350 cycles of LDS latency in front of a step's first
// MFMA (profiles/r06_pp_ablation.txt).  Measured (profiles/r06_ab.txt): step period 4 520 -> 5 116 cycles, step 17.26 -> 17.49 ms,
// forward only 3 412 -> 3 366 images/s.  The four fragments (16 VGPRs live across the step boundary) push the 768-thread workgroup's
// 168-VGPR budget over the edge (3-5 registers to scratch, reloaded inside the MFMA block), and a barrier in the middle of the MFMA
// stream stalls the pipe for the whole wait.  Round 4 had reached the same wall from the other side.
#ifndef DC_PP_PF
#define DC_PP_PF 0
#endif
#ifdef DC_PP_TIMELINE
// per workgroup: s_memrealtime (the chip-wide 100 MHz counter) at kernel entry, barrier 0 passed, last MFMA step done, exit
__device__ unsigned long long g_pp_timeline[1024 * 4];
extern "C" int dc_debug_pp_timeline(unsigned long long* out4096) {
  return hipMemcpyFromSymbol(out4096, HIP_SYMBOL(g_pp_timeline), sizeof(g_pp_timeline)) == hipSuccess ? 0 : -2;
}
#define PP_TL(i) do { if (threadIdx.x == 0 && blockIdx.x < 1024) g_pp_timeline[blockIdx.x * 4 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define PP_TL(i) do {} while (0)
#endif
#ifdef DC_IGEMM_TRACE
__device__ unsigned long long* g_pp_trace = nullptr;
extern "C" int dc_debug_set_pp_trace(unsigned long long* p) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_pp_trace), &p, sizeof(p)) == hipSuccess ? 0 : -2;
}
#define PP_TRACE_INIT()                                                                                          \
  unsigned long long* trc_ = (g_pp_trace && blockIdx.x % 37 == 0 && (threadIdx.x & 255) == 0)                    \
                                 ? g_pp_trace + ((blockIdx.x / 37) * 3 + (threadIdx.x >> 8)) * 512 : nullptr;    \
  int trn_ = 0
#define PP_TRACE() do { if (trc_ && trn_ < 510) trc_[++trn_] = __builtin_readcyclecounter(); if (trc_) trc_[0] = trn_; } while (0)
#else
#define PP_TRACE_INIT() do {} while (0)
#define PP_TRACE() do {} while (0)
#endif

namespace pp {
constexpr int KW = 3, TAPS = 9, TW = 32, WAVES_M = 4, CK = 16;
constexpr int G4 = CK / 4, G8 = CK / 8;
constexpr int TMP_BYTES = 64;
constexpr int TABLE_BYTES = 8 * 1024;          // BN-on-load table: 2 x Cin floats, Cin <= 1024
constexpr int EP_COLS = 512;                   // epilogue parameter table: 3 x Ncols floats, Ncols <= 512
constexpr int DZ_CIN = 512;                    // dz-on-load table: 6 x Cin floats, Cin <= 512
constexpr int DZ_BYTES = 6 * DZ_CIN * 4;
constexpr int THREADS = 768;
constexpr unsigned OOB = 0x80000000u;
// The two tile shapes of the 256-thread kernel (same BatchNorm-partial tile counts): <MB 2, NB 2> = 8 x 32 pixels x 64
// columns, <MB 4, NB 1> = 16 x 32 pixels x 32 columns (layers with <= 32 output columns: the 512^2 layers).
template <int MB_, int NB_>
struct Cfg {
  static constexpr int MB = MB_, NB = NB_;
  static constexpr int RPM = 32 / TW, TH = WAVES_M * MB * RPM, BN = NB * 32;
  static constexpr int THI = TH + 2, TWI = TW + 2, NPIXH = THI * TWI;
  static constexpr int PS = ((NPIXH + 5) / 8) * 8 + 2;
  static constexpr int NA = (NPIXH * G4 + 255) / 256;
  static constexpr int BROWS = TAPS * G8 * 2;
  static constexpr int NBV = (BROWS * BN + 255) / 256;
  static constexpr int A_SLOTS = 2 * G8 * PS;
  static constexpr int STAGE_SLOTS = A_SLOTS + BROWS * BN;
  static constexpr int STAGE_BYTES = STAGE_SLOTS * 16;
  static constexpr int RED_ENTRIES = WAVES_M * NB * 32;                      // per consumer set
  static constexpr int RED_BYTES = 2 * RED_ENTRIES * (int)sizeof(DcMoments);
  static constexpr int FIXED_LDS = 2 * STAGE_BYTES + RED_BYTES + TMP_BYTES;
  static constexpr int NBLK = MB * NB;                                       // 32x32 accumulator blocks per wave
  static_assert(256 % BN == 0 && NBLK == 4 && (MB == 2 || MB == 4), "two shapes: 2x2 and 4x1 blocks per wave");
};

// hi = fp16(x*s), lo = fp16(x*s - hi): two v_fma_mix per element (igemm_f16x3.hip split_f16)
__device__ __forceinline__ void split(const f32x4 v, float s, u32x2& hi, u32x2& lo) {
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

struct Item {          // one output tile x column block (wave-uniform)
  int tile_id, img, n0, oy0, ox0;
};
}  // namespace pp

// VARIANT 1 (BNRED): the data-gradient variant that also emits the BatchNorm-backward sums of the layer it writes `da` of.
// VARIANT 2 (POOL): the inference variant whose output also feeds a 2x2 max-pool: the epilogue writes the pooled tensor
// too.  Each is its own instantiation: the other epilogues are compiled out of it, and its code out of the base kernel.
// DZIN: the data-gradient launch whose input operand dz is formed on load from (da, z) -- see IgemmParams::in2.
template <int MB_, int NB_, int VARIANT = 0, bool DZIN = false>
__global__ __launch_bounds__(pp::THREADS, 1) void igemm_pp_kernel(IgemmParams p) {
  constexpr bool BNRED = VARIANT == 1, POOL = VARIANT == 2, PERWG = VARIANT == 3;
  constexpr bool PF = DC_PP_PF && !POOL;     // (the pooled-epilogue instantiation sits at the 168-VGPR budget: 34 spills with it)
  using namespace pp;
  using C = Cfg<MB_, NB_>;
  constexpr int MB = C::MB, NB = C::NB, RPM = C::RPM, TH = C::TH, BN = C::BN, THI = C::THI, TWI = C::TWI, NPIXH = C::NPIXH;
  constexpr int PS = C::PS, NA = C::NA, BROWS = C::BROWS, NBV = C::NBV, A_SLOTS = C::A_SLOTS, STAGE_BYTES = C::STAGE_BYTES;
  constexpr int RED_ENTRIES = C::RED_ENTRIES, RED_BYTES = C::RED_BYTES, FIXED_LDS = C::FIXED_LDS, NBLK = C::NBLK;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  DcMoments* red_all = reinterpret_cast<DcMoments*>(smem + 2 * STAGE_BYTES);
  float* tmp = reinterpret_cast<float*>(smem + 2 * STAGE_BYTES + RED_BYTES);
  float* lds_sc = reinterpret_cast<float*>(smem + FIXED_LDS);                 // BN-on-load (scale, shift) per input channel
  float* lds_ep = reinterpret_cast<float*>(smem + FIXED_LDS + TABLE_BYTES);   // epilogue bias | scale | shift per column
  float* lds_dz = lds_ep + 3 * EP_COLS;                                       // DZIN: sc | sh | mu | A | D | E per input channel
  // statsPerWg (forward launches with BatchNorm partials; never DZIN: the same LDS): running moments of this workgroup's tiles,
  // [consumer set][column] -- column n is only ever touched by thread n % BN of its set
  // Its own instantiation (VARIANT 3), like the other epilogue variants: the base kernel stays the verified code (DESIGN 5c:
  // one more epilogue path compiled INTO it once pushed hipcc over an SGPR-spill cliff; tests/test_abi.py).
#define PP_RUN_M (reinterpret_cast<DcMoments*>(smem + FIXED_LDS + TABLE_BYTES + 3 * EP_COLS * 4))

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // 0..11
  const int role = wave >> 2, wr = wave & 3;                      // 0, 1: consumer sets; 2: producers
  PP_TRACE_INIT();
  PP_TL(0);

  // ---- this workgroup's items: each XCD walks a contiguous range of (tile, column block) pairs, its workgroups
  // striding through it (ids b and b+8 share an L2: the workgroups that read one input patch for different columns, and
  // neighbouring patches, stay on one L2 -- speed only)
  const int nblk = (p.Ncols + BN - 1) / BN;
  const int total = p.N * p.tilesX * p.tilesY * nblk;
  const int G = (int)gridDim.x, xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
  const int nx = (G + 7 - xcd) >> 3;                               // workgroups on this XCD
  const int qq = total >> 3, rr = total & 7;
  const int xstart = xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq;
  const int xcount = qq + (xcd < rr ? 1 : 0);
  const int n_items = seq < xcount ? (xcount - seq + nx - 1) / nx : 0;
  const int nch = p.Cin / CK;
  const int K = n_items * nch;                                     // steps of this workgroup
  auto decode = [&](int j) __attribute__((always_inline)) {
    Item it;
    const int work = xstart + seq + j * nx;
    const int wpos = work / nblk;                               // position in the walk over the pixel tiles
    it.n0 = (work - wpos * nblk) * BN;
    int tx, ty;
    if (p.walk) { ty = wpos % p.tilesY; const int t = wpos / p.tilesY; tx = t % p.tilesX; it.img = t / p.tilesX; }
    else { tx = wpos % p.tilesX; const int t = wpos / p.tilesX; ty = t % p.tilesY; it.img = t / p.tilesY; }
    it.tile_id = (it.img * p.tilesY + ty) * p.tilesX + tx;      // BatchNorm-partial index: row-major whatever the walk
    it.oy0 = ty * TH; it.ox0 = tx * TW;
    return it;
  };

  const int Cin8 = p.Cin >> 3;
  const float w_scale = p.wp[(long)TAPS * Cin8 * 2 * p.Ncols * 4];     // trailer of the packed weights
  const bool bnin = p.inSc != nullptr;
  const int Cinp = (p.Cin + 3) & ~3;
  if (bnin)
    for (int i = tid; i < p.Cin; i += THREADS) { lds_sc[i] = p.inSc[i]; lds_sc[Cinp + i] = p.inSh[i]; }
  // per-column epilogue parameters live in LDS for the whole launch: a global load inside an epilogue slice would make the
  // slice wait (vmcnt counts stores too) for the previous slice's stores to drain
  if constexpr (PERWG)
    for (int i = tid; i < 2 * EP_COLS; i += THREADS) PP_RUN_M[i] = DcMoments{0.f, 0.f, 0.f};
  for (int i = tid; i < p.Ncols; i += THREADS) {
    lds_ep[i] = p.bias ? p.bias[i] : 0.f;
    lds_ep[EP_COLS + i] = p.scale ? p.scale[i] : 1.f;
    lds_ep[2 * EP_COLS + i] = p.shift ? p.shift[i] : 0.f;
    if constexpr (BNRED) { // BatchNorm-backward sums of the layer this launch writes `da` of: its gate affine, mean, 1/std
      float gsc, gsh;      // (a data-gradient launch has no epilogue scale / shift and no BN-on-load table: their slots)
      dc_bn_affine(p.bnMean[i], p.bnInvstd[i], p.bnGamma[i], p.bnBeta[i], gsc, gsh);
      lds_ep[EP_COLS + i] = gsc;
      lds_ep[2 * EP_COLS + i] = gsh;
      lds_sc[i] = p.bnMean[i];
      lds_sc[EP_COLS + i] = p.bnInvstd[i];
    }
  }
  float in_scale;
  if constexpr (DZIN) {
    for (int i = tid; i < 6 * p.Cin; i += THREADS) lds_dz[i] = p.dzCoef[i];
    in_scale = dc_block_guard_scale(p.dzCoef + 6 * p.Cin, p.Cin, tmp);       // (its barriers publish the tables)
  } else {
    in_scale = (p.inAbsmax ? dc_block_absmax_scale(p.inAbsmax, p.inAbsmaxN, 1024.f, tmp) : (p.inScale ? *p.inScale : 1.f)) * dc_block_guard_scale(p.inAbound, p.Cin, tmp, p.inAboundLd);
    if (p.inAbound == nullptr) __syncthreads();               // (the guard's own barriers publish the tables otherwise)
  }

  if (role == 2) {
    // =========================== producers: HBM -> registers -> fp16 hi/lo operand images ==========================
    const int t = tid & 255;
    const __amdgpu_buffer_rsrc_t rsrcB = dc_make_rsrc(p.wp, (unsigned)(TAPS * Cin8 * 2 * p.Ncols) * 16u);
    constexpr int A_STEP = 256 / G4, B_STEP = 256 / BN;
    const int a_g = t % G4, a_pix0 = t / G4;
    const int b_j = t % BN, b_row0 = t / BN;
    // tile-independent parts of the byte offsets: kept in registers, except in the DZIN instantiations (their second
    // input tensor needs the registers: recomputed per item there, ~6 VALU per offset)
    auto a_rel_of = [&](int it, int pix0) __attribute__((always_inline)) {
      const int pix = pix0 + it * A_STEP;
      const int r = pix / TWI, c = pix - r * TWI;
      return ((r * p.Win + c) * p.Cin + 4 * a_g) * 4;
    };
    auto b_rel_of = [&](int it, int row0) __attribute__((always_inline)) {
      const int row = row0 + it * B_STEP;    // (tap, g8, hl)
      const int tap = row / (2 * G8), g8 = (row >> 1) % G8, hl = row & 1;
      return row < BROWS ? (((tap * Cin8 + g8) * 2 + hl) * p.Ncols + b_j) * 16 : -1;
    };
    int a_rel[DZIN ? 1 : NA], b_rel[DZIN ? 1 : NBV];
    if constexpr (!DZIN) {
#pragma unroll
      for (int it = 0; it < NA; ++it) a_rel[it] = a_rel_of(it, a_pix0);
#pragma unroll
      for (int it = 0; it < NBV; ++it) b_rel[it] = b_rel_of(it, b_row0);
    }
    unsigned a_voff[NA], b_voff[NBV];
    __amdgpu_buffer_rsrc_t rsrcA = rsrcB, rsrcZ = rsrcB, rsrcO = rsrcB;
    // DZIN + dzOut: bit k = staging slot k of this thread is an INTERIOR pixel of the halo'd patch (every image pixel is interior
    // to exactly one tile); store_dz = the item staged next holds the tile's first column block (wave-uniform)
    // (64-column instantiations only: the <4,1> ones sit at the 168-VGPR budget of a 768-thread workgroup)
    constexpr bool DZOUT = DZIN && MB == 2;
    unsigned int_mask = 0;
    bool store_dz = false;
    if constexpr (DZOUT) {
#pragma unroll
      for (int k = 0; k < NA; ++k) {
        const int pix = a_pix0 + k * A_STEP, r = pix / TWI, c = pix - r * TWI;
        if (pix < NPIXH && r >= 1 && r <= THI - 2 && c >= 1 && c <= TWI - 2) int_mask |= 1u << k;
      }
    }
    auto setup_item = [&](int j) __attribute__((always_inline)) {
      const Item it = decode(j);
      rsrcA = dc_make_rsrc(p.in + (long)it.img * p.Hin * p.Win * p.Cin, (unsigned)(p.Hin * p.Win * p.Cin) * 4u);
      if constexpr (DZIN) {
        rsrcZ = dc_make_rsrc(p.in2 + (long)it.img * p.Hin * p.Win * p.Cin, (unsigned)(p.Hin * p.Win * p.Cin) * 4u);
        if constexpr (DZOUT) {
          store_dz = p.dzOut != nullptr && it.n0 == 0;
          if (p.dzOut) rsrcO = dc_make_rsrc(p.dzOut + (long)it.img * p.Hin * p.Win * p.Cin, (unsigned)(p.Hin * p.Win * p.Cin) * 4u);
        }
      }
      const int iy0 = it.oy0 - 1, ix0 = it.ox0 - 1;
      const int base = (iy0 * p.Win + ix0) * p.Cin * 4;
      const bool inside = iy0 >= 0 && ix0 >= 0 && iy0 + THI <= p.Hin && ix0 + TWI <= p.Win;     // wave-uniform
      int pix0 = a_pix0, row0 = b_row0;
      if constexpr (DZIN) {      // re-derived from the thread id, opaque per item: or the offsets are hoisted out of the loop and spilled
        int t2 = t;
        asm volatile("" : "+v"(t2));
        pix0 = t2 / G4; row0 = t2 / BN;
      }
#pragma unroll
      for (int k = 0; k < NA; ++k) {
        const int pix = pix0 + k * A_STEP;
        bool ok = pix < NPIXH;
        if (!inside) {
          const int r = pix / TWI, c = pix - r * TWI;
          const int y = iy0 + r, x = ix0 + c;
          ok = ok && y >= 0 && y < p.Hin && x >= 0 && x < p.Win;
        }
        a_voff[k] = ok ? (unsigned)(base + (DZIN ? a_rel_of(k, pix0) : a_rel[DZIN ? 0 : k])) : OOB;
      }
      const bool col_ok = (it.n0 + b_j) < p.Ncols;
#pragma unroll
      for (int k = 0; k < NBV; ++k) {
        const int br = DZIN ? b_rel_of(k, row0) : b_rel[DZIN ? 0 : k];
        b_voff[k] = (br >= 0 && col_ok) ? (unsigned)(br + it.n0 * 16) : OOB;
      }
    };
    f32x4 ra[NA], rz[DZIN ? NA : 1];
    u32x4 rb[NBV];
    auto load_chunk = [&](int c0) __attribute__((always_inline)) {
      const unsigned a_add = (unsigned)c0 * 4u, b_add = (unsigned)(c0 >> 3) * 2u * (unsigned)p.Ncols * 16u;
#pragma unroll
      for (int k = 0; k < NA; ++k)
        ra[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrcA, a_voff[k], (int)a_add, 0));
      if constexpr (DZIN) {
#pragma unroll
        for (int k = 0; k < NA; ++k)
          rz[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrcZ, a_voff[k], (int)a_add, 0));
      }
#pragma unroll
      for (int k = 0; k < NBV; ++k) rb[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrcB, b_voff[k], (int)b_add, 0);
    };
    auto stage = [&](char* st, int c0) __attribute__((always_inline)) {
      u32x4* ldsB = reinterpret_cast<u32x4*>(st) + A_SLOTS;
      f32x4 csc = {1.f, 1.f, 1.f, 1.f}, csh = {0.f, 0.f, 0.f, 0.f};
      if (bnin) {
        csc = *reinterpret_cast<const f32x4*>(lds_sc + c0 + 4 * a_g);
        csh = *reinterpret_cast<const f32x4*>(lds_sc + Cinp + c0 + 4 * a_g);
      }
      f32x4 cmu = csh, cA = csc, cD = csh, cE = csh;
      if constexpr (DZIN) {
        const float* t = lds_dz + c0 + 4 * a_g;
        csc = *reinterpret_cast<const f32x4*>(t);
        csh = *reinterpret_cast<const f32x4*>(t + p.Cin);
        cmu = *reinterpret_cast<const f32x4*>(t + 2 * p.Cin);
        cA = *reinterpret_cast<const f32x4*>(t + 3 * p.Cin);
        cD = *reinterpret_cast<const f32x4*>(t + 4 * p.Cin);
        cE = *reinterpret_cast<const f32x4*>(t + 5 * p.Cin);
      }
#pragma unroll
      for (int k = 0; k < NA; ++k) {
        const int pix = a_pix0 + k * A_STEP;
        if (pix < NPIXH) {
          f32x4 v = ra[k];
          if constexpr (DZIN) {
            const bool live = !(a_voff[k] >> 31);              // outside the image dz is 0, not E - D*mu
            const f32x4 zz = rz[k];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float y = __builtin_fmaf(zz[e], csc[e], csh[e]);       // the forward's own expression: identical ReLU gate
              const float dy = y > 0.f ? v[e] : 0.f;
              const float dzv = __builtin_fmaf(cA[e], dy, __builtin_fmaf(cD[e], zz[e] - cmu[e], cE[e]));
              v[e] = live ? dzv : 0.f;
            }
            // the weight gradient's copy of dz (a store at an out-of-image offset is dropped by the descriptor)
            if constexpr (DZOUT)
              if (store_dz && ((int_mask >> k) & 1u))
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrcO, a_voff[k], c0 * 4, 0);
          }
          if (bnin) {
            const bool live = !(a_voff[k] >> 31);              // zero padding stays zero
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float y = fmaxf(__builtin_fmaf(v[e], csc[e], csh[e]), 0.f);
              v[e] = live ? y : 0.f;
            }
          }
          u32x2 hi, lo;
          if (DC_PP_ABL & 32) {      // ablation: what an operand stored PRE-SPLIT in HBM would leave -- the same LDS writes, no split VALU
            hi = u32x2{__builtin_bit_cast(unsigned, v[0]), __builtin_bit_cast(unsigned, v[1])};
            lo = u32x2{__builtin_bit_cast(unsigned, v[2]), __builtin_bit_cast(unsigned, v[3])};
          } else {
            split(v, in_scale, hi, lo);
          }
          char* base = st + ((a_g >> 1) * PS + pix) * 16 + (a_g & 1) * 8;
          *reinterpret_cast<u32x2*>(base) = hi;
          *reinterpret_cast<u32x2*>(base + G8 * PS * 16) = lo;
        }
      }
#pragma unroll
      for (int k = 0; k < NBV; ++k) {
        const int row = b_row0 + k * B_STEP;
        if (row < BROWS) ldsB[row * BN + b_j] = rb[k];
      }
    };
    // (item, chunk) of the next load
    int jl = 0, cl = 0;
    auto advance = [&]() __attribute__((always_inline)) { if (++cl == nch) { cl = 0; ++jl; } };
    if (K > 0) {
      setup_item(0);
      load_chunk(0);
      advance();
      stage(smem, 0);                                           // step 0 -> stage 0
      if (K > 1) {
        if (cl == 0) setup_item(jl);
        load_chunk(cl * CK);
      }
    }
    PP_TRACE();
    __syncthreads();                                            // barrier 0: stage 0 is ready
    PP_TRACE();
    int cs = cl;                                                // chunk of the step held in the registers (step k+1)
    for (int k = 0; k < K; ++k) {
      if (k + 1 < K) {
        if (!(DC_PP_ABL & 2)) stage(smem + ((k + 1) & 1) * STAGE_BYTES, cs * CK);     // uses the a_voff of that step's item (padding mask)
        advance();
        if (k + 2 < K) {
          if (cl == 0) setup_item(jl);
          if (!(DC_PP_ABL & 8)) load_chunk(cl * CK);
        }
        cs = cl;
      }
      PP_TRACE();
      __syncthreads();
      PP_TRACE();
    }
#pragma unroll 1
    for (int c = 0; c < 5; ++c) __syncthreads();                // the consumers' tail: 4 epilogue slices + the merge
    return;
  }

  // ================================ consumers: LDS fragments -> MFMA, sliced epilogue ================================
  const int li = lane & 31, h = lane >> 5;
  const int wave_m = wr;
  DcMoments* red = red_all + role * RED_ENTRIES;
  int a_base[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int mblk = wave_m * MB + mb;
    const int row = mblk * RPM + li / TW, col = li % TW;
    a_base[mb] = h * PS + row * TWI + col;                      // g8 = 2*ks + h for k-step ks
  }
  const int b_base = (h * 2) * BN + li;

  f32x16 acc[MB][NB];
  const float out_scale = 1.f / (in_scale * w_scale);
  const int ld = (int)p.outLd;
  const int sy = p.Wout * ld, sx = ld;
  Item cur = {0, 0, 0, 0, 0}, pend = {0, 0, 0, 0, 0};
  bool pending = false;

  // DC_PP_PF: the first four fragments of the NEXT step (tap 0, block (0, 0)), requested under this step's last tap
  f16x8 pf_ah, pf_al, pf_bh, pf_bl;
  bool have_pf = false;
  auto mfma_block = [&](const char* st, const char* st_next, bool want_pf) __attribute__((always_inline)) {
    const u32x4* ldsA = reinterpret_cast<const u32x4*>(st);
    const u32x4* ldsB = ldsA + A_SLOTS;
    f16x8 ah[2][MB], al[2][MB], bh[2][NB], bl[2][NB];
    auto fetch = [&](int tap, int buf) __attribute__((always_inline)) {
      const int toff = (tap / KW) * TWI + (tap % KW);
      const int boff = (tap * G8) * 2 * BN;
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
    if (PF && have_pf) {              // block (0, 0) of tap 0 is in registers; the rest of the tap's fragments are requested now
      ah[0][0] = pf_ah; al[0][0] = pf_al; bh[0][0] = pf_bh; bl[0][0] = pf_bl;
#pragma unroll
      for (int mb = 1; mb < MB; ++mb) {
        al[0][mb] = __builtin_bit_cast(f16x8, ldsA[a_base[mb] + G8 * PS]);
        ah[0][mb] = __builtin_bit_cast(f16x8, ldsA[a_base[mb]]);
      }
#pragma unroll
      for (int nb = 1; nb < NB; ++nb) {
        bh[0][nb] = __builtin_bit_cast(f16x8, ldsB[b_base + nb * 32]);
        bl[0][nb] = __builtin_bit_cast(f16x8, ldsB[b_base + BN + nb * 32]);
      }
    } else {
      fetch(0, 0);
    }
    if (DC_PP_ABL & 4) fetch(0, 1);
    if (DC_PP_ABL & 16) { __builtin_amdgcn_s_waitcnt(0xc07f); PP_TRACE(); }    // detail stamp 2: first fragments landed (lgkmcnt 0)
    // Hand-ordered schedule (every statement pinned by sched_barrier): the 8 fragment reads of tap t+1 go one per gap
    // between the first 8 MFMAs of tap t, in the order tap t+1 will use them -- a burst of 8 b128 reads in front of the
    // group kept the wave from issuing MFMAs for ~50 cycles per tap and the pipe ran dry.
#define PP_SB() __builtin_amdgcn_sched_barrier(0)
#define PP_MFMA(mb, nb, A, B) acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[b][mb], B[b][nb], acc[mb][nb], 0, 0, 0); PP_SB()
#define PP_RA(X, mb, off) if (more && !(DC_PP_ABL & 4)) { X[nx][mb] = __builtin_bit_cast(f16x8, ldsA[a_base[mb] + toff + (off)]); PP_SB(); }
#define PP_RB(X, nb, off) if (more && !(DC_PP_ABL & 4)) { X[nx][nb] = __builtin_bit_cast(f16x8, ldsB[b_base + boff + (off)]); PP_SB(); }
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
      const int b = tap & 1, nx = b ^ 1;
      const bool more = tap + 1 < TAPS;
      const int toff = ((tap + 1) / KW) * TWI + ((tap + 1) % KW);
      const int boff = ((tap + 1) * G8) * 2 * BN;
      PP_SB();
      if (PF && tap == TAPS - 1) {
        // every fragment of this stage has been requested (tap 8's during tap 7): once they have landed the stage may be overwritten,
        // and the producers have completed the other one -- the step's barrier, one tap early
        __syncthreads();
        PP_SB();
        if (want_pf) {
          const u32x4* nA = reinterpret_cast<const u32x4*>(st_next);
          const u32x4* nB = nA + A_SLOTS;
          pf_al = __builtin_bit_cast(f16x8, nA[a_base[0] + G8 * PS]);
          pf_bh = __builtin_bit_cast(f16x8, nB[b_base]);
          pf_ah = __builtin_bit_cast(f16x8, nA[a_base[0]]);
          pf_bl = __builtin_bit_cast(f16x8, nB[b_base + BN]);
          PP_SB();
        }
      }
      if constexpr (MB == 2) {          // 2 x 2 blocks: 8 reads in the first 8 gaps
        PP_MFMA(0, 0, al, bh); PP_RA(al, 0, G8 * PS);
        PP_MFMA(0, 0, ah, bl); PP_RB(bh, 0, 0);
        PP_MFMA(0, 0, ah, bh); PP_RA(ah, 0, 0);
        PP_MFMA(0, 1, al, bh); PP_RB(bl, 0, BN);
        PP_MFMA(0, 1, ah, bl); PP_RB(bh, 1, 32);
        PP_MFMA(0, 1, ah, bh); PP_RB(bl, 1, BN + 32);
        PP_MFMA(1, 0, al, bh); PP_RA(al, 1, G8 * PS);
        PP_MFMA(1, 0, ah, bl); PP_RA(ah, 1, 0);
        PP_MFMA(1, 0, ah, bh);
        PP_MFMA(1, 1, al, bh);
        PP_MFMA(1, 1, ah, bl);
        PP_MFMA(1, 1, ah, bh);
      } else {                          // 4 x 1 blocks: 10 reads in the first 10 gaps
        PP_MFMA(0, 0, al, bh); PP_RA(al, 0, G8 * PS);
        PP_MFMA(0, 0, ah, bl); PP_RB(bh, 0, 0);
        PP_MFMA(0, 0, ah, bh); PP_RA(ah, 0, 0);
        PP_MFMA(1, 0, al, bh); PP_RB(bl, 0, BN);
        PP_MFMA(1, 0, ah, bl); PP_RA(al, 1, G8 * PS);
        PP_MFMA(1, 0, ah, bh); PP_RA(ah, 1, 0);
        PP_MFMA(2, 0, al, bh); PP_RA(al, 2, G8 * PS);
        PP_MFMA(2, 0, ah, bl); PP_RA(ah, 2, 0);
        PP_MFMA(2, 0, ah, bh); PP_RA(al, 3, G8 * PS);
        PP_MFMA(3, 0, al, bh); PP_RA(ah, 3, 0);
        PP_MFMA(3, 0, ah, bl);
        PP_MFMA(3, 0, ah, bh);
      }
    }
#undef PP_RA
#undef PP_RB
#undef PP_MFMA
#undef PP_SB
    have_pf = PF && want_pf;
  };

  // One 32x32 accumulator block (mb, nb) of the pending tile per slice: bias, BatchNorm partials (mode 1), inference flag
  // (mode 2), stores.  Slices run in the order (nb 0: mb 0, 1), (nb 1: mb 0, 1); the per-column shifted sums of a column
  // block live in four registers across its two slices and go to LDS after the second; the cross-wave merge of the tile's
  // partials happens one step later (merge_pending), whatever that step is for this set.
  const int mode = BNRED ? 4 : (PERWG ? 1 : (p.outAbsmax ? 2 : ((p.stats && !POOL) ? 1 : 0)));          // wave-uniform
  float e_s1 = 0.f, e_s2 = 0.f, e_cnt = 0.f, e_K = 0.f;
  auto epi_values = [&](auto nb_tag, auto mb_tag, auto interior_tag, auto mode_tag) __attribute__((always_inline)) {
    constexpr int nb = decltype(nb_tag)::value;
    constexpr int mb = decltype(mb_tag)::value;
    constexpr bool INT = decltype(interior_tag)::value;
    constexpr int MODE = decltype(mode_tag)::value;                 // border tiles (!INT): MODE == 3, resolved at run time
    const bool m_stats = !BNRED && !POOL && (MODE == 1 || (MODE == 3 && mode == 1)), m_track = !BNRED && (MODE == 2 || (MODE == 3 && mode == 2));
    constexpr bool m_bnred = BNRED && (MODE == 4 || MODE == 3);
    const long out_img_floats = (long)p.Hout * p.Wout * p.outLd;
    const __amdgpu_buffer_rsrc_t rsrcO = dc_make_rsrc(p.out + (long)pend.img * out_img_floats, (unsigned)(out_img_floats * 4));
    const int n = pend.n0 + nb * 32 + li;
    const bool n_ok = INT || n < p.Ncols;
    const int nl = n_ok ? n : 0;
    const float bv = lds_ep[nl], sc = lds_ep[EP_COLS + nl], sh = lds_ep[2 * EP_COLS + nl];
    if constexpr (mb == 0) {
      e_K = __builtin_fmaf(acc[0][nb][0], out_scale, bv);      // shift of this lane's sums: its first value
      e_s1 = 0.f; e_s2 = 0.f; e_cnt = 0.f;
    }
    float amax = 0.f;
    const int mblk = wave_m * MB + mb;
    const int oyb = pend.oy0 + mblk * RPM, oxb = pend.ox0 + 4 * h;
    const unsigned base = (unsigned)((oyb * sy + oxb * sx + n) * 4);
    // BatchNorm-backward sums: the z values of this block are requested first, the stores of da go out while they fly
    float zr[16];
    if (m_bnred) {
      const __amdgpu_buffer_rsrc_t rsrcZ = dc_make_rsrc(p.bnZ + (long)pend.img * out_img_floats, (unsigned)(out_img_floats * 4));
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int mr = (r & 3) + 8 * (r >> 2);
        const int rowc = mr / TW, colc = mr % TW;
        const bool ok = INT || (n_ok && (oyb + rowc) < p.Hout && (oxb + colc) < p.Wout);
        const unsigned off = ok ? base + (unsigned)((rowc * sy + colc * sx) * 4) : OOB;
        zr[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrcZ, off, 0, 0));
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int mr = (r & 3) + 8 * (r >> 2);
      const int rowc = mr / TW, colc = mr % TW;
      float v = __builtin_fmaf(acc[mb][nb][r], out_scale, bv);
      const float d = v - e_K;
      if constexpr (INT) {
        if constexpr (MODE == 1) { e_s1 += d; e_s2 = __builtin_fmaf(d, d, e_s2); }
        if (p.scale) v = v * sc + sh;
        if (p.relu) v = fmaxf(v, 0.f);
        if constexpr (MODE == 2) amax = fmaxf(amax, fabsf(v));
        if (DC_PP_ABL & 64) {     // ablation (garbage results): no output stores -- one conditional store keeps the values alive
          if (v == 12345.678f) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrcO, base, 0, 0);
        } else {
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrcO, base, (rowc * sy + colc * sx) * 4, 0);
        }
      } else {
        const bool ok = n_ok && (oyb + rowc) < p.Hout && (oxb + colc) < p.Wout;
        if (m_stats) { e_s1 += ok ? d : 0.f; e_s2 += ok ? d * d : 0.f; e_cnt += ok ? 1.f : 0.f; }
        if (p.scale) v = v * sc + sh;
        if (p.relu) v = fmaxf(v, 0.f);
        if (m_track) amax = fmaxf(amax, ok ? fabsf(v) : 0.f);
        const unsigned off = ok ? base + (unsigned)((rowc * sy + colc * sx) * 4) : OOB;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrcO, off, 0, 0);
      }
    }
    if constexpr (POOL && (mb & 1)) {
      // 2x2 max-pool of the row pair (mb - 1, mb) of this wave: both blocks' accumulators are still here, registers
      // (r, r + 1) of a lane are x-neighbours; the activations are re-formed with the store loop's own expressions
      const long pool_img_floats = (long)(p.Hout >> 1) * (p.Wout >> 1) * p.Ncols;
      const __amdgpu_buffer_rsrc_t rsrcP = dc_make_rsrc(p.poolOut + (long)pend.img * pool_img_floats, (unsigned)(pool_img_floats * 4));
      const int py = (oyb - RPM) >> 1;                                  // RPM == 1: block rows mb - 1 and mb are y, y + 1
      auto act = [&](float a) __attribute__((always_inline)) {
        float v = __builtin_fmaf(a, out_scale, bv);
        if (p.scale) v = v * sc + sh;
        if (p.relu) v = fmaxf(v, 0.f);
        return v;
      };
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const int mr = (r & 3) + 8 * (r >> 2);
        const int colc = mr % TW;
        const float m0 = fmaxf(act(acc[mb - 1][nb][r]), act(acc[mb - 1][nb][r + 1]));
        const float m1 = fmaxf(act(acc[mb][nb][r]), act(acc[mb][nb][r + 1]));
        const bool ok = INT || (n_ok && oyb < p.Hout && (oxb + colc + 1) < p.Wout);
        const unsigned off = ok ? (unsigned)(((py * (p.Wout >> 1) + ((oxb + colc) >> 1)) * p.Ncols + n) * 4) : OOB;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, fmaxf(m0, m1)), rsrcP, off, 0, 0);
      }
    }
    if (m_bnred) {
      const float mu = lds_sc[nl], is = lds_sc[EP_COLS + nl];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int mr = (r & 3) + 8 * (r >> 2);
        const int rowc = mr / TW, colc = mr % TW;
        const bool ok = INT || (n_ok && (oyb + rowc) < p.Hout && (oxb + colc) < p.Wout);
        const float v = __builtin_fmaf(acc[mb][nb][r], out_scale, bv);
        const float y = __builtin_fmaf(zr[r], sc, sh);          // the forward's own expression: identical ReLU gate
        const float dy = (ok && y > 0.f) ? v : 0.f;
        e_s1 += dy;
        e_s2 = __builtin_fmaf(dy, (zr[r] - mu) * is, e_s2);
        e_cnt = fmaxf(e_cnt, fabsf(dy));                        // (the count slot is free in this variant: max |dy|)
      }
      if (mb == MB - 1) {
        DcMoments m;                                             // container: (max |dy|, sum dy, sum dy*xhat)
        m.n = fmaxf(e_cnt, __shfl_xor(e_cnt, 32));
        m.mean = e_s1 + __shfl_xor(e_s1, 32);
        m.m2 = e_s2 + __shfl_xor(e_s2, 32);
        if (h == 0) red[(wave_m * NB + nb) * 32 + li] = m;
      }
    }
    if (m_stats && mb == MB - 1) {
      DcMoments m;
      if constexpr (INT) {
        constexpr float NL = (float)(16 * MB);
        const float ms = e_s1 * (1.f / NL);
        m.n = NL; m.mean = e_K + ms; m.m2 = fmaxf(__builtin_fmaf(-e_s1, ms, e_s2), 0.f);
        const float om = __shfl_xor(m.mean, 32), o2 = __shfl_xor(m.m2, 32), dd = om - m.mean;
        m.m2 = m.m2 + o2 + dd * dd * (0.5f * NL);
        m.mean = __builtin_fmaf(dd, 0.5f, m.mean);
        m.n = 2.f * NL;
      } else {
        m = dc_moments_from_shifted(e_cnt, e_K, e_s1, e_s2);
        DcMoments o;
        o.n = __shfl_xor(m.n, 32); o.mean = __shfl_xor(m.mean, 32); o.m2 = __shfl_xor(m.m2, 32);
        m = dc_moments_merge(m, o);
      }
      if (h == 0) red[(wave_m * NB + nb) * 32 + li] = m;
    }
    if (m_track) {
      if (!(amax <= DC_F16_SAFE_MAX)) p.outAbsmax[0] = 1.f;
    }
  };
  bool merge_pending = false;
  Item mitem = {0, 0, 0, 0, 0};
  auto epi_merge = [&]() __attribute__((always_inline)) {      // the red[] entries of both column blocks are complete
    if (!BNRED && !POOL && mode == 1) {
      const int ts = tid & 255;                                 // thread within the consumer set
      if (ts < NB * 32) {
        const int nb = ts / 32, l = ts % 32;
        DcMoments m = red[nb * 32 + l];
#pragma unroll
        for (int wm = 1; wm < WAVES_M; ++wm) m = dc_moments_merge(m, red[(wm * NB + nb) * 32 + l]);
        const int n = mitem.n0 + nb * 32 + l;
        if (n < p.Ncols) {
          if constexpr (PERWG) PP_RUN_M[role * EP_COLS + n] = dc_moments_merge(PP_RUN_M[role * EP_COLS + n], m);   // tile order: deterministic
          else dc_moments_store(p.stats + ((long)mitem.tile_id * p.Ncols + n) * 2, m);
        }
      }
    } else if (BNRED) {
      const int ts = tid & 255;
      if (ts < NB * 32) {
        const int nb = ts / 32, l = ts % 32;
        float s1 = 0.f, s2 = 0.f, am = 0.f;
#pragma unroll
        for (int wm = 0; wm < WAVES_M; ++wm) { const DcMoments m = red[(wm * NB + nb) * 32 + l]; s1 += m.mean; s2 += m.m2; am = fmaxf(am, m.n); }
        const int n = mitem.n0 + nb * 32 + l;
        if (n < p.Ncols) {
          float* dst = p.bnPartial + ((long)mitem.tile_id * p.Ncols + n) * 2;
          dst[0] = s1; dst[1] = s2;
          if (p.bnAmax) p.bnAmax[(long)mitem.tile_id * p.Ncols + n] = am;
        }
      }
    }
    merge_pending = false;
  };
  auto epi_block = [&](int blk) __attribute__((always_inline)) {
    using T0 = std::integral_constant<int, 0>;
    using T1 = std::integral_constant<int, 1>;
    using T2 = std::integral_constant<int, 2>;
    using T3 = std::integral_constant<int, 3>;
    using T4 = std::integral_constant<int, 4>;
    const bool interior = (pend.oy0 + TH <= p.Hout) && (pend.ox0 + TW <= p.Wout) && (pend.n0 + BN <= p.Ncols);
    auto run = [&](auto nb_tag, auto mb_tag) __attribute__((always_inline)) {
      if constexpr (BNRED) {
        if (interior) epi_values(nb_tag, mb_tag, std::true_type{}, T4{});
        else epi_values(nb_tag, mb_tag, std::false_type{}, T3{});
      } else if constexpr (PERWG) {
        if (interior) epi_values(nb_tag, mb_tag, std::true_type{}, T1{});
        else epi_values(nb_tag, mb_tag, std::false_type{}, T3{});
      } else if (interior) {
        if (mode == 0) epi_values(nb_tag, mb_tag, std::true_type{}, T0{});
        else if (!POOL && mode == 1) epi_values(nb_tag, mb_tag, std::true_type{}, T1{});
        else epi_values(nb_tag, mb_tag, std::true_type{}, T2{});
      } else {
        epi_values(nb_tag, mb_tag, std::false_type{}, T3{});
      }
    };
    // block order: column block major (the shifted sums of a column block live across its MB blocks)
    if constexpr (MB == 2) {
      if (blk == 0) run(T0{}, T0{});
      else if (blk == 1) run(T0{}, T1{});
      else if (blk == 2) run(T1{}, T0{});
      else run(T1{}, T1{});
    } else {
      if (blk == 0) run(T0{}, T0{});
      else if (blk == 1) run(T0{}, T1{});
      else if (blk == 2) run(T0{}, T2{});
      else run(T0{}, T3{});
    }
    if (blk == NBLK - 1) {
      pending = false;
      merge_pending = true;
      mitem = pend;
    }
  };
  // the NBLK blocks are spread over the partner's steps: one per step for >= 4 steps per tile, two for 2 or 3
  const int bps = nch >= NBLK ? 1 : (NBLK + nch - 1) / nch;
  auto epi_slice = [&](int c) __attribute__((always_inline)) {
    const int b0 = c * bps;
    if (b0 < NBLK) epi_block(b0);
    if (bps > 1 && b0 + 1 < NBLK) epi_block(b0 + 1);
  };

  PP_TRACE();
  __syncthreads();                                              // barrier 0
  PP_TRACE();
  PP_TL(1);
  int j = 0, c = 0;
  for (int k = 0; k < K + 5; ++k) {                             // + 5: the last tile's 4 epilogue slices and its merge
    if (k == K) PP_TL(2);
    bool did_barrier = false;                                   // DC_PP_PF: an MFMA step takes the step's barrier inside mfma_block
    if (merge_pending) epi_merge();                             // partials written one step ago (a barrier in between)
    if (k < K && (j & 1) == role) {
      if (c == 0) {
        cur = decode(j);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;
      }
      if (DC_PP_ABL & 16) PP_TRACE();                           // detail stamp 1: tile setup done
      __builtin_amdgcn_s_setprio(3);                            // the wave that feeds the matrix pipe goes first
      mfma_block(smem + (k & 1) * STAGE_BYTES, smem + ((k + 1) & 1) * STAGE_BYTES, (c + 1 < nch) && (k + 1 < K));
      __builtin_amdgcn_s_setprio(0);
      did_barrier = PF;
      if (c == nch - 1) { pend = cur; pending = true; }
    } else {
      if (PF) {       // never live across an epilogue step (the next MFMA step of this set starts a tile): tell the register allocator
        const f16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
        pf_ah = z8; pf_al = z8; pf_bh = z8; pf_bl = z8;
        have_pf = false;
      }
      if (DC_PP_ABL & 16) { PP_TRACE(); PP_TRACE(); }
      if (pending && !(DC_PP_ABL & 1)) epi_slice(c);
    }
    if (++c == nch) { c = 0; ++j; }
    PP_TRACE();
    if (!did_barrier) __syncthreads();
    PP_TRACE();
  }
  if constexpr (PERWG)
    if (role < 2)
      for (int n = tid & 255; n < p.Ncols; n += 256)
        dc_moments_store(p.stats + ((long)(2 * blockIdx.x + role) * p.Ncols + n) * 2, PP_RUN_M[role * EP_COLS + n]);
#ifdef DC_PP_TIMELINE
  __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0)
  PP_TL(3);
#endif
#undef PP_RUN_M
}

// ---------------------------------------------------------------------------------------------------------------------
// Host side: the shape test and the launch (called from igemm_f16x3.hip's conv3x3 dispatch).
bool dc_igemm_pp_serves(const IgemmParams& p) {
  // DC_IGEMM_PP: 0 off, 1 (default) every served launch, 2 forward launches only (a data-gradient launch runs beside the
  // weight-gradient kernel of the side stream, and this kernel's 138 KB of LDS cannot share a CU with that one's 87 KB:
  // measured, serving the data gradients too is still the faster setting: 788 vs 780 vs 774 images/s for 1 / 2 / 0)
  const int knob = dc_config().igemm_pp;
  const bool is_dgrad = p.inScale != nullptr || p.inAbsmax != nullptr || p.dzCoef != nullptr;
  const bool enabled = knob == 1 || (knob == 2 && !is_dgrad);
  if (p.dzCoef && p.Cin > pp::DZ_CIN) return false;
  const int th = p.Ncols <= 32 ? 16 : 8, bn = p.Ncols <= 32 ? 32 : 64;
  const long total = (long)p.N * dc_cdiv(p.Wout, pp::TW) * dc_cdiv(p.Hout, th) * dc_cdiv(p.Ncols, bn);
  // 2- and 3-step tiles (Cin 32 / 48) take two epilogue blocks per step: with BatchNorm partials that slice is longer than
  // the partner's MFMA step and the 256-thread kernel stays ahead (309 vs 292 us on 512^2 x 32 -> 32); without them it wins
  if (p.Cin < 4 * pp::CK && p.stats) return false;
  return enabled && p.Wout > 16 && p.Cin % pp::CK == 0 && p.Cin >= 2 * pp::CK && p.scatterCo == 0 &&
         !(p.outAbsmax && p.outAbsmaxLd >= 0) && p.Cin <= 1024 && p.Ncols <= pp::EP_COLS && total >= 8;
}

template <int MB_, int NB_, int VARIANT, bool DZIN = false>
static int pp_launch(IgemmParams p, hipStream_t st, const char* name) {
  using C = pp::Cfg<MB_, NB_>;
  static_assert(C::RPM == 1, "the pooled epilogue pairs block rows mb - 1, mb");
  static_assert(!(DZIN && VARIANT == 3), "the per-workgroup running moments use the dz-on-load table's LDS");
  auto kern = igemm_pp_kernel<MB_, NB_, VARIANT, DZIN>;
  static DcLdsAttr lds_attr;
  static_assert(2 * pp::EP_COLS * (int)sizeof(DcMoments) <= pp::DZ_BYTES, "the per-workgroup running moments live in the dz-on-load table's LDS");
  const int lds = C::FIXED_LDS + pp::TABLE_BYTES + 3 * pp::EP_COLS * 4 + ((DZIN || VARIANT == 3) ? pp::DZ_BYTES : 0);
  static_assert(C::FIXED_LDS + pp::TABLE_BYTES + 3 * pp::EP_COLS * 4 + pp::DZ_BYTES <= 160 * 1024, "LDS budget");
  // (the attribute is set once per device: with the most this instantiation ever asks for)
  constexpr int lds_max = C::FIXED_LDS + pp::TABLE_BYTES + 3 * pp::EP_COLS * 4 + pp::DZ_BYTES;
  if (int rc = dc_func_max_lds(lds_attr, reinterpret_cast<const void*>(kern), lds_max, name)) return rc;
  p.tilesX = dc_cdiv(p.Wout, pp::TW);
  p.tilesY = dc_cdiv(p.Hout, C::TH);
  p.walk = dc_tile_walk();
  static int cus[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  if (dev < 0 || dev >= 64) dev = 0;
  if (cus[dev] == 0) {
    int n = 0;
    hipError_t e = hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    DC_REQUIRE(e == hipSuccess && n > 0, DC_EHIP, "%s: hipDeviceGetAttribute: %s", name, hipGetErrorString(e));
    cus[dev] = n;
  }
  const int total = p.N * p.tilesX * p.tilesY * dc_cdiv(p.Ncols, C::BN);
  const int grid = total < cus[dev] ? total : cus[dev];
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(pp::THREADS), lds, st, p);
  DC_CHECK_LAUNCH(name);
  return DC_OK;
}

// Rows of the per-workgroup BatchNorm partials (IgemmParams::statsPerWg) = 2 x the grid pp_launch() would use; 0 when the
// device's CU count cannot be read (no GPU).
int dc_igemm_pp_stats_rows(const IgemmParams& p) {
  int dev = 0, n = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return 0;
  const int th = p.Ncols <= 32 ? pp::Cfg<4, 1>::TH : pp::Cfg<2, 2>::TH, bn = p.Ncols <= 32 ? 32 : 64;
  const long total = (long)p.N * dc_cdiv(p.Wout, pp::TW) * dc_cdiv(p.Hout, th) * dc_cdiv(p.Ncols, bn);
  return 2 * (int)(total < n ? total : n);
}

// same tile-shape choice as igemm_f16x3.hip's conv3x3 dispatch (and therefore the same BatchNorm-partial tile count)
int dc_igemm_pp_launch(IgemmParams p, hipStream_t st, const char* name) {
  if (p.dzCoef) {
    DC_REQUIRE(p.in2 && p.Cin <= pp::DZ_CIN && !p.inSc && !p.poolOut, DC_EINVAL, "%s: bad dz-on-load launch", name);
    DC_REQUIRE(p.dzOut == nullptr || p.Ncols > 32, DC_EUNSUP, "%s: dz_out needs more than 32 GEMM columns (the 64-column instantiation)", name);
    if (p.bnPartial) return p.Ncols <= 32 ? pp_launch<4, 1, 1, true>(p, st, name) : pp_launch<2, 2, 1, true>(p, st, name);
    return p.Ncols <= 32 ? pp_launch<4, 1, 0, true>(p, st, name) : pp_launch<2, 2, 0, true>(p, st, name);
  }
  if (p.bnPartial) return p.Ncols <= 32 ? pp_launch<4, 1, 1>(p, st, name) : pp_launch<2, 2, 1>(p, st, name);
  if (p.poolOut) return p.Ncols <= 32 ? pp_launch<4, 1, 2>(p, st, name) : pp_launch<2, 2, 2>(p, st, name);
  if (p.statsPerWg && p.stats) return p.Ncols <= 32 ? pp_launch<4, 1, 3>(p, st, name) : pp_launch<2, 2, 3>(p, st, name);
  return p.Ncols <= 32 ? pp_launch<4, 1, 0>(p, st, name) : pp_launch<2, 2, 0>(p, st, name);
}
