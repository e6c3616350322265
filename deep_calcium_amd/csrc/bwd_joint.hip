// Joint backward of a 512^2-class conv block (Conv2D(32 -> 32, 3x3) -> BatchNorm -> ReLU, no Dropout:
// /root/reference/deepcalcium/models/neurons/unet_2d_summary.py:163-167, instances e0b and d0b): ONE kernel stages a
// halo'd tile of dz (formed on load from da and z, dc_bn_bwd_finalize_dzin's table) and of the block input x (BN + ReLU
// on load when the producer's activation is not materialised) ONCE and produces BOTH the data gradient dx and the
// weight-gradient slab.  These layers are HBM-bound end to end (arithmetic intensity 72 flop/byte against a machine
// balance of ~130): the separate data- and weight-gradient kernels read (da, z) twice and x once; here every tensor is
// read once -- da + z + x + dx = 4 tensor passes instead of 7.
//
// One persistent 512-thread workgroup per CU, 4 x 32-pixel tiles (halo'd 6 x 34), fixed wave roles:
//   waves 4-7  producers: request tile t+2's rows (buffer loads, zeros outside the image), form dz / relu(bn(x)) of tile
//              t+1, split into fp16 hi/lo and write the two LDS images of the other stage ([hi|lo][pixel][32 ch] rows
//              of 64 bytes, 16-byte chunks XOR-swizzled by (pixel >> 2) & 3);
//   waves 0-1  data gradient: each owns two pixel rows (2 blocks of 32 px x 32 cin), A = dz at the 9 shifted windows
//              (ds_read_b128, conflict-free through the swizzle), B = the packed weights, resident in LDS for the whole
//              launch; per tile: 108 MFMAs, then the epilogue (dx stores + the BatchNorm-backward sums of the layer in
//              front, accumulated over ALL tiles of the workgroup in registers: one partial row per wave);
//   waves 2-3  weight gradient: each owns two pixel rows as its slice of the contraction, fragments through the
//              transposing ds_read_b64_tr_b16 from the SAME two images, 9 x 16 accumulators kept for the whole launch;
//              per tile: 108 MFMAs.  The two slices are summed through LDS at the end, one slab per workgroup.
// Numerics: those of igemm_f16x3.hip / wgrad_f16x3.hip (split-fp16 operands, power-of-two range-guard scales undone in
// the epilogues, fp32 accumulation); the summation order differs from the separate kernels (tile shape), same tolerance.
#include "wgrad_common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((__vector_size__(4 * sizeof(short)))) short tr_v4i16;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// DC_JOINT_ABL (timing experiments only, results are garbage): 1 data-gradient waves skip their MFMA block, 2 weight-gradient
// waves skip theirs, 4 producers do not split / write LDS, 8 producers do not load
#ifndef DC_JOINT_ABL
#define DC_JOINT_ABL 0
#endif
namespace bj {
constexpr int C = 32;
constexpr int TH = 4, TW = 32, THI = TH + 2, TWI = TW + 2, NPIX = THI * TWI;      // 204 halo'd pixels
constexpr int PLANE = 13312;                 // NPIX * 64 bytes rounded up to a multiple of 1024 (the swizzle reads address bits 8-9)
constexpr int IMG = 2 * PLANE;               // hi | lo
constexpr int RAW = TH * TW * C * 4;          // the tile's interior x values as loaded (fp32): the pre-BN tensor of the layer in front
constexpr int STAGE = 2 * IMG + RAW;         // dz image | x image | raw interior x
constexpr int W_SLOTS = 9 * 4 * 2 * C;       // packed data-gradient weights: [tap][k8][hi|lo][col] 16-byte slots
constexpr int LDS_BYTES = 2 * STAGE + 64;
constexpr int THREADS = 512;
constexpr int NL = (NPIX * 8 + 255) / 256;   // float4 loads per producer thread and tensor (7)
constexpr unsigned OOB = 0x80000000u;
static_assert(LDS_BYTES <= 160 * 1024 && NPIX * 64 <= PLANE && PLANE % 1024 == 0 && STAGE % 1024 == 0, "LDS plan");

// 16-byte chunk c of pixel p lives at chunk c ^ ((p >> 2) & 3): eight consecutive pixels read the same chunk of
// their 64-byte rows from eight different bank groups; a 4-aligned pixel group keeps its chunks together (tr reads)
__device__ __forceinline__ int swz(int rel) { return rel ^ ((rel >> 4) & 0x30); }

__device__ __forceinline__ void split4(const f32x4 v, float s, u32x2& hi, u32x2& lo) {
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

__device__ __forceinline__ f16x8 tr_frag(const char* base, int off1, int off2) {
  typedef __attribute__((address_space(3))) tr_v4i16* lds_p;
  const tr_v4i16 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(base + off1));
  const tr_v4i16 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(base + off2));
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  s16x8 v = {r0[0], r0[1], r0[2], r0[3], r1[0], r1[1], r1[2], r1[3]};
  return __builtin_bit_cast(f16x8, v);
}
}  // namespace bj

struct JointParams {
  const float* x; const float* xSc; const float* xSh; const float* xAbound;
  const float* da; const float* z; const float* dzCoef;
  const float* wp;
  float* dx;
  const float* redZ; const float* redMean; const float* redInvstd; const float* redGamma; const float* redBeta;
  float* bnPartial; float* bnAmax;
  float* slabs;
  int N, H, W, tilesX, tilesY;
};

__global__ __launch_bounds__(bj::THREADS, 1) void bwd_joint32_kernel(JointParams p) {
  using namespace bj;
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  float* tmp = reinterpret_cast<float*>(smem + 2 * STAGE);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // 0..7

  const float x_scale = dc_block_guard_scale(p.xAbound, C, tmp);
  const float dz_scale = dc_block_guard_scale(p.dzCoef + 6 * C, C, tmp);
  const float w_scale = p.wp[W_SLOTS * 4];                         // trailer of the packed weights
  // the layer in front's pre-BN values for its fused sums: when that tensor IS the x operand (BN + ReLU on load: x = its z) the
  // producers leave the tile's interior values in LDS as they loaded them -- no second trip to HBM (537 MB per launch less)
  const bool red_lds = p.redZ != nullptr && p.redZ == p.x;

  // this workgroup's tiles: each XCD walks a contiguous range of the (image, column, row) list -- consecutive positions
  // are vertically adjacent tiles whose halo rows overlap --, its workgroups striding through it (igemm_pp.hip)
  const int total = p.N * p.tilesX * p.tilesY;
  const int G = (int)gridDim.x, xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
  const int nx = (G + 7 - xcd) >> 3;
  const int qq = total >> 3, rr = total & 7;
  const int xstart = xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq;
  const int xcount = qq + (xcd < rr ? 1 : 0);
  const int nt = seq < xcount ? (xcount - seq + nx - 1) / nx : 0;
  struct Tile { int img, y0, x0; };
  auto decode = [&](int j) __attribute__((always_inline)) {
    const int work = xstart + seq + j * nx;
    const int ty = work % p.tilesY, t2 = work / p.tilesY;
    Tile t;
    t.x0 = (t2 % p.tilesX) * TW; t.img = t2 / p.tilesX; t.y0 = ty * TH;
    return t;
  };
  const long img_floats = (long)p.H * p.W * C;

  if (wave >= 4) {
    // ============================ producers: HBM -> registers -> the two fp16 hi/lo images =============================
    const int t = tid & 255, q = t & 7, pb = t >> 3;               // channel quad, first pixel
    const bool xbn = p.xSc != nullptr;
    f32x4 x_sc = {1.f, 1.f, 1.f, 1.f}, x_sh = {0.f, 0.f, 0.f, 0.f};
    if (xbn) { x_sc = *reinterpret_cast<const f32x4*>(p.xSc + 4 * q); x_sh = *reinterpret_cast<const f32x4*>(p.xSh + 4 * q); }
    const float* ct = p.dzCoef + 4 * q;
    const f32x4 g_sc = *reinterpret_cast<const f32x4*>(ct), g_sh = *reinterpret_cast<const f32x4*>(ct + C),
                d_mu = *reinterpret_cast<const f32x4*>(ct + 2 * C), d_A = *reinterpret_cast<const f32x4*>(ct + 3 * C),
                d_D = *reinterpret_cast<const f32x4*>(ct + 4 * C), d_E = *reinterpret_cast<const f32x4*>(ct + 5 * C);

    // lane-constant part of the global offsets / the image positions of this thread's NL pixels are recomputed per request
    // (registers: the two x sets + one (da, z) set + the per-channel tables are what the producers hold)
    auto voff = [&](const Tile& tl, int k, bool& ok) {
      const int pix = pb + 32 * k;
      const int r = __umul24(pix, (65536 + TWI - 1) / TWI) >> 16;          // pix / TWI
      const int c = pix - __umul24(r, TWI);
      const int y = tl.y0 - 1 + r, xx = tl.x0 - 1 + c;
      ok = pix < NPIX && (unsigned)y < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
      return ok ? (unsigned)(((y * p.W + xx) * C + 4 * q) * 4) : OOB;
    };
    auto request_x = [&](int j, f32x4 (&rx)[NL], unsigned& live) {
      live = 0u;
      const Tile tl = decode(j);
      const __amdgpu_buffer_rsrc_t rsX = dc_make_rsrc(p.x + tl.img * img_floats, (unsigned)(img_floats * 4));
#pragma unroll
      for (int k = 0; k < NL; ++k) {
        bool ok;
        const unsigned off = voff(tl, k, ok);
        rx[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX, off, 0, 0));
        if (ok) live |= 1u << k;
      }
    };
    auto request_dz = [&](int j, f32x4 (&ra)[NL], f32x4 (&rz)[NL]) {
      const Tile tl = decode(j);
      const __amdgpu_buffer_rsrc_t rsA = dc_make_rsrc(p.da + tl.img * img_floats, (unsigned)(img_floats * 4));
      const __amdgpu_buffer_rsrc_t rsZ = dc_make_rsrc(p.z + tl.img * img_floats, (unsigned)(img_floats * 4));
#pragma unroll
      for (int k = 0; k < NL; ++k) {
        bool ok;
        const unsigned off = voff(tl, k, ok);
        ra[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, off, 0, 0));
        rz[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsZ, off, 0, 0));
      }
    };
    auto stage_x = [&](const f32x4 (&rx)[NL], unsigned live, char* set) {
#pragma unroll
      for (int k = 0; k < NL; ++k) {
        const int pix = pb + 32 * k;
        const bool lv = (live >> k) & 1u;
        f32x4 xv;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float xr = rx[k][e];
          xv[e] = xbn ? (lv ? fmaxf(__builtin_fmaf(xr, x_sc[e], x_sh[e]), 0.f) : 0.f) : xr;     // zero padding stays zero
        }
        u32x2 xh, xl;
        split4(xv, x_scale, xh, xl);
        if (pix < NPIX) {
          const int off = swz(pix * 64 + q * 8);
          *reinterpret_cast<u32x2*>(set + IMG + off) = xh;
          *reinterpret_cast<u32x2*>(set + IMG + PLANE + off) = xl;
          if (red_lds) {
            const int r = __umul24(pix, (65536 + TWI - 1) / TWI) >> 16, c = pix - __umul24(r, TWI);
            if (r >= 1 && r <= TH && c >= 1 && c <= TW)
              *reinterpret_cast<f32x4*>(set + 2 * IMG + (((r - 1) * TW + (c - 1)) * C + 4 * q) * 4) = rx[k];
          }
        }
      }
    };
    auto stage_dz = [&](const f32x4 (&ra)[NL], const f32x4 (&rz)[NL], unsigned live, char* set) {
#pragma unroll
      for (int k = 0; k < NL; ++k) {
        const int pix = pb + 32 * k;
        const bool lv = (live >> k) & 1u;
        f32x4 dzv;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float zz = rz[k][e];
          const float y = __builtin_fmaf(zz, g_sc[e], g_sh[e]);             // the forward's own expression: identical ReLU gate
          const float dy = y > 0.f ? ra[k][e] : 0.f;
          const float v = __builtin_fmaf(d_A[e], dy, __builtin_fmaf(d_D[e], zz - d_mu[e], d_E[e]));
          dzv[e] = lv ? v : 0.f;                                           // outside the image dz is zero, not E - D*mu
        }
        u32x2 dh, dl;
        split4(dzv, dz_scale, dh, dl);
        if (pix < NPIX) {
          const int off = swz(pix * 64 + q * 8);
          *reinterpret_cast<u32x2*>(set + off) = dh;
          *reinterpret_cast<u32x2*>(set + PLANE + off) = dl;
        }
      }
    };
    // x: two register sets (tile t+2 in flight while t+1 is split); (da, z): one set, requested right after the previous
    // tile's has been split -- one tile of MFMA time ahead (wgrad_f16x3.hip's B1 scheme: 2 x (x + da + z) does not fit)
    f32x4 rx0[NL], rx1[NL], ra[NL], rz[NL];
    unsigned lv0 = 0u, lv1 = 0u;
    if (nt > 0) {
      request_x(0, rx0, lv0);
      request_dz(0, ra, rz);
      if (nt > 1) request_x(1, rx1, lv1);
      stage_x(rx0, lv0, smem);
      stage_dz(ra, rz, lv0, smem);
      if (nt > 1) request_dz(1, ra, rz);
    }
    __syncthreads();
    for (int i = 0; i < nt; i += 2) {
      if (!(DC_JOINT_ABL & 8) && i + 2 < nt) request_x(i + 2, rx0, lv0);
      if (!(DC_JOINT_ABL & 4) && i + 1 < nt) { stage_x(rx1, lv1, smem + STAGE); stage_dz(ra, rz, lv1, smem + STAGE); }
      if (!(DC_JOINT_ABL & 8) && i + 2 < nt) request_dz(i + 2, ra, rz);
      __syncthreads();
      if (i + 1 < nt) {
        if (!(DC_JOINT_ABL & 8) && i + 3 < nt) request_x(i + 3, rx1, lv1);
        if (!(DC_JOINT_ABL & 4) && i + 2 < nt) { stage_x(rx0, lv0, smem); stage_dz(ra, rz, lv0, smem); }
        if (!(DC_JOINT_ABL & 8) && i + 3 < nt) request_dz(i + 3, ra, rz);
        __syncthreads();
      }
    }
#pragma unroll 1
    for (int k = 0; k < 2 * 9; ++k) __syncthreads();             // the weight-gradient waves' cross-slice reduction
    return;
  }

  const int li = lane & 31, h = lane >> 5;
  if (wave < 2) {
    // ================================ data gradient: 2 rows x 32 px x 32 cin per tile ==================================
    const int wd = wave;
    const float out_scale = 1.f / (dz_scale * w_scale);
    const bool red = p.redZ != nullptr;
    float gsc = 0.f, gsh = 0.f, rmu = 0.f, ris = 0.f;
    if (red) {
      rmu = p.redMean[li]; ris = p.redInvstd[li];
      dc_bn_affine(rmu, ris, p.redGamma[li], p.redBeta[li], gsc, gsh);
    }
    float s1 = 0.f, s2 = 0.f, amax = 0.f;                        // this lane's channel, all tiles of the workgroup
    int a_rel[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) a_rel[mb] = ((2 * wd + mb) * TWI + li) * 64 + h * 16;
    // the 36 weight fragments of this lane (tap, 16-channel half, hi | lo) live in registers for the whole launch: they are the
    // same for every tile, and the LDS they occupied now holds the raw interior x values
    u32x4 wreg[18][2];
    {
      const u32x4* wsrc = reinterpret_cast<const u32x4*>(p.wp) + h * 64 + li;      // slot ((tap*4 + ks*2 + h)*2 + hl)*32 + li
#pragma unroll
      for (int g = 0; g < 18; ++g) {
        wreg[g][0] = wsrc[(g >> 1) * 256 + (g & 1) * 128];
        wreg[g][1] = wsrc[(g >> 1) * 256 + (g & 1) * 128 + 32];
      }
    }
    __syncthreads();                                             // stage 0 is in LDS
    for (int i = 0; i < nt; ++i) {
      const char* cur = smem + (i & 1) * STAGE;
      int ar0 = a_rel[0], ar1 = a_rel[1];
      asm volatile("" : "+v"(ar0), "+v"(ar1));     // opaque per tile: or the 36 swizzled offsets are hoisted and spilled
      f32x16 acc[2];
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;
      const Tile tl = decode(i);
      const __amdgpu_buffer_rsrc_t rsO = dc_make_rsrc(p.dx + tl.img * img_floats, (unsigned)(img_floats * 4));
      __builtin_amdgcn_s_setprio(2);
      {
        // 18 (tap, 16-channel half) groups of 6 MFMAs; the 6 fragments of group g+1 are requested from LDS before the MFMAs
        // of group g issue (one wave per SIMD feeds the pipe: without the explicit double buffer every group paid the LDS
        // latency -- the consumer side alone ran at 45 % of the MFMA issue rate)
        constexpr int NG = (DC_JOINT_ABL & 1) ? 2 : 18;
        f16x8 ah[2][2], al[2][2];
        auto fetch = [&](int g, int buf) __attribute__((always_inline)) {
          const int tap = g >> 1, ks = g & 1;
          const int toff = ((tap / 3) * TWI + (tap % 3)) * 64 + ks * 32;
#pragma unroll
          for (int mb = 0; mb < 2; ++mb) {
            const int ao = swz((mb ? ar1 : ar0) + toff);
            ah[buf][mb] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(cur + ao));
            al[buf][mb] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(cur + PLANE + ao));
          }
        };
        fetch(0, 0);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          const int b = g & 1;
          if (g + 1 < NG) fetch(g + 1, b ^ 1);
          __builtin_amdgcn_sched_barrier(0);
          const f16x8 bh = __builtin_bit_cast(f16x8, wreg[g][0]), bl = __builtin_bit_cast(f16x8, wreg[g][1]);
#pragma unroll
          for (int mb = 0; mb < 2; ++mb) {
            acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[b][mb], bh, acc[mb], 0, 0, 0);
            acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[b][mb], bl, acc[mb], 0, 0, 0);
            acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[b][mb], bh, acc[mb], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      __builtin_amdgcn_s_setprio(0);
      // ---- epilogue: dx stores (lane = cin channel li, registers = pixels) + the sums of the layer in front ----------
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        const int oy = tl.y0 + 2 * wd + mb, oxb = tl.x0 + 4 * h;
        const bool row_ok = oy < p.H;
        unsigned offs[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int colc = (r & 3) + 8 * (r >> 2);
          const bool ok = row_ok && (oxb + colc) < p.W;
          offs[r] = ok ? (unsigned)(((oy * p.W + oxb + colc) * C + li) * 4) : OOB;
        }
        float zr[16];
        if (red_lds) {                        // the producers left this tile's interior x (= that layer's z) in LDS
          const float* raw = reinterpret_cast<const float*>(cur + 2 * IMG) + ((2 * wd + mb) * TW + 4 * h) * C + li;
#pragma unroll
          for (int r = 0; r < 16; ++r) zr[r] = raw[((r & 3) + 8 * (r >> 2)) * C];
        } else if (red) {                     // another tensor: requested here, the stores of this block go out while they fly
          // (requesting them a tile ahead / before the MFMA block was measured slower: they compete with the producers' prefetch)
          const __amdgpu_buffer_rsrc_t rsR = dc_make_rsrc(p.redZ + tl.img * img_floats, (unsigned)(img_floats * 4));
#pragma unroll
          for (int r = 0; r < 16; ++r) zr[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsR, offs[r], 0, 0));
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = acc[mb][r] * out_scale;
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsO, offs[r], 0, 0);
        }
        if (red) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float v = acc[mb][r] * out_scale;
            const float y = __builtin_fmaf(zr[r], gsc, gsh);
            const float dy = (!(offs[r] >> 31) && y > 0.f) ? v : 0.f;
            s1 += dy;
            s2 = __builtin_fmaf(dy, (zr[r] - rmu) * ris, s2);
            amax = fmaxf(amax, fabsf(dy));
          }
        }
      }
      __syncthreads();
    }
    if (red) {
      s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32); amax = fmaxf(amax, __shfl_xor(amax, 32));
      if (h == 0) {
        const long row = (long)blockIdx.x * 2 + wd;
        p.bnPartial[(row * C + li) * 2] = s1;
        p.bnPartial[(row * C + li) * 2 + 1] = s2;
        if (p.bnAmax) p.bnAmax[row * C + li] = amax;
      }
    }
#pragma unroll 1
    for (int k = 0; k < 2 * 9; ++k) __syncthreads();
    return;
  }

  // ================================== weight gradient: 2 rows of every tile as the contraction slice ====================
  const int ww = wave - 2;
  const int cb = (lane >> 4) & 1, c16 = lane & 15, q4 = c16 >> 2, pp = c16 & 3;
  const int frag = (cb * 2 + (pp >> 1)) * 16 + (pp & 1) * 8;     // 16-channel half, channel quad inside it
  // lane's byte offset inside the image: its wave's first row, its first group of 4 contraction pixels (x = 8h + q4), its chunk
  const int lane_rel = (2 * ww * TWI + 8 * h + q4) * 64 + frag;
  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  __syncthreads();
  for (int i = 0; i < nt; ++i) {
    const char* cur = smem + (i & 1) * STAGE;
    int lrel = lane_rel;
    asm volatile("" : "+v"(lrel));               // opaque per tile: or the 80 swizzled offsets are hoisted and spilled
    __builtin_amdgcn_s_setprio(2);
    {
      // 36 (k-step, tap) groups of 3 MFMAs: k-step = (row of this wave's two, 16-pixel half); the x fragments of group g+1
      // (and the dz fragments of the next k-step) are requested before the MFMAs of group g issue
      constexpr int NG = (DC_JOINT_ABL & 2) ? 2 : 36;
      f16x8 ah[2], al[2], bh[2], bl[2];
      auto fetch_a = [&](int g, int buf) __attribute__((always_inline)) {
        const int ks = g / 9, tap = g % 9;
        const int r = ks >> 1, xs = ks & 1;                      // (row relative to the wave's first: in lane_rel)
        const int pbase = lrel + ((r + tap / 3) * TWI + 16 * xs + tap % 3) * 64;
        const int a0 = swz(pbase), a1 = swz(pbase + 4 * 64);
        ah[buf] = tr_frag(cur + IMG, a0, a1);
        al[buf] = tr_frag(cur + IMG + PLANE, a0, a1);
      };
      auto fetch_b = [&](int ks, int buf) __attribute__((always_inline)) {
        const int r = ks >> 1, xs = ks & 1;
        const int brel = lrel + ((r + 1) * TWI + 16 * xs + 1) * 64;
        const int b0 = swz(brel), b1 = swz(brel + 4 * 64);
        bh[buf] = tr_frag(cur, b0, b1);
        bl[buf] = tr_frag(cur + PLANE, b0, b1);
      };
      fetch_b(0, 0);
      fetch_a(0, 0);
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int ca = g & 1, ks = g / 9, tap = g % 9, cbuf = ks & 1;
        if (g + 1 < NG) {
          fetch_a(g + 1, ca ^ 1);
          if ((g + 1) % 9 == 0) fetch_b(ks + 1, cbuf ^ 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ca], bh[cbuf], acc[tap], 0, 0, 0);
        acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ca], bl[cbuf], acc[tap], 0, 0, 0);
        acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ca], bh[cbuf], acc[tap], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_s_setprio(0);
    __syncthreads();
  }
  WgradParams wp;
  wp.slabs = p.slabs; wp.Cm = C; wp.Cn = C;
  wgrad_store<9, 1, 1, 2>(wp, acc, smem, (int)blockIdx.x, 0, 0, 0, 0, ww, lane, 1.f / (x_scale * dz_scale));
}

// ---------------------------------------------------------------------------------------------------------------------
// The 64 -> 32 block of the same resolution (d0a: x = the 64-channel concat buffer, materialised; dx = its gradient, which
// feeds a Dropout block and the max-pool backward: no fused sums).  Same tiles, same dz image; differences:
//   * the x image has 128-byte pixel rows (64 channels: 8 chunks, XOR-swizzled with bit 1 of the pixel index: the
//     transposing reads of 4 consecutive pixels then cover all 64 banks); two stages of (dz + x) fill the LDS (156 KB), so
//   * the data-gradient weights (74 KB) are NOT LDS-resident: the 72 fragments a tile needs are read from global memory
//     (L2-resident, the same for every workgroup), three (tap, k-step) groups ahead of the MFMAs that use them;
//   * data gradient: waves 0-1, two rows each x 64 cin columns (2 x 2 blocks: the weight fragments are shared by two row
//     blocks, the dz fragments by two column blocks); weight gradient: wave 2 + ww owns the 32 input channels [32 ww, 32 ww + 32)
//     for ALL four rows and all 9 taps (144 accumulators): no cross-wave reduction, each writes its half of the slab.
// Per tile and consumer wave 216 MFMAs; fragment reads 464 KB per tile = 52 % of the LDS bandwidth at full MFMA rate.
namespace bj64 {
using namespace bj;
constexpr int CX = 64;                         // x / dx channels
constexpr int XPLANE = 26624;                  // NPIX * 128 bytes rounded up to a multiple of 1024
constexpr int XIMG = 2 * XPLANE;
constexpr int STAGE64 = IMG + XIMG;            // dz image | x image
constexpr int LDS64 = 2 * STAGE64 + 64;
constexpr int NLX = (NPIX * 16 + 255) / 256;   // float4 loads of x per producer thread (13)
static_assert(LDS64 <= 160 * 1024 && NPIX * 128 <= XPLANE && XPLANE % 1024 == 0 && IMG % 1024 == 0, "LDS plan");
__device__ __forceinline__ int swzx(int rel) { return rel ^ ((rel >> 2) & 0x40); }     // chunk ^= 4 * bit 1 of the pixel
}  // namespace bj64

__global__ __launch_bounds__(bj::THREADS, 1) void bwd_joint64_kernel(JointParams p) {
  using namespace bj64;
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  float* tmp = reinterpret_cast<float*>(smem + 2 * STAGE64);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float x_scale = dc_block_guard_scale(p.xAbound, CX, tmp);
  const float dz_scale = dc_block_guard_scale(p.dzCoef + 6 * C, C, tmp);
  constexpr int W_SLOTS64 = 9 * 4 * 2 * CX;     // [tap][k8][hi|lo][64 cols]
  const float w_scale = p.wp[W_SLOTS64 * 4];

  const int total = p.N * p.tilesX * p.tilesY;
  const int G = (int)gridDim.x, xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
  const int nx = (G + 7 - xcd) >> 3;
  const int qq = total >> 3, rr = total & 7;
  const int xstart = xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq;
  const int xcount = qq + (xcd < rr ? 1 : 0);
  const int nt = seq < xcount ? (xcount - seq + nx - 1) / nx : 0;
  struct Tile { int img, y0, x0; };
  auto decode = [&](int j) __attribute__((always_inline)) {
    const int work = xstart + seq + j * nx;
    const int ty = work % p.tilesY, t2 = work / p.tilesY;
    Tile t;
    t.x0 = (t2 % p.tilesX) * TW; t.img = t2 / p.tilesX; t.y0 = ty * TH;
    return t;
  };
  const long dz_img = (long)p.H * p.W * C, x_img = (long)p.H * p.W * CX;

  if (wave >= 4) {
    // ============================ producers ===========================================================================
    const int t = tid & 255;
    const int q = t & 7, pb = t >> 3;               // dz: channel quad (8), first pixel (32 pixels per pass)
    const int qx = t & 15, pbx = t >> 4;            // x: channel quad (16), first pixel (16 pixels per pass)
    const float* ct = p.dzCoef + 4 * q;
    const f32x4 g_sc = *reinterpret_cast<const f32x4*>(ct), g_sh = *reinterpret_cast<const f32x4*>(ct + C),
                d_mu = *reinterpret_cast<const f32x4*>(ct + 2 * C), d_A = *reinterpret_cast<const f32x4*>(ct + 3 * C),
                d_D = *reinterpret_cast<const f32x4*>(ct + 4 * C), d_E = *reinterpret_cast<const f32x4*>(ct + 5 * C);
    auto pix_ok = [&](const Tile& tl, int pix, int& y, int& xx) {
      const int r = __umul24(pix, (65536 + TWI - 1) / TWI) >> 16;
      const int c = pix - __umul24(r, TWI);
      y = tl.y0 - 1 + r; xx = tl.x0 - 1 + c;
      return pix < NPIX && (unsigned)y < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
    };
    auto request_x = [&](int j, f32x4 (&rx)[NLX]) {
      const Tile tl = decode(j);
      const __amdgpu_buffer_rsrc_t rsX = dc_make_rsrc(p.x + tl.img * x_img, (unsigned)(x_img * 4));
#pragma unroll
      for (int k = 0; k < NLX; ++k) {
        int y, xx;
        const bool ok = pix_ok(tl, pbx + 16 * k, y, xx);
        const unsigned off = ok ? (unsigned)(((y * p.W + xx) * CX + 4 * qx) * 4) : OOB;
        rx[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsX, off, 0, 0));      // zeros outside the image
      }
    };
    auto request_dz = [&](int j, f32x4 (&ra)[NL], f32x4 (&rz)[NL], unsigned& live) {
      live = 0u;
      const Tile tl = decode(j);
      const __amdgpu_buffer_rsrc_t rsA = dc_make_rsrc(p.da + tl.img * dz_img, (unsigned)(dz_img * 4));
      const __amdgpu_buffer_rsrc_t rsZ = dc_make_rsrc(p.z + tl.img * dz_img, (unsigned)(dz_img * 4));
#pragma unroll
      for (int k = 0; k < NL; ++k) {
        int y, xx;
        const bool ok = pix_ok(tl, pb + 32 * k, y, xx);
        const unsigned off = ok ? (unsigned)(((y * p.W + xx) * C + 4 * q) * 4) : OOB;
        ra[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, off, 0, 0));
        rz[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsZ, off, 0, 0));
        if (ok) live |= 1u << k;
      }
    };
    auto stage_x = [&](const f32x4 (&rx)[NLX], char* set) {
#pragma unroll
      for (int k = 0; k < NLX; ++k) {
        const int pix = pbx + 16 * k;
        u32x2 xh, xl;
        split4(rx[k], x_scale, xh, xl);
        if (pix < NPIX) {
          const int off = swzx(pix * 128 + qx * 8);
          *reinterpret_cast<u32x2*>(set + IMG + off) = xh;
          *reinterpret_cast<u32x2*>(set + IMG + XPLANE + off) = xl;
        }
      }
    };
    auto stage_dz = [&](const f32x4 (&ra)[NL], const f32x4 (&rz)[NL], unsigned live, char* set) {
#pragma unroll
      for (int k = 0; k < NL; ++k) {
        const int pix = pb + 32 * k;
        const bool lv = (live >> k) & 1u;
        f32x4 dzv;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float zz = rz[k][e];
          const float y = __builtin_fmaf(zz, g_sc[e], g_sh[e]);
          const float dy = y > 0.f ? ra[k][e] : 0.f;
          const float v = __builtin_fmaf(d_A[e], dy, __builtin_fmaf(d_D[e], zz - d_mu[e], d_E[e]));
          dzv[e] = lv ? v : 0.f;
        }
        u32x2 dh, dl;
        split4(dzv, dz_scale, dh, dl);
        if (pix < NPIX) {
          const int off = swz(pix * 64 + q * 8);
          *reinterpret_cast<u32x2*>(set + off) = dh;
          *reinterpret_cast<u32x2*>(set + PLANE + off) = dl;
        }
      }
    };
    f32x4 rx0[NLX], rx1[NLX], ra[NL], rz[NL];
    unsigned lv = 0u;
    if (nt > 0) {
      request_x(0, rx0);
      request_dz(0, ra, rz, lv);
      if (nt > 1) request_x(1, rx1);
      stage_x(rx0, smem);
      stage_dz(ra, rz, lv, smem);
      if (nt > 1) request_dz(1, ra, rz, lv);
    }
    __syncthreads();
    for (int i = 0; i < nt; i += 2) {
      if (i + 2 < nt) request_x(i + 2, rx0);
      if (i + 1 < nt) { stage_x(rx1, smem + STAGE64); stage_dz(ra, rz, lv, smem + STAGE64); }
      if (i + 2 < nt) request_dz(i + 2, ra, rz, lv);
      __syncthreads();
      if (i + 1 < nt) {
        if (i + 3 < nt) request_x(i + 3, rx1);
        if (i + 2 < nt) { stage_x(rx0, smem); stage_dz(ra, rz, lv, smem); }
        if (i + 3 < nt) request_dz(i + 3, ra, rz, lv);
        __syncthreads();
      }
    }
    return;
  }

  const int li = lane & 31, h = lane >> 5;
  if (wave < 2) {
    // ================================ data gradient: 2 rows x 32 px x 64 cin per tile ==================================
    const int wd = wave;
    const float out_scale = 1.f / (dz_scale * w_scale);
    int a_rel[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) a_rel[mb] = ((2 * wd + mb) * TWI + li) * 64 + h * 16;
    const __amdgpu_buffer_rsrc_t rsW = dc_make_rsrc(p.wp, (unsigned)W_SLOTS64 * 16u);
    const unsigned w_rel = (unsigned)((h * 2 * CX + li) * 16);      // slot ((tap*4 + ks*2 + h)*2 + hl)*64 + nb*32 + li
    __syncthreads();
    for (int i = 0; i < nt; ++i) {
      const char* cur = smem + (i & 1) * STAGE64;
      int ar0 = a_rel[0], ar1 = a_rel[1];
      asm volatile("" : "+v"(ar0), "+v"(ar1));
      f32x16 acc[2][2];
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[mb][nb][r] = 0.f;
      __builtin_amdgcn_s_setprio(2);
      {
        constexpr int NG = 18, AHEAD = 3;                          // weight fragments: global (L2) loads AHEAD groups ahead
        u32x4 wf[AHEAD + 1][2][2];                                 // [ring][nb][hi|lo]
        f16x8 ah[2][2], al[2][2];
        auto fetch_w = [&](int g, int slot) __attribute__((always_inline)) {
          const int tap = g >> 1, ks = g & 1;
          const int soff = ((tap * 4 + ks * 2) * 2) * CX * 16;
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) {
            wf[slot][nb][0] = __builtin_amdgcn_raw_buffer_load_b128(rsW, w_rel + nb * 512, soff, 0);
            wf[slot][nb][1] = __builtin_amdgcn_raw_buffer_load_b128(rsW, w_rel + nb * 512, soff + CX * 16, 0);
          }
        };
        auto fetch_a = [&](int g, int buf) __attribute__((always_inline)) {
          const int tap = g >> 1, ks = g & 1;
          const int toff = ((tap / 3) * TWI + (tap % 3)) * 64 + ks * 32;
#pragma unroll
          for (int mb = 0; mb < 2; ++mb) {
            const int ao = swz((mb ? ar1 : ar0) + toff);
            ah[buf][mb] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(cur + ao));
            al[buf][mb] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(cur + PLANE + ao));
          }
        };
#pragma unroll
        for (int g = 0; g < AHEAD; ++g) fetch_w(g, g);
        fetch_a(0, 0);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          const int b = g & 1, ws = g % (AHEAD + 1);
          if (g + AHEAD < NG) fetch_w(g + AHEAD, (g + AHEAD) % (AHEAD + 1));
          if (g + 1 < NG) fetch_a(g + 1, b ^ 1);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
              const f16x8 bh = __builtin_bit_cast(f16x8, wf[ws][nb][0]), bl = __builtin_bit_cast(f16x8, wf[ws][nb][1]);
              acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[b][mb], bh, acc[mb][nb], 0, 0, 0);
              acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[b][mb], bl, acc[mb][nb], 0, 0, 0);
              acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[b][mb], bh, acc[mb][nb], 0, 0, 0);
            }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      __builtin_amdgcn_s_setprio(0);
      const Tile tl = decode(i);
      const __amdgpu_buffer_rsrc_t rsO = dc_make_rsrc(p.dx + tl.img * x_img, (unsigned)(x_img * 4));
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        const int oy = tl.y0 + 2 * wd + mb, oxb = tl.x0 + 4 * h;
        const bool row_ok = oy < p.H;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int colc = (r & 3) + 8 * (r >> 2);
          const bool ok = row_ok && (oxb + colc) < p.W;
          const unsigned off = ok ? (unsigned)(((oy * p.W + oxb + colc) * CX + li) * 4) : OOB;
#pragma unroll
          for (int nb = 0; nb < 2; ++nb)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, acc[mb][nb][r] * out_scale), rsO, off, nb * 128, 0);
        }
      }
      __syncthreads();
    }
    return;
  }

  // ============== weight gradient: wave 2 + ww = input channels [32 ww, 32 ww + 32), all four rows, all taps ==============
  const int ww = wave - 2;
  const int cb = (lane >> 4) & 1, c16 = lane & 15, q4 = c16 >> 2, pp = c16 & 3;
  const int fragz = (cb * 2 + (pp >> 1)) * 16 + (pp & 1) * 8;                 // dz image: 4 chunks per pixel row
  const int fragx = (ww * 4 + cb * 2 + (pp >> 1)) * 16 + (pp & 1) * 8;        // x image: 8 chunks, this wave's 32-channel half
  const int lane_z = (8 * h + q4) * 64 + fragz, lane_x = (8 * h + q4) * 128 + fragx;
  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  __syncthreads();
  for (int i = 0; i < nt; ++i) {
    const char* cur = smem + (i & 1) * STAGE64;
    int lz = lane_z, lx = lane_x;
    asm volatile("" : "+v"(lz), "+v"(lx));
    __builtin_amdgcn_s_setprio(2);
    {
      constexpr int NG = 72;                                       // 8 k-steps (row, 16-pixel half) x 9 taps
      f16x8 ah[2], al[2], bh[2], bl[2];
      auto fetch_a = [&](int g, int buf) __attribute__((always_inline)) {
        const int ks = g / 9, tap = g % 9;
        const int r = ks >> 1, xs = ks & 1;
        const int pbase = lx + ((r + tap / 3) * TWI + 16 * xs + tap % 3) * 128;
        const int a0 = swzx(pbase), a1 = swzx(pbase + 4 * 128);
        ah[buf] = tr_frag(cur + IMG, a0, a1);
        al[buf] = tr_frag(cur + IMG + XPLANE, a0, a1);
      };
      auto fetch_b = [&](int ks, int buf) __attribute__((always_inline)) {
        const int r = ks >> 1, xs = ks & 1;
        const int brel = lz + ((r + 1) * TWI + 16 * xs + 1) * 64;
        const int b0 = swz(brel), b1 = swz(brel + 4 * 64);
        bh[buf] = tr_frag(cur, b0, b1);
        bl[buf] = tr_frag(cur + PLANE, b0, b1);
      };
      fetch_b(0, 0);
      fetch_a(0, 0);
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int ca = g & 1, ks = g / 9, tap = g % 9, cbuf = ks & 1;
        if (g + 1 < NG) {
          fetch_a(g + 1, ca ^ 1);
          if ((g + 1) % 9 == 0) fetch_b(ks + 1, cbuf ^ 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ca], bh[cbuf], acc[tap], 0, 0, 0);
        acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ca], bl[cbuf], acc[tap], 0, 0, 0);
        acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ca], bh[cbuf], acc[tap], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_s_setprio(0);
    __syncthreads();
  }
  // slab [9][64][32] of this workgroup: rows m = input channel, columns n = dz channel (C/D: col = lane & 31, row = (r&3) + 8 (r>>2) + 4h)
  const float ws_scale = 1.f / (x_scale * dz_scale);
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    float* dst = p.slabs + ((long)blockIdx.x * 9 + tap) * CX * C;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = 32 * ww + (r & 3) + 8 * (r >> 2) + 4 * h;
      dst[m * C + li] = acc[tap][r] * ws_scale;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
static int joint_grid(int N, int H, int W) {
  static int cus[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (cus[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return 0;
    cus[dev] = n;
  }
  const long total = (long)N * dc_cdiv(W, bj::TW) * dc_cdiv(H, bj::TH);
  return (int)(total < cus[dev] ? total : cus[dev]);
}
static bool joint_serves(int N, int H, int W, int Cin, int Cout) {
  return N > 0 && (Cin == bj::C || Cin == bj64::CX) && Cout == bj::C && W >= 32 && H >= 4 && (long)H * W * Cin * 4 < (1L << 31);
}

// rows of bn_partial / amax_partial (2 per workgroup) when the joint kernel serves this shape, else 0
extern "C" int dc_conv3x3_bwd_joint_blocks(int N, int H, int W, int Cin, int Cout) {
  if (!joint_serves(N, H, W, Cin, Cout)) return 0;
  return 2 * joint_grid(N, H, W);
}
extern "C" long dc_conv3x3_bwd_joint_ws_floats(int N, int H, int W, int Cin, int Cout) {
  if (!joint_serves(N, H, W, Cin, Cout)) return 0;
  const long L = 9L * Cin * Cout;
  return (long)joint_grid(N, H, W) * L + 32 * L;
}
extern "C" int dc_conv3x3_bwd_joint_f16x3(const float* x, const float* in_sc, const float* in_sh, const float* x_abound,
                                          const float* da, const float* z, const float* dz_coef, const void* wp16, float* dx,
                                          const float* red_z, const float* red_mean, const float* red_invstd,
                                          const float* red_gamma, const float* red_beta, float* bn_partial,
                                          float* amax_partial, float* dw, float* ws, int N, int H, int W, int Cin, int Cout,
                                          dc_stream_t stream) {
  DC_REQUIRE(x && da && z && dz_coef && wp16 && dx && dw && ws, DC_EINVAL, "dc_conv3x3_bwd_joint_f16x3: null pointer");
  DC_REQUIRE(dc_aligned16(x) && dc_aligned16(da) && dc_aligned16(z) && dc_aligned16(dz_coef) && dc_aligned16(wp16) && dc_aligned16(dx),
             DC_EINVAL, "dc_conv3x3_bwd_joint_f16x3: pointers must be 16-byte aligned");
  DC_REQUIRE((in_sc == nullptr) == (in_sh == nullptr) && (!in_sc || (dc_aligned16(in_sc) && dc_aligned16(in_sh))), DC_EINVAL,
             "dc_conv3x3_bwd_joint_f16x3: in_scale / in_shift go together, 16-byte aligned");
  DC_REQUIRE(red_z == nullptr || (red_mean && red_invstd && red_gamma && red_beta && bn_partial), DC_EINVAL,
             "dc_conv3x3_bwd_joint_f16x3: red_z needs red_mean / red_invstd / red_gamma / red_beta / bn_partial");
  DC_REQUIRE(joint_serves(N, H, W, Cin, Cout), DC_EUNSUP,
             "dc_conv3x3_bwd_joint_f16x3: shape not served (dc_conv3x3_bwd_joint_blocks() == 0): use the separate kernels");
  const int grid = joint_grid(N, H, W);
  DC_REQUIRE(grid > 0, DC_EHIP, "dc_conv3x3_bwd_joint_f16x3: no device");
  static DcLdsAttr lds_attr, lds_attr64;
  if (Cin == bj::C) {
    if (int rc = dc_func_max_lds(lds_attr, reinterpret_cast<const void*>(bwd_joint32_kernel), bj::LDS_BYTES, "conv3x3_bwd_joint_f16x3"))
      return rc;
  } else {
    DC_REQUIRE(!in_sc && !red_z, DC_EUNSUP, "dc_conv3x3_bwd_joint_f16x3: the 64 -> 32 kernel takes a materialised x and emits no sums");
    if (int rc = dc_func_max_lds(lds_attr64, reinterpret_cast<const void*>(bwd_joint64_kernel), bj64::LDS64, "conv3x3_bwd_joint_f16x3"))
      return rc;
  }
  JointParams p;
  p.x = x; p.xSc = in_sc; p.xSh = in_sh; p.xAbound = x_abound; p.da = da; p.z = z; p.dzCoef = dz_coef;
  p.wp = reinterpret_cast<const float*>(wp16); p.dx = dx;
  p.redZ = red_z; p.redMean = red_mean; p.redInvstd = red_invstd; p.redGamma = red_gamma; p.redBeta = red_beta;
  p.bnPartial = bn_partial; p.bnAmax = amax_partial; p.slabs = ws;
  p.N = N; p.H = H; p.W = W; p.tilesX = dc_cdiv(W, bj::TW); p.tilesY = dc_cdiv(H, bj::TH);
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  const bool bracket = dc_take_bracket(&ev0, &ev1);          // dc_bracket_next_launch: the matrix kernel without the slab reduction
  if (bracket && ev0) (void)hipEventRecord(ev0, (hipStream_t)stream);
  if (Cin == bj::C)
    hipLaunchKernelGGL(bwd_joint32_kernel, dim3((unsigned)grid), dim3(bj::THREADS), bj::LDS_BYTES, (hipStream_t)stream, p);
  else
    hipLaunchKernelGGL(bwd_joint64_kernel, dim3((unsigned)grid), dim3(bj::THREADS), bj64::LDS64, (hipStream_t)stream, p);
  if (bracket && ev1) (void)hipEventRecord(ev1, (hipStream_t)stream);
  DC_CHECK_LAUNCH("dc_conv3x3_bwd_joint_f16x3");
  const long L = 9L * Cin * Cout;
  return dc_reduce_partials(ws, grid, L, 1.0f, dw, ws + (long)grid * L, stream);
}
