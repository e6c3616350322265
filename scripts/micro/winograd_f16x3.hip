// Winograd F(2x2, 3x3) conv3x3 forward in the TRANSFORMED domain on the split-fp16 matrix path -- round-5 review item 3:
// "measure it, with a gate".  3x3 convolutions are 95.5 % of the network's FLOPs (Conv2D(nf, (3,3)),
// /root/reference/deepcalcium/models/neurons/unet_2d_summary.py:163-167, deep instances :183-184,:189-190,:195-196,:201-202);
// Winograd needs 16 products per 2x2 outputs instead of 36: 2.25x fewer MFMAs.  The question is what the transforms cost.
//
//   x [N][H][W][C] fp32 NHWC, w [3][3][C][K] HWIO  ->  y [N][H][W][K]   ('same', zero padding), H, W multiples of 16, C % 16 == 0, K % 64 == 0
//   U[xi] = G g G^T  (16 x [C x K], packed once: fp16 hi/lo pairs, power-of-two scale)          -- pack kernel
//   V[xi] = B^T d B  per 4x4 input tile (stride 2), M[xi] = V[xi] U[xi] (16 GEMMs), Y = A^T M A  -- main kernel
// Main kernel (v2): 512 threads, one workgroup = 8 x 8 tiles (16 x 16 output pixels of one image) x 64 output channels, K loop over 16-channel
// chunks.  Per chunk: the raw 18 x 18 x 16 patch goes global -> registers (one chunk ahead) -> LDS; every thread transforms one
// (tile, channel quad, half of the 4 x 4) in fp32 (adds only), splits into fp16 hi/lo (exact: hi + lo = 22 significant bits) and writes
// V to LDS as [xi][hi|lo][8-channel group][tile] 16-byte slots; wave (i, nb) = row i of the 4 x 4 (xi = 4 i + j) x 32 output channels x
// both 32-tile blocks: 24 x v_mfma_f32_32x32x16_f16 per chunk (hi*hi + hi*lo + lo*hi), A fragments from LDS, B fragments (transformed
// weights) straight from L2 into registers.  Epilogue: row transform in registers, column transform across the 4 waves of a column
// block through LDS, coalesced stores.
// Build + run:  hipcc --offload-arch=gfx950 -O3 scripts/micro/winograd_f16x3.hip -o scripts/micro/winograd_f16x3 -ldl && ./scripts/micro/winograd_f16x3
// (baseline: the product's own dc_conv3x3_fwd_f16x3 -- the persistent role-split kernel igemm_pp_kernel<2,2,0> on these shapes --
//  from deep_calcium_amd/lib/libdcunet.so, same tensors, same box)
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <math.h>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int TT = 8, NT = TT * TT, KB = 64, CKC = 16, RW = 2 * TT + 2, RWP = RW + 1;      // raw rows padded to an ODD slot count
constexpr int NPXP = ((RW * RWP + 15) / 16) * 16 + 1;          // ... and the quad planes to 1 (mod 16) slots: conflict-free transform reads
constexpr int RAW_BYTES = 4 * NPXP * 16;                       // [channel quad][padded px] 16-byte slots: 22 592
constexpr int V_BYTES = 16 * 2 * 2 * NT * 16;                  // one stage: [xi][hi|lo][cg][(tile + 8 cg) & 63] x 16 B = 65 536
constexpr int EX_BYTES = 4 * 2 * 2 * NT * 32 * 4;              // 131 072: [i][j pair][b][tile][32 couts] floats, one column block at a time
constexpr int LDS_BYTES = RAW_BYTES + 2 * V_BYTES;             // 153 664 (the epilogue exchange aliases it)
static_assert(EX_BYTES <= LDS_BYTES && LDS_BYTES <= 160 * 1024, "LDS budget");

// ---- weights: U = G g G^T, scaled by a power of two, split into fp16 hi / lo: Up[xi][c/8][hi|lo][k][8]
__global__ void wino_pack(const float* __restrict__ w, _Float16* __restrict__ up, int C, int K, float scale) {
  const long idx = blockIdx.x * 256L + threadIdx.x;
  if (idx >= (long)C * K) return;
  const int c = (int)(idx / K), k = (int)(idx % K);
  float g[3][3];
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) g[a][b] = w[((a * 3 + b) * (long)C + c) * K + k];
  float t[4][3];
  for (int b = 0; b < 3; ++b) {
    t[0][b] = g[0][b];
    t[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
    t[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
    t[3][b] = g[2][b];
  }
  for (int i = 0; i < 4; ++i) {
    const float u[4] = {t[i][0], 0.5f * (t[i][0] + t[i][1] + t[i][2]), 0.5f * (t[i][0] - t[i][1] + t[i][2]), t[i][2]};
    for (int j = 0; j < 4; ++j) {
      const float v = u[j] * scale;
      const _Float16 hi = (_Float16)v, lo = (_Float16)(v - (float)hi);
      const long slot = ((long)((i * 4 + j) * (C / 8) + c / 8) * 2) * K + k;
      up[slot * 8 + (c & 7)] = hi;
      up[(slot + K) * 8 + (c & 7)] = lo;
    }
  }
}

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
// hi = fp16(x*s), lo = fp16(x*s - hi): two v_fma_mix per element (as csrc/igemm_pp.hip split)
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

// Schedule: V is double-buffered; SCHED 1 / 2: the two waves of a SIMD run the chunk's two phases in OPPOSITE order (which waves share
// a SIMD is not documented: SCHED 1 assumes w and w + 4, SCHED 2 assumes 2s and 2s + 1); SCHED 0: the same order for every wave.
// v5 (this version): consumers register-blocked 2 xi x 2 tile blocks x 2 column blocks (every A fragment feeds two column blocks: half
// the fragment reads of v1-v3), conflict-free LDS layouts (transform lanes = 8 tiles of a row x 2 quads per 16-lane group over
// [quad][px] planes of 1 (mod 16) slots; the two 8-channel groups of V rotated by 8 tiles), epilogue per column block with 16-byte IO.
// ABL (ablation, wrong results on purpose): 1 no transform, 2 no MFMAs, 4 no B-fragment loads, 8 no raw-patch loads, 16 no epilogue
template <int SCHED, int ABL>
__global__ __launch_bounds__(512, 1) void wino_fwd(const float* __restrict__ x, const _Float16* __restrict__ up, float* __restrict__ y,
                                                   int N, int H, int W, int C, int K, float in_scale, float out_scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x4* raw = reinterpret_cast<f32x4*>(smem);                                  // [q][py * RWP + px]
  char* vstage = smem + RAW_BYTES;                                              // 2 x V
  float* exch = reinterpret_cast<float*>(smem);                                 // epilogue (aliases everything)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, h = lane >> 5;
  const int regs_x = W / (2 * TT), regs_y = H / (2 * TT);
  // output-channel block FASTEST: the K / 64 workgroups of one region run side by side and share its input patch in L2
  const int nkb = K / KB, kb = blockIdx.x % nkb, reg = blockIdx.x / nkb;
  const int rx = reg % regs_x, ry = (reg / regs_x) % regs_y, img = reg / (regs_x * regs_y);
  const int oy0 = ry * 2 * TT, ox0 = rx * 2 * TT, k0 = kb * KB;
  const float* ximg = x + (long)img * H * W * C;

  // ---- raw patch loader: 324 px x 4 quads = 1296 float4 items, 3 per thread (the last round partly idle)
  f32x4 rv[3];
  int roff[3], rdst[3];
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    const int it = tid + 512 * s;
    roff[s] = -1; rdst[s] = -1;
    if (it < RW * RW * 4) {
      const int px = it >> 2, q = it & 3, py = px / RW, pxx = px % RW;
      const int iy = oy0 - 1 + py, ix = ox0 - 1 + pxx;
      rdst[s] = q * NPXP + py * RWP + pxx;
      if (iy >= 0 && iy < H && ix >= 0 && ix < W) roff[s] = (iy * W + ix) * C + 4 * q;
    }
  }
  auto raw_load = [&](int c0) {
#pragma unroll
    for (int s = 0; s < 3; ++s) rv[s] = (roff[s] >= 0 && !(ABL & 8)) ? *reinterpret_cast<const f32x4*>(ximg + roff[s] + c0) : f32x4{0.f, 0.f, 0.f, 0.f};
  };
  auto raw_store = [&]() {
#pragma unroll
    for (int s = 0; s < 3; ++s)
      if (rdst[s] >= 0) raw[rdst[s]] = rv[s];
  };
  // ---- transform item of this thread: (tile, channel quad, half of the 4 x 4); a 16-lane group = the 8 tiles of a tile row x 2 quads
  const int ttx = tid & 7, tq = (tid >> 3) & 3, tty = (tid >> 5) & 7, thalf = __builtin_amdgcn_readfirstlane(tid >> 8), ttile = tty * TT + ttx;
  const f32x4* rsrc = raw + tq * NPXP + (2 * tty + thalf) * RWP + 2 * ttx;
  const int vdst = ((tq >> 1) * NT + ((ttile + 8 * (tq >> 1)) & 63)) * 16 + (tq & 1) * 8;
  auto transform = [&](char* vbase) {
    if (ABL & 1) return;
    // rows of B^T d: half 0 -> (d0 - d2, d1 + d2), half 1 -> (d2 - d1, d1 - d3); raw rows needed: half 0: 0,1,2; half 1: 1,2,3
    f32x4 d[3][4];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int cx = 0; cx < 4; ++cx) d[r][cx] = rsrc[r * RWP + cx];
#pragma unroll
    for (int ii = 0; ii < 2; ++ii) {
      f32x4 t[4];
#pragma unroll
      for (int cx = 0; cx < 4; ++cx) {
        if (thalf == 0) t[cx] = ii == 0 ? d[0][cx] - d[2][cx] : d[1][cx] + d[2][cx];
        else            t[cx] = ii == 0 ? d[1][cx] - d[0][cx] : d[0][cx] - d[2][cx];
      }
      const f32x4 v[4] = {t[0] - t[2], t[1] + t[2], t[2] - t[1], t[1] - t[3]};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int xi = (2 * thalf + ii) * 4 + j;
        u32x2 hi, lo;
        split4(v[j], in_scale, hi, lo);
        char* dst = vbase + (xi * 2 * 2 * NT * 16) + vdst;
        *reinterpret_cast<u32x2*>(dst) = hi;
        *reinterpret_cast<u32x2*>(dst + 2 * NT * 16) = lo;
      }
    }
  };
  // ---- consumer role of this wave: row i of the 4 x 4, column pair jp (xi = 4 i + 2 jp + jj), both tile blocks, both column blocks
  const int wi = wave >> 1, wjp = wave & 1;
  f32x16 acc[2][2][2];
#pragma unroll
  for (int jj = 0; jj < 2; ++jj)
#pragma unroll
    for (int tb = 0; tb < 2; ++tb)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[jj][tb][nb][r] = 0.f;
  const int C8 = C / 8;
  f16x8 bf[2][2][2];
  const _Float16* bsrc = up + ((long)((wi * 4 + 2 * wjp) * C8 + h) * 2 * K + k0 + li) * 8;
  auto b_load = [&](int c0) {
    if (ABL & 4) {
      if (c0 == 0)
        for (int jj = 0; jj < 2; ++jj) for (int nb = 0; nb < 2; ++nb) for (int hl = 0; hl < 2; ++hl) for (int e = 0; e < 8; ++e) bf[jj][nb][hl][e] = (_Float16)(0.001f * (lane + e));
      return;
    }
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int hl = 0; hl < 2; ++hl)
          bf[jj][nb][hl] = *reinterpret_cast<const f16x8*>(bsrc + (((long)(jj * C8 + c0 / 8) * 2 + hl) * K + nb * 32) * 8);
  };
  const int aoff0 = (h * NT + ((li + 8 * h) & 63)) * 16, aoff1 = (h * NT + ((32 + li + 8 * h) & 63)) * 16;
  auto mfma_phase = [&](const char* vbase) {
    if (ABL & 2) return;
    __builtin_amdgcn_s_setprio(3);                           // the wave that feeds the matrix pipe goes first
    // all 8 A fragments first, then the three product terms ACROSS the accumulator blocks: consecutive MFMAs never depend on each
    // other, so ONE wave can keep its SIMD's matrix pipe busy (the other wave of the SIMD is transforming)
    f16x8 ah[2][2], al[2][2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int tb = 0; tb < 2; ++tb) {
        const char* ap = vbase + (wi * 4 + 2 * wjp + jj) * (2 * 2 * NT * 16) + (tb ? aoff1 : aoff0);
        ah[jj][tb] = *reinterpret_cast<const f16x8*>(ap);
        al[jj][tb] = *reinterpret_cast<const f16x8*>(ap + 2 * NT * 16);
      }
#pragma unroll
    for (int term = 0; term < 3; ++term)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb)
            acc[jj][tb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(term == 0 ? al[jj][tb] : ah[jj][tb], bf[jj][nb][term == 1 ? 1 : 0],
                                                                     acc[jj][tb][nb], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };

  const int nch = C / CKC;
  const bool mfma_first = SCHED == 0 || (SCHED == 1 ? wave < 4 : (wave & 1) == 0);
  raw_load(0);
  // iteration k: transform chunk k (k < nch) into stage k & 1; multiply chunk k - 1 (k >= 1) from stage (k - 1) & 1
  for (int k = 0; k <= nch; ++k) {
    if (k < nch) raw_store();
    __syncthreads();                                           // raw(k) visible; V stage (k - 1) & 1 complete (barrier below)
    if (k + 1 < nch) raw_load((k + 1) * CKC);
    char* vt = vstage + (k & 1) * V_BYTES;
    const char* vm = vstage + ((k - 1) & 1) * V_BYTES;
    if (mfma_first) {
      if (k >= 1) mfma_phase(vm);
      if (k < nch) { b_load(k * CKC); transform(vt); }
    } else {
      if (k < nch) transform(vt);
      if (k >= 1) mfma_phase(vm);
      if (k < nch) b_load(k * CKC);
    }
    __syncthreads();                                           // V stage k & 1 complete; raw and stage (k - 1) & 1 free
  }
  // ---- output transform.  R[i][b] = sum_j M[i][j] A[j][b]: the wave holding columns (0, 1) contributes (m0 + m1, m1), the one holding
  // (2, 3) contributes (m2, -m2 - m3); Y[a][b] = sum_i A^T[a][i] R[i][b] across the waves through LDS, one 32-column block at a time.
  if (ABL & 16) {
    float sacc = 0.f;
    for (int jj = 0; jj < 2; ++jj) for (int tb = 0; tb < 2; ++tb) for (int nb = 0; nb < 2; ++nb) for (int r = 0; r < 16; ++r) sacc += acc[jj][tb][nb][r];
    if (sacc == 12345.f) y[tid] = sacc;
    return;
  }
  float* yimg = y + (long)img * H * W * K;
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    if (nb) __syncthreads();                                   // the previous block's readers are done
#pragma unroll
    for (int tb = 0; tb < 2; ++tb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int tile = tb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const float m0 = acc[0][tb][nb][r], m1 = acc[1][tb][nb][r];
        const float p0 = wjp == 0 ? m0 + m1 : m0, p1 = wjp == 0 ? m1 : -m0 - m1;
        exch[(((wi * 2 + wjp) * 2 + 0) * NT + tile) * 32 + li] = p0;
        exch[(((wi * 2 + wjp) * 2 + 1) * NT + tile) * 32 + li] = p1;
      }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int idx = tid + 512 * s, c4 = (idx & 7) * 4, b = (idx >> 3) & 1, tile = idx >> 4;
      f32x4 R[4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        R[i] = *reinterpret_cast<const f32x4*>(exch + (((i * 2 + 0) * 2 + b) * NT + tile) * 32 + c4) +
               *reinterpret_cast<const f32x4*>(exch + (((i * 2 + 1) * 2 + b) * NT + tile) * 32 + c4);
      const int oy = oy0 + 2 * (tile / TT), ox = ox0 + 2 * (tile % TT) + b;
      *reinterpret_cast<f32x4*>(yimg + ((long)oy * W + ox) * K + k0 + nb * 32 + c4) = (R[0] + R[1] + R[2]) * out_scale;
      *reinterpret_cast<f32x4*>(yimg + ((long)(oy + 1) * W + ox) * K + k0 + nb * 32 + c4) = (R[1] - R[2] - R[3]) * out_scale;
    }
  }
}

typedef int (*pack_fn)(const float*, void*, int, int, int, long, long, long, int, void*);
typedef long (*floats_fn)(int, int, int);
typedef int (*conv_fn)(const float*, const void*, const float*, float*, long, double*, int, const float*, const float*, int, const float*, long,
                       float*, long, float*, int, int, int, int, int, void*);

int main(int argc, char** argv) {
  struct Shape { int N, HW, C, K; };
  const Shape shapes[] = {{16, 64, 256, 256}, {16, 32, 512, 512}};
  void* lib = dlopen("deep_calcium_amd/lib/libdcunet.so", RTLD_NOW);
  if (!lib) lib = dlopen("../../deep_calcium_amd/lib/libdcunet.so", RTLD_NOW);
  pack_fn dc_pack = lib ? (pack_fn)dlsym(lib, "dc_pack_weights_f16x3") : nullptr;
  floats_fn dc_floats = lib ? (floats_fn)dlsym(lib, "dc_pack_weights_f16x3_floats") : nullptr;
  conv_fn dc_conv = lib ? (conv_fn)dlsym(lib, "dc_conv3x3_fwd_f16x3") : nullptr;
  if (!dc_conv) printf("(libdcunet.so not found: no baseline)\n");
  typedef void (*kern_t)(const float*, const _Float16*, float*, int, int, int, int, int, float, float);
  struct Var { const char* name; kern_t fn; };
  const Var vars[] = {{"v5 SCHED 1", wino_fwd<1, 0>}, {"v5 SCHED 0 (same phase order for all waves)", wino_fwd<0, 0>}, {"v5 SCHED 2 (opposite order for waves 2s / 2s+1)", wino_fwd<2, 0>},
                      {"v2 - transform", wino_fwd<1, 1>}, {"v2 - MFMAs", wino_fwd<1, 2>}, {"v2 - B loads", wino_fwd<1, 4>},
                      {"v2 - raw loads", wino_fwd<1, 8>}, {"v2 - epilogue", wino_fwd<1, 16>}, {"v2 - transform - MFMAs (loads + barriers)", wino_fwd<1, 3>},
                      {"v2 - MFMAs - B - raw (transform alone)", wino_fwd<1, 14>}, {"v2 - transform - B - raw (MFMAs alone)", wino_fwd<1, 13>}};
  for (const Var& v : vars) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(v.fn), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
  for (const Shape& s : shapes) {
    const int N = s.N, H = s.HW, W = s.HW, C = s.C, K = s.K;
    const long nx = (long)N * H * W * C, ny = (long)N * H * W * K, nw = 9L * C * K;
    std::vector<float> hx(nx), hw(nw);
    std::mt19937 rng(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (auto& v : hx) v = nd(rng);
    const float wstd = sqrtf(2.f / (9 * C));
    for (auto& v : hw) v = nd(rng) * wstd;
    float *dx, *dw, *dy, *dyb;
    _Float16* dup;
    CK(hipMalloc(&dx, nx * 4)); CK(hipMalloc(&dw, nw * 4)); CK(hipMalloc(&dy, ny * 4)); CK(hipMalloc(&dyb, ny * 4));
    CK(hipMalloc(&dup, 16L * C * K * 2 * 2));
    CK(hipMemcpy(dx, hx.data(), nx * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw, hw.data(), nw * 4, hipMemcpyHostToDevice));
    // power-of-two scales: max |U| <= max |g| * 9/4 -> bring into [2^9, 2^11); activations of N(0,1): |V| <= 4 * 5.5 -> x 2^10 < 65504
    float wmax = 0.f;
    for (float v : hw) wmax = fmaxf(wmax, fabsf(v));
    const float wscale = exp2f(floorf(log2f(1024.f / (wmax * 2.25f)))), in_scale = 1024.f;
    hipLaunchKernelGGL(wino_pack, dim3((unsigned)(((long)C * K + 255) / 256)), dim3(256), 0, 0, dw, dup, C, K, wscale);
    const int grid = N * (H / 16) * (W / 16) * (K / KB);
    kern_t cur = vars[0].fn;
    auto run = [&]() {
      hipLaunchKernelGGL(cur, dim3(grid), dim3(512), LDS_BYTES, 0, dx, dup, dy, N, H, W, C, K, in_scale, 1.f / (wscale * in_scale));
    };
    run();
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f, sum = 0.f;
    const int reps = 20;
    for (int r = 0; r < reps; ++r) {
      CK(hipEventRecord(e0, 0)); run(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = fminf(best, ms); sum += ms;
    }
    // baseline: the product's conv3x3 forward (igemm_pp_kernel<2,2,0>) on the same tensors
    float bbest = -1.f, bsum = 0.f;
    if (dc_conv) {
      void* wp; CK(hipMalloc(&wp, dc_floats(9, C, K) * 4));
      dc_pack(dw, wp, 9, C, K, (long)C * K, K, 1, 0, nullptr);
      dc_conv(dx, wp, nullptr, dyb, K, nullptr, 0, nullptr, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, N, H, W, C, K, nullptr);
      CK(hipDeviceSynchronize());
      bbest = 1e9f;
      for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0, 0));
        dc_conv(dx, wp, nullptr, dyb, K, nullptr, 0, nullptr, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, N, H, W, C, K, nullptr);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); bbest = fminf(bbest, ms); bsum += ms;
      }
      CK(hipFree(wp));
    }
    // accuracy: float64 direct convolution at 4000 random output points + all of image 0's first 16 x 16 block; relative to max |y|
    std::vector<float> hy(ny), hyb(ny);
    CK(hipMemcpy(hy.data(), dy, ny * 4, hipMemcpyDeviceToHost));
    if (dc_conv) CK(hipMemcpy(hyb.data(), dyb, ny * 4, hipMemcpyDeviceToHost));
    double ymax = 0.0, ew = 0.0, eb = 0.0;
    auto ref_at = [&](int n, int oy, int ox, int k) {
      double a = 0.0;
      for (int dy_ = 0; dy_ < 3; ++dy_)
        for (int dx_ = 0; dx_ < 3; ++dx_) {
          const int iy = oy + dy_ - 1, ix = ox + dx_ - 1;
          if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
          const float* xp = &hx[(((long)n * H + iy) * W + ix) * C];
          const float* wp = &hw[((dy_ * 3 + dx_) * (long)C) * K + k];
          for (int c = 0; c < C; ++c) a += (double)xp[c] * (double)wp[(long)c * K];
        }
      return a;
    };
    std::uniform_int_distribution<long> pick(0, ny - 1);
    std::vector<long> pts;
    for (int i = 0; i < 4000; ++i) pts.push_back(pick(rng));
    for (int oy = 0; oy < 16; ++oy) for (int ox = 0; ox < 16; ++ox) for (int k = 0; k < K; k += 7) pts.push_back(((long)oy * W + ox) * K + k);
    for (int b = 0; b < 2; ++b) for (int k = 0; k < K; k += 5) pts.push_back(((((long)(N - 1) * H + H - 1) * W) + W - 1 - b) * K + k);      // last corner
    for (long p : pts) {
      const int k = (int)(p % K); long r = p / K; const int ox = (int)(r % W); r /= W; const int oy = (int)(r % H); const int n = (int)(r / H);
      const double ref = ref_at(n, oy, ox, k);
      ymax = fmax(ymax, fabs(ref)); ew = fmax(ew, fabs(hy[p] - ref)); if (dc_conv) eb = fmax(eb, fabs(hyb[p] - ref));
    }
    double dmax = 0.0, allmax = 0.0;
    if (dc_conv) for (long i = 0; i < ny; ++i) { dmax = fmax(dmax, fabs((double)hy[i] - hyb[i])); allmax = fmax(allmax, fabs((double)hyb[i])); }
    const double gf = 2.0 * 9 * C * K * (double)N * H * W * 1e-9;
    printf("%d x %d^2 x %d -> %d  (%.1f GFLOP direct):  winograd %.1f us best / %.1f mean = %.0f TF/s direct-equivalent | baseline %.1f us best / %.1f mean = %.0f TF/s | speed-up %.2fx\n",
           N, H, C, K, gf, best * 1e3, sum / reps * 1e3, gf / best, bbest * 1e3, bsum / reps * 1e3, gf / bbest, bbest / best);
    printf("    max rel error vs float64 at %zu points: winograd %.2e, baseline %.2e (bar 2e-5); winograd vs baseline over the whole tensor: %.2e\n",
           pts.size(), ew / ymax, eb / ymax, dc_conv ? dmax / allmax : 0.0);
    for (size_t vi = 1; vi < sizeof(vars) / sizeof(vars[0]); ++vi) {          // schedule / ablation variants (ablations compute garbage)
      cur = vars[vi].fn;
      run(); CK(hipDeviceSynchronize());
      float vb = 1e9f;
      for (int r = 0; r < 10; ++r) {
        CK(hipEventRecord(e0, 0)); run(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); vb = fminf(vb, ms);
      }
      printf("      %-48s %.1f us\n", vars[vi].name, vb * 1e3);
    }
    CK(hipFree(dx)); CK(hipFree(dw)); CK(hipFree(dy)); CK(hipFree(dyb)); CK(hipFree(dup));
  }
  return 0;
}
