// Winograd F(2x2, 3x3) conv3x3 forward in the TRANSFORMED domain on the split-fp16 matrix path -- round-5 review item 3:
// "measure it, with a gate".  3x3 convolutions are 95.5 % of the network's FLOPs (Conv2D(nf, (3,3)),
// /root/reference/deepcalcium/models/neurons/unet_2d_summary.py:163-167, deep instances :183-184,:189-190,:195-196,:201-202);
// Winograd needs 16 products per 2x2 outputs instead of 36: 2.25x fewer MFMAs.  The question is what the transforms cost.
//
//   x [N][H][W][C] fp32 NHWC, w [3][3][C][K] HWIO  ->  y [N][H][W][K]   ('same', zero padding), H, W multiples of 16, C % 16 == 0, K % 64 == 0
//   U[xi] = G g G^T  (16 x [C x K], packed once: fp16 hi/lo pairs, power-of-two scale)          -- pack kernel
//   V[xi] = B^T d B  per 4x4 input tile (stride 2), M[xi] = V[xi] U[xi] (16 GEMMs), Y = A^T M A  -- main kernel
// Main kernel: 512 threads, one workgroup = 8 x 8 tiles (16 x 16 output pixels of one image) x 64 output channels, K loop over 16-channel
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

constexpr int TT = 8, NT = TT * TT, KB = 64, CKC = 16, RW = 2 * TT + 2, NPX = RW * RW;
constexpr int RAW_BYTES = NPX * CKC * 4;                       // 20 736
constexpr int V_BYTES = 16 * 2 * 2 * NT * 16;                  // 65 536: [xi][hi|lo][cg][tile] x 16 B
constexpr int EX_BYTES = 4 * 2 * NT * KB * 4;                  // 131 072: [i][b][tile][cout] floats
constexpr int LDS_BYTES = EX_BYTES > RAW_BYTES + V_BYTES ? EX_BYTES : RAW_BYTES + V_BYTES;

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

__device__ __forceinline__ void split4(const f32x4 v, float s, f16x4& hi, f16x4& lo) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float x = v[e] * s;
    const _Float16 h = (_Float16)x;
    hi[e] = h;
    lo[e] = (_Float16)(x - (float)h);
  }
}

__global__ __launch_bounds__(512, 1) void wino_fwd(const float* __restrict__ x, const _Float16* __restrict__ up, float* __restrict__ y,
                                                   int N, int H, int W, int C, int K, float in_scale, float out_scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* raw = reinterpret_cast<float*>(smem);                                  // [px][16]
  char* vbase = smem + RAW_BYTES;                                               // V slots
  float* exch = reinterpret_cast<float*>(smem);                                 // epilogue (aliases both)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, h = lane >> 5;
  const int regs_x = W / (2 * TT), regs_y = H / (2 * TT), regs = N * regs_y * regs_x;
  const int kb = blockIdx.x / regs, reg = blockIdx.x % regs;                    // output-channel block slowest: co-running workgroups share U
  const int rx = reg % regs_x, ry = (reg / regs_x) % regs_y, img = reg / (regs_x * regs_y);
  const int oy0 = ry * 2 * TT, ox0 = rx * 2 * TT, k0 = kb * KB;
  const float* ximg = x + (long)img * H * W * C;

  // ---- raw patch loader: 324 px x 4 float4 = 1296 items, 3 per thread (the last round partly idle)
  f32x4 rv[3];
  auto raw_load = [&](int c0) {
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const int it = tid + 512 * s;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (it < NPX * 4) {
        const int px = it >> 2, f4 = it & 3, py = px / RW, pxx = px % RW;
        const int iy = oy0 - 1 + py, ix = ox0 - 1 + pxx;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = *reinterpret_cast<const f32x4*>(ximg + ((long)iy * W + ix) * C + c0 + 4 * f4);
      }
      rv[s] = v;
    }
  };
  auto raw_store = [&]() {
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      const int it = tid + 512 * s;
      if (it < NPX * 4) *reinterpret_cast<f32x4*>(raw + it * 4) = rv[s];
    }
  };
  // ---- transform item of this thread: (tile, channel quad, half)
  const int tq = tid & 3, thalf = (tid >> 2) & 1, ttile = tid >> 3, tty = ttile / TT, ttx = ttile % TT;
  auto transform = [&]() {
    // rows of B^T d: half 0 -> (d0 - d2, d1 + d2), half 1 -> (d2 - d1, d1 - d3); raw rows needed: half 0: 0,1,2; half 1: 1,2,3
    f32x4 d[3][4];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int cx = 0; cx < 4; ++cx)
        d[r][cx] = *reinterpret_cast<const f32x4*>(raw + (((2 * tty + thalf + r) * RW) + 2 * ttx + cx) * CKC + 4 * tq);
    f32x4 t[2][4];
#pragma unroll
    for (int cx = 0; cx < 4; ++cx) {
      if (thalf == 0) { t[0][cx] = d[0][cx] - d[2][cx]; t[1][cx] = d[1][cx] + d[2][cx]; }
      else            { t[0][cx] = d[1][cx] - d[0][cx]; t[1][cx] = d[0][cx] - d[2][cx]; }
    }
#pragma unroll
    for (int ii = 0; ii < 2; ++ii) {
      const f32x4 v[4] = {t[ii][0] - t[ii][2], t[ii][1] + t[ii][2], t[ii][2] - t[ii][1], t[ii][1] - t[ii][3]};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int xi = (2 * thalf + ii) * 4 + j;
        f16x4 hi, lo;
        split4(v[j], in_scale, hi, lo);
        char* dst = vbase + ((((xi * 2 + 0) * 2 + (tq >> 1)) * NT + ttile) * 16) + (tq & 1) * 8;
        *reinterpret_cast<f16x4*>(dst) = hi;
        *reinterpret_cast<f16x4*>(dst + 2 * NT * 16) = lo;
      }
    }
  };
  // ---- consumer role of this wave: row i of the 4 x 4, output-channel block nb
  const int wi = wave & 3, wnb = wave >> 2;
  f32x16 acc[4][2];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int tb = 0; tb < 2; ++tb)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][tb][r] = 0.f;
  const int C8 = C / 8;
  f16x8 bf[4][2];
  auto b_load = [&](int c0) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int hl = 0; hl < 2; ++hl) {
        const long slot = ((long)(((wi * 4 + j) * C8 + c0 / 8 + h) * 2 + hl)) * K + k0 + wnb * 32 + li;
        bf[j][hl] = *reinterpret_cast<const f16x8*>(up + slot * 8);
      }
  };

  const int nch = C / CKC;
  raw_load(0);
  for (int ch = 0; ch < nch; ++ch) {
    raw_store();
    b_load(ch * CKC);                                          // in flight across the transform
    if (ch + 1 < nch) raw_load((ch + 1) * CKC);
    __syncthreads();
    transform();
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int xi = wi * 4 + j;
#pragma unroll
      for (int tb = 0; tb < 2; ++tb) {
        const f16x8 ah = *reinterpret_cast<const f16x8*>(vbase + ((((xi * 2 + 0) * 2 + h) * NT + tb * 32 + li) * 16));
        const f16x8 al = *reinterpret_cast<const f16x8*>(vbase + ((((xi * 2 + 1) * 2 + h) * NT + tb * 32 + li) * 16));
        acc[j][tb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bf[j][0], acc[j][tb], 0, 0, 0);
        acc[j][tb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bf[j][1], acc[j][tb], 0, 0, 0);
        acc[j][tb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bf[j][0], acc[j][tb], 0, 0, 0);
      }
    }
  }
  __syncthreads();                                             // every wave is done reading V: the exchange area may overwrite it
  // ---- output transform: R[i][b] = sum_j M[i][j] A[j][b] in registers, Y[a][b] = sum_i A^T[a][i] R[i][b] through LDS
#pragma unroll
  for (int tb = 0; tb < 2; ++tb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int tile = tb * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      const float r0 = acc[0][tb][r] + acc[1][tb][r] + acc[2][tb][r];
      const float r1 = acc[1][tb][r] - acc[2][tb][r] - acc[3][tb][r];
      exch[((wi * 2 + 0) * NT + tile) * KB + wnb * 32 + li] = r0;
      exch[((wi * 2 + 1) * NT + tile) * KB + wnb * 32 + li] = r1;
    }
  __syncthreads();
  float* yimg = y + (long)img * H * W * K;
#pragma unroll 4
  for (int s = 0; s < 16; ++s) {
    const int idx = tid + 512 * s, co = idx & 63, b = (idx >> 6) & 1, tile = idx >> 7;
    const float q0 = exch[((0 * 2 + b) * NT + tile) * KB + co], q1 = exch[((1 * 2 + b) * NT + tile) * KB + co];
    const float q2 = exch[((2 * 2 + b) * NT + tile) * KB + co], q3 = exch[((3 * 2 + b) * NT + tile) * KB + co];
    const int oy = oy0 + 2 * (tile / TT), ox = ox0 + 2 * (tile % TT) + b;
    yimg[((long)oy * W + ox) * K + k0 + co] = (q0 + q1 + q2) * out_scale;
    yimg[((long)(oy + 1) * W + ox) * K + k0 + co] = (q1 - q2 - q3) * out_scale;
  }
}

typedef int (*pack_fn)(const float*, void*, int, int, int, long, long, long, int, void*);
typedef long (*floats_fn)(int, int, int);
typedef int (*conv_fn)(const float*, const void*, const float*, float*, long, double*, int, const float*, const float*, int, const float*, long,
                       float*, long, float*, int, int, int, int, int, void*);

int main(int argc, char** argv) {
  struct Shape { int N, HW, C, K; };
  const Shape shapes[] = {{16, 64, 256, 256}, {16, 32, 512, 512}, {16, 128, 128, 128}};
  void* lib = dlopen("deep_calcium_amd/lib/libdcunet.so", RTLD_NOW);
  if (!lib) lib = dlopen("../../deep_calcium_amd/lib/libdcunet.so", RTLD_NOW);
  pack_fn dc_pack = lib ? (pack_fn)dlsym(lib, "dc_pack_weights_f16x3") : nullptr;
  floats_fn dc_floats = lib ? (floats_fn)dlsym(lib, "dc_pack_weights_f16x3_floats") : nullptr;
  conv_fn dc_conv = lib ? (conv_fn)dlsym(lib, "dc_conv3x3_fwd_f16x3") : nullptr;
  if (!dc_conv) printf("(libdcunet.so not found: no baseline)\n");
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(wino_fwd), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
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
    auto run = [&]() {
      hipLaunchKernelGGL(wino_fwd, dim3(grid), dim3(512), LDS_BYTES, 0, dx, dup, dy, N, H, W, C, K, in_scale, 1.f / (wscale * in_scale));
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
    CK(hipFree(dx)); CK(hipFree(dw)); CK(hipFree(dy)); CK(hipFree(dyb)); CK(hipFree(dup));
  }
  return 0;
}
