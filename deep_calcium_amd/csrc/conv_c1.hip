// First UNet2DS layer: Conv2D(nfb,(3,3),'same') on the 1-channel image
// (/root/reference/deepcalcium/models/neurons/unet_2d_summary.py:170-172: expand_dims + conv_layer(nfb)).
// K = 9: no matrix shape to speak of and 4.4 flop/byte -- an HBM-bound streaming kernel (the 128-B output
// row per pixel dominates), written on the vector ALU: Cout/4 lanes per pixel, float4 stores so one
// pixel's channels form one contiguous segment, 3x3 taps + bias held in registers.
#include "common.h"

struct C1Params {
  const float* x;
  const float* w;  // HWIO (3,3,1,Cout)
  const float* bias;
  float* z;
  double* stats;
  const float* scale;
  const float* shift;
  float* absmax;   // nullable: per-channel max |output| (atomic max), the next layer's fp16 range-guard bound
  long absmaxLd;   // floats between its DC_ABOUND_SLOTS replicas (common.h)
  int N, H, W, Cout, relu;
  long pixels, zLd;
};

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// Per-thread BatchNorm partials as SHIFTED sums around the thread's first value (common.h DcMoments), merged over the
// block's pixel lanes with Chan's update and stored as doubles (sum v, sum v^2) per (block, channel).
struct C1Acc {
  f32x4 K, s1, s2, amax;
  float n;
  __device__ __forceinline__ void init() {
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    K = z4; s1 = z4; s2 = z4; amax = z4; n = 0.f;
  }
  __device__ __forceinline__ void add(const f32x4& v) {
    if (n == 0.f) K = v;
    const f32x4 d = v - K;
    s1 += d;
    s2 += d * d;
    n += 1.f;
  }
  __device__ __forceinline__ void add_shifted(const f32x4& v) {      // K was set up front
    const f32x4 d = v - K;
    s1 += d;
    s2 += d * d;
    n += 1.f;
  }
  __device__ __forceinline__ void track(const f32x4& v) {
#pragma unroll
    for (int e = 0; e < 4; ++e) amax[e] = fmaxf(amax[e], fabsf(v[e]));
  }
};
__device__ __forceinline__ void c1_finish(const C1Params& p, const C1Acc& a, int tid, int q, int pl, int C4, int PPB) {
  __shared__ DcMoments sm[4][256];
  if (p.absmax && p.absmaxLd < 0) {      // optimistic inference: flag an output beyond fp16's range, nothing else
    const float m = fmaxf(fmaxf(a.amax[0], a.amax[1]), fmaxf(a.amax[2], a.amax[3]));
    if (!(m <= DC_F16_SAFE_MAX)) p.absmax[0] = 1.f;
  } else if (p.absmax) {
    // pixel lanes of one channel quad sit C4 threads apart: fold them through LDS, one atomic per channel per block
    float* fm = reinterpret_cast<float*>(&sm[0][0]);
#pragma unroll
    for (int e = 0; e < 4; ++e) fm[e * 256 + tid] = a.amax[e];
    __syncthreads();
    if (pl == 0) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float m = a.amax[e];
        for (int k = 1; k < PPB; ++k) m = fmaxf(m, fm[e * 256 + k * C4 + q]);
        dc_atomic_absmax(p.absmax + dc_absmax_slot(p.absmaxLd) + 4 * q + e, m);
      }
    }
    __syncthreads();
  }
  if (p.stats) {
#pragma unroll
    for (int e = 0; e < 4; ++e) sm[e][tid] = dc_moments_from_shifted(a.n, a.K[e], a.s1[e], a.s2[e]);
    __syncthreads();
    if (pl == 0) {
      double* dst = p.stats + ((long)blockIdx.x * p.Cout + 4 * q) * 2;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        DcMoments m = sm[e][q];
        for (int k = 1; k < PPB; ++k) m = dc_moments_merge(m, sm[e][k * C4 + q]);
        dc_moments_store(dst + 2 * e, m);
      }
    }
  }
}

__global__ __launch_bounds__(256) void conv_c1_fwd_kernel(C1Params p) {
  const int C4 = p.Cout >> 2, PPB = 256 / C4;
  const int tid = threadIdx.x, q = tid % C4, pl = tid / C4;
  f32x4 w[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) w[t] = ld4(p.w + t * p.Cout + 4 * q);
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f}, one4 = {1.f, 1.f, 1.f, 1.f};
  const f32x4 b = p.bias ? ld4(p.bias + 4 * q) : z4;
  const f32x4 sc = p.scale ? ld4(p.scale + 4 * q) : one4;
  const f32x4 sh = p.shift ? ld4(p.shift + 4 * q) : z4;
  C1Acc a;
  a.init();
  const long HW = (long)p.H * p.W;
  for (long pix = (long)blockIdx.x * PPB + pl; pix < p.pixels; pix += (long)gridDim.x * PPB) {
    const long img = pix / HW;
    const int rem = (int)(pix - img * HW);
    const int y = rem / p.W, x = rem - y * p.W;
    const float* xi = p.x + img * HW;
    f32x4 v = b;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int yy = y + dy - 1;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int xx = x + dx - 1;
        const float xv = (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) ? xi[yy * p.W + xx] : 0.f;
        v += xv * w[dy * 3 + dx];
      }
    }
    a.add(v);
    if (p.scale) v = v * sc + sh;
    if (p.relu) {
      v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
    }
    if (p.absmax) a.track(v);
    *reinterpret_cast<f32x4*>(p.z + pix * p.zLd + 4 * q) = v;
  }
  c1_finish(p, a, tid, q, pl, C4, PPB);
}

// "dz on load" (dcunet.h, dc_bn_bwd_finalize_dzin): dz = fmaf(A, [fmaf(z,sc,sh) > 0] * da, fmaf(D, z - mu, E)), this
// thread's channel quad of the table dz_coef[7][Cout] in registers
struct C1Dz {
  f32x4 sc, sh, mu, A, D, E;
  __device__ __forceinline__ void load(const float* __restrict__ coef, int Cout, int q) {
    sc = ld4(coef + 4 * q); sh = ld4(coef + Cout + 4 * q); mu = ld4(coef + 2 * Cout + 4 * q);
    A = ld4(coef + 3 * Cout + 4 * q); D = ld4(coef + 4 * Cout + 4 * q); E = ld4(coef + 5 * Cout + 4 * q);
  }
  __device__ __forceinline__ f32x4 dz(const f32x4& da, const f32x4& z) const {
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float y = __builtin_fmaf(z[e], sc[e], sh[e]);
      const float dy = y > 0.f ? da[e] : 0.f;
      v[e] = __builtin_fmaf(A[e], dy, __builtin_fmaf(D[e], z[e] - mu[e], E[e]));
    }
    return v;
  }
};

// dW[tap][co] partial per block = sum_p x[p+tap] * dz[p][co]   (DZIN: `dz` holds da, dz is formed from (da, zt, coef))
template <bool DZIN>
__global__ __launch_bounds__(256) void conv_c1_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dz,
                                                            float* __restrict__ partial, int N, int H, int W, int Cout,
                                                            long pixels, const float* __restrict__ zt,
                                                            const float* __restrict__ coef) {
  __shared__ f32x4 sm[256];
  const int C4 = Cout >> 2, PPB = 256 / C4;
  const int tid = threadIdx.x, q = tid % C4, pl = tid / C4;
  C1Dz cz;
  if constexpr (DZIN) cz.load(coef, Cout, q);
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = z4;
  const long HW = (long)H * W;
  for (long pix = (long)blockIdx.x * PPB + pl; pix < pixels; pix += (long)gridDim.x * PPB) {
    const long img = pix / HW;
    const int rem = (int)(pix - img * HW);
    const int y = rem / W, xq = rem - y * W;
    const float* xi = x + img * HW;
    f32x4 g = ld4(dz + pix * Cout + 4 * q);
    if constexpr (DZIN) g = cz.dz(g, ld4(zt + pix * Cout + 4 * q));
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int yy = y + dy - 1;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const int xx = xq + dx - 1;
        const float xv = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? xi[yy * W + xx] : 0.f;
        acc[dy * 3 + dx] += xv * g;
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    sm[tid] = acc[t];
    __syncthreads();
    if (pl == 0) {
      f32x4 s = acc[t];
      for (int k = 1; k < PPB; ++k) s += sm[k * C4 + q];
      *reinterpret_cast<f32x4*>(partial + ((long)blockIdx.x * 9 + t) * Cout + 4 * q) = s;
    }
    __syncthreads();
  }
}

// W % 4 == 0 (every size this network uses): a thread owns FOUR consecutive pixels of a row for its channel quad.  The
// 3 x 6 input window is read once (one 16-byte + two 4-byte loads per row) and slides across the 4 pixels, and the
// pixel -> (image, row, column) division is paid once per 4 pixels: ~4x fewer instructions per output than the generic
// kernel above, which was VALU-bound at 2.5 TB/s.
__device__ __forceinline__ void c1_window(const float* __restrict__ xi, int y, int x0, int H, int W, float (&win)[3][6]) {
#pragma unroll
  for (int dy = 0; dy < 3; ++dy) {
    const int yy = y + dy - 1;
    const bool rok = yy >= 0 && yy < H;
    const float* row = xi + (long)yy * W + x0;
    const f32x4 m = rok ? *reinterpret_cast<const f32x4*>(row) : f32x4{0.f, 0.f, 0.f, 0.f};
    win[dy][0] = (rok && x0 > 0) ? row[-1] : 0.f;
    win[dy][1] = m[0]; win[dy][2] = m[1]; win[dy][3] = m[2]; win[dy][4] = m[3];
    win[dy][5] = (rok && x0 + 4 < W) ? row[4] : 0.f;
  }
}

__global__ __launch_bounds__(256) void conv_c1_fwd4_kernel(C1Params p) {
  const int C4 = p.Cout >> 2, PPB = 256 / C4;
  const int tid = threadIdx.x, q = tid % C4, pl = tid / C4;
  f32x4 w[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) w[t] = ld4(p.w + t * p.Cout + 4 * q);
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f}, one4 = {1.f, 1.f, 1.f, 1.f};
  const f32x4 b = p.bias ? ld4(p.bias + 4 * q) : z4;
  const f32x4 sc = p.scale ? ld4(p.scale + 4 * q) : one4;
  const f32x4 sh = p.shift ? ld4(p.shift + 4 * q) : z4;
  C1Acc a;
  a.init();
  const int W4 = p.W >> 2, HW4 = p.H * W4;
  const int groups = (int)(p.pixels >> 2);
  // shift of the BatchNorm partial sums = the bias + the thread's first window centre times the centre tap: one fma,
  // no per-pixel select (any finite shift is valid; this one tracks a DC offset of the image as well as a large bias)
  {
    const int g0 = blockIdx.x * PPB + pl;
    if (g0 < groups) {
      const int img = g0 / HW4, rem = g0 - img * HW4;
      const int y = rem / W4, x0 = (rem - y * W4) * 4;
      const float xc = p.x[((long)img * p.H + y) * p.W + x0];
      float ws[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) ws[e] += w[t][e];
#pragma unroll
      for (int e = 0; e < 4; ++e) a.K[e] = __builtin_fmaf(xc, ws[e], b[e]);
    }
  }
  // The next group's window is loaded BEFORE this group's stores are issued: loads and stores retire in order through one
  // counter on this part, so a window load issued behind the stores waits for their write acknowledgements as well (the
  // pure-write sweep ran at 3.4 TB/s that way).
  float nxt[3][6];
  {
    const int g = blockIdx.x * PPB + pl;
    if (g < groups) {
      const int img = g / HW4, rem = g - img * HW4;
      const int y = rem / W4;
      c1_window(p.x + (long)img * p.H * p.W, y, (rem - y * W4) * 4, p.H, p.W, nxt);
    }
  }
  for (int g = blockIdx.x * PPB + pl; g < groups; g += gridDim.x * PPB) {
    float win[3][6];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 6; ++dx) win[dy][dx] = nxt[dy][dx];
    {
      const int gn = g + gridDim.x * PPB;
      if (gn < groups) {
        const int img = gn / HW4, rem = gn - img * HW4;
        const int y = rem / W4;
        c1_window(p.x + (long)img * p.H * p.W, y, (rem - y * W4) * 4, p.H, p.W, nxt);
      }
    }
    const long pix0 = (long)g * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f32x4 v = b;
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) v += win[dy][i + dx] * w[dy * 3 + dx];
      a.add_shifted(v);
      if (p.scale) v = v * sc + sh;
      if (p.relu) {
        v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
      }
      if (p.absmax) a.track(v);
      *reinterpret_cast<f32x4*>(p.z + (pix0 + i) * p.zLd + 4 * q) = v;
    }
  }
  c1_finish(p, a, tid, q, pl, C4, PPB);
}

template <bool DZIN>
__global__ __launch_bounds__(256) void conv_c1_wgrad4_kernel(const float* __restrict__ x, const float* __restrict__ dz,
                                                             float* __restrict__ partial, int N, int H, int W, int Cout,
                                                             long pixels, const float* __restrict__ zt,
                                                             const float* __restrict__ coef) {
  __shared__ f32x4 sm[256];
  const int C4 = Cout >> 2, PPB = 256 / C4;
  const int tid = threadIdx.x, q = tid % C4, pl = tid / C4;
  C1Dz cz;
  if constexpr (DZIN) cz.load(coef, Cout, q);
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = z4;
  const int W4 = W >> 2, HW4 = H * W4;
  const int groups = (int)(pixels >> 2);
  for (int g = blockIdx.x * PPB + pl; g < groups; g += gridDim.x * PPB) {
    const int img = g / HW4, rem = g - img * HW4;
    const int y = rem / W4, x0 = (rem - y * W4) * 4;
    float win[3][6];
    c1_window(x + (long)img * H * W, y, x0, H, W, win);
    const long pix0 = (long)g * 4;
    f32x4 gz[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) gz[i] = ld4(dz + (pix0 + i) * Cout + 4 * q);
    if constexpr (DZIN) {
      f32x4 zz[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) zz[i] = ld4(zt + (pix0 + i) * Cout + 4 * q);
#pragma unroll
      for (int i = 0; i < 4; ++i) gz[i] = cz.dz(gz[i], zz[i]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) acc[dy * 3 + dx] += win[dy][i + dx] * gz[i];
  }
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    sm[tid] = acc[t];
    __syncthreads();
    if (pl == 0) {
      f32x4 s = acc[t];
      for (int k = 1; k < PPB; ++k) s += sm[k * C4 + q];
      *reinterpret_cast<f32x4*>(partial + ((long)blockIdx.x * 9 + t) * Cout + 4 * q) = s;
    }
    __syncthreads();
  }
}

static int c1_blocks(long pixels, int Cout) {
  const int PPB = 256 / (Cout / 4);
  long b = (pixels + PPB - 1) / PPB;
  return (int)(b > 2048 ? 2048 : b);
}

extern "C" int dc_conv3x3_c1_tiles(int N, int H, int W, int Cout) { return c1_blocks((long)N * H * W, Cout); }

static int check_c1(const char* fn, int N, int H, int W, int Cout) {
  DC_REQUIRE(N > 0 && H > 0 && W > 0, DC_EINVAL, "%s: non-positive dimension", fn);
  DC_REQUIRE(Cout >= 4 && Cout <= 1024 && dc_is_pow2(Cout), DC_EUNSUP, "%s: Cout=%d must be a power of two in [4,1024]", fn, Cout);
  return DC_OK;
}

extern "C" int dc_conv3x3_c1_fwd(const float* x, const float* w, const float* bias, float* z, long z_ld, double* stats,
                                 const float* scale, const float* shift, int relu, float* out_absmax, long out_absmax_ld,
                                 int N, int H, int W, int Cout, dc_stream_t stream) {
  DC_REQUIRE(x && w && z, DC_EINVAL, "dc_conv3x3_c1_fwd: null pointer");
  DC_REQUIRE(dc_aligned16(w) && dc_aligned16(z), DC_EINVAL, "dc_conv3x3_c1_fwd: w and z must be 16-byte aligned");
  DC_REQUIRE((scale == nullptr) == (shift == nullptr), DC_EINVAL, "dc_conv3x3_c1_fwd: scale and shift go together");
  int rc = check_c1("dc_conv3x3_c1_fwd", N, H, W, Cout);
  if (rc) return rc;
  C1Params p;
  p.x = x; p.w = w; p.bias = bias; p.z = z; p.stats = stats; p.scale = scale; p.shift = shift; p.absmax = out_absmax; p.absmaxLd = out_absmax_ld;
  p.N = N; p.H = H; p.W = W; p.Cout = Cout; p.relu = relu; p.pixels = (long)N * H * W; p.zLd = z_ld;
  DC_REQUIRE(z_ld >= Cout && z_ld % 4 == 0, DC_EINVAL, "dc_conv3x3_c1_fwd: bad z_ld");
  // same grid (= the BN-partial row count dc_conv3x3_c1_tiles reports) for both kernels
  const bool fast = (W % 4 == 0) && dc_aligned16(x) && p.pixels < (1L << 31);
  hipLaunchKernelGGL(fast ? conv_c1_fwd4_kernel : conv_c1_fwd_kernel, dim3(c1_blocks(p.pixels, Cout)), dim3(256), 0,
                     (hipStream_t)stream, p);
  DC_CHECK_LAUNCH("dc_conv3x3_c1_fwd");
  return DC_OK;
}

long dc_conv3x3_c1_wgrad_ws(int N, int H, int W, int Cout) {
  const long L = 9L * Cout;
  return (long)c1_blocks((long)N * H * W, Cout) * L + 32 * L;
}

int dc_conv3x3_c1_wgrad(const float* x, const float* dz, float* dw, float* ws, int N, int H, int W, int Cout,
                        hipStream_t st) {
  int rc = check_c1("dc_conv3x3_wgrad(Cin=1)", N, H, W, Cout);
  if (rc) return rc;
  const long pixels = (long)N * H * W;
  const int blocks = c1_blocks(pixels, Cout);
  const bool fast = (W % 4 == 0) && dc_aligned16(x) && pixels < (1L << 31);
  const float* none = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  const bool bracket = dc_take_bracket(&ev0, &ev1);          // dc_bracket_next_launch
  if (bracket && ev0) (void)hipEventRecord(ev0, st);
  hipLaunchKernelGGL(fast ? conv_c1_wgrad4_kernel<false> : conv_c1_wgrad_kernel<false>, dim3(blocks), dim3(256), 0, st, x, dz,
                     ws, N, H, W, Cout, pixels, none, none);
  if (bracket && ev1) (void)hipEventRecord(ev1, st);
  DC_CHECK_LAUNCH("dc_conv3x3_wgrad(Cin=1)");
  const long L = 9L * Cout;
  return dc_reduce_partials(ws, blocks, L, 1.0f, dw, ws + (long)blocks * L, (dc_stream_t)st);
}

// first layer's weight gradient with dz formed on load from (da, z, dz_coef)  (dc_conv3x3_wgrad_dzin_f16x3, Cin == 1)
int dc_conv3x3_c1_wgrad_dzin(const float* x, const float* da, const float* z, const float* dz_coef, float* dw, float* ws,
                             int N, int H, int W, int Cout, hipStream_t st) {
  int rc = check_c1("dc_conv3x3_wgrad_dzin(Cin=1)", N, H, W, Cout);
  if (rc) return rc;
  const long pixels = (long)N * H * W;
  const int blocks = c1_blocks(pixels, Cout);
  const bool fast = (W % 4 == 0) && dc_aligned16(x) && pixels < (1L << 31);
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  const bool bracket = dc_take_bracket(&ev0, &ev1);          // dc_bracket_next_launch
  if (bracket && ev0) (void)hipEventRecord(ev0, st);
  hipLaunchKernelGGL(fast ? conv_c1_wgrad4_kernel<true> : conv_c1_wgrad_kernel<true>, dim3(blocks), dim3(256), 0, st, x, da,
                     ws, N, H, W, Cout, pixels, z, dz_coef);
  if (bracket && ev1) (void)hipEventRecord(ev1, st);
  DC_CHECK_LAUNCH("dc_conv3x3_wgrad_dzin(Cin=1)");
  const long L = 9L * Cout;
  return dc_reduce_partials(ws, blocks, L, 1.0f, dw, ws + (long)blocks * L, (dc_stream_t)st);
}
