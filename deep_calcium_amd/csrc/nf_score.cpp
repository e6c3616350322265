// Host-side Neurofinder scoring of a predicted mask (no device code): what _ValidationMetricsCB runs 6 x n times per epoch
// (/root/reference/deepcalcium/models/neurons/unet_2d_summary.py:90-91 -> datasets/nf.py:153-174 -> neurofinder.centers /
// neurofinder.shapes).  deep_calcium_amd/nf_metrics.py holds the literal Python restatement (region lists, sets of
// coordinate tuples: ~6 ms per 128 x 512 stripe with 60 regions, and it holds the GIL); this is the same arithmetic in one
// native call (~0.1 ms, GIL released by ctypes, so a few scoring threads run beside the GPU's validation forwards).
// Bit-identical by construction and by test (tests/test_nf_matching.py): 8-connected components numbered in raster order of
// their first pixel (= scipy.ndimage.label / skimage.measure.label), centres = integer coordinate sums / counts in double,
// distances sqrt(dy*dy + dx*dx) in double without contraction, greedy matching in the order of the truth's regions with
// the first-minimum tie-break of numpy.argmin, overlaps as pixel counts.  The MEANS over the matched pairs are left to
// numpy (its pairwise summation order is not restated here): the per-pair ratios are returned.
#include "common.h"
#include <math.h>

namespace {

// One labelled mask inside the caller's workspace: label image, region count, per-region pixel count and centre.
struct Labelled {
  int* lbl;          // H*W, 0 = background
  int* fg;           // flat indices of the foreground pixels, raster order
  int nfg;
  int n;
  double *cnt, *cy, *cx;
};

inline int find_root(int* parent, int a) {
  while (parent[a] != a) {
    parent[a] = parent[parent[a]];
    a = parent[a];
  }
  return a;
}

// 8-connected components, numbered in raster order of their first pixel.  Masks are mostly background (neurons cover ~12 %
// of a field of view): zero runs are skipped 8 bytes at a time and every later pass walks the foreground list only.
void label8(const uint8_t* m, int H, int W, Labelled& out, int* parent, int* final_of) {
  int np = 1, nfg = 0;
  parent[0] = 0;
  for (int y = 0; y < H; ++y) {
    const int* up = y ? out.lbl + (size_t)(y - 1) * W : nullptr;
    int* row = out.lbl + (size_t)y * W;
    const uint8_t* mr = m + (size_t)y * W;
    for (int x = 0; x < W; ++x) {
      if (!mr[x]) {
        if (x + 8 <= W) {
          uint64_t w8;
          __builtin_memcpy(&w8, mr + x, 8);
          if (w8 == 0) {
            for (int e = 0; e < 8; ++e) row[x + e] = 0;
            x += 7;
            continue;
          }
        }
        row[x] = 0;
        continue;
      }
      int best = 0;
      const int nb[4] = {x ? row[x - 1] : 0, (up && x) ? up[x - 1] : 0, up ? up[x] : 0, (up && x + 1 < W) ? up[x + 1] : 0};
      for (int l : nb) {
        if (!l) continue;
        const int r = find_root(parent, l);
        if (!best) best = r;
        else if (r != best) {
          const int lo = r < best ? r : best, hi = r < best ? best : r;
          parent[hi] = lo;
          best = lo;
        }
      }
      if (!best) {
        best = np;
        parent[np++] = best;
      }
      row[x] = best;
      out.fg[nfg++] = y * W + x;
    }
  }
  out.nfg = nfg;
  for (int i = 0; i < np; ++i) final_of[i] = 0;
  int n = 0;
  for (int k = 0; k < nfg; ++k) {
    const int i = out.fg[k];
    const int r = find_root(parent, out.lbl[i]);
    if (!final_of[r]) final_of[r] = ++n;
    out.lbl[i] = final_of[r];
  }
  out.n = n;
  for (int i = 0; i < n; ++i) out.cnt[i] = out.cy[i] = out.cx[i] = 0.0;
  for (int k = 0; k < nfg; ++k) {
    const int i = out.fg[k], l = out.lbl[i], y = i / W, x = i - y * W;
    out.cnt[l - 1] += 1.0; out.cy[l - 1] += (double)y; out.cx[l - 1] += (double)x;      // integer sums: exact
  }
  for (int i = 0; i < n; ++i) { out.cy[i] /= out.cnt[i]; out.cx[i] /= out.cnt[i]; }
}

// neurofinder.match: for every a in order, the nearest REMAINING b (first minimum) if closer than threshold
void greedy(const Labelled& a, const Labelled& b, double threshold, int* out, char* alive) {
  for (int i = 0; i < a.n; ++i) out[i] = -1;
  for (int j = 0; j < b.n; ++j) alive[j] = 1;
  int left = b.n;
  for (int i = 0; i < a.n && left > 0; ++i) {
    int bj = -1;
    double bd = 0.0;
    for (int j = 0; j < b.n; ++j) {
      if (!alive[j]) continue;
      const double dy = b.cy[j] - a.cy[i], dx = b.cx[j] - a.cx[i];
      const double d = sqrt(dy * dy + dx * dx);
      if (bj < 0 || d < bd) { bj = j; bd = d; }
    }
    if (bj >= 0 && bd < threshold) { out[i] = bj; alive[bj] = 0; --left; }
  }
}

inline long max_regions(int H, int W) { return ((long)(H + 1) / 2) * ((W + 1) / 2); }      // 8-connectivity

struct Carve {
  char* p;
  template <class T> T* take(size_t n) {
    T* r = reinterpret_cast<T*>(p);
    p += (n * sizeof(T) + 15) & ~(size_t)15;
    return r;
  }
};

}  // namespace

extern "C" long dc_host_nf_ws_bytes(int H, int W) {
  if (H <= 0 || W <= 0) return 0;
  const size_t hw = (size_t)H * W, mr = (size_t)max_regions(H, W);
  auto al = [](size_t b) { return (b + 15) & ~(size_t)15; };
  return (long)(4 * al(hw * 4) + 2 * al((hw + 1) * 4) + 6 * al(mr * 8) + 2 * al(mr * 4) + al(mr) + al(mr * 8) + 64);
}

static void carve_all(void* ws, int H, int W, Labelled& a, Labelled& b, int*& parent, int*& final_of, int*& m1, int*& m2,
                      char*& alive, long*& hit) {
  const size_t hw = (size_t)H * W, mr = (size_t)max_regions(H, W);
  Carve c{reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(ws) + 15) & ~(uintptr_t)15)};
  a.lbl = c.take<int>(hw); b.lbl = c.take<int>(hw);
  a.fg = c.take<int>(hw); b.fg = c.take<int>(hw);
  parent = c.take<int>(hw + 1); final_of = c.take<int>(hw + 1);
  a.cnt = c.take<double>(mr); a.cy = c.take<double>(mr); a.cx = c.take<double>(mr);
  b.cnt = c.take<double>(mr); b.cy = c.take<double>(mr); b.cx = c.take<double>(mr);
  m1 = c.take<int>(mr); m2 = c.take<int>(mr);
  alive = c.take<char>(mr);
  hit = c.take<long>(mr);
}

extern "C" int dc_host_nf_pairs(const uint8_t* truth, const uint8_t* pred, int H, int W, double threshold, int* counts,
                                double* inc, double* exc, int cap, void* ws) {
  DC_REQUIRE(truth && pred && counts && inc && exc && ws && H > 0 && W > 0 && cap >= 0, DC_EINVAL,
             "dc_host_nf_pairs: bad arguments");
  Labelled a, b;
  int *parent, *final_of, *mt, *mi;
  char* alive;
  long* hit;
  carve_all(ws, H, W, a, b, parent, final_of, mt, mi, alive, hit);
  label8(truth, H, W, a, parent, final_of);
  label8(pred, H, W, b, parent, final_of);
  counts[0] = a.n; counts[1] = b.n; counts[2] = 0; counts[3] = 0;
  if (a.n == 0 || b.n == 0) return DC_OK;
  greedy(a, b, threshold, mt, alive);
  int hits = 0;
  for (int i = 0; i < a.n; ++i) hits += mt[i] >= 0;
  counts[2] = hits;
  greedy(a, b, INFINITY, mi, alive);
  // |A_i & B_j| of the matched pairs only: one pass over the pixels
  for (int i = 0; i < a.n; ++i) hit[i] = 0;
  for (int k = 0; k < b.nfg; ++k) {
    const int p = b.fg[k], la = a.lbl[p];
    if (la && mi[la - 1] == b.lbl[p] - 1) hit[la - 1] += 1;
  }
  int np = 0;
  for (int i = 0; i < a.n; ++i) {
    const int j = mi[i];
    if (j < 0) continue;
    DC_REQUIRE(np < cap, DC_EINVAL, "dc_host_nf_pairs: more than cap=%d matched pairs", cap);
    inc[np] = (double)hit[i] / a.cnt[i];
    exc[np] = (double)hit[i] / b.cnt[j];
    ++np;
  }
  counts[3] = np;
  return DC_OK;
}

// the label image itself (int32 [H][W], components numbered in raster order of their first pixel) -- for the labelling test
extern "C" int dc_host_label8(const uint8_t* mask, int H, int W, int* labels, int* n, void* ws) {
  DC_REQUIRE(mask && labels && n && ws && H > 0 && W > 0, DC_EINVAL, "dc_host_label8: bad arguments");
  Labelled a, b;
  int *parent, *final_of, *m1, *m2;
  char* alive;
  long* hit;
  carve_all(ws, H, W, a, b, parent, final_of, m1, m2, alive, hit);
  label8(mask, H, W, a, parent, final_of);
  const size_t total = (size_t)H * W;
  for (size_t i = 0; i < total; ++i) labels[i] = a.lbl[i];
  *n = a.n;
  return DC_OK;
}
