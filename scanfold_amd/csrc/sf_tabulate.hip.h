// sf_tabulate.hip.h — base-pair tabulation of a scan table on the device.
//
// What it replaces: the per-window loop of the Fold stage that walks every window's dot-bracket string, numbers the
// nesting levels, and appends the window's (z-score, MFE, ED) to a Python dict keyed by (nucleotide, partner)
// (/root/reference/ScanFold-Fold.py:583-682; ScanFold.py:564-677), followed by np.sum / np.mean over every list
// (ScanFold-Fold.py:704-846).  Here:
//   sf_tab_partner_kernel   one thread per window: bracket matching with a stack in LDS -> partner[window][position]
//   sf_tab_groups_kernel    one wave per nucleotide coordinate k: the windows that cover k, in window order, grouped by
//                           the partner they give k (partner == k: unpaired); per group the window count, the first
//                           window (the dict's insertion order) and the three sums
//   sf_tab_scan_kernel      exclusive scan of the per-coordinate group counts (output offsets)
// The sums are the reference's bit for bit: numpy adds a list with its pairwise scheme (eight running partial sums over
// blocks of eight, combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), the tail added one by one; lists above 128 entries
// split in two halves, the first a multiple of eight long) — restated in sf_tab_pairwise.  Only additions: nothing
// the compiler could contract.  Means, rounding and everything after the sums stay on the host (scanfold_amd/fold.py).
//
// Bytes: this is integer / byte work bound by HBM and latency, not arithmetic: per window W structure bytes in,
// 2 W partner bytes out and in again (L2), 24 bytes of metrics; per group 40 bytes out.
#pragma once
#include "sf_launch.h"

#define SF_TAB_MAXM 512  // windows that can cover one coordinate: W <= SF_MAX_W = 400 at step >= 1
#define SF_TAB_ST_UNBALANCED 4
#define SF_TAB_ST_TOO_MANY 8

__global__ void sf_tab_partner_kernel(const char *__restrict__ structs, const int row_stride, const int n_win, const int W,
                                      int16_t *__restrict__ partner, int *__restrict__ status) {
  SF_DYN_SMEM(smem);
  int16_t *stack = (int16_t *)smem;  // [depth][lane]: a lane's stack never shares a bank with a neighbour's
  const int lane = threadIdx.x;
  const int w = blockIdx.x * 64 + lane;
  if (w >= n_win) return;
  const char *s = structs + (size_t)w * row_stride;
  int16_t *out = partner + (size_t)w * W;
  int depth = 0, bad = 0;
  for (int p = 0; p < W; p++) {
    const char ch = s[p];
    int q = -1;
    if (ch == '(') {
      // the stack holds W/2 + 1 entries per lane: a row with more open brackets than that cannot balance
      // (rows come from a user's TSV in the Fold stage)
      if (depth > W / 2) bad = 1;
      else stack[depth * 64 + lane] = (int16_t)p;
      depth++;
    } else if (ch == ')') {
      if (depth == 0) {
        bad = 1;
      } else {
        depth--;
        if (depth <= W / 2) {
          q = stack[depth * 64 + lane];
          out[q] = (int16_t)p;
        }
      }
    }
    out[p] = (int16_t)q;
  }
  if (bad || depth) atomicOr(status, SF_TAB_ST_UNBALANCED);
}

// numpy's pairwise sum of at(0..n-1), n <= 128: the additions in numpy's order
template <class F>
__device__ __forceinline__ double sf_tab_pairwise_leaf(F at, const int n) {
  if (n < 8) {
    double res = 0.;
    for (int i = 0; i < n; i++) res = res + at(i);
    return res;
  }
  double r0 = at(0), r1 = at(1), r2 = at(2), r3 = at(3), r4 = at(4), r5 = at(5), r6 = at(6), r7 = at(7);
  int i = 8;
  for (; i < n - (n % 8); i += 8) {
    r0 = r0 + at(i + 0); r1 = r1 + at(i + 1); r2 = r2 + at(i + 2); r3 = r3 + at(i + 3);
    r4 = r4 + at(i + 4); r5 = r5 + at(i + 5); r6 = r6 + at(i + 6); r7 = r7 + at(i + 7);
  }
  double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
  for (; i < n; i++) res = res + at(i);
  return res;
}
// ... n <= SF_TAB_MAXM: two levels of halving reach leaves of at most 128
template <class F>
__device__ __forceinline__ double sf_tab_pairwise_mid(F at, const int n) {
  if (n <= 128) return sf_tab_pairwise_leaf(at, n);
  int n2 = n / 2;
  n2 -= n2 % 8;
  const double a = sf_tab_pairwise_leaf(at, n2);
  const double b = sf_tab_pairwise_leaf([&](int i) { return at(n2 + i); }, n - n2);
  return a + b;
}
template <class F>
__device__ __forceinline__ double sf_tab_pairwise(F at, const int n) {
  if (n <= 128) return sf_tab_pairwise_leaf(at, n);
  int n2 = n / 2;
  n2 -= n2 % 8;
  const double a = sf_tab_pairwise_mid(at, n2);
  const double b = sf_tab_pairwise_mid([&](int i) { return at(n2 + i); }, n - n2);
  return a + b;
}

// first index with starts[idx] > v (starts ascending)
__device__ __forceinline__ int sf_tab_upper(const int32_t *__restrict__ starts, const int n, const int v) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (starts[mid] <= v) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}

// WRITE = false: counts[c] = number of groups of coordinate lo + c.  WRITE = true: counts holds the exclusive scan;
// the groups of coordinate c go to offsets counts[c].. in first-window order.
template <bool WRITE>
__global__ void sf_tab_groups_kernel(const int16_t *__restrict__ partner, const int32_t *__restrict__ starts, const int n_win,
                                     const int W, const int lo, int32_t *__restrict__ counts,
                                     const double *__restrict__ z, const double *__restrict__ mfe,
                                     const double *__restrict__ ed, int32_t *__restrict__ gk, int32_t *__restrict__ gj,
                                     int32_t *__restrict__ gcount, int32_t *__restrict__ gfirst,
                                     double *__restrict__ gsum_z, double *__restrict__ gsum_mfe,
                                     double *__restrict__ gsum_ed, int *__restrict__ status) {
  __shared__ int32_t jl[SF_TAB_MAXM];     // partner coordinate given by the a-th covering window
  __shared__ int16_t gidx[SF_TAB_MAXM];   // first a' with the same partner
  __shared__ int16_t gno[SF_TAB_MAXM];    // group number of entry a
  __shared__ int16_t ord[SF_TAB_MAXM];    // entries by group, window order inside a group
  __shared__ int16_t gfa[SF_TAB_MAXM], gcnt[SF_TAB_MAXM], goff[SF_TAB_MAXM], gfill[SF_TAB_MAXM];
  __shared__ int G_s;
  const int t = threadIdx.x;
  const int k = lo + (int)blockIdx.x;
  const int w_hi = sf_tab_upper(starts, n_win, k) - 1;  // last window that starts at or before k
  const int w_lo = sf_tab_upper(starts, n_win, k - W);  // first window that still reaches k
  int m = w_hi - w_lo + 1;
  if (m > SF_TAB_MAXM) {
    if (t == 0) atomicOr(status, SF_TAB_ST_TOO_MANY);
    m = 0;
  }
  if (m <= 0) {
    if (!WRITE && t == 0) counts[blockIdx.x] = 0;
    return;
  }
  for (int a = t; a < m; a += 64) {
    const int w = w_lo + a;
    const int st = starts[w];
    const int q = partner[(size_t)w * W + (k - st)];
    jl[a] = q < 0 ? k : st + q;
  }
  __syncthreads();
  for (int a = t; a < m; a += 64) {
    const int ja = jl[a];
    int f = a;
    for (int b = 0; b < a; b++)
      if (jl[b] == ja) { f = b; break; }
    gidx[a] = (int16_t)f;
  }
  __syncthreads();
  if (t == 0) {
    int G = 0;
    for (int a = 0; a < m; a++) {
      const int f = gidx[a];
      int g;
      if (f == a) {
        g = G++;
        gfa[g] = (int16_t)a;
        gcnt[g] = 0;
      } else {
        g = gno[f];
      }
      gno[a] = (int16_t)g;
      gcnt[g]++;
    }
    int o = 0;
    for (int g = 0; g < G; g++) {
      goff[g] = (int16_t)o;
      gfill[g] = 0;
      o += gcnt[g];
    }
    for (int a = 0; a < m; a++) {
      const int g = gno[a];
      ord[goff[g] + gfill[g]] = (int16_t)a;
      gfill[g]++;
    }
    G_s = G;
  }
  __syncthreads();
  const int G = G_s;
  if (!WRITE) {
    if (t == 0) counts[blockIdx.x] = G;
    return;
  }
  const int base = counts[blockIdx.x];
  for (int g = t; g < G; g += 64) {
    const int n = gcnt[g];
    const int16_t *o = ord + goff[g];
    const int dst = base + g;
    gk[dst] = k;
    gj[dst] = jl[gfa[g]];
    gcount[dst] = n;
    gfirst[dst] = w_lo + gfa[g];
    gsum_z[dst] = sf_tab_pairwise([&](int i) { return z[w_lo + o[i]]; }, n);
    gsum_mfe[dst] = sf_tab_pairwise([&](int i) { return mfe[w_lo + o[i]]; }, n);
    gsum_ed[dst] = sf_tab_pairwise([&](int i) { return ed[w_lo + o[i]]; }, n);
  }
}

// in-place exclusive scan of counts[0..n), total to counts[n]; one workgroup of 256 threads
__global__ void sf_tab_scan_kernel(int32_t *__restrict__ counts, const int n) {
  __shared__ int part[256];
  const int t = threadIdx.x;
  const int per = (n + 255) / 256;
  const int a = t * per, b = (a + per < n) ? a + per : n;
  int s = 0;
  for (int i = a; i < b; i++) s += counts[i];
  part[t] = s;
  __syncthreads();
  if (t == 0) {
    int acc = 0;
    for (int i = 0; i < 256; i++) {
      const int v = part[i];
      part[i] = acc;
      acc += v;
    }
    counts[n] = acc;
  }
  __syncthreads();
  int acc = part[t];
  for (int i = a; i < b; i++) {
    const int v = counts[i];
    counts[i] = acc;
    acc += v;
  }
}
