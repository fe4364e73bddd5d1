// sf_mfe_full.hip.h — Zuker MFE fill + traceback with int32 tables in device memory ("FULL" kernel).
//
// Replaces RNA.fold(seq) / RNA.fold_compound(seq, md).mfe() for the windows whose structure string the TSV
// needs (ScanFold-Scan.py:385,394) and is the exact-arithmetic fallback of the LDS-resident int16 kernel
// (sf_mfe_fast.hip.h).  One workgroup per sequence; thread t owns cell (i = t+1, j = i+d) of anti-diagonal d;
// one barrier per diagonal.  Tables are diagonal-major, T(d,i) = tab[d*(W+1)+i], so the threads of a
// diagonal touch consecutive addresses.  Recurrences and traceback order: SURVEY.md A.3.
#pragma once
#include "sf_energy.h"

#define SF_FULL_SCRATCH_INTS(W) (3 * (size_t)(W) * ((W) + 1) + 3 * (4 * (size_t)(W) + 8))

__device__ inline int sf_block_min(int v, int *red) {
  for (int m = 32; m >= 1; m >>= 1) {
    int o = __shfl_xor(v, m);
    v = o < v ? o : v;
  }
  const int tid = threadIdx.x;
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  int r = red[0];
  const int nw = ((int)blockDim.x + 63) >> 6;
  for (int w = 1; w < nw; w++) r = red[w] < r ? red[w] : r;
  return r;
}

// items: k in [0, n_items); sequence row = idx_list ? idx_list[k] : k*row_stride;
// energy goes to mfe_out[idx_list ? idx_list[k] : k*mfe_stride] (if mfe_out); structure (if db_out) goes to
// db_out + k*(W+1) when db_stride == 0, else only rows that are multiples of db_stride are traced, into
// db_out + (row/db_stride)*(W+1) (how the fast kernel's overflow list maps back to native windows).
__global__ void sf_mfe_full_kernel(const uint8_t *__restrict__ seqs, const int *__restrict__ idx_list,
                                   const int *__restrict__ count_ptr, int n, int row_stride, int mfe_stride, int W,
                                   const SfDevParams *__restrict__ D, int32_t *__restrict__ scratch,
                                   int32_t *__restrict__ mfe_out, char *__restrict__ db_out, int db_stride,
                                   int *__restrict__ status, const char *__restrict__ cons_rows,
                                   const int32_t *__restrict__ sc_rows) {
  // cons_rows: item k's hard constraint = W characters at cons_rows + k*W (fc.hc_add_from_db, ScanFold-Scan.py:405-410);
  // sc_rows: item k's Deigan pseudo-energies (dcal/mol per nucleotide, added to every stack the nucleotide is part of:
  // fc.sc_add_SHAPE_deigan, ScanFold.py:533-539) at sc_rows + k*W.  Null = none (the plain RNA.fold of the hot path).
  __shared__ char hcC[SF_MAX_W + 2];
  __shared__ int16_t hcP[SF_MAX_W + 2], hcE[SF_MAX_W + 2], hcStk[SF_MAX_W + 2];
  __shared__ int scS[SF_MAX_W + 2];
  __shared__ int hcBad;
  __shared__ uint8_t S[SF_MAX_W + 2];
  __shared__ int f5s[SF_MAX_W + 1];
  __shared__ int red[8];
  const int tid = threadIdx.x;
  const int nthreads = blockDim.x;
  const int W1 = W + 1;
  int32_t *c = scratch + (size_t)blockIdx.x * SF_FULL_SCRATCH_INTS(W);
  int32_t *fML = c + (size_t)W * W1;
  int32_t *DML = fML + (size_t)W * W1;
  int32_t *stk = DML + (size_t)W * W1;
#define FT(tab, d, i) tab[(d)*W1 + (i)]
  const sf_params_blob &P = D->P;
  const int n_items = count_ptr ? *count_ptr : n;

  for (int k = blockIdx.x; k < n_items; k += gridDim.x) {
    const int row = idx_list ? idx_list[k] : k * row_stride;
    const uint8_t *src = seqs + (size_t)row * W;
    __syncthreads();
    for (int x = tid; x < W; x += nthreads) S[x + 1] = sf_encode_nt(src[x]);
    if (tid == 0) { S[0] = 0; S[W + 1] = 0; f5s[0] = 0; }
    for (int x = tid; x < 4 * W1 && x < W * W1; x += nthreads) { c[x] = SFD_INF; fML[x] = SFD_INF; DML[x] = SFD_INF; }
    SfHc hc;
    hc.c = cons_rows ? hcC : nullptr; hc.partner = hcP; hc.encl = hcE;
    // (constraint / pseudo-energy rows go by SEQUENCE ROW: with an index list — the folds the int16 kernel hands back —
    // item k is row idx_list[k]; without one row == k * row_stride, and constrained callers use row_stride 1)
    if (cons_rows && tid == 0) hcBad = sf_hc_parse(cons_rows + (size_t)row * W, W, hcC, hcP, hcE, hcStk);
    if (sc_rows)
      for (int x = tid; x < W; x += nthreads) scS[x + 1] = sc_rows[(size_t)row * W + x];
    __syncthreads();
    if (cons_rows && hcBad) {  // unbalanced brackets: report, fold nothing (ViennaRNA aborts the process here)
      if (tid == 0) {
        if (status) atomicOr(status, 2);
        if (mfe_out) mfe_out[idx_list ? idx_list[k] : k * mfe_stride] = 0;
      }
      continue;
    }
    // pair type under the window's constraint; stack term of the soft constraint
    auto PTY = [&](int a, int b) -> int {
      const bool ok = b - a <= D->max_pair_dist;
      return sf_hc_type(hc, ok ? D->pair[S[a]][S[b]] : 0, a, b, ok);
    };
    auto SCS = [&](int a, int b, int u1, int u2) -> int {
      return (sc_rows && (u1 | u2) == 0) ? scS[a] + scS[a + 1] + scS[b - 1] + scS[b] : 0;
    };

    for (int d = SFD_TURN + 1; d < W; d++) {
      const int i = tid + 1, j = i + d;
      if (j <= W) {
        const int type = PTY(i, j);  // max_bp_span, hard constraint
        int cij = SFD_INF;
        if (type) {
          int e = sfd_hairpin(D, S, i, j, type);
          const int umax = sfd_min(SFD_MAXLOOP, d - 2 - (SFD_TURN + 1));
          const int si1 = S[i + 1], sj1 = S[j - 1];
          for (int u1 = 0; u1 <= umax; u1++) {
            const int p = i + 1 + u1;
            for (int u2 = 0; u2 <= umax - u1; u2++) {
              const int q = j - 1 - u2;
              const int t2 = PTY(p, q);
              if (!t2) continue;
              const int en = sfd_intloop(D, u1, u2, type, sfd_rtype(t2), si1, sj1, S[p - 1], S[q + 1]) + FT(c, q - p, p) +
                             SCS(i, j, u1, u2);
              e = sfd_min(e, en);
            }
          }
          const int dml = FT(DML, d - 2, i + 1);
          if (dml < SFD_INF) e = sfd_min(e, dml + sfd_mlstem(D, sfd_rtype(type), sj1, si1) + P.MLclosing);
          cij = e;
        }
        FT(c, d, i) = cij;
        int f = SFD_INF;
        const int a = FT(fML, d - 1, i + 1), b = FT(fML, d - 1, i);
        if (a < SFD_INF) f = sfd_min(f, a + P.MLbase);
        if (b < SFD_INF) f = sfd_min(f, b + P.MLbase);
        if (type) f = sfd_min(f, cij + sfd_mlstem(D, type, i > 1 ? S[i - 1] : -1, j < W ? S[j + 1] : -1));
        int dec = SFD_INF;
        for (int m = SFD_TURN + 1; m <= d - SFD_TURN - 2; m++) {
          const int x = FT(fML, m, i), y = FT(fML, d - m - 1, i + m + 1);
          if (x < SFD_INF && y < SFD_INF) dec = sfd_min(dec, x + y);
        }
        FT(DML, d, i) = dec;
        FT(fML, d, i) = sfd_min(f, dec);
      }
      __syncthreads();
    }

    // exterior loop: f5[j] = min(f5[j-1], min_i f5[i-1] + c[i,j] + ExtLoop(i,j))
    for (int j = 1; j <= W; j++) {
      int v = SFD_INF;
      const int i = tid + 1;
      if (i + SFD_TURN + 1 <= j && j - i <= D->max_pair_dist) {
        const int type = PTY(i, j);
        if (type) v = f5s[i - 1] + FT(c, j - i, i) + sfd_extloop(D, type, i > 1 ? S[i - 1] : -1, j < W ? S[j + 1] : -1);
      }
      v = sf_block_min(v, red);
      if (tid == 0) f5s[j] = sfd_min(f5s[j - 1], v);
      __syncthreads();
    }
    if (tid == 0 && mfe_out) mfe_out[idx_list ? idx_list[k] : k * mfe_stride] = f5s[W];

    if (db_out && tid == 0 && (db_stride == 0 || row % db_stride == 0)) {
      char *db = db_out + (size_t)(db_stride ? row / db_stride : k) * W1;
      for (int x = 0; x < W; x++) db[x] = '.';
      db[W] = 0;
      int s = 0, bad = 0;
      stk[0] = 1; stk[1] = W; stk[2] = 0; s = 1;
      while (s > 0 && !bad) {
        --s;
        int i = stk[3 * s], j = stk[3 * s + 1];
        const int ml = stk[3 * s + 2];
        bool have_pair = false;
        if (ml == 0) {
          while (j > 0 && f5s[j] == f5s[j - 1]) j--;
          if (j < SFD_TURN + 2) continue;
          const int fij = f5s[j];
          int kk, found = 0;
          for (kk = j - SFD_TURN - 1; kk >= 1; kk--) {
            const int type = PTY(kk, j);
            if (!type) continue;
            if (fij == f5s[kk - 1] + FT(c, j - kk, kk) +
                           sfd_extloop(D, type, kk > 1 ? S[kk - 1] : -1, j < W ? S[j + 1] : -1)) { found = 1; break; }
          }
          if (!found) { bad = 1; break; }
          stk[3 * s] = 1; stk[3 * s + 1] = kk - 1; stk[3 * s + 2] = 0; s++;
          i = kk;
          have_pair = true;
        } else {
          if (j - i < SFD_TURN + 1) { bad = 1; break; }
          while (j - i > SFD_TURN + 1 && FT(fML, j - 1 - i, i) < SFD_INF &&
                 FT(fML, j - i, i) == FT(fML, j - 1 - i, i) + P.MLbase) j--;
          while (j - i > SFD_TURN + 1 && FT(fML, j - i - 1, i + 1) < SFD_INF &&
                 FT(fML, j - i, i) == FT(fML, j - i - 1, i + 1) + P.MLbase) i++;
          const int fij = FT(fML, j - i, i);
          const int type = PTY(i, j);
          if (type && fij == FT(c, j - i, i) + sfd_mlstem(D, type, i > 1 ? S[i - 1] : -1, j < W ? S[j + 1] : -1)) {
            have_pair = true;
          } else {
            int kk, found = 0;
            for (kk = i + SFD_TURN + 1; kk <= j - SFD_TURN - 2; kk++) {
              const int x = FT(fML, kk - i, i), y = FT(fML, j - kk - 1, kk + 1);
              if (x < SFD_INF && y < SFD_INF && fij == x + y) { found = 1; break; }
            }
            if (!found) { bad = 1; break; }
            stk[3 * s] = i; stk[3 * s + 1] = kk; stk[3 * s + 2] = 1; s++;
            stk[3 * s] = kk + 1; stk[3 * s + 1] = j; stk[3 * s + 2] = 1; s++;
          }
        }
        while (have_pair) {
          db[i - 1] = '(';
          db[j - 1] = ')';
          const int type = PTY(i, j);
          const int cij = FT(c, j - i, i);
          if (cij == sfd_hairpin(D, S, i, j, type)) break;
          const int d = j - i;
          const int umax = sfd_min(SFD_MAXLOOP, d - 2 - (SFD_TURN + 1));
          int found = 0, fp = 0, fq = 0;
          for (int u1 = 0; u1 <= umax && !found; u1++) {
            const int p = i + 1 + u1;
            for (int u2 = 0; u2 <= umax - u1; u2++) {
              const int q = j - 1 - u2;
              const int t2 = PTY(p, q);
              if (!t2) continue;
              const int en = sfd_intloop(D, u1, u2, type, sfd_rtype(t2), S[i + 1], S[j - 1], S[p - 1], S[q + 1]) +
                             FT(c, q - p, p) + SCS(i, j, u1, u2);
              if (cij == en) { found = 1; fp = p; fq = q; break; }
            }
          }
          if (found) { i = fp; j = fq; continue; }
          const int mm = P.MLclosing + sfd_mlstem(D, sfd_rtype(type), S[j - 1], S[i + 1]);
          int kk, ok = 0;
          for (kk = i + 1 + SFD_TURN + 1; kk <= j - 1 - SFD_TURN - 2; kk++) {
            const int x = FT(fML, kk - (i + 1), i + 1), y = FT(fML, j - 1 - (kk + 1), kk + 1);
            if (x < SFD_INF && y < SFD_INF && cij == x + y + mm) { ok = 1; break; }
          }
          if (!ok) { bad = 1; break; }
          stk[3 * s] = i + 1; stk[3 * s + 1] = kk; stk[3 * s + 2] = 1; s++;
          stk[3 * s] = kk + 1; stk[3 * s + 1] = j - 1; stk[3 * s + 2] = 1; s++;
          break;
        }
      }
      if (bad && status) atomicOr(status, 1);
    }
  }
#undef FT
}
