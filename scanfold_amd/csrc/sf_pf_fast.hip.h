// sf_pf_fast.hip.h — McCaskill partition function for W <= 256 with O(MAXLOOP) interior-loop work per cell.
//
// Same mathematics and outputs as sf_pf.hip.h (which stays as the W > 256 path); replaces fc.pf() /
// fc.centroid() / fc.mean_bp_distance() for the native windows (ScanFold-Scan.py:383-389).  Differences:
//  * centre-based thread mapping (see sf_mfe_fast.hip.h): the cell a thread handles two diagonals later
//    encloses the current one, so the generic interior-loop SUM is carried in registers:
//      inside   H[i,j,u] = sum_{u1+u2=u; u1,u2>=2} qbI[i+1+u1, j-1-u2] w(|u1-u2|) = H[i+1,j-1,u-2] + 2 edge terms
//      outside  G[i,j,u] = sum_{u1+u2=u; u1,u2>=2} obI[i-1-u1, j+1+u2] w(|u1-u2|) = G[i-1,j+1,u-2] + 2 edge terms
//    (qbI / obI = qb / ob pre-multiplied by the pair's own interior mismatch weight), 6 table reads per loop
//    size instead of one per (u1,u2);
//  * every O(W) multiloop sum is indexed by diagonal so that the threads of a wave read consecutive addresses.
// Tables are FP64, triangular diagonal-major T(d,i), in device memory (13 tables x W(W+1)/2 doubles per
// workgroup).  Sums are re-associated with respect to the oracle, so results agree to ~1e-12 relative.
#pragma once
#include "sf_energy.h"
#include "sf_pf.hip.h"
#include <type_traits>

#define SF_PFF_NTABLES 13
#define SF_PFF_TABLE_DOUBLES(W) ((size_t)(W) * ((W) + 1) / 2 + 8)
#define SF_PFF_SCRATCH_DOUBLES(W) (SF_PFF_NTABLES * SF_PFF_TABLE_DOUBLES(W))
#define SF_PFF_MAXW 256
#ifndef SF_PFF_UNROLL
#define SF_PFF_UNROLL 4  // terms of a cooperative multiloop sum in flight per thread
#endif

struct SfPfTabs {
  double *QB, *QBI, *QB1N, *QBB, *QM, *QM1, *OB, *OBI, *OB1N, *OBB, *OBW, *A0, *A1;
};

// Parts a cell's O(d) multiloop sums are split into on diagonal d (n = W - d cells, NT threads): NT / n, capped — the cell's
// owner adds the partial sums one after the other from LDS, so on the last diagonals (n -> 1) an uncapped NT / n would trade a
// chain of d global loads for a chain of 256 LDS reads in ONE thread while the workgroup waits at the barrier.  With the cap a
// part walks <= ceil(d / 16) terms and the owner adds <= 16.  (The summation order differs from the LDS kernel's and the
// oracle's at ~1e-13; a pair probability within that of 0.5 could fall on either side of the centroid's threshold — DESIGN.md.)
#ifndef SF_PFF_MAXPARTS
#define SF_PFF_MAXPARTS 16
#endif
__host__ __device__ static inline int sf_pff_parts(int nt, int n) {
  const int k = nt / n;
  return k < SF_PFF_MAXPARTS ? k : SF_PFF_MAXPARTS;
}

// HC: fold k has its own hard constraint, W characters at cons_rows + k * row_stride * W (fc.hc_add_from_db before fc.pf(),
// ScanFold-Scan.py:405-412).  As in the LDS kernels the constraint acts where a cell's OWN pair type is made (inside, the
// two exterior sums, outside): a forbidden pair has qb = 0 and drops out of every sum that reads it; the types of enclosed /
// enclosing pairs in the special-loop look-ups stay sequence-only.  W <= 250 (partners are bytes); a bracket pair of
// non-complementary bases never comes here (the host routes such batches to sf_pf_kernel).  Until round 4 a constrained
// scan at 120 < W <= 256 ran its partition functions on sf_pf_kernel — O(W^2 L^2) per fold, ~90 x slower.
template <int NT, int WT, bool HC = false>
__global__ __launch_bounds__(NT, 2) void sf_pf_fast_kernel(const uint8_t *__restrict__ seqs, int n, int row_stride, int Wrt,
                                                        const SfDevParams *__restrict__ D,
                                                        const SfDevParamsPF *__restrict__ X,
                                                        double *__restrict__ scratch, double *__restrict__ ens_dG,
                                                        double *__restrict__ mean_bp_dist, char *__restrict__ centroid,
                                                        double *__restrict__ centroid_dist,
                                                        const char *__restrict__ cons_rows, int *__restrict__ status) {
  __shared__ uint8_t S[SF_PFF_MAXW + 2];
  __shared__ char hcC[HC ? SF_PFF_MAXW + 2 : 1];
  __shared__ uint8_t hcP[HC ? SF_PFF_MAXW + 2 : 1], hcE[HC ? SF_PFF_MAXW + 2 : 1];
  SfHc8 hc;
  hc.c = (HC && cons_rows) ? hcC : nullptr; hc.partner = hcP; hc.encl = hcE;
  __shared__ double q5[SF_PFF_MAXW + 2];
  __shared__ double q3[SF_PFF_MAXW + 3];
  __shared__ double red[8];
  // partial sums of the O(W) multiloop sums of one diagonal, [part][cell] (see inside_sums / outside_sums)
  __shared__ double ps0[NT], ps1[NT];
  __shared__ double mlbS[SF_PFF_MAXW + 2];
  const int W = WT ? WT : Wrt;  // WT > 0: width known at compile time
  const int tid = threadIdx.x;
  const int W1 = W + 1;
  const size_t TS = SF_PFF_TABLE_DOUBLES(W);
  SfPfTabs T;
  {
    double *b = scratch + (size_t)blockIdx.x * SF_PFF_SCRATCH_DOUBLES(W);
    T.QB = b; T.QBI = b + TS; T.QB1N = b + 2 * TS; T.QBB = b + 3 * TS; T.QM = b + 4 * TS; T.QM1 = b + 5 * TS;
    T.OB = b + 6 * TS; T.OBI = b + 7 * TS; T.OB1N = b + 8 * TS; T.OBB = b + 9 * TS; T.OBW = b + 10 * TS;
    T.A0 = b + 11 * TS; T.A1 = b + 12 * TS;
  }
// triangular, diagonal-major: diagonal d holds the cells i = 1..W-d
#define PT(tab, d, i) tab[(d)*W - ((d) * ((d)-1)) / 2 + (i)-1]
// the pair type of a cell's OWN pair (a, b): sequence, max_bp_span and — HC — the fold's constraint
#define OWNT(Dp, a, b) (HC ? sf_hc_type8(hc, ((b) - (a) <= (Dp)->max_pair_dist ? (Dp)->pair[S[a]][S[b]] : 0), (a), (b), (b) - (a) <= (Dp)->max_pair_dist) \
                           : ((b) - (a) <= (Dp)->max_pair_dist ? (Dp)->pair[S[a]][S[b]] : 0))
  for (int x = threadIdx.x; x <= W + 1; x += NT) mlbS[x] = X->mlbase_pow[x];  // (read with per-lane indices in the cooperative sums)
  const double *mlb = mlbS;
  const int OFF = (((W + 1) >> 1) - 32 + NT) & (NT - 1);
  const int v = (tid + OFF) & (NT - 1);
  const double xTAU = X->TermAU;

  for (int k = blockIdx.x; k < n; k += gridDim.x) {
    const uint8_t *src = seqs + (size_t)k * row_stride * W;
    __syncthreads();
    for (int x = tid; x < W; x += NT) S[x + 1] = sf_encode_nt(src[x]);
    if (tid == 0) { S[0] = 0; S[W + 1] = 0; }
    if (HC && cons_rows) {
      for (int x = tid; x < W; x += NT) hcC[x + 1] = cons_rows[(size_t)k * row_stride * W + x];
      __syncthreads();
      if (tid == 0 && sf_hc_parse8(W, hcC, hcP, hcE)) atomicOr(status, 2);  // unbalanced: reported; runs with what matched
    }
    for (size_t x = tid; x < (size_t)4 * W && x < TS; x += NT) {
      T.QB[x] = 0.0; T.QBI[x] = 0.0; T.QB1N[x] = 0.0; T.QBB[x] = 0.0; T.QM[x] = 0.0; T.QM1[x] = 0.0;
    }
    __syncthreads();

    // ================= inside =================
    double Ha[27], Hb[27];
#pragma unroll
    for (int u = 0; u < 27; u++) { Ha[u] = 0.0; Hb[u] = 0.0; }
    // GT = std::true_type: loop sizes are tested against the limit d-6.  (A false_type instantiation makes the
    // candidate code straight-line; in FP64 that needs >256 VGPRs and spills, so it is not used.)
    // The two O(d) sums of a cell — the multiloop closing sum  sum_a qm[i+1, .] qm1[., j-1]  and the qm recurrence
    // sum_a (MLbase^a + qm[i, .]) qm1[., j] — read diagonals < d only.  One thread per cell walked them alone: d / 4 dependent
    // round trips to the L2 per diagonal, with W - d of the NT threads at work — the whole kernel waited on that chain (41 ms
    // per 200-mer).  Now a diagonal's NT threads split into k = NT / (W - d) parts per cell (part-major, so the lanes of a part
    // still read consecutive cells), every part sums its share of the range, and the cell's owner adds the k partial sums
    // from LDS.  Early diagonals (k = 1) are as before; from d = W - NT / 2 on the chain shrinks with the diagonal.
    // W = 200: 323 -> 226 ms per 4 096 folds (12.7 k -> 18.1 k folds/s), W = 160 1.50 x, W = 250 1.38 x; results equal to 1e-13
    // (profiles/r04/pf_fast_cooperative.txt).  What bounds it now is the fabric: 120 MB fetched per 200-mer (2 x FETCH_SIZE),
    // 2.2 TB/s at that rate — the tables of the 512 resident folds (2 MB each) are far beyond the L2s; the terms in flight per
    // thread (2 / 4 / 8) no longer matter.
    auto inside_sums = [&](const int d) {
      const SfDevParams *const Dc = sf_const_base(D);
      const int n = W - d, kk = sf_pff_parts(NT, n);
      const int p = tid / n, c = tid - p * n;
      if (p < kk) {
        const int i = c + 1, j = i + d;
        double sm = 0.0, sq = 0.0;
        if (OWNT(Dc, i, j)) {
          const int lo = SFD_TURN + 2, hi = d - SFD_TURN - 2;
          const int per = (hi - lo + kk) / kk;  // ceil((hi - lo + 1) / kk); <= 0 for an empty range
          const int a0 = lo + p * per, a1 = sfd_min(a0 + per - 1, hi);
#pragma unroll SF_PFF_UNROLL
          for (int a = a0; a <= a1; a++) sm += PT(T.QM, a - 2, i + 1) * PT(T.QM1, d - 1 - a, i + a);
        }
        {
          const int lo = 1, hi = d - SFD_TURN - 1;
          const int per = (hi - lo + kk) / kk;
          const int a0 = lo + p * per, a1 = sfd_min(a0 + per - 1, hi);
#pragma unroll SF_PFF_UNROLL
          for (int a = a0; a <= a1; a++) sq += (mlb[a] + PT(T.QM, a - 1, i)) * PT(T.QM1, d - a, i + a);
        }
        ps0[tid] = sm; ps1[tid] = sq;
      }
    };
    auto inside_step = [&](const int d, double(&H)[27], auto GT) {
      // (parameter blocks through sf_const_base: their tables addressed "one scalar base + offset" — the compiler otherwise keeps
      // one hoisted 64-bit address per table row in scalar registers it does not have: thousands of v_readlane restores)
      const SfDevParamsPF *const Xc = sf_const_base(X);
      const SfDevParams *const Dc = sf_const_base(D);
      constexpr bool G = decltype(GT)::value;
      const int i = v - (d >> 1), j = i + d;
      const bool valid = (i >= 1) && (j <= W);
      if (valid) {
        const int umax = G ? sfd_min(SFD_MAXLOOP, d - 2 - (SFD_TURN + 1)) : SFD_MAXLOOP;
        const int type = OWNT(Dc, i, j);
        const int si1 = S[i + 1], sj1 = S[j - 1];
        // generic interior sums of this cell from those of the enclosed cell
#pragma unroll
        for (int u = 30; u >= 6; --u)
          if (!G || u <= umax) H[u - 4] = H[u - 6] + (PT(T.QBI, d - 2 - u, i + 3) + PT(T.QBI, d - 2 - u, i + u - 1)) * Xc->ninio[u - 4];
        if (!G || umax >= 5) H[1] = (PT(T.QBI, d - 7, i + 3) + PT(T.QBI, d - 7, i + 4)) * Xc->ninio[1];
        if (!G || umax >= 4) H[0] = PT(T.QBI, d - 6, i + 3) * Xc->ninio[0];
        double qbij = 0.0;
        if (type) {
          double z = sfx_hairpin(Dc, Xc, S, i, j, type);
          if (!G || umax >= 0) {
            const double tau_out = type > 2 ? xTAU : 1.0;
            z += PT(T.QB, d - 2, i + 1) * Xc->stack[type][sfd_rtype(Dc->pair[si1][sj1])];
            if (!G || umax >= 1) {
              const int ta = sfd_rtype(Dc->pair[si1][S[j - 2]]), tb = sfd_rtype(Dc->pair[S[i + 2]][sj1]);
              z += (PT(T.QB, d - 3, i + 1) * Xc->stack[type][ta] + PT(T.QB, d - 3, i + 2) * Xc->stack[type][tb]) * Xc->bulge[1];
            }
            if (!G || umax >= 2) {
              const int t2r = sfd_rtype(Dc->pair[S[i + 2]][S[j - 2]]);
              z += PT(T.QB, d - 4, i + 2) * Xc->int11[type][t2r][si1][sj1];
            }
            if (!G || umax >= 3) {
              const int ta = sfd_rtype(Dc->pair[S[i + 2]][S[j - 3]]), tb = sfd_rtype(Dc->pair[S[i + 3]][S[j - 2]]);
              z += PT(T.QB, d - 5, i + 2) * Xc->int21[type][ta][si1][S[j - 2]][sj1] +
                   PT(T.QB, d - 5, i + 3) * Xc->int21[tb][type][sj1][si1][S[i + 2]];
            }
            if (!G || umax >= 4) {
              const int t2r = sfd_rtype(Dc->pair[S[i + 3]][S[j - 3]]);
              z += PT(T.QB, d - 6, i + 3) * Xc->int22[type][t2r][si1][S[i + 2]][S[j - 2]][sj1];
            }
            if (!G || umax >= 5) {
              const int ta = sfd_rtype(Dc->pair[S[i + 3]][S[j - 4]]), tb = sfd_rtype(Dc->pair[S[i + 4]][S[j - 3]]);
              const double m23 = Xc->internal_loop[5] * Xc->ninio[1] * Xc->mismatch23I[type][si1][sj1];
              z += m23 * (PT(T.QB, d - 7, i + 3) * Xc->mismatch23I[ta][S[j - 3]][S[i + 2]] +
                          PT(T.QB, d - 7, i + 4) * Xc->mismatch23I[tb][S[j - 2]][S[i + 3]]);
            }
            double gb = 0.0, g1 = 0.0, gg = 0.0;
#pragma unroll
            for (int u = 2; u <= 30; ++u)
              if (!G || u <= umax) {
                gb += (PT(T.QBB, d - 2 - u, i + 1) + PT(T.QBB, d - 2 - u, i + 1 + u)) * Xc->bulge[u];
                if (u >= 4)
                  g1 += (PT(T.QB1N, d - 2 - u, i + 2) + PT(T.QB1N, d - 2 - u, i + u)) * Xc->internal_loop[u] * Xc->ninio[u - 2];
                if (u >= 6) gg += H[u - 4] * Xc->internal_loop[u];
              }
            z += gb * tau_out + g1 * Xc->mismatch1nI[type][si1][sj1] + gg * Xc->mismatchI[type][si1][sj1];
          }
          double ml = 0.0;
          {
            const int n = W - d, kk = sf_pff_parts(NT, n);
            for (int q = 0; q < kk; q++) ml += ps0[q * n + i - 1];
          }
          z += ml * Xc->MLclosing * sfx_mlstem(Xc, sfd_rtype(type), sj1, si1);
          qbij = z;
        }
        {
          const int tr = sfd_rtype(type);
          const int sp1 = S[i - 1], sq1 = S[j + 1];
          PT(T.QB, d, i) = qbij;
          PT(T.QBI, d, i) = type ? qbij * Xc->mismatchI[tr][sq1][sp1] : 0.0;
          PT(T.QB1N, d, i) = type ? qbij * Xc->mismatch1nI[tr][sq1][sp1] : 0.0;
          PT(T.QBB, d, i) = (type && tr > 2) ? qbij * xTAU : qbij;
          double m1 = PT(T.QM1, d - 1, i) * Xc->MLbase;
          if (type) m1 += qbij * sfx_mlstem(Xc, type, i > 1 ? sp1 : -1, j < W ? sq1 : -1);
          PT(T.QM1, d, i) = m1;
          double m = m1;
          {
            const int n = W - d, kk = sf_pff_parts(NT, n);
            for (int q = 0; q < kk; q++) m += ps1[q * n + i - 1];
          }
          PT(T.QM, d, i) = m;
        }
      }
    };
    for (int d = SFD_TURN + 1; d < W; d += 2) {  // even d -> Ha, odd d -> Hb
      inside_sums(d);
      __syncthreads();
      inside_step(d, Ha, std::true_type{});
      __syncthreads();
      if (d + 1 < W) inside_sums(d + 1);
      __syncthreads();
      if (d + 1 < W) {
        inside_step(d + 1, Hb, std::true_type{});
      }
      __syncthreads();
    }

    // ================= exterior =================
    if (tid == 0) { q5[0] = 1.0; q3[W + 1] = 1.0; }
    __syncthreads();
    for (int j = 1; j <= W; j++) {
      double val = 0.0;
      const int i = tid + 1;
      if (i + SFD_TURN + 1 <= j) {
        const int type = OWNT(D, i, j);
        if (type) val = q5[i - 1] * PT(T.QB, j - i, i) * sfx_extloop(X, type, i > 1 ? S[i - 1] : -1, j < W ? S[j + 1] : -1);
      }
      val = sf_block_sum(val, red);
      if (tid == 0) q5[j] = q5[j - 1] + val;
      __syncthreads();
    }
    for (int i = W; i >= 1; i--) {
      double val = 0.0;
      const int j = tid + 1;
      if (j <= W && i + SFD_TURN + 1 <= j) {
        const int type = OWNT(D, i, j);
        if (type) val = PT(T.QB, j - i, i) * sfx_extloop(X, type, i > 1 ? S[i - 1] : -1, j < W ? S[j + 1] : -1) * q3[j + 1];
      }
      val = sf_block_sum(val, red);
      if (tid == 0) q3[i] = q3[i + 1] + val;
      __syncthreads();
    }
    const double Z = q5[W];
    if (centroid)
      for (int x = tid; x <= W; x += NT) centroid[(size_t)k * W1 + x] = (x < W) ? '.' : 0;

    // ================= outside (descending d) =================
#pragma unroll
    for (int u = 0; u < 27; u++) { Ha[u] = 0.0; Hb[u] = 0.0; }
    double mbd = 0.0, cd = 0.0;
    // (the outside pass's two O(W - d) sums — A1 over the closers (k, j) and the multiloop-stem sum over the closers' spans —
    // read diagonals > d only, all complete: split over the idle threads like the inside sums)
    auto outside_sums = [&](const int d) {
      const SfDevParams *const Dc = sf_const_base(D);
      const int n = W - d, kk = sf_pff_parts(NT, n);
      const int p = tid / n, c = tid - p * n;
      if (p < kk) {
        const int i = c + 1, j = i + d;
        double sa = 0.0, sm = 0.0;
        if (i > 1) {
          const int lo = d + SFD_TURN + 3, hi = sfd_min(W - 1, j - 1);
          const int per = (hi - lo + kk) / kk;
          const int d0 = lo + p * per, d1 = sfd_min(d0 + per - 1, hi);
#pragma unroll SF_PFF_UNROLL
          for (int dd = d0; dd <= d1; dd++) sa += PT(T.OBW, dd, j - dd) * PT(T.QM, dd - d - 2, j - dd + 1);
        }
        if (i > 1 && j < W && OWNT(Dc, i, j) && PT(T.QB, d, i) != 0.0) {
          const int lo = d + 2, hi = sfd_min(W - 1, W - i);
          const int per = (hi - lo + kk) / kk;
          const int d0 = lo + p * per, d1 = sfd_min(d0 + per - 1, hi);
#pragma unroll SF_PFF_UNROLL
          for (int dd = d0; dd <= d1; dd++) {
            const double q0 = PT(T.QM, dd - d - 2, j + 1);
            sm += PT(T.A1, dd, i) * (mlb[dd - d - 1] + q0) + PT(T.A0, dd, i) * q0;
          }
        }
        ps0[tid] = sa; ps1[tid] = sm;
      }
    };
    auto outside_step = [&](const int d, double(&G)[27], auto GT) {
      // (parameter blocks through sf_const_base: their tables addressed "one scalar base + offset" — the compiler otherwise keeps
      // one hoisted 64-bit address per table row in scalar registers it does not have: thousands of v_readlane restores)
      const SfDevParamsPF *const Xc = sf_const_base(X);
      const SfDevParams *const Dc = sf_const_base(D);
      constexpr bool GU = decltype(GT)::value;  // true: some enclosing diagonals d+2+u fall outside the table
      const int i = v - (d >> 1), j = i + d;
      const bool valid = (i >= 1) && (j <= W);
      if (valid) {
        const bool inner = (i > 1) && (j < W);  // (i,j) can be enclosed by another pair
        // largest loop size whose enclosing diagonal d+2+u still exists
        const int uomax = GU ? sfd_min(SFD_MAXLOOP, W - 1 - d - 2) : SFD_MAXLOOP;
        if (!inner) {
#pragma unroll
          for (int u = 0; u < 27; u++) G[u] = 0.0;
        } else {
#pragma unroll
          for (int u = 30; u >= 6; --u)
            if (!GU || u <= uomax) {
              const int dd = d + 2 + u;
              const double e1 = (i - 3 >= 1 && j + u - 1 <= W) ? PT(T.OBI, dd, i - 3) : 0.0;      // u1 = 2
              const double e2 = (i - u + 1 >= 1 && j + 3 <= W) ? PT(T.OBI, dd, i - u + 1) : 0.0;  // u2 = 2
              G[u - 4] = G[u - 6] + (e1 + e2) * Xc->ninio[u - 4];
            } else {
              G[u - 4] = 0.0;
            }
          if (!GU || uomax >= 5) {
            const double e1 = (i - 3 >= 1 && j + 4 <= W) ? PT(T.OBI, d + 7, i - 3) : 0.0;
            const double e2 = (i - 4 >= 1 && j + 3 <= W) ? PT(T.OBI, d + 7, i - 4) : 0.0;
            G[1] = (e1 + e2) * Xc->ninio[1];
          } else G[1] = 0.0;
          G[0] = ((!GU || uomax >= 4) && i - 3 >= 1 && j + 3 <= W) ? PT(T.OBI, d + 6, i - 3) * Xc->ninio[0] : 0.0;
        }
        // helper tables for multiloops closed by (k, j), k < i: indexed by the closer's span dd = j - k
        double a0 = 0.0, a1 = 0.0;
        if (i > 1) {
          a0 = PT(T.A0, d + 1, i - 1) * Xc->MLbase + PT(T.OBW, d + 1, i - 1);
          {
            const int n = W - d, kk = sf_pff_parts(NT, n);
            for (int q = 0; q < kk; q++) a1 += ps0[q * n + i - 1];
          }
        }
        PT(T.A0, d, i) = a0;
        PT(T.A1, d, i) = a1;
        const int type = OWNT(Dc, i, j);
        const double qbij = PT(T.QB, d, i);
        double o = 0.0;
        if (type && qbij != 0.0) {
          o = q5[i - 1] * q3[j + 1] * sfx_extloop(Xc, type, i > 1 ? S[i - 1] : -1, j < W ? S[j + 1] : -1);
          if (inner) {
            const int rt = sfd_rtype(type);
            const int sp1 = S[i - 1], sq1 = S[j + 1];
            // special loops; (kk,l) is the enclosing pair, its type tk, its inner neighbours S[kk+1], S[l-1]
#define OBV(kk, l) (((kk) >= 1 && (l) <= W) ? PT(T.OB, (l) - (kk), (kk)) : 0.0)
#define TK(kk, l) (((kk) >= 1 && (l) <= W) ? Dc->pair[S[kk]][S[l]] : 0)
#define SS(x) S[(x) < 0 ? 0 : ((x) > W + 1 ? W + 1 : (x))]  /* neighbours of pairs that may not exist */
            {
              const int tk = TK(i - 1, j + 1);
              o += OBV(i - 1, j + 1) * Xc->stack[tk][rt];
            }
            {
              const int ta = TK(i - 1, j + 2), tb = TK(i - 2, j + 1);
              o += (OBV(i - 1, j + 2) * Xc->stack[ta][rt] + OBV(i - 2, j + 1) * Xc->stack[tb][rt]) * Xc->bulge[1];
            }
            {
              const int tk = TK(i - 2, j + 2);
              o += OBV(i - 2, j + 2) * Xc->int11[tk][rt][S[i - 1]][S[j + 1]];
            }
            {
              const int ta = TK(i - 2, j + 3);  // u1 = 1, u2 = 2
              o += OBV(i - 2, j + 3) * Xc->int21[ta][rt][S[i - 1]][sq1][SS(j + 2)];
              const int tb = TK(i - 3, j + 2);  // u1 = 2, u2 = 1
              o += OBV(i - 3, j + 2) * Xc->int21[rt][tb][sq1][SS(i - 2)][sp1];
            }
            {
              const int tk = TK(i - 3, j + 3);
              o += OBV(i - 3, j + 3) * Xc->int22[tk][rt][SS(i - 2)][sp1][sq1][SS(j + 2)];
            }
            {
              const double m23 = Xc->internal_loop[5] * Xc->ninio[1] * Xc->mismatch23I[rt][sq1][sp1];
              const int ta = TK(i - 3, j + 4), tb = TK(i - 4, j + 3);
              o += m23 * (OBV(i - 3, j + 4) * Xc->mismatch23I[ta][SS(i - 2)][SS(j + 3)] +
                          OBV(i - 4, j + 3) * Xc->mismatch23I[tb][SS(i - 3)][SS(j + 2)]);
            }
#undef OBV
#undef TK
#undef SS
            double gb = 0.0, g1 = 0.0, gg = 0.0;
#pragma unroll
            for (int u = 2; u <= 30; ++u)
              if (!GU || u <= uomax) {
                const int dd = d + 2 + u;
                const double b1 = (j + 1 + u <= W) ? PT(T.OBB, dd, i - 1) : 0.0;          // u1 = 0
                const double b2 = (i - 1 - u >= 1) ? PT(T.OBB, dd, i - 1 - u) : 0.0;      // u2 = 0
                gb += (b1 + b2) * Xc->bulge[u];
                if (u >= 4) {
                  const double n1 = (i - 2 >= 1 && j + u <= W) ? PT(T.OB1N, dd, i - 2) : 0.0;  // u1 = 1
                  const double n2 = (i - u >= 1 && j + 2 <= W) ? PT(T.OB1N, dd, i - u) : 0.0;  // u2 = 1
                  g1 += (n1 + n2) * Xc->internal_loop[u] * Xc->ninio[u - 2];
                }
                if (u >= 6) gg += G[u - 4] * Xc->internal_loop[u];
              }
            o += gb * (rt > 2 ? xTAU : 1.0) + g1 * Xc->mismatch1nI[rt][sq1][sp1] + gg * Xc->mismatchI[rt][sq1][sp1];
            // (i,j) as a stem of a multiloop closed by (k,l): indexed by the span dd = l - i
            double mlsum = 0.0;
            {
              // closers (k, l) with l = i + dd <= W; dd = d+1 has an empty right part, the spans >= d + 2 come from outside_sums
              const int ddmax = sfd_min(W - 1, W - i);
              if (d + 1 <= ddmax) mlsum += PT(T.A1, d + 1, i) * mlb[0];
              const int n = W - d, kk = sf_pff_parts(NT, n);
              for (int q = 0; q < kk; q++) mlsum += ps1[q * n + i - 1];
            }
            o += mlsum * sfx_mlstem(Xc, type, sp1, sq1);
          }
        }
        {
          const int si1 = S[i + 1], sj1 = S[j - 1];
          PT(T.OB, d, i) = o;
          PT(T.OBI, d, i) = type ? o * Xc->mismatchI[type][si1][sj1] : 0.0;
          PT(T.OB1N, d, i) = type ? o * Xc->mismatch1nI[type][si1][sj1] : 0.0;
          PT(T.OBB, d, i) = (type > 2) ? o * xTAU : o;
          PT(T.OBW, d, i) = type ? o * Xc->MLclosing * sfx_mlstem(Xc, sfd_rtype(type), sj1, si1) : 0.0;
          const double p = o * qbij / Z;
          mbd += p * (1.0 - p);
          if (p > 0.5) {
            cd += 1.0 - p;
            if (centroid) { centroid[(size_t)k * W1 + i - 1] = '('; centroid[(size_t)k * W1 + j - 1] = ')'; }
          } else cd += p;
        }
      }
    };
    {
      int d = W - 1;
      if (d & 1) {
        outside_sums(d);
        __syncthreads();
        outside_step(d, Hb, std::true_type{});
        __syncthreads();
        d--;
      }
      for (; d >= SFD_TURN + 1; d -= 2) {
        outside_sums(d);
        __syncthreads();
        outside_step(d, Ha, std::true_type{});
        __syncthreads();
        if (d - 1 >= SFD_TURN + 1) outside_sums(d - 1);
        __syncthreads();
        if (d - 1 >= SFD_TURN + 1) {
          outside_step(d - 1, Hb, std::true_type{});
        }
        __syncthreads();
      }
    }
    mbd = sf_block_sum(mbd, red);
    __syncthreads();
    cd = sf_block_sum(cd, red);
    if (tid == 0) {
      if (ens_dG) ens_dG[k] = -log(Z) * X->kT / 1000.0;
      if (mean_bp_dist) mean_bp_dist[k] = 2.0 * mbd;
      if (centroid_dist) centroid_dist[k] = cd;
    }
  }
#undef PT
#undef OWNT
}

template <typename... A>
static inline void sf_pf_fast_launch(int grid, int W, hipStream_t st, A... args) {
  if (W == 120) SF_LAUNCH((sf_pf_fast_kernel<128, 120>), grid, 128, 0, st, args..., (const char *)nullptr, (int *)nullptr);
  else if (W <= 128) SF_LAUNCH((sf_pf_fast_kernel<128, 0>), grid, 128, 0, st, args..., (const char *)nullptr, (int *)nullptr);
  else if (W == 200) SF_LAUNCH((sf_pf_fast_kernel<256, 200>), grid, 256, 0, st, args..., (const char *)nullptr, (int *)nullptr);
  else SF_LAUNCH((sf_pf_fast_kernel<256, 0>), grid, 256, 0, st, args..., (const char *)nullptr, (int *)nullptr);
}
// constrained folds (a constraint row per fold); W <= 250
#define SF_PFF_HC_MAXW 250
template <typename... A>
static inline void sf_pf_fast_launch_hc(int grid, int W, hipStream_t st, A... args) {
  if (W <= 128) SF_LAUNCH((sf_pf_fast_kernel<128, 0, true>), grid, 128, 0, st, args...);
  else SF_LAUNCH((sf_pf_fast_kernel<256, 0, true>), grid, 256, 0, st, args...);
}
