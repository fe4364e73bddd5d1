// sf_pf_lds.hip.h — McCaskill partition function with every table resident in LDS (W <= SF_PFL_MAXW).
//
// Same mathematics and outputs as sf_pf_fast.hip.h / sf_pf.hip.h; replaces fc.pf() / fc.centroid() /
// fc.mean_bp_distance() for the native windows (ScanFold-Scan.py:383-389).  What is different is the order of
// evaluation, chosen so that the whole working set of one fold fits the 160 KB of one CU:
//  * cells are visited COLUMN by column (j ascending inside, descending outside; all rows of a column in
//    parallel) instead of by anti-diagonal.  In that order qm1 is a rolling vector (only column j-1 is read)
//    and the outside multiloop helpers collapse to three vectors over the closing pair's 5' end:
//      R0[i] = sum_{m>l} w(i,m) MLbase^(m-l-1)      (one multiply-add per column)
//      R1[i] = sum_{m>l} w(i,m) qm[l+1,m-1]          w(i,m) = ob[i,m] MLclosing stem'(i,m)
//      ob[k,l] += stem(k,l) sum_{i<k} ( qm[i+1,k-1] (R0[i]+R1[i]) + MLbase^(k-i-1) R1[i] )
//    so only TWO full tables remain: qb (overwritten in place by ob during the outside pass; qb[k,l] is last
//    read by the cell that replaces it) and qm.
//  * a thread owns a centre s = i+j (cell (s-j, j) of column j), so the generic interior-loop sums are carried
//    in registers from the enclosed / enclosing cell exactly as in sf_pf_fast.hip.h.
//  * qb/ob are stored column-major (a column's rows are consecutive: the lanes of a wave read consecutive
//    doubles for every interior-loop candidate), qm diagonal-major (multiloop sums run over the offset from the
//    thread's own row).  Interior-loop candidates come in two families:
//      B: the column is one of the last three, the row runs with the loop size — read from three small
//         rolling buffers that hold qb (ob) pre-multiplied by the pair's own mismatch / terminal weights;
//      A: the row is fixed per thread, the column runs with the loop size — qb (ob) times a weight taken
//         from a 25x25 table indexed by (nucleotides at the column: wave-uniform row) x (nucleotides at the
//         thread's row: 25 consecutive doubles, so the gather is bank-conflict free).
// One barrier per column in either pass.  FP64 throughout; sums are re-associated with respect to the oracle
// (agreement ~1e-12 relative).
#pragma once
#include "sf_energy.h"
#include "sf_pf.hip.h"

#define SF_PFL_NT 256
#define SF_PFL_PAD 32
#define SF_PFL_LDS_LIMIT (160 * 1024)
// A per-column table of wave-uniform values, one entry per lane, read back with v_readlane (no LDS round trip
// per use).  The table is filled outside divergent control flow.  The CPU emulation keeps it as a plain array.
#ifdef SF_EMUL
#define SF_LANE_TABLE(name, L, expr) int name[64]; for (int L = 0; L < 64; L++) name[L] = (expr)
#define SF_LANE_GET(name, idx) name[idx]
#else
// (the empty asm pins the load here, with every lane active: the compiler must not sink it into the divergent
// region where the entries are read back, or inactive lanes would hold stale values)
#define SF_LANE_TABLE(name, L, expr) int name; { const int L = threadIdx.x & 63; name = (expr); asm volatile("" : "+v"(name)); }
#define SF_LANE_GET(name, idx) __builtin_amdgcn_readlane(name, idx)
#endif
// packed neighbour codes: bits 0-9 code*25 (row offset into a 25x25 weight table), 12-16 code, 20-22 nucleotide
#define SF_PK_ROW(p) ((p) & 0x3ff)
#define SF_PK_CODE(p) (((p) >> 12) & 31)
#define SF_PK_NT(p) ((p) >> 20)

__host__ __device__ inline size_t sf_pfl_lds_bytes(int W) {
  const size_t NC = (size_t)(W - 4) * (W - 3) / 2, RP = W + 2 * SF_PFL_PAD, VW = W + 8;
  const size_t dbl = 2 * NC + 12 * RP + 3 * 625 + 8 * VW + 8 + (W + 2) + (W + 3) + 8;
  return dbl * sizeof(double) + 2 * (size_t)(W + 2) * sizeof(int) + (size_t)(W + 8);
}
static inline bool sf_pfl_supported(int W) { return W >= 16 && 2 * W - 4 < SF_PFL_NT && sf_pfl_lds_bytes(W) <= SF_PFL_LDS_LIMIT; }

template <int WT>
__global__ __launch_bounds__(SF_PFL_NT) void sf_pf_lds_kernel(const uint8_t *__restrict__ seqs, int n, int row_stride,
                                                              int Wrt, const SfDevParams *__restrict__ D,
                                                              const SfDevParamsPF *__restrict__ X,
                                                              double *__restrict__ ens_dG,
                                                              double *__restrict__ mean_bp_dist,
                                                              char *__restrict__ centroid,
                                                              double *__restrict__ centroid_dist) {
  SF_DYN_SMEM(smem);
  const int W = WT ? WT : Wrt;
  const int tid = threadIdx.x;
  const int W1 = W + 1;
  const int NC = ((W - 4) * (W - 3)) >> 1, RP = W + 2 * SF_PFL_PAD, VW = W + 8;
  double *QB = (double *)smem;    // qb, then ob: column-major, column j holds rows 1..j-4
  double *QM = QB + NC;           // qm: diagonal-major, diagonal d >= 4 holds rows 1..W-d
  double *DER = QM + NC;          // [3 kinds][4 column slots][RP rows, row r at r + PAD]
  double *FAC = DER + 12 * RP;    // [3][25][25] family-A weights
  double *QM1 = FAC + 3 * 625;    // [2][VW]
  double *RV = QM1 + 2 * VW + 8;  // R0[2], R1[2], R01[2], each VW, row r at r + 8
  double *q5 = RV + 6 * VW + 0;       // [W+2] (the 8 pad rows of RV are taken from the tail of its last vector)
  double *q3 = q5 + (W + 2);      // [W+3]
  double *red = q3 + (W + 3);     // [8]
  int *FWD = (int *)(red + 8);    // [W+2]  S[x]*5 + S[x+1]
  int *BWD = FWD + (W + 2);       // [W+2]  S[x]*5 + S[x-1]
  uint8_t *S = (uint8_t *)(BWD + (W + 2));
#define COFF(j) ((((j)-5) * ((j)-4)) >> 1)
#define DOFF(d) (((d)-4) * W - ((((d) * ((d)-1)) >> 1) - 6))
#define QBC(i, j) QB[COFF(j) + (i)-1]
#define QMD(d, i) QM[DOFF(d) + (i)-1]
#define DERP(kind, col) (DER + ((kind)*4 + ((col)&3)) * RP + SF_PFL_PAD)
  const double *mlb = X->mlbase_pow;
  const double xTAU = X->TermAU;
  const double xMLbase = X->MLbase;
  // speculative (discarded or zero-weighted) reads below may land anywhere in the tables: keep them finite
  for (int x = tid; x < 2 * NC; x += SF_PFL_NT) QB[x] = 0.0;

  for (int fold = blockIdx.x; fold < n; fold += gridDim.x) {
    const uint8_t *src = seqs + (size_t)fold * row_stride * W;
    __syncthreads();
    for (int x = tid; x < W; x += SF_PFL_NT) S[x + 1] = sf_encode_nt(src[x]);
    if (tid == 0) { S[0] = 0; S[W + 1] = 0; }
    for (int x = tid; x < 12 * RP; x += SF_PFL_NT) DER[x] = 0.0;
    for (int x = tid; x < 8 * VW + 8; x += SF_PFL_NT) QM1[x] = 0.0;  // QM1 and the six R vectors are contiguous
    __syncthreads();
    for (int x = tid; x <= W + 1; x += SF_PFL_NT) {
      const int cf = S[x] * 5 + (x <= W ? S[x + 1] : 0), cb = S[x] * 5 + (x >= 1 ? S[x - 1] : 0);
      FWD[x] = cf * 25 | cf << 12 | S[x] << 20;
      BWD[x] = cb * 25 | cb << 12 | S[x] << 20;
    }
    // family-A weights, inside orientation: row = nucleotides at the column q (S[q], S[q+1]), entry = the
    // thread's row p (S[p], S[p-1]);  value = mismatch of the inner pair (p,q) towards the loop
    for (int e = tid; e < 625; e += SF_PFL_NT) {
      const int f = e / 25, b = e - f * 25;
      const int a = f / 5, a1 = f - a * 5, bs = b / 5, b1 = b - bs * 5;
      const int t = D->pair[a][bs];
      FAC[e] = t ? X->mismatchI[t][a1][b1] : 0.0;
      FAC[625 + e] = t ? X->mismatch1nI[t][a1][b1] : 0.0;
    }
    __syncthreads();

    // ================= inside: columns j ascending =================
    double H[27];
#pragma unroll
    for (int u = 0; u < 27; u++) H[u] = 0.0;
    for (int j = SFD_TURN + 2; j <= W; j++) {
      const int i = tid - j, d = j - i;
      const bool valid = (i >= 1) && (d >= SFD_TURN + 1);
      double *qm1c = QM1 + (j & 1) * VW, *qm1p = QM1 + ((j & 1) ^ 1) * VW;
      SF_LANE_TABLE(tq, L, FWD[sfd_max(j - L, 5)]);  // entry L: column j-L
      if (valid) {
        const int umax = sfd_min(SFD_MAXLOOP, d - 2 - (SFD_TURN + 1));
        const int type = D->pair[S[i]][S[j]];
        const int si1 = S[i + 1], sj1 = S[j - 1];
        const double *dI3 = DERP(0, j - 3), *d1N2 = DERP(1, j - 2), *dB1 = DERP(2, j - 1);
        const double *fI = FAC + SF_PK_CODE(BWD[i + 3]), *f1N = FAC + 625 + SF_PK_CODE(BWD[i + 2]);
        // generic interior sums of this cell from those of the enclosed cell (same thread, previous column)
        // Straight-line: family-A values are loaded speculatively (column clamped to an existing one) and
        // dropped by a select when the inner span would be < TURN+1; family-B values are 0 there by themselves.
        // Sizes beyond umax therefore stay exactly 0 and need no separate bookkeeping.
#pragma unroll
        for (int u = 30; u >= 6; --u) {
          const int q = sfd_max(j - u + 1, 5);
          const double a = QBC(i + 3, q) * fI[SF_PK_ROW(SF_LANE_GET(tq, u - 1))];
          H[u - 4] = H[u - 6] + ((u <= umax ? a : 0.0) + dI3[i + u - 1]) * X->ninio[u - 4];
        }
        {
          const int q = sfd_max(j - 4, 5);
          const double a = QBC(i + 3, q) * fI[SF_PK_ROW(SF_LANE_GET(tq, 4))];
          H[1] = ((umax >= 5 ? a : 0.0) + dI3[i + 4]) * X->ninio[1];
          H[0] = dI3[i + 3] * X->ninio[0];
        }
        double qbij = 0.0;
        if (type) {
          double z = sfx_hairpin(D, X, S, i, j, type);
          if (umax >= 0) {
            const double tau_out = type > 2 ? xTAU : 1.0;
            z += QBC(i + 1, j - 1) * X->stack[type][sfd_rtype(D->pair[si1][sj1])];
            if (umax >= 1) {
              const int ta = sfd_rtype(D->pair[si1][S[j - 2]]), tb = sfd_rtype(D->pair[S[i + 2]][sj1]);
              z += (QBC(i + 1, j - 2) * X->stack[type][ta] + QBC(i + 2, j - 1) * X->stack[type][tb]) * X->bulge[1];
            }
            if (umax >= 2) {
              const int t2r = sfd_rtype(D->pair[S[i + 2]][S[j - 2]]);
              z += QBC(i + 2, j - 2) * X->int11[type][t2r][si1][sj1];
            }
            if (umax >= 3) {
              const int ta = sfd_rtype(D->pair[S[i + 2]][S[j - 3]]), tb = sfd_rtype(D->pair[S[i + 3]][S[j - 2]]);
              z += QBC(i + 2, j - 3) * X->int21[type][ta][si1][S[j - 2]][sj1] +
                   QBC(i + 3, j - 2) * X->int21[tb][type][sj1][si1][S[i + 2]];
            }
            if (umax >= 4) {
              const int t2r = sfd_rtype(D->pair[S[i + 3]][S[j - 3]]);
              z += QBC(i + 3, j - 3) * X->int22[type][t2r][si1][S[i + 2]][S[j - 2]][sj1];
            }
            if (umax >= 5) {
              const int ta = sfd_rtype(D->pair[S[i + 3]][S[j - 4]]), tb = sfd_rtype(D->pair[S[i + 4]][S[j - 3]]);
              const double m23 = X->internal_loop[5] * X->ninio[1] * X->mismatch23I[type][si1][sj1];
              z += m23 * (QBC(i + 3, j - 4) * X->mismatch23I[ta][S[j - 3]][S[i + 2]] +
                          QBC(i + 4, j - 3) * X->mismatch23I[tb][S[j - 2]][S[i + 3]]);
            }
            double gb = 0.0, g1 = 0.0, gg = 0.0;
            const int sp = S[i + 1];  // row of the u1 = 0 bulge candidates
#pragma unroll
            for (int u = 2; u <= 30; ++u) {
              const int qb_ = sfd_max(j - 1 - u, 5);
              const double ta_ = (sp * SF_PK_NT(SF_LANE_GET(tq, u + 1)) == 6) ? 1.0 : xTAU;  // C-G / G-C: no terminal penalty
              const double ab = QBC(i + 1, qb_) * ta_;
              gb += ((u <= umax ? ab : 0.0) + dB1[i + 1 + u]) * X->bulge[u];
              if (u >= 4) {
                const int qn = sfd_max(j - u, 5);
                const double an = QBC(i + 2, qn) * f1N[SF_PK_ROW(SF_LANE_GET(tq, u))];
                g1 += ((u <= umax ? an : 0.0) + d1N2[i + u]) * (X->internal_loop[u] * X->ninio[u - 2]);
              }
              if (u >= 6) gg += H[u - 4] * X->internal_loop[u];
            }
            z += gb * tau_out + g1 * X->mismatch1nI[type][si1][sj1] + gg * X->mismatchI[type][si1][sj1];
          }
          double ml = 0.0;
          {
            // sum_a qm[i+1,i+a-1] qm1[i+a,j-1], a = 6..d-5; eight terms per trip, the overshoot reads rows of
            // qm1 that are still 0 (rows > j-5 of column j-1)
            double ml1 = 0.0;
            const double *qmr = QM + i, *q1 = qm1p + i;
            int off = 0, st = W - 4;  // DOFF(a-2) and its increment, a = 6
            for (int a = SFD_TURN + 3; a <= d - SFD_TURN - 2; a += 8) {
              double t0 = 0.0, t1 = 0.0;
#pragma unroll
              for (int t = 0; t < 8; t += 2) {
                t0 += qmr[off] * q1[a + t];
                off += st--;
                t1 += qmr[off] * q1[a + t + 1];
                off += st--;
              }
              ml += t0;
              ml1 += t1;
            }
            ml += ml1;
          }
          z += ml * X->MLclosing * sfx_mlstem(X, sfd_rtype(type), sj1, si1);
          qbij = z;
        }
        {
          const int tr = sfd_rtype(type);
          const int sp1 = S[i - 1], sq1 = S[j + 1];
          QBC(i, j) = qbij;
          DERP(0, j)[i] = type ? qbij * X->mismatchI[tr][sq1][sp1] : 0.0;
          DERP(1, j)[i] = type ? qbij * X->mismatch1nI[tr][sq1][sp1] : 0.0;
          DERP(2, j)[i] = (type && tr > 2) ? qbij * xTAU : qbij;
          double m1 = qm1p[i] * xMLbase;
          if (type) m1 += qbij * sfx_mlstem(X, type, i > 1 ? sp1 : -1, j < W ? sq1 : -1);
          qm1c[i] = m1;
        }
      }
      __syncthreads();
      if (valid) {
        // qm[i,j] = sum_{a>=0} MLbase^a qm1[i+a,j] + sum_{a>=5} qm[i,i+a-1] qm1[i+a,j]
        double m = qm1c[i];
        const int amax = d - SFD_TURN - 1;
        for (int a = 1; a <= sfd_min(amax, 4); a++) m += mlb[a] * qm1c[i + a];
        {
          // eight terms per trip; the overshoot reads rows > j-4 of qm1 (column j), which are 0
          double m2 = 0.0;
          const double *qmr = QM + i - 1, *q1 = qm1c + i;
          int off = 0, st = W - 4;  // DOFF(a-1) and its increment, a = 5
          for (int a = 5; a <= amax; a += 8) {
            double t0 = 0.0, t1 = 0.0;
#pragma unroll
            for (int t = 0; t < 8; t += 2) {
              t0 += (mlb[a + t] + qmr[off]) * q1[a + t];
              off += st--;
              t1 += (mlb[a + t + 1] + qmr[off]) * q1[a + t + 1];
              off += st--;
            }
            m += t0;
            m2 += t1;
          }
          m += m2;
        }
        QMD(d, i) = m;
      }
    }
    __syncthreads();

    // ================= exterior =================
    if (tid == 0) { q5[0] = 1.0; q3[W + 1] = 1.0; }
    __syncthreads();
    for (int j = 1; j <= W; j++) {
      double val = 0.0;
      const int i = tid + 1;
      if (i + SFD_TURN + 1 <= j) {
        const int type = D->pair[S[i]][S[j]];
        if (type) val = q5[i - 1] * QBC(i, j) * sfx_extloop(X, type, i > 1 ? S[i - 1] : -1, j < W ? S[j + 1] : -1);
      }
      val = sf_block_sum(val, red);
      if (tid == 0) q5[j] = q5[j - 1] + val;
      __syncthreads();
    }
    for (int i = W; i >= 1; i--) {
      double val = 0.0;
      const int j = tid + 1;
      if (j <= W && i + SFD_TURN + 1 <= j) {
        const int type = D->pair[S[i]][S[j]];
        if (type) val = QBC(i, j) * sfx_extloop(X, type, i > 1 ? S[i - 1] : -1, j < W ? S[j + 1] : -1) * q3[j + 1];
      }
      val = sf_block_sum(val, red);
      if (tid == 0) q3[i] = q3[i + 1] + val;
      __syncthreads();
    }
    const double Z = q5[W];
    if (centroid)
      for (int x = tid; x <= W; x += SF_PFL_NT) centroid[(size_t)fold * W1 + x] = (x < W) ? '.' : 0;
    // family-A weights, outside orientation: row = nucleotides at the enclosing pair's column l' (S[l'],
    // S[l'-1]), entry = its row k' (S[k'], S[k'+1]); third table = multiloop closing weight of (k', l')
    for (int e = tid; e < 625; e += SF_PFL_NT) {
      const int b = e / 25, f = e - b * 25;
      const int a = f / 5, a1 = f - a * 5, bs = b / 5, b1 = b - bs * 5;
      const int t = D->pair[a][bs];
      FAC[e] = t ? X->mismatchI[t][a1][b1] : 0.0;
      FAC[625 + e] = t ? X->mismatch1nI[t][a1][b1] : 0.0;
      FAC[1250 + e] = t ? X->MLclosing * sfx_mlstem(X, sfd_rtype(t), b1, a1) : 0.0;
    }
    __syncthreads();

    // ================= outside: columns l descending =================
#pragma unroll
    for (int u = 0; u < 27; u++) H[u] = 0.0;
    double mbd = 0.0, cd = 0.0;
    for (int l = W; l >= SFD_TURN + 2; l--) {
      const int k = tid - l, d = l - k;
      const bool valid = (k >= 1) && (d >= SFD_TURN + 1);
      const double *R1c = RV + (2 + (l & 1)) * VW, *R01c = RV + (4 + (l & 1)) * VW, *R0c = RV + (l & 1) * VW;
      double *R0n = RV + ((l & 1) ^ 1) * VW, *R1n = RV + (2 + ((l & 1) ^ 1)) * VW, *R01n = RV + (4 + ((l & 1) ^ 1)) * VW;
      SF_LANE_TABLE(tl, L, BWD[sfd_min(l + L, W)]);  // entry L: column l+L
      if (valid) {
        const bool inner = (k > 1) && (l < W);
        const double *dI3 = DERP(0, l + 3), *d1N2 = DERP(1, l + 2), *dB1 = DERP(2, l + 1);
        const bool c3 = l + 3 <= W, c2 = l + 2 <= W;  // the column exists (its slot holds outside values)
        const bool r3 = k - 3 >= 1, r2 = k - 2 >= 1;   // the row exists
        const int kr3 = r3 ? k - 3 : 1, kr2 = r2 ? k - 2 : 1;  // rows for speculative reads
        const double *fI = FAC + SF_PK_CODE(FWD[kr3]), *f1N = FAC + 625 + SF_PK_CODE(FWD[kr2]);
        if (!inner) {
#pragma unroll
          for (int u = 0; u < 27; u++) H[u] = 0.0;
        } else {
#pragma unroll
          for (int u = 30; u >= 6; --u) {
            const int lp = sfd_min(l + u - 1, W);
            const double a = QBC(kr3, lp) * fI[SF_PK_ROW(SF_LANE_GET(tl, u - 1))];
            const double e1 = (r3 && l + u - 1 <= W) ? a : 0.0;  // u1 = 2
            const double e2 = c3 ? dI3[k - u + 1] : 0.0;         // u2 = 2
            H[u - 4] = H[u - 6] + (e1 + e2) * X->ninio[u - 4];
          }
          {
            const int lp = sfd_min(l + 4, W);
            const double a = QBC(kr3, lp) * fI[SF_PK_ROW(SF_LANE_GET(tl, 4))];
            const double e1 = (r3 && l + 4 <= W) ? a : 0.0;
            const double e2 = c3 ? dI3[k - 4] : 0.0;
            H[1] = (e1 + e2) * X->ninio[1];
          }
          H[0] = c3 ? dI3[k - 3] * X->ninio[0] : 0.0;
        }
        const int type = D->pair[S[k]][S[l]];
        const double qbkl = QBC(k, l);
        double o = 0.0;
        if (type && qbkl != 0.0) {
          o = q5[k - 1] * q3[l + 1] * sfx_extloop(X, type, k > 1 ? S[k - 1] : -1, l < W ? S[l + 1] : -1);
          if (inner) {
            const int rt = sfd_rtype(type);
            const int sp1 = S[k - 1], sq1 = S[l + 1];
#define OBV(kk, ll) (((kk) >= 1 && (ll) <= W) ? QBC((kk) >= 1 ? (kk) : 1, (ll) <= W ? (ll) : W) : 0.0)
#define TK(kk, ll) (((kk) >= 1 && (ll) <= W) ? D->pair[S[kk]][S[ll]] : 0)
#define SS(x) S[(x) < 0 ? 0 : ((x) > W + 1 ? W + 1 : (x))] /* neighbours of pairs that may not exist */
            {
              const int tk = TK(k - 1, l + 1);
              o += OBV(k - 1, l + 1) * X->stack[tk][rt];
            }
            {
              const int ta = TK(k - 1, l + 2), tb = TK(k - 2, l + 1);
              o += (OBV(k - 1, l + 2) * X->stack[ta][rt] + OBV(k - 2, l + 1) * X->stack[tb][rt]) * X->bulge[1];
            }
            {
              const int tk = TK(k - 2, l + 2);
              o += OBV(k - 2, l + 2) * X->int11[tk][rt][S[k - 1]][S[l + 1]];
            }
            {
              const int ta = TK(k - 2, l + 3);  // u1 = 1, u2 = 2
              o += OBV(k - 2, l + 3) * X->int21[ta][rt][S[k - 1]][sq1][SS(l + 2)];
              const int tb = TK(k - 3, l + 2);  // u1 = 2, u2 = 1
              o += OBV(k - 3, l + 2) * X->int21[rt][tb][sq1][SS(k - 2)][sp1];
            }
            {
              const int tk = TK(k - 3, l + 3);
              o += OBV(k - 3, l + 3) * X->int22[tk][rt][SS(k - 2)][sp1][sq1][SS(l + 2)];
            }
            {
              const double m23 = X->internal_loop[5] * X->ninio[1] * X->mismatch23I[rt][sq1][sp1];
              const int ta = TK(k - 3, l + 4), tb = TK(k - 4, l + 3);
              o += m23 * (OBV(k - 3, l + 4) * X->mismatch23I[ta][SS(k - 2)][SS(l + 3)] +
                          OBV(k - 4, l + 3) * X->mismatch23I[tb][SS(k - 3)][SS(l + 2)]);
            }
#undef OBV
#undef TK
#undef SS
            double gb = 0.0, g1 = 0.0, gg = 0.0;
#pragma unroll
            for (int u = 2; u <= 30; ++u) {
              const int lb = sfd_min(l + 1 + u, W);
              const double ab = QBC(k - 1, lb) * ((sp1 * SF_PK_NT(SF_LANE_GET(tl, u + 1)) == 6) ? 1.0 : xTAU);
              const double b1 = (l + 1 + u <= W) ? ab : 0.0;  // u1 = 0
              const double b2 = dB1[k - 1 - u];               // u2 = 0
              gb += (b1 + b2) * X->bulge[u];
              if (u >= 4) {
                const int ln = sfd_min(l + u, W);
                const double an = QBC(kr2, ln) * f1N[SF_PK_ROW(SF_LANE_GET(tl, u))];
                const double n1 = (r2 && l + u <= W) ? an : 0.0;  // u1 = 1
                const double n2 = c2 ? d1N2[k - u] : 0.0;         // u2 = 1
                g1 += (n1 + n2) * (X->internal_loop[u] * X->ninio[u - 2]);
              }
              if (u >= 6) gg += H[u - 4] * X->internal_loop[u];
            }
            o += gb * (rt > 2 ? xTAU : 1.0) + g1 * X->mismatch1nI[rt][sq1][sp1] + gg * X->mismatchI[rt][sq1][sp1];
            // (k,l) as a stem of a multiloop closed by (i,m), i < k, m > l
            double ms = 0.0;
            for (int a = 1; a <= sfd_min(k - 1, 5); a++) ms += mlb[a - 1] * R1c[k - a];
            {
              // eight closers per trip; the overshoot reads rows <= 0 of R1 / R01, which are 0
              double ms2 = 0.0;
              const double *qmr = QM + k, *r1p = R1c + k, *r01p = R01c + k;
              int off = -6, st = W - 5;  // DOFF(a-2) - a and its increment, a = 6
              for (int a = 6; a <= k - 1; a += 8) {
                double t0 = 0.0, t1 = 0.0;
#pragma unroll
                for (int t = 0; t < 8; t += 2) {
                  t0 += mlb[a + t - 1] * r1p[-a - t] + qmr[off] * r01p[-a - t];
                  off += st--;
                  t1 += mlb[a + t] * r1p[-a - t - 1] + qmr[off] * r01p[-a - t - 1];
                  off += st--;
                }
                ms += t0;
                ms2 += t1;
              }
              ms += ms2;
            }
            o += ms * sfx_mlstem(X, type, sp1, sq1);
          }
        }
        {
          const int si1 = S[k + 1], sj1 = S[l - 1];
          QBC(k, l) = o;
          DERP(0, l)[k] = type ? o * X->mismatchI[type][si1][sj1] : 0.0;
          DERP(1, l)[k] = type ? o * X->mismatch1nI[type][si1][sj1] : 0.0;
          DERP(2, l)[k] = (type > 2) ? o * xTAU : o;
          const double w = type ? o * X->MLclosing * sfx_mlstem(X, sfd_rtype(type), sj1, si1) : 0.0;
          const double r0 = w + xMLbase * R0c[k];
          // R1 of the next column l-1: closers (k, m), m >= l+5, right part qm[l, m-1]
          double r1 = 0.0;
          const double *fW = FAC + 1250 + SF_PK_CODE(FWD[k]);
          {
            double r1b = 0.0;
            int m = l + SFD_TURN + 2;
            for (; m + 3 <= W; m += 4) {
              r1 += QBC(k, m) * fW[SF_PK_ROW(BWD[m])] * QMD(m - 1 - l, l) +
                    QBC(k, m + 2) * fW[SF_PK_ROW(BWD[m + 2])] * QMD(m + 1 - l, l);
              r1b += QBC(k, m + 1) * fW[SF_PK_ROW(BWD[m + 1])] * QMD(m - l, l) +
                     QBC(k, m + 3) * fW[SF_PK_ROW(BWD[m + 3])] * QMD(m + 2 - l, l);
            }
            for (; m <= W; m++) r1 += QBC(k, m) * fW[SF_PK_ROW(BWD[m])] * QMD(m - 1 - l, l);
            r1 += r1b;
          }
          R0n[k] = r0;
          R1n[k] = r1;
          R01n[k] = r0 + r1;
          const double p = o * qbkl / Z;
          mbd += p * (1.0 - p);
          if (p > 0.5) {
            cd += 1.0 - p;
            if (centroid) { centroid[(size_t)fold * W1 + k - 1] = '('; centroid[(size_t)fold * W1 + l - 1] = ')'; }
          } else cd += p;
        }
      }
      __syncthreads();
    }
    mbd = sf_block_sum(mbd, red);
    __syncthreads();
    cd = sf_block_sum(cd, red);
    if (tid == 0) {
      if (ens_dG) ens_dG[fold] = -log(Z) * X->kT / 1000.0;
      if (mean_bp_dist) mean_bp_dist[fold] = 2.0 * mbd;
      if (centroid_dist) centroid_dist[fold] = cd;
    }
  }
#undef COFF
#undef DOFF
#undef QBC
#undef QMD
#undef DERP
}

static inline hipError_t sf_pfl_configure() {
  hipError_t e = hipFuncSetAttribute((const void *)sf_pf_lds_kernel<120>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     SF_PFL_LDS_LIMIT);
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute((const void *)sf_pf_lds_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize,
                             SF_PFL_LDS_LIMIT);
}

template <typename... A>
static inline void sf_pf_lds_launch(int grid, int W, hipStream_t st, A... args) {
  const size_t lds = sf_pfl_lds_bytes(W);
  if (W == 120) SF_LAUNCH((sf_pf_lds_kernel<120>), grid, SF_PFL_NT, lds, st, args...);
  else SF_LAUNCH((sf_pf_lds_kernel<0>), grid, SF_PFL_NT, lds, st, args...);
}
