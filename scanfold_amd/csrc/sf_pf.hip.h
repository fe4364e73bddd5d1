// sf_pf.hip.h — McCaskill partition function, base-pair probabilities, centroid and mean base-pair distance.
//
// Replaces fc.pf() / RNA.pf_fold() / fc.centroid() / fc.mean_bp_distance() for the native window
// (ScanFold-Scan.py:383-384,388-389,395,400-401).  FP64 throughout, no per-nucleotide scaling (|F| stays far
// below the exponent range for W <= SF_MAX_W at 37 C with Turner-type parameters).  One workgroup per
// sequence, thread t owns cell (i = t+1, j = i+d); tables are diagonal-major T(d,i) in device memory.
//
// Inside (ascending d), unambiguous decomposition with dangles=2:
//   qb[i,j]  = hairpin + sum_{p,q} int(i,j,p,q) qb[p,q] + MLclosing*stem'(i,j) * sum_u qm[i+1,u-1] qm1[u,j-1]
//   qm1[i,j] = qm1[i,j-1]*MLbase + qb[i,j]*stem(i,j)
//   qm[i,j]  = sum_u (MLbase^(u-i) + qm[i,u-1]) qm1[u,j]
//   q5[j]    = q5[j-1] + sum_i q5[i-1] qb[i,j] ext(i,j)          q3 mirrored
// Outside (descending d), O(W^3) through two helper tables over enclosing multiloop closers (k,l):
//   w(k,l)   = ob[k,l]*MLclosing*stem'(k,l)
//   A0[i,l]  = sum_{k<i} w(k,l) MLbase^(i-k-1)  = A0[i-1,l]*MLbase + w(i-1,l)
//   A1[i,l]  = sum_{k<i} w(k,l) qm[k+1,i-1]
//   ob[i,j]  = q5[i-1] q3[j+1] ext(i,j) + sum_{k,l} ob[k,l] int(k,l,i,j)
//            + stem(i,j) * sum_{l>j} ( A1[i,l] (MLbase^(l-1-j) + qm[j+1,l-1]) + A0[i,l] qm[j+1,l-1] )
//   p[i,j]   = ob[i,j] qb[i,j] / Z
#pragma once
#include "sf_energy.h"

#define SF_PF_NTABLES 7
#define SF_PF_SCRATCH_DOUBLES(W) (SF_PF_NTABLES * (size_t)(W) * ((W) + 1))

__device__ inline double sf_block_sum(double v, double *red) {
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  const int tid = threadIdx.x;
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  double r = 0.0;
  const int nw = ((int)blockDim.x + 63) >> 6;
  for (int w = 0; w < nw; w++) r += red[w];
  return r;
}

__global__ void sf_pf_kernel(const uint8_t *__restrict__ seqs, int n, int row_stride, int W,
                             const SfDevParams *__restrict__ D, const SfDevParamsPF *__restrict__ X,
                             double *__restrict__ scratch, double *__restrict__ ens_dG,
                             double *__restrict__ mean_bp_dist, char *__restrict__ centroid,
                             double *__restrict__ centroid_dist, const char *__restrict__ cons_rows,
                             int *__restrict__ status) {
  // cons_rows: item k's hard constraint = W characters at cons_rows + k*W (fc.hc_add_from_db before fc.pf(),
  // ScanFold-Scan.py:405-417); null = none
  __shared__ char hcC[SF_MAX_W + 2];
  __shared__ int16_t hcP[SF_MAX_W + 2], hcE[SF_MAX_W + 2], hcStk[SF_MAX_W + 2];
  __shared__ int hcBad;
  __shared__ uint8_t S[SF_MAX_W + 2];
  __shared__ double q5[SF_MAX_W + 2];
  __shared__ double q3[SF_MAX_W + 3];
  __shared__ double red[8];
  const int tid = threadIdx.x;
  const int nthreads = blockDim.x;
  const int W1 = W + 1;
  const size_t TS = (size_t)W * W1;
  double *QB = scratch + (size_t)blockIdx.x * SF_PF_SCRATCH_DOUBLES(W);
  double *QM = QB + TS, *QM1 = QM + TS, *OB = QM1 + TS, *OBW = OB + TS, *A0 = OBW + TS, *A1 = A0 + TS;
#define PT(tab, d, i) tab[(size_t)(d)*W1 + (i)]
  const double *mlb = X->mlbase_pow;

  for (int k = blockIdx.x; k < n; k += gridDim.x) {
    const uint8_t *src = seqs + (size_t)k * row_stride * W;
    __syncthreads();
    for (int x = tid; x < W; x += nthreads) S[x + 1] = sf_encode_nt(src[x]);
    if (tid == 0) { S[0] = 0; S[W + 1] = 0; }
    for (size_t x = tid; x < (size_t)4 * W1 && x < TS; x += nthreads) { QB[x] = 0.0; QM[x] = 0.0; QM1[x] = 0.0; }
    SfHc hc;
    hc.c = cons_rows ? hcC : nullptr; hc.partner = hcP; hc.encl = hcE;
    if (cons_rows && tid == 0) hcBad = sf_hc_parse(cons_rows + (size_t)k * W, W, hcC, hcP, hcE, hcStk);
    __syncthreads();
    if (cons_rows && hcBad) {  // unbalanced brackets
      if (tid == 0 && status) atomicOr(status, 2);
      continue;
    }
    auto PTY = [&](int a, int b) -> int {
      const bool ok = b - a <= D->max_pair_dist;
      return sf_hc_type(hc, ok ? D->pair[S[a]][S[b]] : 0, a, b, ok);
    };

    // ---------------- inside ----------------
    for (int d = SFD_TURN + 1; d < W; d++) {
      const int i = tid + 1, j = i + d;
      if (j <= W) {
        const int type = PTY(i, j);
        double qbij = 0.0;
        if (type) {
          double z = sfx_hairpin(D, X, S, i, j, type);
          const int umax = sfd_min(SFD_MAXLOOP, d - 2 - (SFD_TURN + 1));
          const int si1 = S[i + 1], sj1 = S[j - 1];
          for (int u1 = 0; u1 <= umax; u1++) {
            const int p = i + 1 + u1;
            for (int u2 = 0; u2 <= umax - u1; u2++) {
              const int q = j - 1 - u2;
              const int t2 = PTY(p, q);
              if (!t2) continue;
              z += sfx_intloop(X, u1, u2, type, sfd_rtype(t2), si1, sj1, S[p - 1], S[q + 1]) * PT(QB, q - p, p);
            }
          }
          double ml = 0.0;
          for (int u = i + 2 + SFD_TURN; u <= j - SFD_TURN - 2; u++) ml += PT(QM, u - i - 2, i + 1) * PT(QM1, j - 1 - u, u);
          z += ml * X->MLclosing * sfx_mlstem(X, sfd_rtype(type), sj1, si1);
          qbij = z;
        }
        PT(QB, d, i) = qbij;
        double m1 = PT(QM1, d - 1, i) * X->MLbase;
        if (type) m1 += qbij * sfx_mlstem(X, type, i > 1 ? S[i - 1] : -1, j < W ? S[j + 1] : -1);
        PT(QM1, d, i) = m1;
        double m = m1;  // u == i: MLbase^0 * qm1[i,j]
        for (int u = i + 1; u + SFD_TURN + 1 <= j; u++) m += (mlb[u - i] + PT(QM, u - 1 - i, i)) * PT(QM1, j - u, u);
        PT(QM, d, i) = m;
      }
      __syncthreads();
    }

    // ---------------- exterior ----------------
    if (tid == 0) { q5[0] = 1.0; q3[W + 1] = 1.0; }
    __syncthreads();
    for (int j = 1; j <= W; j++) {
      double v = 0.0;
      const int i = tid + 1;
      if (i + SFD_TURN + 1 <= j) {
        const int type = PTY(i, j);
        if (type) v = q5[i - 1] * PT(QB, j - i, i) * sfx_extloop(X, type, i > 1 ? S[i - 1] : -1, j < W ? S[j + 1] : -1);
      }
      v = sf_block_sum(v, red);
      if (tid == 0) q5[j] = q5[j - 1] + v;
      __syncthreads();
    }
    for (int i = W; i >= 1; i--) {
      double v = 0.0;
      const int j = tid + 1;
      if (j <= W && i + SFD_TURN + 1 <= j) {
        const int type = PTY(i, j);
        if (type) v = PT(QB, j - i, i) * sfx_extloop(X, type, i > 1 ? S[i - 1] : -1, j < W ? S[j + 1] : -1) * q3[j + 1];
      }
      v = sf_block_sum(v, red);
      if (tid == 0) q3[i] = q3[i + 1] + v;
      __syncthreads();
    }
    const double Z = q5[W];

    // ---------------- outside ----------------
    for (int d = W - 1; d >= SFD_TURN + 1; d--) {
      const int i = tid + 1, j = i + d;
      if (j <= W) {
        // helper tables for multiloops closed by (k,j), k < i
        double a0 = 0.0, a1 = 0.0;
        if (i > 1) {
          a0 = PT(A0, d + 1, i - 1) * X->MLbase + PT(OBW, d + 1, i - 1);
          for (int kk = 1; kk <= i - 2 - SFD_TURN - 1; kk++) a1 += PT(OBW, j - kk, kk) * PT(QM, i - kk - 2, kk + 1);
        }
        PT(A0, d, i) = a0;
        PT(A1, d, i) = a1;
        const int type = PTY(i, j);
        double o = 0.0, ow = 0.0;
        const double qbij = PT(QB, d, i);
        if (type && qbij != 0.0) {
          o = q5[i - 1] * q3[j + 1] * sfx_extloop(X, type, i > 1 ? S[i - 1] : -1, j < W ? S[j + 1] : -1);
          if (i > 1 && j < W) {
            const int rt = sfd_rtype(type);
            const int sp1 = S[i - 1], sq1 = S[j + 1];
            const int u1max = sfd_min(SFD_MAXLOOP, i - 2);
            for (int u1 = 0; u1 <= u1max; u1++) {
              const int kk = i - 1 - u1;
              const int u2max = sfd_min(SFD_MAXLOOP - u1, W - j - 1);
              for (int u2 = 0; u2 <= u2max; u2++) {
                const int l = j + 1 + u2;
                const int tk = PTY(kk, l);
                if (!tk) continue;
                o += PT(OB, l - kk, kk) * sfx_intloop(X, u1, u2, tk, rt, S[kk + 1], S[l - 1], sp1, sq1);
              }
            }
            double mlsum = 0.0;
            for (int l = j + 1; l <= W; l++) {
              const double qmr = (l - j - 2 >= 0) ? PT(QM, l - j - 2, j + 1) : 0.0;
              mlsum += PT(A1, l - i, i) * (mlb[l - 1 - j] + qmr) + PT(A0, l - i, i) * qmr;
            }
            o += mlsum * sfx_mlstem(X, type, sp1, sq1);
          }
          ow = o * X->MLclosing * sfx_mlstem(X, sfd_rtype(type), S[j - 1], S[i + 1]);
        }
        PT(OB, d, i) = o;
        PT(OBW, d, i) = ow;
      }
      __syncthreads();
    }

    // ---------------- probabilities -> centroid, distances ----------------
    if (centroid)
      for (int x = tid; x <= W; x += nthreads) centroid[(size_t)k * W1 + x] = (x < W) ? '.' : 0;
    __syncthreads();
    double mbd = 0.0, cd = 0.0;
    for (int d = SFD_TURN + 1; d < W; d++) {
      const int i = tid + 1, j = i + d;
      if (j <= W) {
        const double p = PT(OB, d, i) * PT(QB, d, i) / Z;
        mbd += p * (1.0 - p);
        if (p > 0.5) {
          cd += 1.0 - p;
          if (centroid) { centroid[(size_t)k * W1 + i - 1] = '('; centroid[(size_t)k * W1 + j - 1] = ')'; }
        } else cd += p;
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
}
