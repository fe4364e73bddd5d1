/*
 * sf_cpu_twin.c — a CPU engine for the same per-window job, written for speed.  BASELINE / TEST INFRASTRUCTURE ONLY.
 * (#included at the end of sf_oracle.c: it reuses that file's parameter tables and loop-energy functions.  Same rule as
 * the oracle: only tests/ and bench.py's cpu_baseline leg may call it; the product never does, and its symbols carry
 * the sfo_ prefix so that nothing can load it in place of libscanfold_hip.so.)
 *
 * Why it exists: the reference's CPU path is ViennaRNA, which is absent here, and
 * the checker in sf_oracle.c is written to be read, not to be fast — O(n^4) outside pass, three mallocs per fold
 * (128 threads serialise in the allocator), a function call per pair-type look-up.  Timing THAT says little about what
 * a CPU does on this job.  This twin is the same algorithm as ViennaRNA's fold / pf_fold organised the usual way:
 *   - one workspace per thread, allocated once;
 *   - a pair-type matrix per sequence, so non-pairing cells cost one byte load;
 *   - dense interior-loop search over (p, q) with u1 + u2 <= 30, as ViennaRNA does (no Lyngso recurrence); the generic
 *     loops (both sides >= 2, not 2x2 / 2x3) read a copy of c with the inner pair's mismatch term added when the cell
 *     was stored, so a candidate is three table look-ups and the scan over q has no branch;
 *   - the multiloop split as a contiguous min-plus scan (row of fML against a row of its transpose: vectorisable);
 *   - McCaskill inside with O(1) qm1 recurrence, outside in O(n^3) through two helper tables over the closing pair
 *     (the formulation of scanfold_amd/csrc/sf_pf.hip.h);
 *   - OpenMP over windows, one thread per physical core.
 * It is checked against the oracle (tests/test_oracle.py: same energies, structures, centroids; ensemble values to
 * 1e-9).  It is still not ViennaRNA: "kind": "port" in bench.py.
 */
typedef struct {
  int cap; /* sequence length the buffers are sized for */
  int *c, *ci, *fML, *fMLT, *DML, *f5;
  unsigned char *pt;
  double *qb, *qm, *qm1, *ob, *obw, *a0, *a1, *q5, *q3, *mlb;
} twin_ws;

static _Thread_local twin_ws TW = {0};

static void twin_reserve(int n) {
  if (TW.cap >= n) return;
  free(TW.c); free(TW.ci); free(TW.fML); free(TW.fMLT); free(TW.DML); free(TW.f5); free(TW.pt);
  free(TW.qb); free(TW.qm); free(TW.qm1); free(TW.ob); free(TW.obw); free(TW.a0); free(TW.a1);
  free(TW.q5); free(TW.q3); free(TW.mlb);
  const size_t sz = (size_t)(n + 2) * (size_t)(n + 2);
  TW.c = (int *)malloc(sz * sizeof(int)); TW.ci = (int *)malloc(sz * sizeof(int)); TW.fML = (int *)malloc(sz * sizeof(int));
  TW.fMLT = (int *)malloc(sz * sizeof(int)); TW.DML = (int *)malloc(sz * sizeof(int));
  TW.f5 = (int *)malloc((size_t)(n + 2) * sizeof(int));
  TW.pt = (unsigned char *)malloc(sz);
  TW.qb = (double *)malloc(sz * sizeof(double)); TW.qm = (double *)malloc(sz * sizeof(double));
  TW.qm1 = (double *)malloc(sz * sizeof(double)); TW.ob = (double *)malloc(sz * sizeof(double));
  TW.obw = (double *)malloc(sz * sizeof(double)); TW.a0 = (double *)malloc(sz * sizeof(double));
  TW.a1 = (double *)malloc(sz * sizeof(double));
  TW.q5 = (double *)malloc((size_t)(n + 3) * sizeof(double)); TW.q3 = (double *)malloc((size_t)(n + 3) * sizeof(double));
  TW.mlb = (double *)malloc((size_t)(n + 3) * sizeof(double));
  TW.cap = n;
}

#define TX(i, j) ((size_t)(i) * (size_t)(n + 2) + (size_t)(j))

static void twin_pair_matrix(const seq_t *q) {
  const int n = q->n;
  memset(TW.pt, 0, (size_t)(n + 2) * (size_t)(n + 2));
  for (int i = 1; i <= n; i++)
    for (int j = i + TURN + 1; j <= n; j++) TW.pt[TX(i, j)] = (unsigned char)ptype(q, i, j);
}

/* Zuker fill; leaves c, fML, f5 in the workspace (same values as mfe_fill of the oracle) */
static int twin_mfe_fill(const seq_t *q) {
  const int n = q->n;
  const int *S = q->S;
  int *c = TW.c, *ci = TW.ci, *fML = TW.fML, *fMLT = TW.fMLT, *DML = TW.DML, *f5 = TW.f5;
  const unsigned char *pt = TW.pt;
  int IL[MAXLOOP + 1], NIN[MAXLOOP + 1];
  for (int u = 0; u <= MAXLOOP; u++) {
    IL[u] = P.internal_loop[u];
    NIN[u] = MIN2(P.max_ninio, u * P.ninio);
  }
  /* only the entries with j - i < TURN + 1 next to the first diagonals are read before they are written */
  for (int i = 0; i <= n + 1; i++)
    for (int j = i; j <= n + 1 && j <= i + TURN + 2; j++) {
      c[TX(i, j)] = fML[TX(i, j)] = DML[TX(i, j)] = INF;
      fMLT[TX(j, i)] = INF;
    }
  for (int d = TURN + 1; d < n; d++) {
    for (int i = 1; i + d <= n; i++) {
      const int j = i + d;
      const int type = pt[TX(i, j)];
      int cij = INF;
      if (type) {
        int e = E_hairpin(d - 1, type, S[i + 1], S[j - 1], q->str + i - 1);
        const int si1 = S[i + 1], sj1 = S[j - 1];
        const int pmax = MIN2(j - 2 - TURN, i + MAXLOOP + 1);
        const int mi = P.mismatchI[type][si1][sj1];
        int gen = 2 * INF; /* best generic candidate without the outer pair's mismatch term */
        for (int p = i + 1; p <= pmax; p++) {
          int minq = j - i + p - MAXLOOP - 2;
          if (minq < p + 1 + TURN) minq = p + 1 + TURN;
          const unsigned char *ptp = pt + TX(p, 0);
          const int *cp = c + TX(p, 0), *cip = ci + TX(p, 0);
          const int sp1 = S[p - 1], u1 = p - i - 1;
          int qq = j - 1;
          /* u2 = 0, 1 always, and every u2 while u1 < 2: the tabulated / bulge / 1xn rules */
          const int slow_until = u1 >= 2 ? j - 2 : minq; /* last q handled by the general function */
          for (; qq >= slow_until && qq >= minq; qq--) {
            const int t2 = ptp[qq];
            if (!t2) continue;
            const int en = E_intloop(u1, j - qq - 1, type, rtype[t2], si1, sj1, sp1, S[qq + 1]) +
                           sc_stack_term(q, i, j, p, qq) + cp[qq];
            if (en < e) e = en;
          }
          /* u1 >= 2, u2 >= 2: 2x2, 2x3 and 3x2 are tabulated too, the rest is generic */
          for (; qq >= minq; qq--) {
            const int u2 = j - 1 - qq;
            if ((u1 == 2 && u2 <= 3) || (u1 == 3 && u2 == 2)) {
              const int t2 = ptp[qq];
              if (!t2) continue;
              const int en = E_intloop(u1, u2, type, rtype[t2], si1, sj1, sp1, S[qq + 1]) + cp[qq];
              if (en < e) e = en;
              continue;
            }
            const int v = cip[qq] + IL[u1 + u2] + NIN[u1 > u2 ? u1 - u2 : u2 - u1];
            gen = v < gen ? v : gen;
          }
        }
        if (gen < INF / 2 && gen + mi < e) e = gen + mi;
        const int dml = DML[TX(i + 1, j - 1)];
        if (dml < INF) {
          const int en = dml + E_mlstem(rtype[type], sj1, si1) + P.MLclosing;
          if (en < e) e = en;
        }
        cij = e;
      }
      c[TX(i, j)] = cij;
      ci[TX(i, j)] = type ? cij + P.mismatchI[rtype[type]][ml_nb3(q, j) < 0 ? 0 : S[j + 1]][ml_nb5(q, i) < 0 ? 0 : S[i - 1]] : INF;
      int f = INF;
      if (fML[TX(i + 1, j)] < INF) f = MIN2(f, fML[TX(i + 1, j)] + P.MLbase);
      if (fML[TX(i, j - 1)] < INF) f = MIN2(f, fML[TX(i, j - 1)] + P.MLbase);
      if (type) f = MIN2(f, cij + E_mlstem(type, ml_nb5(q, i), ml_nb3(q, j)));
      /* split: min_k fML[i,k] + fML[k+1,j] = row i of fML against row j of the transpose, both contiguous in k */
      int dec = 2 * INF;
      {
        const int *a = fML + TX(i, 0), *b = fMLT + TX(j, 0);
        for (int k = i + TURN + 1; k <= j - TURN - 2; k++) {
          const int v = a[k] + b[k + 1];
          dec = v < dec ? v : dec;
        }
      }
      if (dec >= INF / 2) dec = INF; /* INF + a (negative) energy is still "no structure" */
      DML[TX(i, j)] = dec;
      const int fv = MIN2(f, dec);
      fML[TX(i, j)] = fv;
      fMLT[TX(j, i)] = fv;
    }
  }
  f5[0] = 0;
  for (int j = 1; j <= n; j++) {
    int v = f5[j - 1];
    for (int i = 1; i + TURN + 1 <= j; i++) {
      const int type = pt[TX(i, j)];
      if (!type) continue;
      const int en = f5[i - 1] + c[TX(i, j)] + E_extloop(type, ml_nb5(q, i), ml_nb3(q, j));
      if (en < v) v = en;
    }
    f5[j] = v;
  }
  return f5[n];
}

/* same traceback as the oracle's (mfe_traceback), over the workspace tables */
static int twin_traceback(const seq_t *q, char *db) {
  const int n = q->n;
  mfe_tabs t;
  t.n = n;
  /* the oracle's traceback indexes with IX(i,j) = i*(n+2)+j as well */
  t.c = TW.c; t.fML = TW.fML; t.DML = TW.DML; t.f5 = TW.f5;
  return mfe_traceback(q, &t, db);
}

/* McCaskill inside / outside -> centroid, mean bp distance, ensemble free energy */
static void twin_pf(const seq_t *q, char *centroid, double *mean_bp_dist, double *ens_dG) {
  const int n = q->n;
  const int *S = q->S;
  const unsigned char *pt = TW.pt;
  double *qb = TW.qb, *qm = TW.qm, *qm1 = TW.qm1, *ob = TW.ob, *obw = TW.obw, *A0 = TW.a0, *A1 = TW.a1;
  double *q5 = TW.q5, *q3 = TW.q3, *mlb = TW.mlb;
  const size_t sz = (size_t)(n + 2) * (size_t)(n + 2);
  memset(qb, 0, sz * sizeof(double)); memset(qm, 0, sz * sizeof(double)); memset(qm1, 0, sz * sizeof(double));
  memset(ob, 0, sz * sizeof(double)); memset(obw, 0, sz * sizeof(double));
  memset(A0, 0, sz * sizeof(double)); memset(A1, 0, sz * sizeof(double));
  mlb[0] = 1.0;
  for (int k = 1; k <= n + 1; k++) mlb[k] = mlb[k - 1] * XP->MLbase;
  for (int d = TURN + 1; d < n; d++) {
    for (int i = 1; i + d <= n; i++) {
      const int j = i + d;
      const int type = pt[TX(i, j)];
      double qbij = 0.0;
      if (type) {
        double z = X_hairpin(d - 1, type, S[i + 1], S[j - 1], q->str + i - 1);
        const int si1 = S[i + 1], sj1 = S[j - 1];
        const int pmax = MIN2(j - 2 - TURN, i + MAXLOOP + 1);
        for (int p = i + 1; p <= pmax; p++) {
          int minq = j - i + p - MAXLOOP - 2;
          if (minq < p + 1 + TURN) minq = p + 1 + TURN;
          const unsigned char *ptp = pt + TX(p, 0);
          const double *qbp = qb + TX(p, 0);
          const int sp1 = S[p - 1], u1 = p - i - 1;
          for (int qq = j - 1; qq >= minq; qq--) {
            const int t2 = ptp[qq];
            if (!t2) continue;
            z += X_intloop(u1, j - qq - 1, type, rtype[t2], si1, sj1, sp1, S[qq + 1]) * qbp[qq];
          }
        }
        double ml = 0.0;
        for (int u = i + 2 + TURN; u <= j - 1 - TURN - 1; u++) ml += qm[TX(i + 1, u - 1)] * qm1[TX(u, j - 1)];
        z += ml * XP->MLclosing * X_mlstem(rtype[type], sj1, si1);
        qbij = z;
      }
      qb[TX(i, j)] = qbij;
      double m1 = qm1[TX(i, j - 1)] * XP->MLbase;
      if (type) m1 += qbij * X_mlstem(type, ml_nb5(q, i), ml_nb3(q, j));
      qm1[TX(i, j)] = m1;
      double m = m1; /* u == i */
      for (int u = i + 1; u + TURN + 1 <= j; u++) m += (mlb[u - i] + qm[TX(i, u - 1)]) * qm1[TX(u, j)];
      qm[TX(i, j)] = m;
    }
  }
  q5[0] = 1.0;
  for (int j = 1; j <= n; j++) {
    double z = q5[j - 1];
    for (int i = 1; i + TURN + 1 <= j; i++) {
      const int type = pt[TX(i, j)];
      if (type) z += q5[i - 1] * qb[TX(i, j)] * X_extloop(type, ml_nb5(q, i), ml_nb3(q, j));
    }
    q5[j] = z;
  }
  q3[n + 1] = 1.0;
  for (int i = n; i >= 1; i--) {
    double z = q3[i + 1];
    for (int j = i + TURN + 1; j <= n; j++) {
      const int type = pt[TX(i, j)];
      if (type) z += qb[TX(i, j)] * X_extloop(type, ml_nb5(q, i), ml_nb3(q, j)) * q3[j + 1];
    }
    q3[i] = z;
  }
  const double Z = q5[n];
  if (ens_dG) *ens_dG = -log(Z) * XP->kT / 1000.0;
  /* outside, widest pairs first; A0 / A1 are indexed by the cell (i, l): sums over closers (k, l), k < i */
  for (int d = n - 1; d >= TURN + 1; d--) {
    for (int i = 1; i + d <= n; i++) {
      const int j = i + d;
      double a0 = 0.0, a1 = 0.0;
      if (i > 1) {
        a0 = A0[TX(i - 1, j)] * XP->MLbase + obw[TX(i - 1, j)];
        for (int k = 1; k <= i - 2 - TURN - 1; k++) a1 += obw[TX(k, j)] * qm[TX(k + 1, i - 1)];
      }
      A0[TX(i, j)] = a0;
      A1[TX(i, j)] = a1;
      const int type = pt[TX(i, j)];
      const double qbij = qb[TX(i, j)];
      if (!type || qbij == 0.0) continue;
      double o = q5[i - 1] * q3[j + 1] * X_extloop(type, ml_nb5(q, i), ml_nb3(q, j));
      if (i > 1 && j < n) {
        const int rt = rtype[type], sp1 = S[i - 1], sq1 = S[j + 1];
        for (int k = MAX2(1, i - MAXLOOP - 1); k < i; k++) {
          const int u1 = i - k - 1;
          const unsigned char *ptk = pt + TX(k, 0);
          const double *obk = ob + TX(k, 0);
          for (int l = j + 1; l <= n && (l - j - 1) + u1 <= MAXLOOP; l++) {
            const int tk = ptk[l];
            if (!tk || obk[l] == 0.0) continue;
            o += obk[l] * X_intloop(u1, l - j - 1, tk, rt, S[k + 1], S[l - 1], sp1, sq1);
          }
        }
        double mlsum = 0.0;
        for (int l = j + 1; l <= n; l++) {
          const double qmr = (l - 1 >= j + 1) ? qm[TX(j + 1, l - 1)] : 0.0;
          mlsum += A1[TX(i, l)] * (mlb[l - 1 - j] + qmr) + A0[TX(i, l)] * qmr;
        }
        o += mlsum * X_mlstem(type, sp1, sq1);
      }
      ob[TX(i, j)] = o;
      obw[TX(i, j)] = o * XP->MLclosing * X_mlstem(rtype[type], S[j - 1], S[i + 1]);
    }
  }
  double mbd = 0.0;
  if (centroid) {
    for (int k = 0; k < n; k++) centroid[k] = '.';
    centroid[n] = 0;
  }
  for (int i = 1; i <= n; i++)
    for (int j = i + TURN + 1; j <= n; j++) {
      const double p = ob[TX(i, j)] * qb[TX(i, j)] / Z;
      mbd += p * (1.0 - p);
      if (p > 0.5 && centroid) { centroid[i - 1] = '('; centroid[j - 1] = ')'; }
    }
  if (mean_bp_dist) *mean_bp_dist = 2.0 * mbd;
}
#undef TX

/* Same contract as sfo_scan_windows: rows = n_win * (r+1) sequences of W characters; native row: MFE + traceback +
 * partition function, shuffle rows: MFE only.  One OpenMP thread per window. */
int sfo_twin_scan_windows(const char *rows, int n_win, int r, int W, int *energies, char *structures, char *centroids,
                          double *ens_div, int nthreads) {
  if (!have_params) return -10;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
  (void)nthreads;
  int bad = 0;
  g_cons = NULL; g_sc_stack = NULL;
#pragma omp parallel for schedule(dynamic, 1)
  for (int w = 0; w < n_win; w++) {
    twin_reserve(W);
    const char *base = rows + (size_t)w * (r + 1) * W;
    for (int k = 0; k <= r; k++) {
      seq_t q;
      seq_init(&q, base + (size_t)k * W, W);
      twin_pair_matrix(&q);
      energies[(size_t)w * (r + 1) + k] = twin_mfe_fill(&q);
      if (k == 0) {
        if (structures && twin_traceback(&q, structures + (size_t)w * (W + 1))) bad = 1;
        if (centroids || ens_div) twin_pf(&q, centroids ? centroids + (size_t)w * (W + 1) : NULL,
                                          ens_div ? &ens_div[w] : NULL, NULL);
      }
      seq_free(&q);
    }
  }
  return bad ? -1 : 0;
}

int sfo_twin_mfe_batch(const char *seqs, int nseq, int W, int *out, int nthreads) {
  if (!have_params) return -10;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
  (void)nthreads;
  g_cons = NULL; g_sc_stack = NULL;
#pragma omp parallel for schedule(dynamic, 8)
  for (int k = 0; k < nseq; k++) {
    twin_reserve(W);
    seq_t q;
    seq_init(&q, seqs + (size_t)k * W, W);
    twin_pair_matrix(&q);
    out[k] = twin_mfe_fill(&q);
    seq_free(&q);
  }
  return 0;
}
