/* sf_cpu_abi.c — the C ABI of include/scanfold_hip.h on the HOST CPU: oracle/libscanfold_cpu.so.
 *
 * TEST / BASELINE INFRASTRUCTURE, like everything under oracle/: the same entry points as libscanfold_hip.so, every one
 * of them implemented with the CPU restatement (sf_oracle.c, its faster twin sf_cpu_twin.c, sf_shuffle_oracle.c).  It is
 * what SURVEY.md §8(b) calls "the same symbols from a libscanfold_cpu.so twin": BASELINE config 1 ("CPU reference path,
 * plumbing, no GPU") runs the unchanged host code of scanfold_amd over it when a user points SCANFOLD_LIB_PATH at it
 * (INTEGRATION.md), and tests/test_cpu_twin_abi.py checks that it exports every symbol the header declares and that
 * the command line on top of it writes the bytes the oracle's values give.  The product never loads it by itself:
 * scanfold_amd/_lib.py looks for libscanfold_hip.so only and fails without it.
 *
 * "Device" pointers are host pointers here, streams are ignored, device ordinal 0 is the host.
 * Not reentrant (the oracle keeps the current constraint in a global), like the HIP library. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "../include/scanfold_hip.h"
#include "../include/sf_params_blob.h"

/* the restatement's entry points (sf_oracle.c, sf_cpu_twin.c, sf_shuffle_oracle.c) */
int sfo_set_params(const void *blob, size_t n);
int sfo_set_params_exact(const void *blob, size_t n, const void *blob37, const void *blob_dH);
int sfo_set_constraint(const char *cons, const int *sc_stack_dcal);
int sfo_set_max_bp_span(int span);
int sfo_mfe(const char *seq, int n, int *mfe_dcal, char *structure);
int sfo_pf(const char *seq, int n, double *ensemble_dG, double *bpp_out, char *centroid, double *centroid_dist,
           double *mean_bp_dist);
int sfo_twin_mfe_batch(const char *seqs, int nseq, int W, int *out, int nthreads);
int sfo_twin_scan_windows(const char *rows, int n_win, int r, int W, int *energies, char *structures, char *centroids,
                          double *ens_div, int nthreads);
int sfo_shuffle_windows(const unsigned char *transcript, int L, int W, int step, int win_begin, int n_win, int r, int kind,
                        uint64_t seed, unsigned char *out);

static int g_init = 0, g_have_params = 0;
static double g_prof_ms = 0.0;
static long long g_prof_launches = 0, g_prof_folds = 0;
static int g_prof_on = 0;

const char *sf_strerror(int status) {
  switch (status) {
    case SF_OK: return "ok";
    case SF_ERR_NOT_INIT: return "sf_init has not been called";
    case SF_ERR_NO_PARAMS: return "no energy parameters loaded (sf_params_load)";
    case SF_ERR_BAD_ARG: return "bad argument";
    case SF_ERR_BAD_PARAMS: return "parameter blob has the wrong size, magic or version";
    case SF_ERR_TEMPERATURE: return "temperature differs from the one the parameter blob is valid at";
    case SF_ERR_HIP: return "HIP runtime error (see sf_last_hip_error)";
    case SF_ERR_NO_DEVICE: return "no usable GPU device";
    case SF_ERR_INTERNAL: return "internal error: traceback found no decomposition";
    case SF_ERR_TABLE: return "scan table: unbalanced structure string, or window starts not ascending";
    case SF_ERR_CONSTRAINT: return "unbalanced brackets in a window's constraint string";
    default: return "unknown status";
  }
}
const char *sf_last_hip_error(void) { return ""; }

int sf_init(int device_ordinal) {
  if (device_ordinal != 0) return SF_ERR_BAD_ARG; /* the host is the only "device" */
  g_init = 1;
  return SF_OK;
}
int sf_shutdown(void) {
  g_init = 0;
  g_have_params = 0;
  return SF_OK;
}
int sf_device_name(char *buf, size_t n) {
  if (!g_init) return SF_ERR_NOT_INIT;
  if (!buf || n == 0) return SF_ERR_BAD_ARG;
  snprintf(buf, n, "host CPU (oracle/libscanfold_cpu.so: the CPU restatement behind the scanfold_hip.h ABI)");
  return SF_OK;
}
int sf_params_load(const void *blob, size_t nbytes, double temperature_c) {
  return sf_params_load_rescaled(blob, nbytes, temperature_c, NULL, NULL);
}
int sf_params_load_rescaled(const void *blob, size_t nbytes, double temperature_c, const void *blob_37c,
                            const void *blob_enthalpy) {
  if (!g_init) return SF_ERR_NOT_INIT;
  if ((blob_37c == NULL) != (blob_enthalpy == NULL)) return SF_ERR_BAD_ARG;
  if (!blob || nbytes != sizeof(sf_params_blob)) return SF_ERR_BAD_PARAMS;
  const sf_params_blob *P = (const sf_params_blob *)blob;
  if (P->magic != SF_PARAMS_MAGIC || P->version != SF_PARAMS_VERSION) return SF_ERR_BAD_PARAMS;
  if (fabs(P->temperature - temperature_c) > 1e-9) return SF_ERR_TEMPERATURE;
  if (sfo_set_params_exact(blob, nbytes, blob_37c, blob_enthalpy)) return SF_ERR_BAD_PARAMS;
  g_have_params = 1;
  return SF_OK;
}
static int ready(void) {
  if (!g_init) return SF_ERR_NOT_INIT;
  if (!g_have_params) return SF_ERR_NO_PARAMS;
  return SF_OK;
}

/* rows of codes 0..4 or ASCII -> ASCII (the restatement reads characters) */
static char *ascii_rows(const uint8_t *seqs, size_t nbytes) {
  char *out = (char *)malloc(nbytes + 1);
  if (!out) return NULL;
  for (size_t k = 0; k < nbytes; k++) out[k] = seqs[k] < 5 ? "NACGU"[seqs[k]] : (char)seqs[k];
  out[nbytes] = 0;
  return out;
}
static double now_ms(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}
static void prof_add(double t0, long long folds) {
  if (!g_prof_on) return;
  g_prof_ms += now_ms() - t0;
  g_prof_launches++;
  g_prof_folds += folds;
}

int sf_mfe_batch(const uint8_t *seqs, int n, int W, int32_t *out) {
  int rc = ready();
  if (rc) return rc;
  if (n < 0 || W < 1 || W > SF_MAX_W || (n > 0 && (!seqs || !out))) return SF_ERR_BAD_ARG;
  if (n == 0) return SF_OK;
  char *rows = ascii_rows(seqs, (size_t)n * W);
  if (!rows) return SF_ERR_BAD_ARG;
  const double t0 = now_ms();
  sfo_set_constraint(NULL, NULL);
  rc = sfo_twin_mfe_batch(rows, n, W, (int *)out, 0);
  prof_add(t0, n);
  free(rows);
  return rc ? SF_ERR_INTERNAL : SF_OK;
}
int sf_mfe_batch_dev(const uint8_t *d_seqs, int n, int W, int32_t *d_out, void *stream) {
  (void)stream;
  return sf_mfe_batch(d_seqs, n, W, d_out);
}

int sf_fold_constrained(const uint8_t *seqs, int n, int W, const char *cons, const int32_t *sc, unsigned flags,
                        int32_t *mfe_out, char *db_out, double *ens_dG, double *mbd, char *centroid, double *cdist) {
  int rc = ready();
  if (rc) return rc;
  if (n < 0 || W < 1 || W > SF_MAX_W || (n > 0 && !seqs)) return SF_ERR_BAD_ARG;
  if (n == 0) return SF_OK;
  char *rows = ascii_rows(seqs, (size_t)n * W);
  char *crow = (char *)malloc((size_t)W + 1), *tmp_db = (char *)malloc((size_t)W + 1), *tmp_cen = (char *)malloc((size_t)W + 1);
  int status = SF_OK;
  for (int k = 0; k < n && status == SF_OK; k++) { /* serial: the current constraint is a global of the restatement */
    if (cons) {
      memcpy(crow, cons + (size_t)k * W, (size_t)W);
      crow[W] = 0;
    }
    sfo_set_constraint(cons ? crow : NULL, sc ? (const int *)(sc + (size_t)k * W) : NULL);
    if (!(flags & SF_FOLD_NO_MFE)) {
      int e = 0;
      const int r1 = sfo_mfe(rows + (size_t)k * W, W, &e, tmp_db);
      if (r1 == -3) status = SF_ERR_CONSTRAINT;
      else if (r1) status = SF_ERR_INTERNAL;
      if (mfe_out) mfe_out[k] = e;
      if (db_out) memcpy(db_out + (size_t)k * (W + 1), tmp_db, (size_t)W + 1);
    }
    if (status == SF_OK && !(flags & SF_FOLD_NO_PF)) {
      double dG = 0, cd = 0, m = 0;
      const int r2 = sfo_pf(rows + (size_t)k * W, W, &dG, NULL, tmp_cen, &cd, &m);
      if (r2 == -3) status = SF_ERR_CONSTRAINT;
      else if (r2) status = SF_ERR_INTERNAL;
      if (ens_dG) ens_dG[k] = dG;
      if (cdist) cdist[k] = cd;
      if (mbd) mbd[k] = m;
      if (centroid) memcpy(centroid + (size_t)k * (W + 1), tmp_cen, (size_t)W + 1);
    }
  }
  sfo_set_constraint(NULL, NULL);
  free(rows); free(crow); free(tmp_db); free(tmp_cen);
  return status;
}
int sf_mfe_trace_batch(const uint8_t *seqs, int n, int W, int32_t *mfe_out, char *db_out) {
  if (n > 0 && (!mfe_out || !db_out)) return ready() ? ready() : SF_ERR_BAD_ARG;
  return sf_fold_constrained(seqs, n, W, NULL, NULL, SF_FOLD_NO_PF, mfe_out, db_out, NULL, NULL, NULL, NULL);
}
int sf_pf_batch(const uint8_t *seqs, int n, int W, double *ens_dG, double *mbd, char *centroid, double *cdist) {
  return sf_fold_constrained(seqs, n, W, NULL, NULL, SF_FOLD_NO_MFE, NULL, NULL, ens_dG, mbd, centroid, cdist);
}

int sf_shuffle_windows(const uint8_t *transcript, int L, int W, int step, int win_begin, int n_win, int r, int kind,
                       uint64_t seed, uint8_t *seqs_out) {
  if (!g_init) return SF_ERR_NOT_INIT;
  if (!transcript || !seqs_out || L < 1 || W < 1 || W > SF_MAX_W || step < 1 || win_begin < 0 || n_win < 0 || r < 0 ||
      (kind != SF_SHUFFLE_MONO && kind != SF_SHUFFLE_DI))
    return SF_ERR_BAD_ARG;
  if (n_win > 0 && (long long)(win_begin + n_win - 1) * step + W > L) return SF_ERR_BAD_ARG;
  if (n_win == 0) return SF_OK;
  return sfo_shuffle_windows(transcript, L, W, step, win_begin, n_win, r, kind, seed, seqs_out) ? SF_ERR_BAD_ARG : SF_OK;
}

int sf_scan(const uint8_t *transcript, int L, int W, int step, int win_begin, int n_win, int r, int kind, uint64_t seed,
            unsigned flags, int32_t *energies, char *structure, char *centroid, double *ens_div, double *ens_dG) {
  int rc = ready();
  if (rc) return rc;
  if (!energies) return SF_ERR_BAD_ARG;
  if (n_win == 0) return SF_OK;
  const size_t nrows = (size_t)n_win * (size_t)(r + 1);
  uint8_t *codes = (uint8_t *)malloc(nrows * W);
  if (!codes) return SF_ERR_BAD_ARG;
  rc = sf_shuffle_windows(transcript, L, W, step, win_begin, n_win, r, kind, seed, codes);
  if (rc) { free(codes); return rc; }
  char *rows = ascii_rows(codes, nrows * W);
  free(codes);
  const int want_pf = !(flags & SF_SCAN_NO_PF), want_db = !(flags & SF_SCAN_NO_TRACE);
  const double t0 = now_ms();
  sfo_set_constraint(NULL, NULL);
  rc = sfo_twin_scan_windows(rows, n_win, r, W, (int *)energies, want_db ? structure : NULL, want_pf ? centroid : NULL,
                             want_pf ? ens_div : NULL, 0);
  prof_add(t0, (long long)nrows);
  if (!rc && want_pf && ens_dG) { /* the twin's scan keeps no ensemble energy: the restatement's partition function gives it */
    for (int w = 0; w < n_win && !rc; w++) rc = sfo_pf(rows + (size_t)w * (r + 1) * W, W, &ens_dG[w], NULL, NULL, NULL, NULL);
  }
  free(rows);
  return rc ? SF_ERR_INTERNAL : SF_OK;
}
int sf_scan_dev(const uint8_t *d_tr, int L, int W, int step, int win_begin, int n_win, int r, int kind, uint64_t seed,
                unsigned flags, int32_t *d_energies, char *d_structure, char *d_centroid, double *d_ens_div, double *d_ens_dG,
                void *stream) {
  (void)stream;
  return sf_scan(d_tr, L, W, step, win_begin, n_win, r, kind, seed, flags, d_energies, d_structure, d_centroid, d_ens_div,
                 d_ens_dG);
}
int sf_last_status(void) { return g_init ? SF_OK : SF_ERR_NOT_INIT; }

/* ---- pair tabulation (ScanFold-Fold.py:583-682,704-760): groups (nucleotide k, partner j) over the covering windows, in
 * (k, first window) order; sums in numpy's pairwise order so that they equal np.sum bit for bit (scanfold_amd/fold.py) ---- */
static double pw_leaf(const double *v, int n) {
  if (n < 8) {
    double res = 0.;
    for (int i = 0; i < n; i++) res = res + v[i];
    return res;
  }
  double r0 = v[0], r1 = v[1], r2 = v[2], r3 = v[3], r4 = v[4], r5 = v[5], r6 = v[6], r7 = v[7];
  int i = 8;
  for (; i < n - (n % 8); i += 8) {
    r0 = r0 + v[i + 0]; r1 = r1 + v[i + 1]; r2 = r2 + v[i + 2]; r3 = r3 + v[i + 3];
    r4 = r4 + v[i + 4]; r5 = r5 + v[i + 5]; r6 = r6 + v[i + 6]; r7 = r7 + v[i + 7];
  }
  double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
  for (; i < n; i++) res = res + v[i];
  return res;
}
static double pw_sum(const double *v, int n) {
  if (n <= 128) return pw_leaf(v, n);
  int n2 = n / 2;
  n2 -= n2 % 8;
  return pw_sum(v, n2) + pw_sum(v + n2, n - n2);
}
static struct { long long n; int32_t *k, *j, *cnt, *first; double *sz, *sm, *se; } g_tab = {-1, 0, 0, 0, 0, 0, 0, 0};
static void tab_free(void) {
  free(g_tab.k); free(g_tab.j); free(g_tab.cnt); free(g_tab.first); free(g_tab.sz); free(g_tab.sm); free(g_tab.se);
  memset(&g_tab, 0, sizeof g_tab);
  g_tab.n = -1;
}
int sf_tabulate_pairs(const char *structures, int row_stride, int on_device, int n_win, int W, const int32_t *starts,
                      const double *z, const double *mfe, const double *ed, int64_t *n_groups) {
  (void)on_device;
  if (!g_init) return SF_ERR_NOT_INIT;
  if (!structures || !starts || !z || !mfe || !ed || !n_groups || n_win < 1 || W < 1 || W > SF_MAX_W || row_stride < W)
    return SF_ERR_BAD_ARG;
  for (int w = 1; w < n_win; w++)
    if (starts[w] <= starts[w - 1]) return SF_ERR_TABLE;
  tab_free();
  int16_t *partner = (int16_t *)malloc(sizeof(int16_t) * (size_t)n_win * W);
  int *stack = (int *)malloc(sizeof(int) * (size_t)(W + 1));
  for (int w = 0; w < n_win; w++) {
    const char *s = structures + (size_t)w * row_stride;
    int16_t *out = partner + (size_t)w * W;
    int depth = 0, bad = 0;
    for (int p = 0; p < W; p++) {
      out[p] = -1;
      if (s[p] == '(') stack[depth++] = p;
      else if (s[p] == ')') {
        if (!depth) { bad = 1; break; }
        const int q = stack[--depth];
        out[q] = (int16_t)p;
        out[p] = (int16_t)q;
      }
    }
    if (bad || depth) { free(partner); free(stack); return SF_ERR_TABLE; }
  }
  free(stack);
  const int lo = starts[0], hi = starts[n_win - 1] + W - 1;
  const size_t cap = (size_t)(hi - lo + 1) * 8 + 1024;
  size_t room = cap, G = 0;
  g_tab.k = (int32_t *)malloc(4 * room); g_tab.j = (int32_t *)malloc(4 * room); g_tab.cnt = (int32_t *)malloc(4 * room);
  g_tab.first = (int32_t *)malloc(4 * room);
  g_tab.sz = (double *)malloc(8 * room); g_tab.sm = (double *)malloc(8 * room); g_tab.se = (double *)malloc(8 * room);
  int maxm = 0, w_lo = 0;
  int *jl = NULL, *gno = NULL;
  double *bz = NULL, *bm = NULL, *be = NULL;
  for (int k = lo; k <= hi; k++) {
    while (w_lo < n_win && starts[w_lo] + W - 1 < k) w_lo++;
    int w_hi = w_lo;
    while (w_hi < n_win && starts[w_hi] <= k) w_hi++;
    const int m = w_hi - w_lo;
    if (m <= 0) continue;
    if (m > maxm) {
      maxm = m;
      jl = (int *)realloc(jl, sizeof(int) * (size_t)m); gno = (int *)realloc(gno, sizeof(int) * (size_t)m);
      bz = (double *)realloc(bz, 8 * (size_t)m); bm = (double *)realloc(bm, 8 * (size_t)m); be = (double *)realloc(be, 8 * (size_t)m);
    }
    for (int a = 0; a < m; a++) {
      const int st = starts[w_lo + a], q = partner[(size_t)(w_lo + a) * W + (k - st)];
      jl[a] = q < 0 ? k : st + q;
      gno[a] = -1;
    }
    for (int a = 0; a < m; a++) { /* groups in first-window order */
      if (gno[a] >= 0) continue;
      int cnt = 0;
      for (int b = a; b < m; b++)
        if (jl[b] == jl[a]) {
          gno[b] = a;
          bz[cnt] = z[w_lo + b]; bm[cnt] = mfe[w_lo + b]; be[cnt] = ed[w_lo + b];
          cnt++;
        }
      if (G == room) {
        room *= 2;
        g_tab.k = (int32_t *)realloc(g_tab.k, 4 * room); g_tab.j = (int32_t *)realloc(g_tab.j, 4 * room);
        g_tab.cnt = (int32_t *)realloc(g_tab.cnt, 4 * room); g_tab.first = (int32_t *)realloc(g_tab.first, 4 * room);
        g_tab.sz = (double *)realloc(g_tab.sz, 8 * room); g_tab.sm = (double *)realloc(g_tab.sm, 8 * room);
        g_tab.se = (double *)realloc(g_tab.se, 8 * room);
      }
      g_tab.k[G] = k; g_tab.j[G] = jl[a]; g_tab.cnt[G] = cnt; g_tab.first[G] = w_lo + a;
      g_tab.sz[G] = pw_sum(bz, cnt); g_tab.sm[G] = pw_sum(bm, cnt); g_tab.se[G] = pw_sum(be, cnt);
      G++;
    }
  }
  free(jl); free(gno); free(bz); free(bm); free(be); free(partner);
  g_tab.n = (long long)G;
  *n_groups = (int64_t)G;
  return SF_OK;
}
int sf_tabulate_fetch(int32_t *gk, int32_t *gj, int32_t *gw, int32_t *gf, double *sz, double *sm, double *se) {
  if (!g_init) return SF_ERR_NOT_INIT;
  if (g_tab.n < 0 || !gk || !gj || !gw || !gf || !sz || !sm || !se) return SF_ERR_BAD_ARG;
  const size_t n = (size_t)g_tab.n;
  memcpy(gk, g_tab.k, 4 * n); memcpy(gj, g_tab.j, 4 * n); memcpy(gw, g_tab.cnt, 4 * n); memcpy(gf, g_tab.first, 4 * n);
  memcpy(sz, g_tab.sz, 8 * n); memcpy(sm, g_tab.sm, 8 * n); memcpy(se, g_tab.se, 8 * n);
  return SF_OK;
}

int sf_set_max_bp_span(int span) {
  if (!g_init) return SF_ERR_NOT_INIT;
  sfo_set_max_bp_span(span);
  return SF_OK;
}
int sf_set_kernel_mode(int mode) {
  if (!g_init) return SF_ERR_NOT_INIT;
  return (mode == 0 || mode == 1) ? SF_OK : SF_ERR_BAD_ARG; /* one engine here */
}
int sf_prof_reset(void) {
  if (!g_init) return SF_ERR_NOT_INIT;
  g_prof_ms = 0.0; g_prof_launches = 0; g_prof_folds = 0; g_prof_on = 1;
  return SF_OK;
}
int sf_prof_get(double *ms, int64_t *launches, int64_t *folds) {
  if (!g_init) return SF_ERR_NOT_INIT;
  if (ms) *ms = g_prof_ms;
  if (launches) *launches = g_prof_launches;
  if (folds) *folds = g_prof_folds;
  return SF_OK;
}
int sf_prof_stop(void) {
  if (!g_init) return SF_ERR_NOT_INIT;
  g_prof_on = 0;
  return SF_OK;
}
