/*
 * sf_oracle.c — CPU restatement of the thermodynamic engine ScanFold-Scan calls. TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the
 * product (scanfold_amd/, libscanfold_hip.so) never links, imports or calls it.
 *
 * What it restates.  The reference has no arithmetic of its own on this path: it calls the
 * third-party ViennaRNA Python module `RNA` (no pinned version; a 2.4.x release by the era,
 * SURVEY.md F3), which is absent from /root/reference, from this container and from the GPU box:
 *   RNA.fold(seq)                          ScanFold-Scan.py:245        -> sfo_mfe (energy only used)
 *   RNA.fold_compound(seq, md).mfe()       ScanFold-Scan.py:382,385,394; ScanFoldFunctions.py:786-787 -> sfo_mfe
 *   fc.pf(), RNA.pf_fold(seq)              ScanFold-Scan.py:383-384,395 -> sfo_pf (ensemble free energy)
 *   fc.centroid()                          ScanFold-Scan.py:388,400    -> sfo_pf (centroid string, distance)
 *   fc.mean_bp_distance()                  ScanFold-Scan.py:389,401    -> sfo_pf (mean_bp_dist)
 * so this file restates ViennaRNA's published algorithm under RNA.md() defaults (Zuker MFE with
 * dangles=2, noLP=0, special hairpins on, MAXLOOP=30, TURN=3; McCaskill partition function with
 * pf_smooth=1) as recorded in SURVEY.md Appendix A, over the parameter blob of
 * include/sf_params_blob.h.
 *
 * PARITY UNPINNED at the ViennaRNA boundary: the reference holds no golden MFE / structure / centroid /
 * ensemble-diversity value anywhere and ViennaRNA cannot be run here, so nothing
 * in this file is checked against ViennaRNA output.  What pins it instead (tests/test_oracle_*.py):
 *   V1  sfo_brute (exhaustive enumeration, scored by the independent loop evaluator sfo_eval) == sfo_mfe
 *   V2  sfo_eval(traceback structure) == MFE on random sequences up to W=200
 *   V3  brute-force Boltzmann sum == sfo_pf partition function; brute-force pair probabilities == bpp
 *   V4  parameter-table symmetries
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/sf_params_blob.h"

#define INF SF_INF
#define TURN SF_TURN
#define MAXLOOP SF_MAXLOOP
#define MIN2(a, b) ((a) < (b) ? (a) : (b))
#define MAX2(a, b) ((a) > (b) ? (a) : (b))
#define K0 273.15
#define GASCONST 1.98717 /* cal/(K mol) */

static sf_params_blob P;
static int have_params = 0;

static const int rtype[8] = {0, 2, 1, 4, 3, 6, 5, 7};
static int pair_tab[5][5];

/* Boltzmann-weight tables derived from P at P.temperature */
typedef struct {
  double kT; /* cal/mol */
  double stack[8][8], hairpin[31], bulge[31], internal_loop[31];
  double mismatchI[8][5][5], mismatchH[8][5][5], mismatchM[8][5][5], mismatch1nI[8][5][5],
      mismatch23I[8][5][5], mismatchExt[8][5][5];
  double dangle5[8][5], dangle3[8][5];
  double int11[8][8][5][5], int21[8][8][5][5][5], int22[8][8][5][5][5][5];
  double ninio[MAXLOOP + 1];
  double MLbase, MLclosing, MLintern[8], TermAU;
  double tetra[SF_MAX_SPECIAL], tri[SF_MAX_SPECIAL], hexa[SF_MAX_SPECIAL];
} exp_params;
static exp_params *XP = NULL;

/* ViennaRNA's SMOOTH() for stabilising dangle / terminal-mismatch terms in the partition function */
static double smooth(double x) {
  const double SCALE = 10.0;
  if (x / SCALE < -1.2283697) return 0.0;
  if (x / SCALE > 0.8660254) return x;
  double s = sin(x / SCALE - 0.34242663) + 1.0;
  return SCALE * 0.38490018 * s * s;
}

/* Free energy behind one Boltzmann weight.  ViennaRNA's get_boltzmann_factors [EXT] rescales each entry from the 37 C
 * value and its enthalpy in double, dG(T) = dH - (dH - dG37) * (T + K0) / (37 + K0), and does NOT truncate it to an
 * integer the way the MFE tables are (md.pf_smooth, the default).  When the caller handed over the 37 C and enthalpy
 * records (sfo_set_params_exact) the weight of field `*ref` of P is taken from the same field of those; otherwise from
 * the integer in P.  At 37 C both are the same number. */
static const sf_params_blob *P37x = NULL, *PdHx = NULL;
static double exact_energy(const int32_t *ref) {
  if (!P37x) return (double)*ref;
  size_t off = (size_t)((const char *)ref - (const char *)&P);
  int32_t g = *(const int32_t *)((const char *)P37x + off), h = *(const int32_t *)((const char *)PdHx + off);
  if (g >= INF || g <= -INF) return (double)*ref;
  double tempf = (P.temperature + K0) / (37.0 + K0);
  return (double)h - ((double)h - (double)g) * tempf;
}

static void build_exp_params(void) {
  if (!XP) XP = (exp_params *)malloc(sizeof(exp_params));
  exp_params *x = XP;
  x->kT = (P.temperature + K0) * GASCONST;
  const double kT = x->kT;
#define EX(f) exact_energy(&(f))
#define BW(e) exp(-(double)(e)*10.0 / kT)
#define BWS(e) exp(smooth(-(double)(e)) * 10.0 / kT)
  for (int a = 0; a < 8; a++)
    for (int b = 0; b < 8; b++) x->stack[a][b] = BW(EX(P.stack[a][b]));
  for (int i = 0; i <= 30; i++) {
    x->hairpin[i] = BW(EX(P.hairpin[i]));
    x->bulge[i] = BW(EX(P.bulge[i]));
    x->internal_loop[i] = BW(EX(P.internal_loop[i]));
    x->ninio[i] = BW(MIN2((double)P.max_ninio, i * EX(P.ninio)));
  }
  for (int t = 0; t < 8; t++)
    for (int a = 0; a < 5; a++) {
      x->dangle5[t][a] = BWS(EX(P.dangle5[t][a]));
      x->dangle3[t][a] = BWS(EX(P.dangle3[t][a]));
      for (int b = 0; b < 5; b++) {
        x->mismatchI[t][a][b] = BW(EX(P.mismatchI[t][a][b]));
        x->mismatchH[t][a][b] = BW(EX(P.mismatchH[t][a][b]));
        x->mismatch1nI[t][a][b] = BW(EX(P.mismatch1nI[t][a][b]));
        x->mismatch23I[t][a][b] = BW(EX(P.mismatch23I[t][a][b]));
        x->mismatchM[t][a][b] = BWS(EX(P.mismatchM[t][a][b]));
        x->mismatchExt[t][a][b] = BWS(EX(P.mismatchExt[t][a][b]));
      }
    }
  for (int a = 0; a < 8; a++)
    for (int b = 0; b < 8; b++)
      for (int c = 0; c < 5; c++)
        for (int d = 0; d < 5; d++) {
          x->int11[a][b][c][d] = BW(EX(P.int11[a][b][c][d]));
          for (int e = 0; e < 5; e++) {
            x->int21[a][b][c][d][e] = BW(EX(P.int21[a][b][c][d][e]));
            for (int f = 0; f < 5; f++) x->int22[a][b][c][d][e][f] = BW(EX(P.int22[a][b][c][d][e][f]));
          }
        }
  x->MLbase = BW(EX(P.MLbase));
  x->MLclosing = BW(EX(P.MLclosing));
  for (int t = 0; t < 8; t++) x->MLintern[t] = BW(EX(P.MLintern[t]));
  x->TermAU = BW(EX(P.TerminalAU));
  for (int k = 0; k < SF_MAX_SPECIAL; k++) {
    x->tetra[k] = BW(EX(P.tetra_E[k]));
    x->tri[k] = BW(EX(P.tri_E[k]));
    x->hexa[k] = BW(EX(P.hexa_E[k]));
  }
#undef BW
#undef BWS
#undef EX
}

int sfo_params_size(void) { return (int)sizeof(sf_params_blob); }

static int set_params_impl(const void *blob, size_t n);
int sfo_set_params(const void *blob, size_t n) {
  P37x = PdHx = NULL;
  return set_params_impl(blob, n);
}
/* As sfo_set_params, with the records the set was rescaled from: the free energies at 37 C and the enthalpies (same
 * struct).  The MFE model still uses the truncated integers of `blob`; the Boltzmann weights use the exact doubles. */
int sfo_set_params_exact(const void *blob, size_t n, const void *blob37, const void *blob_dH) {
  static sf_params_blob A, B;
  if (n != sizeof(sf_params_blob)) return -1;
  if (!blob37 || !blob_dH) return sfo_set_params(blob, n);
  memcpy(&A, blob37, sizeof A);
  memcpy(&B, blob_dH, sizeof B);
  P37x = &A;
  PdHx = &B;
  int rc = set_params_impl(blob, n);
  P37x = PdHx = NULL;
  return rc;
}
static int set_params_impl(const void *blob, size_t n) {
  if (n != sizeof(sf_params_blob)) return -1;
  memcpy(&P, blob, sizeof(P));
  if (P.magic != SF_PARAMS_MAGIC || P.version != SF_PARAMS_VERSION) return -2;
  memset(pair_tab, 0, sizeof(pair_tab));
  pair_tab[2][3] = 1; /* CG */
  pair_tab[3][2] = 2; /* GC */
  pair_tab[3][4] = 3; /* GU */
  pair_tab[4][3] = 4; /* UG */
  pair_tab[1][4] = 5; /* AU */
  pair_tab[4][1] = 6; /* UA */
  build_exp_params(); /* Boltzmann weights come from the values as given (through SMOOTH) ... */
  /* ... while the MFE model uses dangle / multiloop / exterior mismatch terms clamped to <= 0: ViennaRNA's
   * get_scaled_params stores them as min(0, x) ("must be <= 0") [EXT].  Turner 2004 has no positive entry, so this
   * only matters for user-supplied sets (Turner 1999, Andronescu) and the randomised test tables. */
  for (int t = 0; t < 8; t++)
    for (int a = 0; a < 5; a++) {
      P.dangle5[t][a] = MIN2(0, P.dangle5[t][a]);
      P.dangle3[t][a] = MIN2(0, P.dangle3[t][a]);
      for (int b = 0; b < 5; b++) {
        P.mismatchM[t][a][b] = MIN2(0, P.mismatchM[t][a][b]);
        P.mismatchExt[t][a][b] = MIN2(0, P.mismatchExt[t][a][b]);
      }
    }
  have_params = 1;
  return 0;
}

/* ---------- sequence handling: S[1..n] codes, S[0]=S[n+1]=unused; str[0..n-1] normalised ---------- */
static int encode_char(char c, char *norm) {
  switch (c) {
    case 'A': case 'a': *norm = 'A'; return 1;
    case 'C': case 'c': *norm = 'C'; return 2;
    case 'G': case 'g': *norm = 'G'; return 3;
    case 'U': case 'u': case 'T': case 't': *norm = 'U'; return 4;
    default: *norm = 'N'; return 0;
  }
}

typedef struct {
  int n;
  int *S;    /* 0..n+1 */
  char *str; /* normalised, NUL terminated */
  /* hard constraint of this sequence (NULL: none): the dot-bracket string, bracket partners (0: none) and, per
   * position, the innermost bracket pair that encloses it (index of its opening position, 0: none) */
  const char *cons;
  int *partner, *encl;
  const int *sc_stack; /* soft constraint: pseudo-energy (dcal/mol) of position i when it sits in a stacked pair; 0-based, NULL: none */
  int cons_bad;        /* unbalanced brackets */
} seq_t;

/* fc.hc_add_from_db(window_constraints) (ScanFold-Scan.py:405-410; ScanFold.py:508-512) and
 * fc.sc_add_SHAPE_deigan(...) (ScanFold.py:533-539) for the NEXT single-sequence calls (sfo_mfe, sfo_eval, sfo_brute,
 * sfo_pf); the batch entry points ignore it.  Semantics restated from ViennaRNA 2.4's vrna_hc_add_from_db with its
 * default options — no VRNA_CONSTRAINT_DB_ENFORCE_BP — [EXT, unverifiable here]:
 *   '.'  no constraint          'x'  the position stays unpaired
 *   '|'  no effect (without the enforce option a position is merely ALLOWED to pair in either direction)
 *   '<'  the position may only pair with a position downstream (it cannot be the 3' partner of a pair)
 *   '>'  the position may only pair with a position upstream (it cannot be the 5' partner)
 *   '(' ')'  the two positions may pair with each other only (also if the bases are not complementary: pair
 *        type 7), and no pair may cross theirs; they are not forced to pair
 * Unbalanced brackets are an error (ViennaRNA aborts the process there).  The Deigan term of position i is added to
 * every stack (interior loop without unpaired bases) i takes part in, in the MFE model only: the reference adds SHAPE
 * data after its partition function call (ScanFold.py:525-539). */
static const char *g_cons = NULL;
static const int *g_sc_stack = NULL;
int sfo_set_constraint(const char *cons, const int *sc_stack_dcal) {
  g_cons = cons;
  g_sc_stack = sc_stack_dcal;
  return 0;
}

static void seq_init(seq_t *q, const char *seq, int n) {
  q->n = n;
  q->S = (int *)calloc((size_t)n + 2, sizeof(int));
  q->str = (char *)malloc((size_t)n + 1);
  for (int i = 0; i < n; i++) q->S[i + 1] = encode_char(seq[i], &q->str[i]);
  q->str[n] = 0;
  q->cons = g_cons;
  q->sc_stack = g_sc_stack;
  q->partner = q->encl = NULL;
  q->cons_bad = 0;
  if (q->cons) {
    q->partner = (int *)calloc((size_t)n + 2, sizeof(int));
    q->encl = (int *)calloc((size_t)n + 2, sizeof(int));
    int *stack = (int *)malloc(sizeof(int) * (size_t)(n + 1)), sp = 0;
    for (int i = 1; i <= n; i++) {
      const char ch = q->cons[i - 1];
      if (ch == ')') {
        if (sp == 0) { q->cons_bad = 1; break; }
        const int o = stack[--sp];
        q->partner[o] = i;
        q->partner[i] = o;
      }
      q->encl[i] = sp ? stack[sp - 1] : 0; /* a bracket position itself: the pair around its own pair */
      if (ch == '(') stack[sp++] = i;
    }
    if (sp) q->cons_bad = 1;
    free(stack);
  }
}
static void seq_free(seq_t *q) {
  free(q->S);
  free(q->str);
  free(q->partner);
  free(q->encl);
}
/* RNA.md().max_bp_span (ScanFold.py:214-215; [EXT] ViennaRNA: a pair (i,j) needs j - i + 1 <= max_bp_span); <= 0: none */
static int g_max_bp_span = 0;
int sfo_set_max_bp_span(int span) { g_max_bp_span = span > 0 ? span : 0; return 0; }
static inline int ptype(const seq_t *q, int i, int j) {
  if (g_max_bp_span > 0 && j - i + 1 > g_max_bp_span) return 0;
  const int t = pair_tab[q->S[i]][q->S[j]];
  if (!q->cons) return t;
  if (q->partner[i] || q->partner[j]) return q->partner[i] == j ? (t ? t : 7) : 0;
  const char ci = q->cons[i - 1], cj = q->cons[j - 1];
  if (ci == 'x' || cj == 'x' || ci == '>' || cj == '<') return 0;
  if (q->encl[i] != q->encl[j]) return 0; /* would cross a bracket pair */
  return t;
}
/* Deigan pseudo-energy of the stack (i,j) on (i+1,j-1) */
static inline int sc_stack_term(const seq_t *q, int i, int j, int p, int qq) {
  if (!q->sc_stack || p != i + 1 || qq != j - 1) return 0;
  return q->sc_stack[i - 1] + q->sc_stack[p - 1] + q->sc_stack[qq - 1] + q->sc_stack[j - 1];
}

/* ---------- loop energies ---------- */
static int special_lookup(const char (*tab)[8], int cnt, const char *s, int len) {
  for (int k = 0; k < cnt; k++)
    if (strncmp(tab[k], s, (size_t)len) == 0) return k;
  return -1;
}
static int special_lookup12(const char (*tab)[12], int cnt, const char *s, int len) {
  for (int k = 0; k < cnt; k++)
    if (strncmp(tab[k], s, (size_t)len) == 0) return k;
  return -1;
}

/* loop = pointer to the normalised string at the closing 5' base i (0-based i-1), loop length size+2 */
static int E_hairpin(int size, int type, int si1, int sj1, const char *loop) {
  int ge = (size <= 30) ? P.hairpin[size] : P.hairpin[30] + (int)(P.lxc * log(size / 30.));
  if (size < 3) return ge;
  if (size == 4) {
    int k = special_lookup(P.tetra_seq, P.n_tetra, loop, 6);
    if (k >= 0) return P.tetra_E[k];
  } else if (size == 6) {
    int k = special_lookup12(P.hexa_seq, P.n_hexa, loop, 8);
    if (k >= 0) return P.hexa_E[k];
  } else if (size == 3) {
    int k = special_lookup(P.tri_seq, P.n_tri, loop, 5);
    if (k >= 0) return P.tri_E[k];
    return ge + (type > 2 ? P.TerminalAU : 0);
  }
  return ge + P.mismatchH[type][si1][sj1];
}

static int E_intloop(int n1, int n2, int type, int type_2, int si1, int sj1, int sp1, int sq1) {
  int nl, ns, u, energy;
  if (n1 > n2) { nl = n1; ns = n2; } else { nl = n2; ns = n1; }
  if (nl == 0) return P.stack[type][type_2];
  if (ns == 0) { /* bulge */
    energy = (nl <= MAXLOOP) ? P.bulge[nl] : P.bulge[30] + (int)(P.lxc * log(nl / 30.));
    if (nl == 1) energy += P.stack[type][type_2];
    else {
      if (type > 2) energy += P.TerminalAU;
      if (type_2 > 2) energy += P.TerminalAU;
    }
    return energy;
  }
  if (ns == 1) {
    if (nl == 1) return P.int11[type][type_2][si1][sj1];
    if (nl == 2) {
      if (n1 == 1) return P.int21[type][type_2][si1][sq1][sj1];
      return P.int21[type_2][type][sq1][si1][sp1];
    }
    /* 1 x n */
    energy = (nl + 1 <= MAXLOOP) ? P.internal_loop[nl + 1] : P.internal_loop[30] + (int)(P.lxc * log((nl + 1) / 30.));
    energy += MIN2(P.max_ninio, (nl - ns) * P.ninio);
    energy += P.mismatch1nI[type][si1][sj1] + P.mismatch1nI[type_2][sq1][sp1];
    return energy;
  }
  if (ns == 2) {
    if (nl == 2) return P.int22[type][type_2][si1][sp1][sq1][sj1];
    if (nl == 3) {
      energy = P.internal_loop[5] + P.ninio;
      energy += P.mismatch23I[type][si1][sj1] + P.mismatch23I[type_2][sq1][sp1];
      return energy;
    }
  }
  u = nl + ns;
  energy = (u <= MAXLOOP) ? P.internal_loop[u] : P.internal_loop[30] + (int)(P.lxc * log(u / 30.));
  energy += MIN2(P.max_ninio, (nl - ns) * P.ninio);
  energy += P.mismatchI[type][si1][sj1] + P.mismatchI[type_2][sq1][sp1];
  return energy;
}

/* si1 / sj1 < 0 : neighbour does not exist */
static int E_mlstem(int type, int si1, int sj1) {
  int e = 0;
  if (si1 >= 0 && sj1 >= 0) e += P.mismatchM[type][si1][sj1];
  else if (si1 >= 0) e += P.dangle5[type][si1];
  else if (sj1 >= 0) e += P.dangle3[type][sj1];
  if (type > 2) e += P.TerminalAU;
  return e + P.MLintern[type];
}
static int E_extloop(int type, int si1, int sj1) {
  int e = 0;
  if (si1 >= 0 && sj1 >= 0) e += P.mismatchExt[type][si1][sj1];
  else if (si1 >= 0) e += P.dangle5[type][si1];
  else if (sj1 >= 0) e += P.dangle3[type][sj1];
  if (type > 2) e += P.TerminalAU;
  return e;
}

/* Boltzmann-weight twins */
static double X_hairpin(int size, int type, int si1, int sj1, const char *loop) {
  double q = (size <= 30) ? XP->hairpin[size] : XP->hairpin[30] * exp(-(P.lxc * log(size / 30.)) * 10. / XP->kT);
  if (size < 3) return q;
  if (size == 4) {
    int k = special_lookup(P.tetra_seq, P.n_tetra, loop, 6);
    if (k >= 0) return XP->tetra[k];
  } else if (size == 6) {
    int k = special_lookup12(P.hexa_seq, P.n_hexa, loop, 8);
    if (k >= 0) return XP->hexa[k];
  } else if (size == 3) {
    int k = special_lookup(P.tri_seq, P.n_tri, loop, 5);
    if (k >= 0) return XP->tri[k];
    return (type > 2) ? q * XP->TermAU : q;
  }
  return q * XP->mismatchH[type][si1][sj1];
}
static double X_intloop(int n1, int n2, int type, int type_2, int si1, int sj1, int sp1, int sq1) {
  int nl, ns;
  if (n1 > n2) { nl = n1; ns = n2; } else { nl = n2; ns = n1; }
  if (nl == 0) return XP->stack[type][type_2];
  if (ns == 0) {
    double z = XP->bulge[nl];
    if (nl == 1) z *= XP->stack[type][type_2];
    else {
      if (type > 2) z *= XP->TermAU;
      if (type_2 > 2) z *= XP->TermAU;
    }
    return z;
  }
  if (ns == 1) {
    if (nl == 1) return XP->int11[type][type_2][si1][sj1];
    if (nl == 2) {
      if (n1 == 1) return XP->int21[type][type_2][si1][sq1][sj1];
      return XP->int21[type_2][type][sq1][si1][sp1];
    }
    return XP->internal_loop[nl + 1] * XP->ninio[nl - ns] * XP->mismatch1nI[type][si1][sj1] *
           XP->mismatch1nI[type_2][sq1][sp1];
  }
  if (ns == 2) {
    if (nl == 2) return XP->int22[type][type_2][si1][sp1][sq1][sj1];
    if (nl == 3)
      return XP->internal_loop[5] * XP->ninio[1] * XP->mismatch23I[type][si1][sj1] *
             XP->mismatch23I[type_2][sq1][sp1];
  }
  return XP->internal_loop[nl + ns] * XP->ninio[nl - ns] * XP->mismatchI[type][si1][sj1] *
         XP->mismatchI[type_2][sq1][sp1];
}
static double X_mlstem(int type, int si1, int sj1) {
  double z = 1.0;
  if (si1 >= 0 && sj1 >= 0) z = XP->mismatchM[type][si1][sj1];
  else if (si1 >= 0) z = XP->dangle5[type][si1];
  else if (sj1 >= 0) z = XP->dangle3[type][sj1];
  if (type > 2) z *= XP->TermAU;
  return z * XP->MLintern[type];
}
static double X_extloop(int type, int si1, int sj1) {
  double z = 1.0;
  if (si1 >= 0 && sj1 >= 0) z = XP->mismatchExt[type][si1][sj1];
  else if (si1 >= 0) z = XP->dangle5[type][si1];
  else if (sj1 >= 0) z = XP->dangle3[type][sj1];
  if (type > 2) z *= XP->TermAU;
  return z;
}

/* =================================== MFE fill + traceback =================================== */
typedef struct {
  int n;
  int *c, *fML, *DML; /* (n+2)*(n+2), index [i*(n+2)+j] */
  int *f5;            /* 0..n */
} mfe_tabs;
#define IX(i, j) ((size_t)(i) * (size_t)(n + 2) + (size_t)(j))

static int ml_nb5(const seq_t *q, int i) { return i > 1 ? q->S[i - 1] : -1; }
static int ml_nb3(const seq_t *q, int j) { return j < q->n ? q->S[j + 1] : -1; }

static void mfe_fill(const seq_t *q, mfe_tabs *t) {
  const int n = q->n;
  const int *S = q->S;
  t->n = n;
  size_t sz = (size_t)(n + 2) * (size_t)(n + 2);
  t->c = (int *)malloc(sz * sizeof(int));
  t->fML = (int *)malloc(sz * sizeof(int));
  t->DML = (int *)malloc(sz * sizeof(int));
  t->f5 = (int *)malloc((size_t)(n + 1) * sizeof(int));
  for (size_t k = 0; k < sz; k++) t->c[k] = t->fML[k] = t->DML[k] = INF;
  int *c = t->c, *fML = t->fML, *DML = t->DML;

  for (int d = TURN + 1; d < n; d++) {
    for (int i = 1; i + d <= n; i++) {
      int j = i + d;
      int type = ptype(q, i, j);
      if (type) {
        int e = E_hairpin(d - 1, type, S[i + 1], S[j - 1], q->str + i - 1);
        /* interior loops, (p,q) enclosed with at most MAXLOOP unpaired */
        int pmax = MIN2(j - 2 - TURN, i + MAXLOOP + 1);
        for (int p = i + 1; p <= pmax; p++) {
          int minq = j - i + p - MAXLOOP - 2;
          if (minq < p + 1 + TURN) minq = p + 1 + TURN;
          for (int qq = j - 1; qq >= minq; qq--) {
            int t2 = ptype(q, p, qq);
            if (!t2) continue;
            int en = E_intloop(p - i - 1, j - qq - 1, type, rtype[t2], S[i + 1], S[j - 1], S[p - 1], S[qq + 1]) +
                     sc_stack_term(q, i, j, p, qq) +
                     c[IX(p, qq)];
            e = MIN2(e, en);
          }
        }
        /* multiloop closed by (i,j) */
        int dml = DML[IX(i + 1, j - 1)];
        if (dml < INF) {
          int en = dml + E_mlstem(rtype[type], S[j - 1], S[i + 1]) + P.MLclosing;
          e = MIN2(e, en);
        }
        c[IX(i, j)] = e;
      }
      /* fML */
      int f = INF;
      if (fML[IX(i + 1, j)] < INF) f = MIN2(f, fML[IX(i + 1, j)] + P.MLbase);
      if (fML[IX(i, j - 1)] < INF) f = MIN2(f, fML[IX(i, j - 1)] + P.MLbase);
      if (type) f = MIN2(f, c[IX(i, j)] + E_mlstem(type, ml_nb5(q, i), ml_nb3(q, j)));
      int dec = INF;
      for (int k = i + TURN + 1; k <= j - TURN - 2; k++) {
        int a = fML[IX(i, k)], b = fML[IX(k + 1, j)];
        if (a < INF && b < INF) dec = MIN2(dec, a + b);
      }
      DML[IX(i, j)] = dec;
      f = MIN2(f, dec);
      fML[IX(i, j)] = f;
    }
  }
  int *f5 = t->f5;
  f5[0] = 0;
  for (int j = 1; j <= n; j++) {
    f5[j] = f5[j - 1];
    for (int i = j - TURN - 1; i >= 1; i--) {
      int type = ptype(q, i, j);
      if (!type) continue;
      int en = f5[i - 1] + c[IX(i, j)] + E_extloop(type, ml_nb5(q, i), ml_nb3(q, j));
      f5[j] = MIN2(f5[j], en);
    }
  }
}

static void mfe_free(mfe_tabs *t) {
  free(t->c);
  free(t->fML);
  free(t->DML);
  free(t->f5);
}

/* Traceback; order of alternatives follows SURVEY.md A.3 (ViennaRNA 2.4-style):
 * exterior/fML: 3' base unpaired first; exterior stem partner scanned from j-TURN-1 downwards;
 * fML: then 5' bases unpaired, then the stem (i,j), then the split with ascending k;
 * pair (i,j): hairpin, then interior loops (p ascending, q descending), then the multiloop split
 * with ascending k.  Returns 0, or -1 if no decomposition reproduces a table value. */
static int mfe_traceback(const seq_t *q, const mfe_tabs *t, char *db) {
  const int n = q->n;
  const int *S = q->S;
  const int *c = t->c, *fML = t->fML, *f5 = t->f5;
  typedef struct { int i, j, ml; } sect;
  sect *st = (sect *)malloc(sizeof(sect) * (size_t)(4 * n + 8));
  int s = 0, rc = 0;
  for (int k = 0; k < n; k++) db[k] = '.';
  db[n] = 0;
  st[s++] = (sect){1, n, 0};
  while (s > 0 && rc == 0) {
    sect cur = st[--s];
    int i = cur.i, j = cur.j, ml = cur.ml;
    if (ml == 2) goto paired;
    if (ml == 0) {
      /* exterior: f5[j] */
      while (j > 0 && f5[j] == f5[j - 1]) j--; /* nibble 3' unpaired */
      if (j < TURN + 2) continue;
      int fij = f5[j], k, found = 0;
      for (k = j - TURN - 1; k >= 1; k--) {
        int type = ptype(q, k, j);
        if (!type) continue;
        if (fij == f5[k - 1] + c[IX(k, j)] + E_extloop(type, ml_nb5(q, k), ml_nb3(q, j))) { found = 1; break; }
      }
      if (!found) { rc = -1; break; }
      st[s++] = (sect){1, k - 1, 0};
      i = k;
      goto paired;
    } else {
      /* multiloop part fML[i,j] */
      if (j - i < TURN + 1) { rc = -1; break; }
      while (j - i > TURN + 1 && fML[IX(i, j - 1)] < INF && fML[IX(i, j)] == fML[IX(i, j - 1)] + P.MLbase) j--;
      while (j - i > TURN + 1 && fML[IX(i + 1, j)] < INF && fML[IX(i, j)] == fML[IX(i + 1, j)] + P.MLbase) i++;
      int fij = fML[IX(i, j)];
      int type = ptype(q, i, j);
      if (type && fij == c[IX(i, j)] + E_mlstem(type, ml_nb5(q, i), ml_nb3(q, j))) goto paired;
      int k, found = 0;
      for (k = i + TURN + 1; k <= j - TURN - 2; k++) {
        int a = fML[IX(i, k)], b = fML[IX(k + 1, j)];
        if (a < INF && b < INF && fij == a + b) { found = 1; break; }
      }
      if (!found) { rc = -1; break; }
      st[s++] = (sect){i, k, 1};
      st[s++] = (sect){k + 1, j, 1};
      continue;
    }
  paired:
    for (;;) {
      db[i - 1] = '(';
      db[j - 1] = ')';
      int type = ptype(q, i, j);
      int cij = c[IX(i, j)];
      if (cij == E_hairpin(j - i - 1, type, S[i + 1], S[j - 1], q->str + i - 1)) break;
      int found = 0, p, qq = 0;
      int pmax = MIN2(j - 2 - TURN, i + MAXLOOP + 1);
      for (p = i + 1; p <= pmax && !found; p++) {
        int minq = j - i + p - MAXLOOP - 2;
        if (minq < p + 1 + TURN) minq = p + 1 + TURN;
        for (qq = j - 1; qq >= minq; qq--) {
          int t2 = ptype(q, p, qq);
          if (!t2) continue;
          int en = E_intloop(p - i - 1, j - qq - 1, type, rtype[t2], S[i + 1], S[j - 1], S[p - 1], S[qq + 1]) +
                   sc_stack_term(q, i, j, p, qq) +
                   c[IX(p, qq)];
          if (cij == en) { found = 1; break; }
        }
        if (found) break;
      }
      if (found) { i = p; j = qq; continue; }
      /* multiloop */
      int mm = P.MLclosing + E_mlstem(rtype[type], S[j - 1], S[i + 1]);
      int k, ok = 0;
      for (k = i + 1 + TURN + 1; k <= j - 1 - TURN - 2; k++) {
        int a = fML[IX(i + 1, k)], b = fML[IX(k + 1, j - 1)];
        if (a < INF && b < INF && cij == a + b + mm) { ok = 1; break; }
      }
      if (!ok) { rc = -1; break; }
      st[s++] = (sect){i + 1, k, 1};
      st[s++] = (sect){k + 1, j - 1, 1};
      break;
    }
  }
  free(st);
  return rc;
}

int sfo_mfe(const char *seq, int n, int *mfe_dcal, char *structure) {
  if (!have_params) return -10;
  if (n < 1) { if (mfe_dcal) *mfe_dcal = 0; if (structure) structure[0] = 0; return 0; }
  seq_t q;
  mfe_tabs t;
  seq_init(&q, seq, n);
  if (q.cons_bad) { seq_free(&q); return -3; }
  mfe_fill(&q, &t);
  if (mfe_dcal) *mfe_dcal = t.f5[n];
  int rc = 0;
  if (structure) rc = mfe_traceback(&q, &t, structure);
  mfe_free(&t);
  seq_free(&q);
  return rc;
}

/* many equal-length sequences, energies only — the shape of energies(seq_list) (ScanFold-Scan.py:253-262) */
int sfo_mfe_batch(const char *seqs, int nseq, int W, int *out, int nthreads) {
  if (!have_params) return -10;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
  (void)nthreads;
  const char *keep_c = g_cons; const int *keep_s = g_sc_stack;
  g_cons = NULL; g_sc_stack = NULL;
#pragma omp parallel for schedule(dynamic, 4)
  for (int k = 0; k < nseq; k++) sfo_mfe(seqs + (size_t)k * W, W, &out[k], NULL);
  g_cons = keep_c; g_sc_stack = keep_s;
  return 0;
}

int sfo_pf(const char *seq, int n, double *ensemble_dG, double *bpp_out, char *centroid, double *centroid_dist,
           double *mean_bp_dist);

/* The whole per-window job of ScanFold-Scan.py:382-423 for n_win windows, every window on its own OpenMP
 * thread: rows = n_win*(r+1) sequences of W chars, row 0 of a window is the native one (MFE + traceback +
 * partition function), rows 1..r its shuffles (MFE only).  Used as bench.py's CPU baseline. */
int sfo_scan_windows(const char *rows, int n_win, int r, int W, int *energies, char *structures, char *centroids,
                     double *ens_div, int nthreads) {
  if (!have_params) return -10;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
  (void)nthreads;
  int bad = 0;
  g_cons = NULL; g_sc_stack = NULL; /* batch entry points are unconstrained */
#pragma omp parallel for schedule(dynamic, 1)
  for (int w = 0; w < n_win; w++) {
    const char *base = rows + (size_t)w * (r + 1) * W;
    int rc = sfo_mfe(base, W, &energies[(size_t)w * (r + 1)], structures + (size_t)w * (W + 1));
    double cd;
    rc |= sfo_pf(base, W, NULL, NULL, centroids + (size_t)w * (W + 1), &cd, &ens_div[w]);
    for (int k = 1; k <= r; k++) rc |= sfo_mfe(base + (size_t)k * W, W, &energies[(size_t)w * (r + 1) + k], NULL);
    if (rc) bad = 1;
  }
  return bad ? -1 : 0;
}

/* =================================== independent evaluator =================================== */
static int make_pair_table(const char *db, int n, int *pt) {
  int *stk = (int *)malloc(sizeof(int) * (size_t)(n + 1));
  int s = 0;
  for (int i = 0; i <= n; i++) pt[i] = 0;
  for (int i = 1; i <= n; i++) {
    char ch = db[i - 1];
    if (ch == '(') stk[s++] = i;
    else if (ch == ')') {
      if (!s) { free(stk); return -1; }
      int k = stk[--s];
      pt[k] = i;
      pt[i] = k;
    }
  }
  free(stk);
  return s ? -1 : 0;
}

/* energy of the loop closed by (i,j); *bad set if a pair is non-canonical */
static int eval_loop(const seq_t *q, const int *pt, int i, int j, int *bad) {
  const int *S = q->S;
  int type = ptype(q, i, j);
  if (!type) { *bad = 1; return 0; }
  int p = i + 1, nstems = 0, p1 = 0, q1 = 0, unp = 0;
  int e_stems = 0;
  while (p < j) {
    if (pt[p] == 0) { unp++; p++; continue; }
    int qq = pt[p];
    int t2 = ptype(q, p, qq);
    if (!t2) { *bad = 1; return 0; }
    if (!nstems) { p1 = p; q1 = qq; }
    nstems++;
    e_stems += E_mlstem(t2, S[p - 1], S[qq + 1]);
    p = qq + 1;
  }
  if (nstems == 0) return E_hairpin(j - i - 1, type, S[i + 1], S[j - 1], q->str + i - 1);
  if (nstems == 1)
    return E_intloop(p1 - i - 1, j - q1 - 1, type, rtype[ptype(q, p1, q1)], S[i + 1], S[j - 1], S[p1 - 1], S[q1 + 1]) +
           sc_stack_term(q, i, j, p1, q1);
  return P.MLclosing + E_mlstem(rtype[type], S[j - 1], S[i + 1]) + e_stems + unp * P.MLbase;
}

static int eval_pt(const seq_t *q, const int *pt, int *bad) {
  int n = q->n, e = 0;
  int i = 1;
  while (i <= n) {
    if (pt[i] == 0) { i++; continue; }
    int j = pt[i];
    int type = ptype(q, i, j);
    if (!type) { *bad = 1; return 0; }
    e += E_extloop(type, ml_nb5(q, i), ml_nb3(q, j));
    i = j + 1;
  }
  for (i = 1; i <= n; i++)
    if (pt[i] > i) e += eval_loop(q, pt, i, pt[i], bad);
  return e;
}

int sfo_eval(const char *seq, const char *structure, int n, int *energy) {
  if (!have_params) return -10;
  seq_t q;
  seq_init(&q, seq, n);
  if (q.cons_bad) { seq_free(&q); return -3; }
  int *pt = (int *)malloc(sizeof(int) * (size_t)(n + 2));
  int rc = make_pair_table(structure, n, pt), bad = 0;
  if (rc == 0) {
    *energy = eval_pt(&q, pt, &bad);
    if (bad) rc = -2;
  }
  free(pt);
  seq_free(&q);
  return rc;
}

/* Boltzmann weight of a structure under the partition-function model (smoothed dangles) */
static double weight_loop(const seq_t *q, const int *pt, int i, int j) {
  const int *S = q->S;
  int type = ptype(q, i, j);
  int p = i + 1, nstems = 0, p1 = 0, q1 = 0, unp = 0;
  double w = 1.0;
  while (p < j) {
    if (pt[p] == 0) { unp++; p++; continue; }
    int qq = pt[p];
    if (!nstems) { p1 = p; q1 = qq; }
    nstems++;
    w *= X_mlstem(ptype(q, p, qq), S[p - 1], S[qq + 1]);
    p = qq + 1;
  }
  if (nstems == 0) return X_hairpin(j - i - 1, type, S[i + 1], S[j - 1], q->str + i - 1);
  if (nstems == 1)
    return X_intloop(p1 - i - 1, j - q1 - 1, type, rtype[ptype(q, p1, q1)], S[i + 1], S[j - 1], S[p1 - 1], S[q1 + 1]);
  return XP->MLclosing * X_mlstem(rtype[type], S[j - 1], S[i + 1]) * w * pow(XP->MLbase, unp);
}
static double weight_pt(const seq_t *q, const int *pt) {
  int n = q->n;
  double w = 1.0;
  int i = 1;
  while (i <= n) {
    if (pt[i] == 0) { i++; continue; }
    int j = pt[i];
    w *= X_extloop(ptype(q, i, j), ml_nb5(q, i), ml_nb3(q, j));
    i = j + 1;
  }
  for (i = 1; i <= n; i++)
    if (pt[i] > i) w *= weight_loop(q, pt, i, pt[i]);
  return w;
}

/* =================================== brute force =================================== */
typedef struct {
  const seq_t *q;
  int *pt;
  int best;
  double Z;
  double *bpp; /* (n+1)*(n+1) accumulators or NULL */
  long long count;
} brute_ctx;

static int loops_ok(const seq_t *q, const int *pt) {
  /* the DP only knows interior loops with at most MAXLOOP unpaired bases */
  int n = q->n;
  for (int i = 1; i <= n; i++) {
    if (pt[i] <= i) continue;
    int j = pt[i], p = i + 1, nst = 0, p1 = 0, q1 = 0;
    while (p < j) {
      if (pt[p] == 0) { p++; continue; }
      if (!nst) { p1 = p; q1 = pt[p]; }
      nst++;
      p = pt[p] + 1;
    }
    if (nst == 1 && (p1 - i - 1) + (j - q1 - 1) > MAXLOOP) return 0;
  }
  return 1;
}

/* Simple, obviously-correct enumerator: recursive generation of all nested pairings on [1..n] by
 * choosing for the smallest free position either "unpaired" or a partner k, then recursing on the
 * inside segment and the remainder through a continuation list of segments. */
typedef struct { int lo, hi; } seg;

static void brute_enum(brute_ctx *b, seg *segs, int nseg) {
  const seq_t *q = b->q;
  int n = q->n;
  /* find first non-empty segment */
  while (nseg > 0 && segs[nseg - 1].lo > segs[nseg - 1].hi) nseg--;
  if (nseg == 0) {
    for (int i = 1; i <= n; i++)
      if (b->pt[i] < 0) b->pt[i] = 0;
    if (loops_ok(q, b->pt)) {
      int bad = 0;
      int e = eval_pt(q, b->pt, &bad);
      b->count++;
      if (e < b->best) b->best = e;
      double w = weight_pt(q, b->pt);
      b->Z += w;
      if (b->bpp)
        for (int i = 1; i <= n; i++)
          if (b->pt[i] > i) b->bpp[(size_t)i * (n + 1) + b->pt[i]] += w;
    }
    return;
  }
  seg cur = segs[nseg - 1], above = segs[nseg]; /* every slot written here is restored on exit */
  int pos = cur.lo;
  /* pos unpaired */
  segs[nseg - 1] = (seg){pos + 1, cur.hi};
  b->pt[pos] = 0;
  brute_enum(b, segs, nseg);
  /* pos paired with k in the same segment */
  for (int k = pos + TURN + 1; k <= cur.hi; k++) {
    if (!ptype(q, pos, k)) continue;
    b->pt[pos] = k;
    b->pt[k] = pos;
    segs[nseg - 1] = (seg){k + 1, cur.hi}; /* remainder after k */
    segs[nseg] = (seg){pos + 1, k - 1};    /* inside */
    brute_enum(b, segs, nseg + 1);
    b->pt[pos] = 0;
    b->pt[k] = 0;
  }
  segs[nseg - 1] = cur;
  segs[nseg] = above;
}

int sfo_brute(const char *seq, int n, int *mfe_dcal, double *Z, double *bpp, long long *count) {
  if (!have_params) return -10;
  if (n > 26) return -3;
  seq_t q;
  seq_init(&q, seq, n);
  if (q.cons_bad) { seq_free(&q); return -3; }
  brute_ctx b;
  b.q = &q;
  b.pt = (int *)calloc((size_t)n + 2, sizeof(int));
  b.best = INF;
  b.Z = 0.0;
  b.bpp = bpp;
  b.count = 0;
  if (bpp) memset(bpp, 0, sizeof(double) * (size_t)(n + 1) * (size_t)(n + 1));
  seg *segs = (seg *)calloc((size_t)(n + 4), sizeof(seg));
  segs[0] = (seg){1, n};
  brute_enum(&b, segs, 1);
  if (bpp)
    for (size_t k = 0; k < (size_t)(n + 1) * (size_t)(n + 1); k++) bpp[k] /= b.Z;
  if (mfe_dcal) *mfe_dcal = b.best;
  if (Z) *Z = b.Z;
  if (count) *count = b.count;
  free(segs);
  free(b.pt);
  seq_free(&q);
  return 0;
}

/* =================================== partition function =================================== */
/* Inside: qb, qm, qm1 (unambiguous McCaskill decomposition, dangles=2), q5/q3 exterior.
 * Outside: direct O(n^4) summation over enclosing pairs (an oracle, not a fast path). */
int sfo_pf(const char *seq, int n, double *ensemble_dG, double *bpp_out, char *centroid, double *centroid_dist,
           double *mean_bp_dist) {
  if (!have_params) return -10;
  seq_t q;
  seq_init(&q, seq, n);
  if (q.cons_bad) { seq_free(&q); return -3; }
  const int *S = q.S;
  size_t sz = (size_t)(n + 2) * (size_t)(n + 2);
  double *qb = (double *)calloc(sz, sizeof(double));
  double *qm = (double *)calloc(sz, sizeof(double));
  double *qm1 = (double *)calloc(sz, sizeof(double));
  double *ob = (double *)calloc(sz, sizeof(double));
  double *q5 = (double *)calloc((size_t)n + 2, sizeof(double));
  double *q3 = (double *)calloc((size_t)n + 3, sizeof(double));
  double *mlb = (double *)malloc(sizeof(double) * (size_t)(n + 2)); /* expMLbase^k */
  mlb[0] = 1.0;
  for (int k = 1; k <= n + 1; k++) mlb[k] = mlb[k - 1] * XP->MLbase;

  for (int d = TURN + 1; d < n; d++) {
    for (int i = 1; i + d <= n; i++) {
      int j = i + d;
      int type = ptype(&q, i, j);
      if (type) {
        double z = X_hairpin(d - 1, type, S[i + 1], S[j - 1], q.str + i - 1);
        int pmax = MIN2(j - 2 - TURN, i + MAXLOOP + 1);
        for (int p = i + 1; p <= pmax; p++) {
          int minq = j - i + p - MAXLOOP - 2;
          if (minq < p + 1 + TURN) minq = p + 1 + TURN;
          for (int qq = j - 1; qq >= minq; qq--) {
            int t2 = ptype(&q, p, qq);
            if (!t2) continue;
            z += X_intloop(p - i - 1, j - qq - 1, type, rtype[t2], S[i + 1], S[j - 1], S[p - 1], S[qq + 1]) *
                 qb[IX(p, qq)];
          }
        }
        double ml = 0.0;
        for (int u = i + 2 + TURN; u <= j - 1 - TURN - 1; u++) ml += qm[IX(i + 1, u - 1)] * qm1[IX(u, j - 1)];
        z += ml * XP->MLclosing * X_mlstem(rtype[type], S[j - 1], S[i + 1]);
        qb[IX(i, j)] = z;
      }
      /* qm1[i][j] = sum_l qb[i][l] * stem(i,l) * MLbase^(j-l) */
      double m1 = 0.0;
      for (int l = i + TURN + 1; l <= j; l++) {
        int t2 = ptype(&q, i, l);
        if (t2 && qb[IX(i, l)] != 0.0) m1 += qb[IX(i, l)] * X_mlstem(t2, ml_nb5(&q, i), ml_nb3(&q, l)) * mlb[j - l];
      }
      qm1[IX(i, j)] = m1;
      /* qm[i][j] = sum_u (MLbase^(u-i) + qm[i][u-1]) * qm1[u][j] */
      double m = 0.0;
      for (int u = i; u + TURN + 1 <= j; u++) {
        double left = mlb[u - i] + ((u - 1 >= i) ? qm[IX(i, u - 1)] : 0.0);
        m += left * qm1[IX(u, j)];
      }
      qm[IX(i, j)] = m;
    }
  }
  /* exterior */
  q5[0] = 1.0;
  for (int j = 1; j <= n; j++) {
    double z = q5[j - 1];
    for (int i = 1; i + TURN + 1 <= j; i++) {
      int type = ptype(&q, i, j);
      if (type) z += q5[i - 1] * qb[IX(i, j)] * X_extloop(type, ml_nb5(&q, i), ml_nb3(&q, j));
    }
    q5[j] = z;
  }
  q3[n + 1] = 1.0;
  for (int i = n; i >= 1; i--) {
    double z = q3[i + 1];
    for (int j = i + TURN + 1; j <= n; j++) {
      int type = ptype(&q, i, j);
      if (type) z += qb[IX(i, j)] * X_extloop(type, ml_nb5(&q, i), ml_nb3(&q, j)) * q3[j + 1];
    }
    q3[i] = z;
  }
  double Z = q5[n];
  if (ensemble_dG) *ensemble_dG = -log(Z) * XP->kT / 1000.0;

  /* outside of pairs, widest first */
  for (int d = n - 1; d >= TURN + 1; d--) {
    for (int i = 1; i + d <= n; i++) {
      int j = i + d;
      int type = ptype(&q, i, j);
      if (!type || qb[IX(i, j)] == 0.0) continue;
      double o = q5[i - 1] * q3[j + 1] * X_extloop(type, ml_nb5(&q, i), ml_nb3(&q, j));
      for (int k = MAX2(1, i - MAXLOOP - 1); k < i; k++) {
        int u1 = i - k - 1;
        for (int l = j + 1; l <= n && (l - j - 1) + u1 <= MAXLOOP; l++) {
          int tk = ptype(&q, k, l);
          if (!tk || ob[IX(k, l)] == 0.0) continue;
          o += ob[IX(k, l)] *
               X_intloop(u1, l - j - 1, tk, rtype[type], S[k + 1], S[l - 1], S[i - 1], S[j + 1]);
        }
      }
      /* (i,j) as a stem of a multiloop closed by (k,l) */
      double mlsum = 0.0;
      if (i > 1 && j < n) {
        double stem = X_mlstem(type, S[i - 1], S[j + 1]);
        for (int k = 1; k < i; k++)
          for (int l = j + 1; l <= n; l++) {
            int tk = ptype(&q, k, l);
            if (!tk || ob[IX(k, l)] == 0.0) continue;
            double left_q = (i - 1 >= k + 1) ? qm[IX(k + 1, i - 1)] : 0.0;
            double right_q = (l - 1 >= j + 1) ? qm[IX(j + 1, l - 1)] : 0.0;
            double ctx = left_q * mlb[l - 1 - j] + mlb[i - k - 1] * right_q + left_q * right_q;
            if (ctx == 0.0) continue;
            mlsum += ob[IX(k, l)] * XP->MLclosing * X_mlstem(rtype[tk], S[l - 1], S[k + 1]) * ctx;
          }
        mlsum *= stem;
      }
      o += mlsum;
      ob[IX(i, j)] = o;
    }
  }
  double mbd = 0.0, cdist = 0.0;
  if (bpp_out) memset(bpp_out, 0, sizeof(double) * (size_t)(n + 1) * (size_t)(n + 1));
  if (centroid) {
    for (int k = 0; k < n; k++) centroid[k] = '.';
    centroid[n] = 0;
  }
  for (int i = 1; i <= n; i++)
    for (int j = i + TURN + 1; j <= n; j++) {
      double p = ob[IX(i, j)] * qb[IX(i, j)] / Z;
      if (bpp_out) bpp_out[(size_t)i * (n + 1) + j] = p;
      mbd += p * (1.0 - p);
      if (p > 0.5) {
        if (centroid) { centroid[i - 1] = '('; centroid[j - 1] = ')'; }
        cdist += 1.0 - p;
      } else cdist += p;
    }
  if (mean_bp_dist) *mean_bp_dist = 2.0 * mbd;
  if (centroid_dist) *centroid_dist = cdist;
  free(qb); free(qm); free(qm1); free(ob); free(q5); free(q3); free(mlb);
  seq_free(&q);
  return 0;
}

/* =================================== fast CPU twin (baseline only) =================================== */
#include "sf_cpu_twin.c"
