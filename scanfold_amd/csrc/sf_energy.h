// sf_energy.h — nearest-neighbour loop energies on the device (integer dcal/mol) and their
// Boltzmann-weight twins.  These are the loop rules ViennaRNA applies under RNA.md() defaults
// (SURVEY.md A.2), which is what RNA.fold / fc.mfe / fc.pf evaluate for ScanFold-Scan.py:245,382-389.
#pragma once
#include "sf_dev_params.h"
#include "sf_launch.h"

#define SFD_INF SF_INF
#define SFD_TURN SF_TURN
#define SFD_MAXLOOP SF_MAXLOOP

// reverse pair type: CG <-> GC, GU <-> UG, AU <-> UA; 0 and the non-standard type 7 (a pair forced by a constraint) map to themselves
__device__ __forceinline__ int sfd_rtype(int t) { return (t && t < 7) ? (((t - 1) ^ 1) + 1) : t; }
__device__ __forceinline__ int sfd_min(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ short sfd_min16(short a, short b) { return a < b ? a : b; }  // v_min_i16: full rate (32-bit v_min_i32 is half-rate on MI355X)
__device__ __forceinline__ int sfd_max(int a, int b) { return a > b ? a : b; }

// S points at 1-based codes (S[1..W]); the loop closed by (i,j) has j-i-1 unpaired bases.
__device__ __forceinline__ uint32_t sfd_loop_key(const uint8_t *S, int i, int len) {
  uint32_t k = 0;
  for (int x = 0; x < len; x++) k = (k << 3) | S[i + x];
  return k;
}

__device__ inline int sfd_special_hairpin(const SfDevParams *D, const uint8_t *S, int i, int size, int *found) {
  *found = 0;
  if (size == 4) {
    uint32_t key = sfd_loop_key(S, i, 6);
    for (int k = 0; k < D->P.n_tetra; k++)
      if (D->tetra_key[k] == key) { *found = 1; return k; }
  } else if (size == 6) {
    uint32_t key = sfd_loop_key(S, i, 8);
    for (int k = 0; k < D->P.n_hexa; k++)
      if (D->hexa_key[k] == key) { *found = 1; return k; }
  } else if (size == 3) {
    uint32_t key = sfd_loop_key(S, i, 5);
    for (int k = 0; k < D->P.n_tri; k++)
      if (D->tri_key[k] == key) { *found = 1; return k; }
  }
  return -1;
}

__device__ inline int sfd_hairpin(const SfDevParams *D, const uint8_t *S, int i, int j, int type) {
  const int size = j - i - 1;
  int found;
  const int k = sfd_special_hairpin(D, S, i, size, &found);
  if (found) return size == 4 ? D->P.tetra_E[k] : (size == 6 ? D->P.hexa_E[k] : D->P.tri_E[k]);
  const int ge = D->hp_init[size];
  if (size == 3) return ge + (type > 2 ? D->P.TerminalAU : 0);
  return ge + D->P.mismatchH[type][S[i + 1]][S[j - 1]];
}

// interior loop closed by (i,j) [type] and (p,q) [type_2 = rtype of the inner pair]; n1+n2 <= MAXLOOP
__device__ inline int sfd_intloop(const SfDevParams *D, int n1, int n2, int type, int type_2, int si1, int sj1, int sp1,
                                  int sq1) {
  const sf_params_blob &P = D->P;
  const int nl = n1 > n2 ? n1 : n2, ns = n1 > n2 ? n2 : n1;
  if (nl == 0) return P.stack[type][type_2];
  if (ns == 0) {
    int e = P.bulge[nl];
    if (nl == 1) return e + P.stack[type][type_2];
    if (type > 2) e += P.TerminalAU;
    if (type_2 > 2) e += P.TerminalAU;
    return e;
  }
  if (ns == 1) {
    if (nl == 1) return P.int11[type][type_2][si1][sj1];
    if (nl == 2) return n1 == 1 ? P.int21[type][type_2][si1][sq1][sj1] : P.int21[type_2][type][sq1][si1][sp1];
    return P.internal_loop[nl + 1] + sfd_min(P.max_ninio, (nl - ns) * P.ninio) + P.mismatch1nI[type][si1][sj1] +
           P.mismatch1nI[type_2][sq1][sp1];
  }
  if (ns == 2) {
    if (nl == 2) return P.int22[type][type_2][si1][sp1][sq1][sj1];
    if (nl == 3)
      return P.internal_loop[5] + P.ninio + P.mismatch23I[type][si1][sj1] + P.mismatch23I[type_2][sq1][sp1];
  }
  return P.internal_loop[nl + ns] + sfd_min(P.max_ninio, (nl - ns) * P.ninio) + P.mismatchI[type][si1][sj1] +
         P.mismatchI[type_2][sq1][sp1];
}

// si1 / sj1 < 0: that neighbour does not exist (sequence end)
__device__ inline int sfd_mlstem(const SfDevParams *D, int type, int si1, int sj1) {
  const sf_params_blob &P = D->P;
  int e = P.MLintern[type];
  if (si1 >= 0 && sj1 >= 0) e += P.mismatchM[type][si1][sj1];
  else if (si1 >= 0) e += P.dangle5[type][si1];
  else if (sj1 >= 0) e += P.dangle3[type][sj1];
  if (type > 2) e += P.TerminalAU;
  return e;
}
__device__ inline int sfd_extloop(const SfDevParams *D, int type, int si1, int sj1) {
  const sf_params_blob &P = D->P;
  int e = 0;
  if (si1 >= 0 && sj1 >= 0) e += P.mismatchExt[type][si1][sj1];
  else if (si1 >= 0) e += P.dangle5[type][si1];
  else if (sj1 >= 0) e += P.dangle3[type][sj1];
  if (type > 2) e += P.TerminalAU;
  return e;
}

// ---------------- Boltzmann-weight twins ----------------
__device__ inline double sfx_hairpin(const SfDevParams *D, const SfDevParamsPF *X, const uint8_t *S, int i, int j,
                                     int type) {
  const int size = j - i - 1;
  int found;
  const int k = sfd_special_hairpin(D, S, i, size, &found);
  if (found) return size == 4 ? X->tetra[k] : (size == 6 ? X->hexa[k] : X->tri[k]);
  const double q = X->hp_init[size];
  if (size == 3) return type > 2 ? q * X->TermAU : q;
  return q * X->mismatchH[type][S[i + 1]][S[j - 1]];
}
__device__ inline double sfx_intloop(const SfDevParamsPF *X, int n1, int n2, int type, int type_2, int si1, int sj1,
                                     int sp1, int sq1) {
  const int nl = n1 > n2 ? n1 : n2, ns = n1 > n2 ? n2 : n1;
  if (nl == 0) return X->stack[type][type_2];
  if (ns == 0) {
    double z = X->bulge[nl];
    if (nl == 1) return z * X->stack[type][type_2];
    if (type > 2) z *= X->TermAU;
    if (type_2 > 2) z *= X->TermAU;
    return z;
  }
  if (ns == 1) {
    if (nl == 1) return X->int11[type][type_2][si1][sj1];
    if (nl == 2) return n1 == 1 ? X->int21[type][type_2][si1][sq1][sj1] : X->int21[type_2][type][sq1][si1][sp1];
    return X->internal_loop[nl + 1] * X->ninio[nl - ns] * X->mismatch1nI[type][si1][sj1] *
           X->mismatch1nI[type_2][sq1][sp1];
  }
  if (ns == 2) {
    if (nl == 2) return X->int22[type][type_2][si1][sp1][sq1][sj1];
    if (nl == 3)
      return X->internal_loop[5] * X->ninio[1] * X->mismatch23I[type][si1][sj1] * X->mismatch23I[type_2][sq1][sp1];
  }
  return X->internal_loop[nl + ns] * X->ninio[nl - ns] * X->mismatchI[type][si1][sj1] *
         X->mismatchI[type_2][sq1][sp1];
}
__device__ inline double sfx_mlstem(const SfDevParamsPF *X, int type, int si1, int sj1) {
  double z = X->MLintern[type];
  if (si1 >= 0 && sj1 >= 0) z *= X->mismatchM[type][si1][sj1];
  else if (si1 >= 0) z *= X->dangle5[type][si1];
  else if (sj1 >= 0) z *= X->dangle3[type][sj1];
  if (type > 2) z *= X->TermAU;
  return z;
}
__device__ inline double sfx_extloop(const SfDevParamsPF *X, int type, int si1, int sj1) {
  double z = 1.0;
  if (si1 >= 0 && sj1 >= 0) z = X->mismatchExt[type][si1][sj1];
  else if (si1 >= 0) z = X->dangle5[type][si1];
  else if (sj1 >= 0) z = X->dangle3[type][sj1];
  if (type > 2) z *= X->TermAU;
  return z;
}

// ---------------- hard constraints of one window (fc.hc_add_from_db, ScanFold-Scan.py:405-410) ----------------
// ViennaRNA 2.4's vrna_hc_add_from_db with its default options (no enforce) [EXT]: 'x' stays unpaired; '<' / '>' may only
// pair downstream / upstream; '(' ')' may pair with each other only (type 7 if the bases are not complementary) and
// nothing may cross them; '|' and '.' change nothing.  Parsed once per window by one thread into LDS.
struct SfHc {
  const char *c;           // the window's constraint characters, 1-based (c[1..W]); null: no constraint
  const int16_t *partner;  // bracket partner of a position, 0: none
  const int16_t *encl;     // opening position of the innermost bracket pair around a position, 0: none
};
// returns 0 if the brackets balance.  c / partner / encl / stack: LDS arrays of W + 2 entries, filled for 1..W
__device__ inline int sf_hc_parse(const char *src, int W, char *c, int16_t *partner, int16_t *encl, int16_t *stack) {
  int sp = 0, bad = 0;
  for (int i = 1; i <= W; i++) {
    const char ch = src[i - 1];
    c[i] = ch;
    partner[i] = 0;
  }
  for (int i = 1; i <= W && !bad; i++) {
    const char ch = c[i];
    if (ch == ')') {
      if (sp == 0) { bad = 1; break; }
      const int o = stack[--sp];
      partner[o] = (int16_t)i;
      partner[i] = (int16_t)o;
    }
    encl[i] = sp ? stack[sp - 1] : 0;
    if (ch == '(') stack[sp++] = (int16_t)i;
  }
  return bad || sp != 0;
}
// pair type of (i, j), i < j, under the constraint; t = the unconstrained type (0 where the span limit forbids the pair)
__device__ __forceinline__ int sf_hc_type(const SfHc &h, int t, int i, int j, bool span_ok) {
  if (!h.c) return t;
  if (h.partner[i] || h.partner[j]) return (h.partner[i] == j && span_ok) ? (t ? t : 7) : 0;
  const char ci = h.c[i], cj = h.c[j];
  if (ci == 'x' || cj == 'x' || ci == '>' || cj == '<') return 0;
  if (h.encl[i] != h.encl[j]) return 0;  // would cross a bracket pair
  return t;
}

// The same for the LDS-resident kernels, whose windows are at most 250 nucleotides: bracket partner and enclosing pair fit a
// byte each (LDS is what those kernels run out of).  No separate parse stack: the open positions are chained through their
// own `partner` entries until they close.
struct SfHc8 {
  const char *c;           // 1-based (c[1..W]); null: no constraint
  const uint8_t *partner;  // 0: none
  const uint8_t *encl;     // 0: none
};
// c[1..W] filled by the caller; returns 0 if the brackets balance
__device__ inline int sf_hc_parse8(int W, const char *c, uint8_t *partner, uint8_t *encl) {
  int top = 0, bad = 0;
  for (int i = 1; i <= W; i++) partner[i] = 0;
  for (int i = 1; i <= W && !bad; i++) {
    const char ch = c[i];
    if (ch == ')') {
      if (top == 0) { bad = 1; break; }
      const int o = top;
      top = partner[o];  // the open position below it
      partner[o] = (uint8_t)i;
      partner[i] = (uint8_t)o;
    }
    encl[i] = (uint8_t)top;
    if (ch == '(') {
      partner[i] = (uint8_t)top;  // (chain; overwritten when the pair closes)
      top = i;
    }
  }
  return bad || top != 0;
}
__device__ __forceinline__ int sf_hc_type8(const SfHc8 &h, int t, int i, int j, bool span_ok) {
  if (!h.c) return t;
  const int pi = h.partner[i], pj = h.partner[j];
  if (pi || pj) return (pi == j && span_ok) ? (t ? t : 7) : 0;
  const char ci = h.c[i], cj = h.c[j];
  if (ci == 'x' || cj == 'x' || ci == '>' || cj == '<') return 0;
  if (h.encl[i] != h.encl[j]) return 0;  // would cross a bracket pair
  return t;
}

// ASCII or code -> code 0..4 (N,A,C,G,U)
__device__ __host__ inline uint8_t sf_encode_nt(uint8_t c) {
  if (c <= 4) return c;
  switch (c) {
    case 'A': case 'a': return 1;
    case 'C': case 'c': return 2;
    case 'G': case 'g': return 3;
    case 'U': case 'u': case 'T': case 't': return 4;
    default: return 0;
  }
}
