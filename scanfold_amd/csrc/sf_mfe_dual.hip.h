// sf_mfe_dual.hip.h — the batched Zuker fill for W <= 128 with TWO FOLDS per workgroup sharing every instruction
// of the table passes: all LDS tables hold (fold A, fold B) as the two int16 halves of one 32-bit entry.
//
// Replaces energies(seq_list) -> rna_folder -> RNA.fold(seq) (ScanFold-Scan.py:244-246,253-262;
// ScanFoldFunctions.py:774-789,805-814) like sf_mfe_fast.hip.h — same recurrences, same thread mapping (a lane
// owns a centre, two wave groups work on the even and the odd diagonal of a step, one barrier per step), same
// exterior pass / traceback code, same overflow route to the int32 kernel.  The difference:
//  * the two folds are independent sequences of the same length, so their fills have identical control flow and
//    identical addresses; with the tables interleaved element-wise, ONE aligned 32-bit LDS read fetches the same
//    candidate of both, and v_pk_min_i16 / v_pk_add_i16 clamp do the arithmetic of both.  That covers the three
//    passes that are 76 % of the one-fold kernel's time (generic-loop recurrence, bulge / 1xn minima, multiloop
//    split).  Unlike sf_mfe_pk.hip.h (two neighbouring cells of one fold per lane) no read is ever misaligned, no
//    half is ever invalid on its own, and the lane mapping is unchanged.
//  * only the table look-ups that depend on a fold's own nucleotides (hairpin, special loops, the terms added when
//    the cell is published) run once per fold.
//  * LDS per workgroup doubles (79.4 kB at W=120), so two workgroups = four folds are resident per CU as before,
//    but with 8 waves instead of 16: up to 256 VGPRs per lane, which is what lets the passes be software
//    pipelined (next batch's reads issued before the current batch's arithmetic) instead of load-wait-compute.
// Status (round 1): bit-exact, opt-in (sf_set_kernel_mode(3)), 1.6 M folds/s against 2.4 M for the default kernel.
// Per fold 110 k VALU + 52 k SALU + 20 k LDS wave-instructions (default: 123 k + 86 k + 51 k) — the per-fold look-ups
// run twice and dominate — and hipcc keeps ~110 kernel-lifetime values in scratch (440 B/lane, 9 k scratch reads
// per fold), which is what makes it slower; see DESIGN.md 4.3.
#pragma once
#include "sf_mfe_fast.hip.h"
#include "sf_pk16.h"
#include <type_traits>

#define SF_DUAL_MAXW 128
#define SF_DUAL_NT 256
#define SF_DUAL_NG 128
// fML triangle without diagonals 0..3: base(d) = sum_{k=4}^{d-1} (W-k)   (W: the local window width)
#define FBASE(dd) (((dd)-4) * W - ((dd) * ((dd)-1) / 2 - 6))

// LDS carve (bytes); every table entry is 4 bytes = (fold A, fold B)
struct SfDualLayout {
  int off_fml, off_ci, off_c1n, off_cb, off_dml, off_tab, off_uni, off_flag, off_S, total;
};
static inline __host__ __device__ SfDualLayout sf_dual_layout(int W) {
  SfDualLayout L;
  int tri = FBASE(W);
  if (tri < 0) tri = 0;
  int o = 0;
  L.off_fml = o; o += tri * 4;
  const int RW = W - 4;
  const int roll = SF_FAST_NR * RW * 4;
  L.off_ci = o; o += roll;
  L.off_c1n = o; o += roll;
  L.off_cb = o; o += roll;
  L.off_dml = o; o += 4 * RW * 4;
  L.off_tab = o; o += 5 * 400 + 128 + 64 + 80 + 80;  // mmI, mm1n, mm23, mmM, mmH, stack, pair, d5, d3
  L.off_uni = o; o += 4 * 32 * 4;
  L.off_flag = o; o += 8;
  L.off_S = o; o += 2 * ((W + 2 + 3) & ~3);
  L.total = o;
  return L;
}
static inline bool sf_dual_w_supported(int W) { return W >= 16 && W <= SF_DUAL_MAXW && sf_dual_layout(W).total * 2 <= 160 * 1024; }

struct SfDualTabs {
  uint32_t *fML, *CI, *C1N, *CB, *DML;        // (A, B) entries
  const uint32_t *NIN, *IL, *L1N, *BUL;        // size terms, the value in both halves
};

// One anti-diagonal for one lane, both folds.  XA / XB: per-fold context (nucleotides, c scratch); tables in T.
// HP[x] (x = size - 4): packed per-size minima of the generic candidates of the enclosed cell on entry, of (i, j)
// on exit.  G: d < 36, every size is tested against the (wave-uniform) limit d - 6.
template <bool G, int WT>
__device__ __forceinline__ void sf_dual_cell(const SfFastCtx &XA, const SfFastCtx &XB, const SfDualTabs &T, const int d,
                                             const int i, const int slot2, const int slotd, uint32_t (&HP)[27],
                                             int &ovfA, int &ovfB, const bool final_fml, const uint32_t fnb,
                                             uint32_t &fpart) {
  const int W = WT ? WT : XA.W, RW = W - 4;
  const int j = i + d, i0 = i - 1;
  const int umax = G ? sfd_min(SFD_MAXLOOP, d - 2 - (SFD_TURN + 1)) : SFD_MAXLOOP;
#define ROW(u) ((slot2 - (u) < 0 ? slot2 - (u) + SF_FAST_NR : slot2 - (u)) * RW)
  const uint32_t BIG2 = sf_pk(32767, 32767);
  uint32_t gb = BIG2, g1 = BIG2, gg = BIG2, dec = BIG2;
  if (G) {
    // ---- short diagonals: sizes are tested against the limit; plain loops ----
#pragma unroll
    for (int u = 30; u >= 6; --u) {
      if (u <= umax) {
        const uint32_t *row = T.CI + ROW(u) + i0;
        HP[u - 4] = sf_pkmin(sf_pkadd(sf_pkmin(row[3], row[u - 1]), T.NIN[u - 4]), HP[u - 6]);  // u1 = 2 and u2 = 2
      }
    }
    if (umax >= 5) {
      const uint32_t *row = T.CI + ROW(5) + i0;
      HP[1] = sf_pkadd(sf_pkmin(row[3], row[4]), T.NIN[1]);
    }
    if (umax >= 4) HP[0] = sf_pkadd(T.CI[ROW(4) + i0 + 3], T.NIN[0]);
#pragma unroll
    for (int u = 2; u <= 30; ++u) {
      if (u <= umax) {
        const int rw = ROW(u) + i0;
        gb = sf_pkmin(gb, sf_pkadd(sf_pkmin(T.CB[rw + 1], T.CB[rw + 1 + u]), T.BUL[u]));
        if (u >= 4) g1 = sf_pkmin(g1, sf_pkadd(sf_pkmin(T.C1N[rw + 2], T.C1N[rw + u]), T.L1N[u - 1]));
        if (u >= 6) gg = sf_pkmin(gg, sf_pkadd(HP[u - 4], T.IL[u]));
      }
    }
  } else {
    // ---- all sizes exist: one software pipeline.  Every stage issues the LDS reads of the NEXT batch, then
    // consumes the batch read one stage earlier; SF_PIN ends a stage (the compiler keeps that order). ----
    uint32_t pa[2][5], pb[2][5], pn[2][5];  // pass 1: five sizes per batch, descending from 30
    auto ld1 = [&](const int bt, uint32_t(&a)[5], uint32_t(&b)[5], uint32_t(&n)[5]) {
#pragma unroll
      for (int k = 0; k < 5; k++) {
        const int u = 30 - 5 * bt - k;
        const uint32_t *row = T.CI + ROW(u) + i0;
        a[k] = row[3];      // u1 = 2
        b[k] = row[u - 1];  // u2 = 2
        n[k] = T.NIN[u - 4];
      }
    };
    auto cp1 = [&](const int bt, const uint32_t(&a)[5], const uint32_t(&b)[5], const uint32_t(&n)[5]) {
#pragma unroll
      for (int k = 0; k < 5; k++) {
        const int u = 30 - 5 * bt - k;
        HP[u - 4] = sf_pkmin(sf_pkadd(sf_pkmin(a[k], b[k]), n[k]), HP[u - 6]);
      }
#pragma unroll
      for (int k = 0; k < 5; k++) SF_PIN(HP[30 - 5 * bt - k - 4]);
    };
    // pass 2: three sizes per batch, ascending from 2
    uint32_t qb1[2][3], qb2[2][3], qbt[2][3], qn1[2][3], qn2[2][3], qnt[2][3], qit[2][3];
    auto ld2 = [&](const int bt, const int f) {
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const int u = 2 + 3 * bt + k;
        if (u <= 30) {
          const int rw = ROW(u) + i0;
          qb1[f][k] = T.CB[rw + 1]; qb2[f][k] = T.CB[rw + 1 + u]; qbt[f][k] = T.BUL[u];
          if (u >= 4) { qn1[f][k] = T.C1N[rw + 2]; qn2[f][k] = T.C1N[rw + u]; qnt[f][k] = T.L1N[u - 1]; }
          if (u >= 6) qit[f][k] = T.IL[u];
        }
      }
    };
    auto cp2 = [&](const int bt, const int f) {
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const int u = 2 + 3 * bt + k;
        if (u <= 30) {
          gb = sf_pkmin(gb, sf_pkadd(sf_pkmin(qb1[f][k], qb2[f][k]), qbt[f][k]));
          if (u >= 4) g1 = sf_pkmin(g1, sf_pkadd(sf_pkmin(qn1[f][k], qn2[f][k]), qnt[f][k]));
          if (u >= 6) gg = sf_pkmin(gg, sf_pkadd(HP[u - 4], qit[f][k]));
        }
      }
      SF_PIN(gb); SF_PIN(g1); SF_PIN(gg);
    };
    ld1(0, pa[0], pb[0], pn[0]);
#pragma unroll
    for (int bt = 0; bt < 4; bt++) {
      ld1(bt + 1, pa[(bt + 1) & 1], pb[(bt + 1) & 1], pn[(bt + 1) & 1]);
      cp1(bt, pa[bt & 1], pb[bt & 1], pn[bt & 1]);
    }
    // bridge: sizes 5 and 4 (no recurrence), first batch of pass 2
    const uint32_t *row5 = T.CI + ROW(5) + i0;
    const uint32_t x53 = row5[3], x54 = row5[4], x4 = T.CI[ROW(4) + i0 + 3];
    const uint32_t n1 = T.NIN[1], n0 = T.NIN[0];
    ld2(0, 0);
    cp1(4, pa[0], pb[0], pn[0]);
    HP[1] = sf_pkadd(sf_pkmin(x53, x54), n1);
    HP[0] = sf_pkadd(x4, n0);
#pragma unroll
    for (int bt = 0; bt < 10; bt++) {
      if (bt < 9) ld2(bt + 1, (bt + 1) & 1);
      cp2(bt, bt & 1);
    }
  }

  // ---- multiloop split min_m fML[i, i+m] + fML[i+m+1, j] (both folds at once) ----
  {
    // fML[i, i+m] = fML_tri[FBASE(m) + i0], fML[i+m+1, j] = fML_tri[FBASE(d-m-1) + i0+m+1]; both offsets are
    // wave-uniform and advance by simple differences: FBASE(m+1)-FBASE(m) = W-m.  Eight split points per batch,
    // two batches in flight (ping-pong registers).
    const uint32_t *fa = T.fML + i0;
    const uint32_t *fb2 = T.fML + i0 + 1;
    int ia = 0;                                       // FBASE(4)
    int ib = FBASE(d - SFD_TURN - 2) + SFD_TURN + 1;  // FBASE(d-m-1) + m at m = 4
    int m = SFD_TURN + 1;
    const int mend = d - SFD_TURN - 2;
    uint32_t dec2 = BIG2;
    uint32_t a0[8], b0[8], a1[8], b1[8];
    auto ldm = [&](uint32_t(&a)[8], uint32_t(&b)[8]) {
#pragma unroll
      for (int k = 0; k < 8; k++) {
        a[k] = fa[ia];
        b[k] = fb2[ib];
        ia += W - (m + k);
        ib -= W - d + (m + k) + 1;
      }
      m += 8;
    };
    auto cpm = [&](const uint32_t(&a)[8], const uint32_t(&b)[8]) {
#pragma unroll
      for (int k = 0; k < 8; k += 2) {
        dec = sf_pkmin(dec, sf_pkadd(a[k], b[k]));
        dec2 = sf_pkmin(dec2, sf_pkadd(a[k + 1], b[k + 1]));
      }
    };
    if (m + 7 <= mend) {
      ldm(a0, b0);
      for (;;) {
        if (m + 7 > mend) { cpm(a0, b0); break; }
        ldm(a1, b1);
        cpm(a0, b0);
        if (m + 7 > mend) { cpm(a1, b1); break; }
        ldm(a0, b0);
        cpm(a1, b1);
      }
    }
    for (; m <= mend; m++) {
      dec = sf_pkmin(dec, sf_pkadd(fa[ia], fb2[ib]));
      ia += W - m;
      ib -= W - d + m + 1;
    }
    dec = sf_pkmin(dec, dec2);
  }
  // multiloop closed by the cell: DML of the enclosed cell, diagonal d-2
  const uint32_t dmlc = T.DML[((d - 2) & 3) * RW + i0 + 1];
  // fML neighbours on diagonal d-1: only the even-diagonal group adds them here (fnb, from its own fix-up)
  const uint32_t fn = (final_fml && d > SFD_TURN + 1) ? fnb : BIG2;

  // ---- once per fold: the terms that depend on the fold's own nucleotides, then publish ----
  int cc[2], cI[2], c1n[2], cb[2], ff[2], dd[2];
  const int16_t *CBh = (const int16_t *)T.CB;
#pragma unroll
  for (int h = 0; h < 2; h++) {
    const SfFastCtx &X = h ? XB : XA;
    const uint8_t *S = X.S;
    int c = SF_INF16, f = SF_FAST_BIG;
    int pI = SF_INF16, p1n = SF_INF16, pb = SF_INF16;
    const int type = d <= X.maxd ? X.tPair[S[i] * 8 + S[j]] : 0;
    if (type) {
      const int si1 = S[i + 1], sj1 = S[j - 1];
      const int TAU = X.TAU;
      const sf_params_blob &PB = X.D->P;
      int e;
      if (G && d <= 7) e = sfd_hairpin(X.D, S, i, j, type);  // sizes 3, 4, 6 may be special loops
      else e = X.D->hp_init[d - 1] + X.tH[SF_TIDX(type, si1, sj1)];
      if (!G || umax >= 0) {
        const int tau_out = type > 2 ? TAU : 0;
        const int16_t *st = X.tStack + type * 8;
#define CBV(roff, c) ((int)CBh[((roff) + i0 + (c)) * 2 + h])
        {  // stack
          const int t2r = sfd_rtype(X.tPair[si1 * 8 + sj1]);
          e = sfd_min(e, CBV(ROW(0), 1) - (t2r > 2 ? TAU : 0) + st[t2r]);
        }
        if (!G || umax >= 1) {  // one-nucleotide bulges keep the stack
          const int b1 = sf_lo(T.BUL[1]);
          const int ta = sfd_rtype(X.tPair[si1 * 8 + S[j - 2]]);  // (i+1, j-2)
          e = sfd_min(e, CBV(ROW(1), 1) - (ta > 2 ? TAU : 0) + b1 + st[ta]);
          const int tb = sfd_rtype(X.tPair[S[i + 2] * 8 + sj1]);  // (i+2, j-1)
          e = sfd_min(e, CBV(ROW(1), 2) - (tb > 2 ? TAU : 0) + b1 + st[tb]);
        }
        if (!G || umax >= 2) {  // 1 x 1: (i+2, j-2)
          const int t2r = sfd_rtype(X.tPair[S[i + 2] * 8 + S[j - 2]]);
          e = sfd_min(e, CBV(ROW(2), 2) - (t2r > 2 ? TAU : 0) + PB.int11[type][t2r][si1][sj1]);
        }
        if (!G || umax >= 3) {  // 1 x 2 and 2 x 1
          const int ta = sfd_rtype(X.tPair[S[i + 2] * 8 + S[j - 3]]);  // (i+2, j-3), sq1 = S[j-2]
          e = sfd_min(e, CBV(ROW(3), 2) - (ta > 2 ? TAU : 0) + PB.int21[type][ta][si1][S[j - 2]][sj1]);
          const int tb = sfd_rtype(X.tPair[S[i + 3] * 8 + S[j - 2]]);  // (i+3, j-2), sp1 = S[i+2]
          e = sfd_min(e, CBV(ROW(3), 3) - (tb > 2 ? TAU : 0) + PB.int21[tb][type][sj1][si1][S[i + 2]]);
        }
        if (!G || umax >= 4) {  // 2 x 2: (i+3, j-3)
          const int t2r = sfd_rtype(X.tPair[S[i + 3] * 8 + S[j - 3]]);
          e = sfd_min(e, CBV(ROW(4), 3) - (t2r > 2 ? TAU : 0) + PB.int22[type][t2r][si1][S[i + 2]][S[j - 2]][sj1]);
        }
        if (!G || umax >= 5) {  // 2 x 3 and 3 x 2
          const int m23 = X.t23[SF_TIDX(type, si1, sj1)] + X.F->L23;
          const int ta = sfd_rtype(X.tPair[S[i + 3] * 8 + S[j - 4]]);  // (i+3, j-4); sp1 = S[i+2], sq1 = S[j-3]
          e = sfd_min(e, CBV(ROW(5), 3) - (ta > 2 ? TAU : 0) + m23 + X.t23[SF_TIDX(ta, S[j - 3], S[i + 2])]);
          const int tb = sfd_rtype(X.tPair[S[i + 4] * 8 + S[j - 3]]);  // (i+4, j-3); sp1 = S[i+3], sq1 = S[j-2]
          e = sfd_min(e, CBV(ROW(5), 4) - (tb > 2 ? TAU : 0) + m23 + X.t23[SF_TIDX(tb, S[j - 2], S[i + 3])]);
        }
#undef CBV
        e = sfd_min(e, (h ? sf_hi(gb) : sf_lo(gb)) + tau_out);
        e = sfd_min(e, (h ? sf_hi(g1) : sf_lo(g1)) + X.t1n[SF_TIDX(type, si1, sj1)]);
        e = sfd_min(e, (h ? sf_hi(gg) : sf_lo(gg)) + X.tI[SF_TIDX(type, si1, sj1)]);
      }
      const int tr = sfd_rtype(type);
      e = sfd_min(e, (h ? sf_hi(dmlc) : sf_lo(dmlc)) + X.tM[SF_TIDX(tr, sj1, si1)] + (tr > 2 ? TAU : 0) + X.MLintern + X.MLclosing);
      c = e;
      if (c < SF_FAST_OVF) { if (h) ovfB = 1; else ovfA = 1; }
      const int sp1 = S[i - 1], sq1 = S[j + 1];
      const int tau_in = tr > 2 ? TAU : 0;
      pI = c + X.tI[SF_TIDX(tr, sq1, sp1)];
      p1n = c + X.t1n[SF_TIDX(tr, sq1, sp1)];
      pb = c + tau_in;
      int stem;  // E_MLstem(type, S[i-1], S[j+1]); sequence ends have dangles only
      if (i > 1 && j < W) stem = X.tM[SF_TIDX(type, sp1, sq1)];
      else if (i > 1) stem = X.tD5[type * 5 + sp1];
      else if (j < W) stem = X.tD3[type * 5 + sq1];
      else stem = 0;
      f = c + stem + tau_in + X.MLintern;
    }
    const int dech = h ? sf_hi(dec) : sf_lo(dec);
    f = sfd_min(f, sfd_min(dech, h ? sf_hi(fn) : sf_lo(fn)));
    if (final_fml && f < SF_FAST_OVF) { if (h) ovfB = 1; else ovfA = 1; }
    cc[h] = c; cI[h] = pI; c1n[h] = p1n; cb[h] = pb;
    ff[h] = f > SF_FAST_THRESH ? SF_INF16 : f;
    dd[h] = dech > SF_FAST_THRESH ? SF_INF16 : dech;
  }

  // ---- publish ----
  const int rbd = slotd * RW + i0;
  T.CI[rbd] = sf_pk(cI[0], cI[1]);
  T.C1N[rbd] = sf_pk(c1n[0], c1n[1]);
  T.CB[rbd] = sf_pk(cb[0], cb[1]);
  T.DML[(d & 3) * RW + i0] = sf_pk(dd[0], dd[1]);
  XA.cg[SF_CGIDX(i, j)] = (int16_t)cc[0];
  XB.cg[SF_CGIDX(i, j)] = (int16_t)cc[1];
  fpart = sf_pk(ff[0], ff[1]);
  // final on the even diagonal; provisional (neighbour term still missing) on the odd one
  T.fML[FBASE(d) + i0] = fpart;
#undef ROW
}

template <int WT>
__global__ __launch_bounds__(SF_DUAL_NT, 2) void sf_mfe_dual_kernel(const uint8_t *__restrict__ seqs, int n, int Wrt,
                                                                   const SfDevParams *__restrict__ D,
                                                                   const SfFastParams *__restrict__ F,
                                                                   int16_t *__restrict__ cg_all, int32_t *__restrict__ out,
                                                                   int *__restrict__ ovf_cnt, int *__restrict__ ovf_list,
                                                                   int trace_stride, char *__restrict__ db_out,
                                                                   int *__restrict__ status) {
  constexpr int NT = SF_DUAL_NT, NG = SF_DUAL_NG;
  const int W = WT ? WT : Wrt;
  SF_DYN_SMEM(smem);
  const SfDualLayout Lo = sf_dual_layout(W);
  const int RW = W - 4;
  SfDualTabs T;
  T.fML = (uint32_t *)(smem + Lo.off_fml);
  T.CI = (uint32_t *)(smem + Lo.off_ci);
  T.C1N = (uint32_t *)(smem + Lo.off_c1n);
  T.CB = (uint32_t *)(smem + Lo.off_cb);
  T.DML = (uint32_t *)(smem + Lo.off_dml);
  int16_t *tab = (int16_t *)(smem + Lo.off_tab);
  uint32_t *uni = (uint32_t *)(smem + Lo.off_uni);
  T.NIN = uni; T.IL = uni + 32; T.L1N = uni + 64; T.BUL = uni + 96;
  int32_t *flag = (int32_t *)(smem + Lo.off_flag);  // [2]
  const int s_stride = (W + 2 + 3) & ~3;
  uint8_t *S0 = (uint8_t *)(smem + Lo.off_S);
  SfFastCtx XA;
  XA.fML = (int16_t *)T.fML; XA.CI = nullptr; XA.C1N = nullptr; XA.CB = nullptr; XA.DMLr = nullptr;
  XA.tI = tab; XA.t1n = tab + 200; XA.t23 = tab + 400; XA.tM = tab + 600; XA.tH = tab + 800;
  XA.tStack = tab + 1000;
  uint8_t *tPair = (uint8_t *)(tab + 1064);
  XA.tPair = tPair;
  XA.tD5 = tab + 1096; XA.tD3 = tab + 1136;
  XA.S = S0;
  XA.D = D; XA.F = F; XA.W = W; XA.fml_pad = 0; XA.fst = 2; XA.maxd = D->max_pair_dist; XA.cg_ext = 0; XA.tE = nullptr;
  XA.TAU = D->P.TerminalAU; XA.MLbase = D->P.MLbase; XA.MLclosing = D->P.MLclosing; XA.MLintern = D->P.MLintern[1];
  XA.uNIN = nullptr; XA.uIL = nullptr; XA.uL1N = nullptr; XA.uBUL = nullptr;
  XA.cg = cg_all + (size_t)blockIdx.x * 2 * SF_CG_ENTRIES(W);
  SfFastCtx XB = XA;
  XB.fML = (int16_t *)T.fML + 1;
  XB.S = S0 + s_stride;
  XB.cg = XA.cg + SF_CG_ENTRIES(W);

  const int tid = threadIdx.x;
  // parameter tables -> LDS, once per workgroup
  for (int x = tid; x < 200; x += NT) {
    tab[x] = F->mmI[x]; tab[200 + x] = F->mm1n[x]; tab[400 + x] = F->mm23[x]; tab[600 + x] = F->mmM[x];
    tab[800 + x] = F->mmH[x];
  }
  for (int x = tid; x < 64; x += NT) { tab[1000 + x] = F->stack[x]; tPair[x] = F->pair[x]; }
  for (int x = tid; x < 40; x += NT) { tab[1096 + x] = F->d5[x]; tab[1136 + x] = F->d3[x]; }
  for (int x = tid; x < 32; x += NT) {
    const int a = sfd_min(F->NIN[x], 32000), b = sfd_min(F->IL[x], 32000), c = sfd_min(F->L1N[x], 32000),
              e = sfd_min(F->BUL[x], 32000);
    uni[x] = sf_pk(a, a); uni[32 + x] = sf_pk(b, b); uni[64 + x] = sf_pk(c, c); uni[96 + x] = sf_pk(e, e);
  }
  // group and centre-based mapping inside the group: v = (tg + OFF) mod NG, cell i = v - d/2
  const int grp = SF_WAVE_UNIFORM(tid / NG);
  const int tg = tid - grp * NG;
  const int OFF = (((W + 1) >> 1) - 32 + NG) & (NG - 1);
  const int v = (tg + OFF) & (NG - 1);
  const int npairs = (n + 1) >> 1;

  for (int pr = blockIdx.x; pr < npairs; pr += gridDim.x) {
    const int seqA = 2 * pr, seqB = (2 * pr + 1 < n) ? 2 * pr + 1 : 2 * pr;  // an odd tail folds its last sequence twice
    __syncthreads();
    for (int x = tid; x < W; x += NT) {
      S0[x + 1] = sf_encode_nt(seqs[(size_t)seqA * W + x]);
      S0[s_stride + x + 1] = sf_encode_nt(seqs[(size_t)seqB * W + x]);
    }
    if (tid == 0) { S0[0] = 0; S0[W + 1] = 0; S0[s_stride] = 0; S0[s_stride + W + 1] = 0; flag[0] = 0; flag[1] = 0; }
    for (int x = tid; x < 4 * RW; x += NT) T.DML[x] = sf_pk(SF_INF16, SF_INF16);  // diagonals 2,3 have no multiloop split
    __syncthreads();
    int ovfA = 0, ovfB = 0;
    uint32_t H[27];
#pragma unroll
    for (int k = 0; k < 27; k++) H[k] = sf_pk(SF_INF16, SF_INF16);
    uint32_t fnb = sf_pk(32767, 32767);  // even group: min of the two fML neighbours of the next cell, + MLbase

    int slot2 = (SFD_TURN + 1 + grp - 2) % SF_FAST_NR, slotd = (SFD_TURN + 1 + grp) % SF_FAST_NR;
    auto step = [&](const int d0, auto GT) {
      constexpr bool G = decltype(GT)::value;
      const int d = d0 + grp;
      const int i = v - (d >> 1);
      const bool valid = (d < W) && (i >= 1) && (i + d <= W);
      uint32_t fpart = sf_pk(32767, 32767);
      if (valid) sf_dual_cell<G, WT>(XA, XB, T, d, i, slot2, slotd, H, ovfA, ovfB, grp == 0, fnb, fpart);
      __syncthreads();
      // fML on the odd diagonal d0+1, finished by the EVEN group (see sf_mfe_fast.hip.h): cells i and i-1
      if (grp == 0 && valid) {
        const int d1 = d0 + 1;
        const int fbd = FBASE(d1), fbe = FBASE(d0);
        const uint32_t mlb2 = sf_pk(XA.MLbase, XA.MLbase);
        uint32_t g[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const int x = i - h;  // cell (x, x+d1)
          g[h] = sf_pk(SF_INF16, SF_INF16);
          if (x >= 1 && x + d1 <= W) {
            const uint32_t v2 = sf_pkmin(T.fML[fbd + x - 1], sf_pkadd(sf_pkmin(T.fML[fbe + x], T.fML[fbe + x - 1]), mlb2));
            const int va = sf_lo(v2), vb = sf_hi(v2);
            if (va < SF_FAST_OVF) ovfA = 1;
            if (vb < SF_FAST_OVF) ovfB = 1;
            g[h] = sf_pk(va > SF_FAST_THRESH ? SF_INF16 : va, vb > SF_FAST_THRESH ? SF_INF16 : vb);
            if (h == 0) T.fML[fbd + x - 1] = g[h];
          }
        }
        fnb = sf_pkadd(sf_pkmin(g[0], g[1]), mlb2);
      }
      slot2 += 2; if (slot2 >= SF_FAST_NR) slot2 -= SF_FAST_NR;
      slotd += 2; if (slotd >= SF_FAST_NR) slotd -= SF_FAST_NR;
    };
    int d0 = SFD_TURN + 1;
    for (; d0 < W && d0 < SFD_MAXLOOP + 6; d0 += 2) step(d0, std::true_type{});  // some loop sizes do not fit yet
    for (; d0 < W; d0 += 2) step(d0, std::false_type{});

    if (ovfA) flag[0] = 1;
    if (ovfB) flag[1] = 1;
    __syncthreads();
    // ---- exterior pass + traceback, one wave per fold (waves 0 and 2), in the dead rolling tables ----
    // per fold: f5[] (int32), c + ExtLoop table, traceback stacks, structure string; the mismatchExt table is shared
    int16_t *tExt = (int16_t *)(smem + Lo.off_ci);
    const int per_fold = (((W + 1) * 4 + 3) & ~3) + SF_CG_ENTRIES(W) * 2 + ((3 * (W + 8) * 2 + 3) & ~3) + ((W + 1 + 3) & ~3);
    for (int x = tid; x < 200; x += NT) tExt[x] = F->mmExt[x];
    __syncthreads();
    {
      const int h = tid >> 7;  // waves 0,1 -> fold A; waves 2,3 -> fold B
      const SfFastCtx &X = h ? XB : XA;
      char *base = smem + Lo.off_ci + 400 + h * per_fold;
      int32_t *f5s = (int32_t *)base;
      int16_t *etab = (int16_t *)(base + (((W + 1) * 4 + 3) & ~3));
      int16_t *stk = etab + SF_CG_ENTRIES(W);
      char *dbL = (char *)stk + ((3 * (W + 8) * 2 + 3) & ~3);
      sf_fast_ext_table(X, W, tid & 127, 128, tExt, etab);
      __syncthreads();
      const int seq = h ? seqB : seqA;
      if ((tid & 127) < 64 && (h == 0 || seqB != seqA))
        sf_fast_exterior<2>(X, W, tid & 63, seq, f5s, tExt, etab, flag + h, stk, dbL, out, ovf_cnt, ovf_list, trace_stride,
                            db_out, status);
    }
  }
}

static inline hipError_t sf_dual_configure() {
  hipError_t e = hipFuncSetAttribute((const void *)sf_mfe_dual_kernel<120>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute((const void *)sf_mfe_dual_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

// grid / LDS / scratch for n folds of W nt on a chip with n_cu CUs
static inline void sf_dual_geometry(int W, int n_cu, int n, int *grid, size_t *lds, size_t *scratch) {
  const SfDualLayout L = sf_dual_layout(W);
  int per_cu = (160 * 1024) / L.total;
  if (per_cu > 4) per_cu = 4;
  if (per_cu < 1) per_cu = 1;
  int g = n_cu * per_cu;
  const int npairs = (n + 1) / 2;
  if (g > npairs) g = npairs;
  *grid = g;
  *lds = (size_t)L.total;
  *scratch = (size_t)g * 2 * SF_CG_ENTRIES(W) * sizeof(int16_t);
}
#undef FBASE

template <typename... A>
static inline void sf_dual_launch(int grid, int W, size_t lds, hipStream_t st, A... args) {
  if (W == 120) SF_LAUNCH((sf_mfe_dual_kernel<120>), grid, SF_DUAL_NT, lds, st, args...);
  else SF_LAUNCH((sf_mfe_dual_kernel<0>), grid, SF_DUAL_NT, lds, st, args...);
}
