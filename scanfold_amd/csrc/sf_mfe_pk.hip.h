// sf_mfe_pk.hip.h — the hot kernel for W <= 128: the LDS-resident int16 Zuker fill of sf_mfe_fast.hip.h with TWO
// cells per lane, processed with packed 16-bit arithmetic (v_pk_min_i16 / v_pk_add_i16 clamp).
//
// Replaces energies(seq_list) -> rna_folder -> RNA.fold(seq) (ScanFold-Scan.py:244-246,253-262;
// ScanFoldFunctions.py:774-789,805-814) exactly like sf_mfe_fast.hip.h (same tables, same recurrences, same
// exterior pass / traceback code, same overflow route to the int32 kernel); what changes is the mapping:
//  * a lane owns two neighbouring centres, i.e. the cells (i, j) and (i+1, j+1) of its anti-diagonal.  Every
//    table is diagonal-major, so for any candidate (p, q) of the first cell the same candidate of the second is
//    the next int16 in memory: ONE 32-bit LDS read (gfx950 reads LDS at any 2-byte alignment) feeds both, and
//    min / add run on both halves at once.  Everything that is a minimum over loop sizes — the Lyngso
//    recurrence for generic loops, bulges, 1 x n loops, the multiloop split — is done packed; only the few
//    table look-ups that depend on the cell's own nucleotides (hairpin, special loops, the terms added when the
//    cell is published) run once per half.
//  * an anti-diagonal of a 120-mer has at most 116 cells = 58 lanes: one wave per diagonal, two waves (the even
//    and the odd diagonal of a step) per fold — half the waves of sf_mfe_fast.hip.h for the same work, each
//    executing well under half the instructions per cell.
//  * LDS alignment.  A 32-bit LDS read at an odd int16 index is legal on gfx950 but runs at ONE LANE PER CYCLE
//    (measured: 64.5 cycles per wave-read against 2.5 aligned, tools/micro/lds_unaligned.hip), so no read here is
//    ever misaligned: every table row starts at an even index (the fML triangle pads odd diagonals by one
//    entry), the column of a lane's first cell has a parity P that is the same for the whole workgroup and
//    alternates from step to step (the step loop is unrolled by two and the cell code is instantiated for both),
//    so the parity of every pair address is known at compile time; an odd pair is fetched as the two aligned
//    dwords around it (one ds_read2_b32) and one v_alignbit_b32.
// Additions saturate (clamp), so INF (SF_INF16) stays far above SF_FAST_THRESH through any single sum; values are
// renormalised to SF_INF16 when stored, as in the unpacked kernel.
#pragma once
#include "sf_mfe_fast.hip.h"
#include "sf_pk16.h"
#include <type_traits>

#define SF_PK_MAXW 128
#define SF_PK_BIG 32767
// fML triangle without diagonals 0..3, every diagonal padded to an even length:
// base(d) = sum_{k=4}^{d-1} (W-k rounded up to even)   (W: the local window width)
#define FBASE(dd) ((((dd)-4) * W - ((dd) * ((dd)-1) / 2 - 6)) + ((((dd) + (W & 1)) >> 1) - 2))
#define FLEN(dd) ((W - (dd) + 1) & ~1)  // padded length of diagonal dd = FBASE(dd+1) - FBASE(dd)

// LDS carve (bytes).  Same pieces as SfFastLayout; the five 200-entry mismatch tables overlap by 25 entries
// (pair types 0 and 7 are never looked up), the dangle tables stay in device memory (sequence ends only), the
// size-dependent terms are stored twice per word (packed operand), and the fML triangle starts 4 bytes in so
// that the pair read of a lane whose first cell is column 0 stays inside the allocation.
struct SfPkLayout {
  int off_fml, off_ci, off_c1n, off_cb, off_dml, off_tab, off_uni, off_flag, off_S, total;
};
#define SF_PK_TABSTRIDE 175
static inline __host__ __device__ SfPkLayout sf_pk_layout(int W) {
  SfPkLayout L;
  int tri = FBASE(W);  // padded triangle
  if (tri < 0) tri = 0;
  tri = (tri + 1) & ~1;
  int o = 4;
  L.off_fml = o; o += tri * 2;
  const int RW = (W - 4 + 1) & ~1;  // even row stride
  const int roll = ((SF_FAST_NR * RW + 1) & ~1) * 2;
  L.off_ci = o; o += roll;
  L.off_c1n = o; o += roll;
  L.off_cb = o; o += roll;
  L.off_dml = o; o += ((4 * RW + 1) & ~1) * 2;
  L.off_tab = o; o += (4 * SF_PK_TABSTRIDE + 200) * 2 + 128 + 64;
  L.off_uni = o; o += 4 * 32 * 4;
  L.off_flag = o; o += 4;
  L.off_S = o; o += (W + 2 + 3) & ~3;
  L.total = o;
  return L;
}
static inline bool sf_pk_w_supported(int W) { return W >= 16 && W <= SF_PK_MAXW; }

struct SfPkUni {
  const uint32_t *NIN, *IL, *L1N, *BUL;  // each value in both halves
};

// One anti-diagonal d for one lane: cells (iA, iA+d) and (iA+1, iA+1+d); iA >= 0, at least one of them valid.
// HP[x] (x = size - 4): packed per-size minima of the generic candidates of the two enclosed cells on entry,
// of these two cells on exit.  G: d < 36, every size is tested against the (wave-uniform) limit d - 6.
template <bool G, int WT, int P>
__device__ __forceinline__ void sf_pk_cell(const SfFastCtx &X, const SfPkUni &U, const int d, const int iA,
                                           const bool vA, const bool vB, const int slot2, const int slotd,
                                           uint32_t (&HP)[27], int &ovf, const bool final_fml, uint32_t &fpart) {
  const int W = WT ? WT : X.W, RW = (W - 4 + 1) & ~1;
  const int i0 = iA - 1;  // column index of the first cell; its parity is P
// pair at table row offset `roff` (even), columns i0 + c and i0 + c + 1
#define LDP(tab, roff, c) sf_ld2((tab) + (roff) + i0 + (c), (P + (c)) & 1)
  const uint8_t *S = X.S;
  const int umax = G ? sfd_min(SFD_MAXLOOP, d - 2 - (SFD_TURN + 1)) : SFD_MAXLOOP;
#define ROW(u) ((slot2 - (u) < 0 ? slot2 - (u) + SF_FAST_NR : slot2 - (u)) * RW)
  const uint32_t INF2 = sf_pk(SF_INF16, SF_INF16), BIG2 = sf_pk(SF_PK_BIG, SF_PK_BIG);

  uint32_t gb = BIG2, g1 = BIG2, gg = BIG2, dec = BIG2;
  if (G) {
    // ---- short diagonals: sizes are tested against the limit; plain loops ----
    // pass 1: per-size minima of the generic interior candidates (both cells at once)
#pragma unroll
    for (int u = 30; u >= 6; --u) {
      if (u <= umax) {
        const uint32_t e = sf_pkadd(sf_pkmin(LDP(X.CI, ROW(u), 3), LDP(X.CI, ROW(u), u - 1)), U.NIN[u - 4]);  // u1 = 2 and u2 = 2
        HP[u - 4] = sf_pkmin(e, HP[u - 6]);
      }
    }
    if (umax >= 5) HP[1] = sf_pkadd(sf_pkmin(LDP(X.CI, ROW(5), 3), LDP(X.CI, ROW(5), 4)), U.NIN[1]);
    if (umax >= 4) HP[0] = sf_pkadd(LDP(X.CI, ROW(4), 3), U.NIN[0]);
    // pass 2a: bulges (size u >= 2), 1 x n loops (total size u >= 4), generic minima plus initiation
#pragma unroll
    for (int u = 2; u <= 30; ++u) {
      if (u <= umax) {
        gb = sf_pkmin(gb, sf_pkadd(sf_pkmin(LDP(X.CB, ROW(u), 1), LDP(X.CB, ROW(u), 1 + u)), U.BUL[u]));
        if (u >= 4) g1 = sf_pkmin(g1, sf_pkadd(sf_pkmin(LDP(X.C1N, ROW(u), 2), LDP(X.C1N, ROW(u), u)), U.L1N[u - 1]));
        if (u >= 6) gg = sf_pkmin(gg, sf_pkadd(HP[u - 4], U.IL[u]));
      }
    }
  } else {
    // ---- all sizes exist: one software pipeline.  Every stage issues the LDS reads of the NEXT batch, then
    // consumes the batch read one stage earlier; SF_PIN ends a stage (the compiler keeps that order), so a
    // batch's latency is covered by the previous batch's arithmetic even with only two waves per SIMD. ----
    uint32_t pa[2][5], pb[2][5], pn[2][5];  // pass 1: five sizes per batch, descending from 30
    auto ld1 = [&](const int bt, uint32_t(&a)[5], uint32_t(&b)[5], uint32_t(&n)[5]) {
#pragma unroll
      for (int k = 0; k < 5; k++) {
        const int u = 30 - 5 * bt - k;
        a[k] = LDP(X.CI, ROW(u), 3);      // u1 = 2
        b[k] = LDP(X.CI, ROW(u), u - 1);  // u2 = 2
        n[k] = U.NIN[u - 4];
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
    // pass 2a: three sizes per batch, ascending from 2
    uint32_t qb1[2][3], qb2[2][3], qbt[2][3], qn1[2][3], qn2[2][3], qnt[2][3], qit[2][3];
    auto ld2 = [&](const int bt, const int f) {
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const int u = 2 + 3 * bt + k;
        if (u <= 30) {
          qb1[f][k] = LDP(X.CB, ROW(u), 1); qb2[f][k] = LDP(X.CB, ROW(u), 1 + u); qbt[f][k] = U.BUL[u];
          if (u >= 4) { qn1[f][k] = LDP(X.C1N, ROW(u), 2); qn2[f][k] = LDP(X.C1N, ROW(u), u); qnt[f][k] = U.L1N[u - 1]; }
          if (u >= 6) qit[f][k] = U.IL[u];
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
    // bridge: sizes 5 and 4 (no recurrence), first batch of pass 2a
    const uint32_t x53 = LDP(X.CI, ROW(5), 3), x54 = LDP(X.CI, ROW(5), 4), x4 = LDP(X.CI, ROW(4), 3);
    const uint32_t n1 = U.NIN[1], n0 = U.NIN[0];
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

  // ---- multiloop split min_m fML[i, i+m] + fML[i+m+1, j] (both cells at once) ----
  {
    // fML[i, i+m] = fML_tri[FBASE(m) + i0], fML[i+m+1, j] = fML_tri[FBASE(d-m-1) + i0+m+1]; both offsets are
    // wave-uniform and advance by the (even-padded) diagonal lengths.  Eight split points per batch, two
    // batches in flight (ping-pong registers).
    const int16_t *fa = X.fML + i0;
    const int16_t *fb2 = X.fML + i0 + 1;
    int ia = 0;                                       // FBASE(4)
    int ib = FBASE(d - SFD_TURN - 2) + SFD_TURN + 1;  // FBASE(d-m-1) + m at m = 4
    int m = SFD_TURN + 1;
    const int mend = d - SFD_TURN - 2;
    uint32_t dec2 = BIG2;
    uint32_t a0[8], b0[8], a1[8], b1[8];
    auto ldm = [&](uint32_t(&a)[8], uint32_t(&b)[8]) {
#pragma unroll
      for (int k = 0; k < 8; k++) {
        a[k] = sf_ld2(fa + ia, P);                  // FBASE is even: the parity of the first cell's column
        b[k] = sf_ld2(fb2 + ib, (P + 1 + k) & 1);   // column i0 + m + 1, m = 4 + 8t + k
        ia += FLEN(m + k);
        ib -= FLEN(d - (m + k) - 2) - 1;
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
      dec = sf_pkmin(dec, sf_pkadd(sf_ld2(fa + ia, P), sf_ld2r(fb2 + ib, (P + 1 + m) & 1)));
      ia += FLEN(m);
      ib -= FLEN(d - m - 2) - 1;
    }
    dec = sf_pkmin(dec, dec2);
  }
  // multiloop closed by the cell: DML of the enclosed cell, diagonal d-2
  const uint32_t dmlc = LDP(X.DMLr, ((d - 2) & 3) * RW, 1);
  // fML neighbours on diagonal d-1 (final only for the even-diagonal group, see the kernel)
  uint32_t fn = BIG2;
  if (final_fml && d > SFD_TURN + 1) {
    fn = sf_pkadd(sf_pkmin(LDP(X.fML, FBASE(d - 1), 1), LDP(X.fML, FBASE(d - 1), 0)), sf_pk(X.MLbase, X.MLbase));
  }

  // ---- pass 2b, once per half: the terms that depend on the cell's own nucleotides, then publish ----
  int cc[2], cI[2], c1n[2], cb[2], ff[2], dd[2];
#pragma unroll
  for (int h = 0; h < 2; h++) {
    const int i = iA + h, j = i + d;
    const bool valid = h ? vB : vA;
    int c = SF_INF16, f = SF_FAST_BIG;
    int pI = SF_INF16, p1n = SF_INF16, pb = SF_INF16;
    if (valid) {
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
          const int ih = i - 1;
          {  // stack
            const int t2r = sfd_rtype(X.tPair[si1 * 8 + sj1]);
            e = sfd_min(e, X.CB[ROW(0) + ih + 1] - (t2r > 2 ? TAU : 0) + st[t2r]);
          }
          if (!G || umax >= 1) {  // one-nucleotide bulges keep the stack
            const int b1 = sf_lo(U.BUL[1]);
            const int16_t *row = X.CB + ROW(1) + ih;
            const int ta = sfd_rtype(X.tPair[si1 * 8 + S[j - 2]]);  // (i+1, j-2)
            e = sfd_min(e, row[1] - (ta > 2 ? TAU : 0) + b1 + st[ta]);
            const int tb = sfd_rtype(X.tPair[S[i + 2] * 8 + sj1]);  // (i+2, j-1)
            e = sfd_min(e, row[2] - (tb > 2 ? TAU : 0) + b1 + st[tb]);
          }
          if (!G || umax >= 2) {  // 1 x 1: (i+2, j-2)
            const int t2r = sfd_rtype(X.tPair[S[i + 2] * 8 + S[j - 2]]);
            e = sfd_min(e, X.CB[ROW(2) + ih + 2] - (t2r > 2 ? TAU : 0) + PB.int11[type][t2r][si1][sj1]);
          }
          if (!G || umax >= 3) {  // 1 x 2 and 2 x 1
            const int16_t *row = X.CB + ROW(3) + ih;
            const int ta = sfd_rtype(X.tPair[S[i + 2] * 8 + S[j - 3]]);  // (i+2, j-3), sq1 = S[j-2]
            e = sfd_min(e, row[2] - (ta > 2 ? TAU : 0) + PB.int21[type][ta][si1][S[j - 2]][sj1]);
            const int tb = sfd_rtype(X.tPair[S[i + 3] * 8 + S[j - 2]]);  // (i+3, j-2), sp1 = S[i+2]
            e = sfd_min(e, row[3] - (tb > 2 ? TAU : 0) + PB.int21[tb][type][sj1][si1][S[i + 2]]);
          }
          if (!G || umax >= 4) {  // 2 x 2: (i+3, j-3)
            const int t2r = sfd_rtype(X.tPair[S[i + 3] * 8 + S[j - 3]]);
            e = sfd_min(e, X.CB[ROW(4) + ih + 3] - (t2r > 2 ? TAU : 0) + PB.int22[type][t2r][si1][S[i + 2]][S[j - 2]][sj1]);
          }
          if (!G || umax >= 5) {  // 2 x 3 and 3 x 2
            const int16_t *row = X.CB + ROW(5) + ih;
            const int m23 = X.t23[SF_TIDX(type, si1, sj1)] + X.F->L23;
            const int ta = sfd_rtype(X.tPair[S[i + 3] * 8 + S[j - 4]]);  // (i+3, j-4); sp1 = S[i+2], sq1 = S[j-3]
            e = sfd_min(e, row[3] - (ta > 2 ? TAU : 0) + m23 + X.t23[SF_TIDX(ta, S[j - 3], S[i + 2])]);
            const int tb = sfd_rtype(X.tPair[S[i + 4] * 8 + S[j - 3]]);  // (i+4, j-3); sp1 = S[i+3], sq1 = S[j-2]
            e = sfd_min(e, row[4] - (tb > 2 ? TAU : 0) + m23 + X.t23[SF_TIDX(tb, S[j - 2], S[i + 3])]);
          }
          e = sfd_min(e, (h ? sf_hi(gb) : sf_lo(gb)) + tau_out);
          e = sfd_min(e, (h ? sf_hi(g1) : sf_lo(g1)) + X.t1n[SF_TIDX(type, si1, sj1)]);
          e = sfd_min(e, (h ? sf_hi(gg) : sf_lo(gg)) + X.tI[SF_TIDX(type, si1, sj1)]);
        }
        const int tr = sfd_rtype(type);
        e = sfd_min(e, (h ? sf_hi(dmlc) : sf_lo(dmlc)) + X.tM[SF_TIDX(tr, sj1, si1)] + (tr > 2 ? TAU : 0) + X.MLintern + X.MLclosing);
        c = e;
        if (c < SF_FAST_OVF) ovf = 1;
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
    }
    const int dech = h ? sf_hi(dec) : sf_lo(dec);
    f = sfd_min(f, sfd_min(dech, h ? sf_hi(fn) : sf_lo(fn)));
    cc[h] = c; cI[h] = pI; c1n[h] = p1n; cb[h] = pb; ff[h] = f;
    dd[h] = dech > SF_FAST_THRESH ? SF_INF16 : dech;
  }

  // ---- publish both cells ----
  const int rbd = slotd * RW + i0;
  if (vA && vB) {
    sf_st2(X.CI + rbd, P, sf_pk(cI[0], cI[1]));
    sf_st2(X.C1N + rbd, P, sf_pk(c1n[0], c1n[1]));
    sf_st2(X.CB + rbd, P, sf_pk(cb[0], cb[1]));
    sf_st2(X.DMLr + (d & 3) * RW + i0, P, sf_pk(dd[0], dd[1]));
  } else {
    const int h = vA ? 0 : 1;
    X.CI[rbd + h] = (int16_t)cI[h];
    X.C1N[rbd + h] = (int16_t)c1n[h];
    X.CB[rbd + h] = (int16_t)cb[h];
    X.DMLr[(d & 3) * RW + i0 + h] = (int16_t)dd[h];
  }
  // c by (row j, column i): the two cells are in different rows of the triangular scratch
  if (vA) X.cg[SF_CGIDX(iA, iA + d)] = (int16_t)cc[0];
  if (vB) X.cg[SF_CGIDX(iA + 1, iA + 1 + d)] = (int16_t)cc[1];
  fpart = sf_pk(sfd_min(ff[0], SF_PK_BIG), sfd_min(ff[1], SF_PK_BIG));
  if (final_fml) {
#pragma unroll
    for (int h = 0; h < 2; h++) {
      if (h ? vB : vA) {
        if (ff[h] < SF_FAST_OVF) ovf = 1;
        X.fML[FBASE(d) + i0 + h] = (int16_t)(ff[h] > SF_FAST_THRESH ? SF_INF16 : ff[h]);
      }
    }
  }
  (void)INF2;
#undef ROW
#undef LDP
}

template <int WT>
__global__ __launch_bounds__(128, 2) void sf_mfe_pk_kernel(const uint8_t *__restrict__ seqs, int n, int Wrt,
                                                        const SfDevParams *__restrict__ D,
                                                        const SfFastParams *__restrict__ F,
                                                        int16_t *__restrict__ cg_all, int32_t *__restrict__ out,
                                                        int *__restrict__ ovf_cnt, int *__restrict__ ovf_list,
                                                        int trace_stride, char *__restrict__ db_out,
                                                        int *__restrict__ status) {
  constexpr int NT = 128;
  const int W = WT ? WT : Wrt;
  SF_DYN_SMEM(smem);
  const SfPkLayout Lo = sf_pk_layout(W);
  const int RW = (W - 4 + 1) & ~1;
  SfFastCtx X;
  X.fML = (int16_t *)(smem + Lo.off_fml);
  X.CI = (int16_t *)(smem + Lo.off_ci);
  X.C1N = (int16_t *)(smem + Lo.off_c1n);
  X.CB = (int16_t *)(smem + Lo.off_cb);
  X.DMLr = (int16_t *)(smem + Lo.off_dml);
  int16_t *tab = (int16_t *)(smem + Lo.off_tab);
  X.tI = tab; X.t1n = tab + SF_PK_TABSTRIDE; X.t23 = tab + 2 * SF_PK_TABSTRIDE; X.tM = tab + 3 * SF_PK_TABSTRIDE;
  X.tH = tab + 4 * SF_PK_TABSTRIDE;
  int16_t *tStack = tab + 4 * SF_PK_TABSTRIDE + 200;
  X.tStack = tStack;
  X.tD5 = F->d5; X.tD3 = F->d3;  // sequence ends only: device memory
  uint8_t *tPair = (uint8_t *)(tStack + 64);
  X.tPair = tPair;
  uint32_t *uni = (uint32_t *)(smem + Lo.off_uni);
  SfPkUni U;
  U.NIN = uni; U.IL = uni + 32; U.L1N = uni + 64; U.BUL = uni + 96;
  X.uNIN = nullptr; X.uIL = nullptr; X.uL1N = nullptr; X.uBUL = nullptr;
  int32_t *flag = (int32_t *)(smem + Lo.off_flag);
  uint8_t *S = (uint8_t *)(smem + Lo.off_S);
  X.S = S;
  X.D = D; X.F = F; X.W = W; X.fml_pad = 1; X.fst = 1; X.maxd = D->max_pair_dist; X.cg_ext = 0; X.tE = nullptr;
  X.TAU = D->P.TerminalAU; X.MLbase = D->P.MLbase; X.MLclosing = D->P.MLclosing; X.MLintern = D->P.MLintern[1];
  // exterior pass aliases (the rolling CI area is dead by then)
  int32_t *f5s = (int32_t *)(smem + Lo.off_ci);
  int16_t *tExt = (int16_t *)(smem + Lo.off_ci + (((W + 1) * 4 + 3) & ~3));

  const int tid = threadIdx.x;
  X.cg = cg_all + (size_t)blockIdx.x * SF_CG_ENTRIES(W);
  // parameter tables -> LDS, once per workgroup.  Entries of pair type 0 / 7 are never read, so table k+1 may
  // start where the type-7 block of table k would be (written in ascending order: the later table wins there).
  if (tid == 0) { smem[0] = 0; smem[1] = 0; smem[2] = 0; smem[3] = 0; }
  for (int x = tid; x < 200; x += NT) {
    if (x < SF_PK_TABSTRIDE) {
      tab[x] = F->mmI[x]; tab[SF_PK_TABSTRIDE + x] = F->mm1n[x]; tab[2 * SF_PK_TABSTRIDE + x] = F->mm23[x];
      tab[3 * SF_PK_TABSTRIDE + x] = F->mmM[x];
    }
    tab[4 * SF_PK_TABSTRIDE + x] = F->mmH[x];
  }
  for (int x = tid; x < 64; x += NT) { tStack[x] = F->stack[x]; tPair[x] = F->pair[x]; }
  for (int x = tid; x < 32; x += NT) {
    const int a = sfd_min(F->NIN[x], 32000), b = sfd_min(F->IL[x], 32000), c = sfd_min(F->L1N[x], 32000),
              e = sfd_min(F->BUL[x], 32000);
    uni[x] = sf_pk(a, a); uni[32 + x] = sf_pk(b, b); uni[64 + x] = sf_pk(c, c); uni[96 + x] = sf_pk(e, e);
  }
  // wave g handles the diagonal d0+g of a step; lane t owns the half-centres 2t+2 and 2t+3: cells
  // iA = 2t + 2 - d/2 and iA + 1
  const int grp = SF_WAVE_UNIFORM(tid >> 6);
  const int lane = tid & 63;

  for (int seq = blockIdx.x; seq < n; seq += gridDim.x) {
    const uint8_t *src = seqs + (size_t)seq * W;
    __syncthreads();
    for (int x = tid; x < W; x += NT) S[x + 1] = sf_encode_nt(src[x]);
    if (tid == 0) { S[0] = 0; S[W + 1] = 0; flag[0] = 0; }
    for (int x = tid; x < 4 * RW; x += NT) X.DMLr[x] = SF_INF16;  // diagonals 2,3 have no multiloop split
    __syncthreads();
    int ovf = 0;
    uint32_t H[27];
#pragma unroll
    for (int k = 0; k < 27; k++) H[k] = sf_pk(SF_INF16, SF_INF16);

    int slot2 = (SFD_TURN + 1 + grp - 2) % SF_FAST_NR, slotd = (SFD_TURN + 1 + grp) % SF_FAST_NR;
    // one step = the even diagonal d0 (wave 0) and the odd diagonal d0+1 (wave 1)
    // PT: parity of the column of every lane's first cell in this step, (d0/2 + 1) & 1
    auto step = [&](const int d0, auto GT, auto PT) {
      constexpr bool G = decltype(GT)::value;
      constexpr int P = decltype(PT)::value;
      const int d = d0 + grp;
      const int iA = 2 * lane + 2 - (d >> 1);
      const bool vA = (d < W) && (iA >= 1) && (iA + d <= W);
      const bool vB = (d < W) && (iA + 1 >= 1) && (iA + 1 + d <= W);
      uint32_t fpart = sf_pk(SF_PK_BIG, SF_PK_BIG);
      if (vA || vB) sf_pk_cell<G, WT, P>(X, U, d, iA, vA, vB, slot2, slotd, H, ovf, grp == 0, fpart);
      __syncthreads();
      if (grp == 1 && (vA || vB)) {  // fML on the odd diagonal: add the neighbours on diagonal d-1, now final
        const int i0 = iA - 1;
        const int16_t *fr = X.fML + FBASE(d - 1) + i0;
        const uint32_t fn = sf_pkadd(sf_pkmin(sf_ld2(fr + 1, (P + 1) & 1), sf_ld2(fr, P)), sf_pk(X.MLbase, X.MLbase));
        const uint32_t f2 = sf_pkmin(fpart, fn);
#pragma unroll
        for (int h = 0; h < 2; h++) {
          if (h ? vB : vA) {
            const int f = h ? sf_hi(f2) : sf_lo(f2);
            if (f < SF_FAST_OVF) ovf = 1;
            X.fML[FBASE(d) + i0 + h] = (int16_t)(f > SF_FAST_THRESH ? SF_INF16 : f);
          }
        }
      }
      __syncthreads();
      slot2 += 2; if (slot2 >= SF_FAST_NR) slot2 -= SF_FAST_NR;
      slotd += 2; if (slotd >= SF_FAST_NR) slotd -= SF_FAST_NR;
    };
    // d0 = 4, 6, 8, ...: the parity alternates 1, 0, 1, ...; two steps per trip
    typedef std::integral_constant<int, 0> P0;
    typedef std::integral_constant<int, 1> P1;
    int d0 = SFD_TURN + 1;
    for (; d0 < W && d0 < SFD_MAXLOOP + 6; d0 += 4) {  // some loop sizes do not fit yet
      step(d0, std::true_type{}, P1{});
      if (d0 + 2 < W) step(d0 + 2, std::true_type{}, P0{});
    }
    for (; d0 < W; d0 += 4) {
      step(d0, std::false_type{}, P1{});
      if (d0 + 2 < W) step(d0 + 2, std::false_type{}, P0{});
    }

    if (ovf) flag[0] = 1;
    for (int x = tid; x < 200; x += NT) tExt[x] = F->mmExt[x];
    __syncthreads();
    // The rolling tables are dead now.  When a table of c + ExtLoop fits into the CI + C1N areas behind f5[] and the
    // mismatchExt table, the whole workgroup builds it in LDS (sf_fast_ext_table), so the single-wave exterior
    // sweep neither waits for device memory nor looks anything up.
    int16_t *etab = nullptr;
    {
      const int e_off = (int)(((char *)tExt - smem) + 400 + 3) & ~3;
      if (SF_EXT_TABLE && e_off + SF_CG_ENTRIES(W) * 2 <= Lo.off_cb) {
        etab = (int16_t *)(smem + e_off);
        sf_fast_ext_table(X, W, tid, NT, tExt, etab);
      }
    }
    __syncthreads();
    if (tid < 64)
      sf_fast_exterior<2>(X, W, tid, seq, f5s, tExt, etab, flag, (int16_t *)(smem + Lo.off_cb),
                      (char *)(smem + Lo.off_cb + ((3 * (W + 8) * 2 + 3) & ~3)), out, ovf_cnt, ovf_list, trace_stride,
                      db_out, status);
  }
}

static inline hipError_t sf_pk_configure() {
  hipError_t e = hipFuncSetAttribute((const void *)sf_mfe_pk_kernel<120>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute((const void *)sf_mfe_pk_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

// grid / LDS / scratch for n folds of W nt on a chip with n_cu CUs
static inline void sf_pk_geometry(int W, int n_cu, int n, int *grid, size_t *lds, size_t *scratch) {
  const SfPkLayout L = sf_pk_layout(W);
  int per_cu = (160 * 1024) / L.total;
  if (per_cu > 8) per_cu = 8;
  if (per_cu < 1) per_cu = 1;
  int g = n_cu * per_cu;
  if (g > n) g = n;
  *grid = g;
  *lds = (size_t)L.total;
  *scratch = (size_t)g * SF_CG_ENTRIES(W) * sizeof(int16_t);
}

#undef FBASE
#undef FLEN

template <typename... A>
static inline void sf_pk_launch(int grid, int W, size_t lds, hipStream_t st, A... args) {
  if (W == 120) SF_LAUNCH((sf_mfe_pk_kernel<120>), grid, 128, lds, st, args...);
  else SF_LAUNCH((sf_mfe_pk_kernel<0>), grid, 128, lds, st, args...);
}
