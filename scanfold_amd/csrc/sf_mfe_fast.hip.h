// sf_mfe_fast.hip.h — the hot kernel: batched Zuker MFE fill (energy only) with LDS-resident int16 tables.
//
// Replaces energies(seq_list) -> rna_folder -> RNA.fold(seq) (ScanFold-Scan.py:244-246,253-262;
// ScanFoldFunctions.py:774-789,805-814): r+1 folds per window whose structures the caller throws away.
//
// One workgroup folds one sequence at a time (persistent grid; a workgroup's first fold is its block index, every further one comes
// from a device-wide counter); two anti-diagonals per step (two thread groups), one barrier per step (two on the long diagonals).
// The steps of a fold run as one loop PER KIND of step — size-tested / guarded / unsplit / split (and finer: SF_LOOPS_BY_KIND, the
// loops at the end of the kernel) — because each loop then gets its own register allocation and schedule: +5-8 % at every width.
// Thread mapping: a thread owns a CENTRE s = i+j (two centres, one per parity of d):
// on diagonal d it handles the cell i = v - d/2, j = i+d with v = (tid+OFF) mod NT.  The cell of the same
// thread two diagonals later is (i-1, j+1), the cell that encloses it — which makes the interior-loop search
// incremental (below) with all of its state in registers.
//
// LDS per workgroup (W=120: 40.9 kB -> 4 workgroups per CU):
//   fML   int16, every diagonal (the O(W^3) multiloop split reads them all).  W <= 128: a triangle, diagonal after
//         diagonal, each starting at an even index (two neighbouring cells = one aligned word: sf_fast_dml2).  W > 128 (FOLD): folded into a rectangle — diagonal x <= H = (W+3)/2 is the left part of row x-4,
//         diagonal W+3-x the right part of the same row (their lengths add up to the row length W-3) — so that a
//         multiloop split walks both operands with constant strides (sf_fast_split_stretch): +6 % at W = 200; at
//         W = 120 the same change measured -5 % (profiles/r02), so the narrow kernel keeps the triangle
//   CI, C1N, CB  int16, rolling window of SF_FAST_NR diagonals of c, pre-added with the inner pair's terms:
//           CI  = c + mismatchI [rtype][S[j+1]][S[i-1]]   generic loops
//           C1N = c + mismatch1nI[rtype][S[j+1]][S[i-1]]  1 x n loops
//           CB  = c + TerminalAU(rtype)                   bulges (and, minus that term, the few special loops)
//           (this kernel interleaves CB and C1N as one 32-bit word per cell, SfFastCtx::BN)
//   small int16 parameter tables (mismatches — rows of the six pair types only —, stack, dangles, pair types)
//   (min_k fML[i,k]+fML[k+1,j] of the enclosed cell, needed by the multiloop closing term, is the thread's own
//   previous result: a register)
//
// Interior loops.  Of the 496 (u1,u2) candidates of a cell, 375 are "generic" (both sides >= 2, not 2x2/2x3):
//   E = CI[p,q] + internal_loop[u1+u2] + min(max_ninio, ninio*|u1-u2|) + mismatchI[type][S[i+1]][S[j-1]].
// For fixed total size u, the candidates of (i,j) with u1,u2 >= 3 are exactly the candidates of (i+1,j-1)
// with total u-2 and the same asymmetry (Lyngso et al. 1999), so with
//   H[i,j,u] = min_{u1+u2=u, u1,u2>=2} CI[i+1+u1, j-1-u2] + ninio(|u1-u2|)
// H[i,j,u] = min(H[i+1,j-1,u-2], the two edge candidates u1=2 and u2=2): 2 LDS reads per u instead of u-3,
// exact (same minimum over the same set).  H lives in registers because (i+1,j-1) is the same thread's
// previous cell.  Bulges (2 per size), 1 x n loops (2 per size) and the 9 special candidates are direct.
// Size-dependent terms are wave-uniform reads of small tables in device memory (scalar loads; guarded copies for the short
// diagonals, SfFastParams::uniG).
//
// c + ExtLoop is streamed to a device scratch table (int16, L2-resident) for the exterior-loop sweep, which trails the fill
// inside the same fold (SfTrail: the 3' -> 5' recurrence on the rows that are already complete; folds whose structure is
// wanted keep the 5' -> 3' sweep at their end, whose f5[] the traceback reads).
//
// int16 is exact while |energy| < 12000 dcal/mol; a fold that leaves that range (a >120 kcal/mol helix) is
// appended to an overflow list and redone by the int32 kernel (sf_mfe_full.hip.h), so results never depend
// on the storage width.  INF is SF_INF16; any sum above SF_FAST_THRESH means "no structure" — parameter
// sets whose entries exceed SF_FAST_MAXPARAM in magnitude are routed to the int32 kernel entirely.
#pragma once
#include <type_traits>
#include "sf_energy.h"
#include "sf_pk16.h"

#define SF_FAST_NR 34
// even diagonals below this one run the size-tested cell code; from here on the straight-line code with guarded size tables.
// (12 until round 5 — "some special loops do not exist yet" below d = 12 — but what the straight-line code reads for them are ring rows
// no diagonal of the fold has written, which every fold initialises to "none"; only the two unpaired sizes 4 / 5 needed a clamp.
// 8 / 10 / 12: 54.6 / 53.9 / 54.2 ms per 262 144 120-mers, W = 200 55.2 / 52.7 / 52.9 per 65 536)
#ifndef SF_FAST_TINY_D0
#define SF_FAST_TINY_D0 10
#endif
// even diagonals below this one skip whole batches of loop sizes above the limit (CH); from here to 36 the skipped
// work is small and the unbroken straight-line code is faster (measured: 20 / 24 / 28 / 36 -> 87.5 / 87.1 / 87.0 / 87.9 ms)
#define SF_FAST_CHUNK_D0 36
#ifndef SF_FAST_ROWTAB
#define SF_FAST_ROWTAB 1  // rolling-row offsets from a table (scalar loads, SfFastRows) instead of per-row ring arithmetic
#endif
#ifndef SF_FAST_UNPACK
#define SF_FAST_UNPACK 1  // narrow kernels, from d0 = 36 on: the generic-loop recurrence on full-rate 16-bit instructions (SfHU)
#endif
#ifndef SF_LOOPS_BY_KIND
#define SF_LOOPS_BY_KIND 1  // one loop per kind of step instead of one loop with the kinds as branches (0: the round-4 structure)
#endif
#ifndef SF_UNP_PB
#define SF_UNP_PB 4  // size pairs per batch of reads in the unpacked recurrence (3 / 4 / 6 / 12: 58.0 / 57.5 / 58.6 / 58.3 ms per 262 144 folds)
#endif
#define SF_FAST_SPLIT 1  // long diagonals: the idle second wave of a group takes part of the cell's work
#define SF_FAST_DML2 1   // ... and the first runs the multiloop split two cells per lane, the lanes in chunks over the terms
// split steps at W <= 128: 1 = ONE helper wave serves both diagonals of a step, on a compacted list of the cells that
// can pair (3 of 8): the special / bulge / 1xn block runs once per step instead of twice
// (W < SF_HELP_MERGE_MAXW, and the W = 120 instantiation: SF_MG120.  With folds handed out dynamically the merge measures +3.6 % at W=80, +2.3 % at W=100,
// +1.1 % at W=110, +1.5 % at W=116 and -2.9 % at W=120, -1.4 % at W=128: the compacted lanes read scattered words, three or four deep
// in the LDS banks, and at the widths where LDS cycles are tightest that costs more than the saved pass.)
#define SF_HELP_MERGE_MAXW 118
#define SF_HELP_MERGE 1
// the W = 120 instantiation runs the merged helper too (round 4: +1.1 %, 59.5 -> 58.8 ms per 262 144 folds) — its cell lists take the
// place of the LDS copy of the size tables (see cell_list in the kernel); the generic instantiation keeps the W < 118 rule
#ifndef SF_MG120
#define SF_MG120 1
#endif
// split steps at W > 128 (four waves per diagonal): 1 = the group's outer waves help the two middle ones
#define SF_FAST_SPLIT_256 1
#ifndef SF_FAST_NARROW_256
#define SF_FAST_NARROW_256 1  // ... and from the diagonal where a diagonal's cells fit one wave on, one main + one helper wave (NARROW in the kernel)
#endif
// split steps of the wide kernel (NG = 256, where the split is most of a cell): the helper waves take (terms / 2 - bias)
// terms of the multiloop split, the ones with the largest m.  The narrow kernel's helpers take none: their special-loop
// work already balances the main wave (any share measured the same or worse in rounds 2 and 3).
// (the W = 200 instantiation: 18 / 26 / 34 / 42 / 50 / 60 / 80 -> 142.4 / 140.4 / 138.1 / 136.5 / 135.8 / 136.2 / 137.2 ms per 131 072 folds at
// the end of round 3; the generic wide instantiation at W = 160: 26 / 42 / 50 / 60 -> 110.6 / 112.0 / 113.1 / 114.2 ms)
#define SF_DML_HELPER_BIAS_256 26
#define SF_DML_HELPER_BIAS_W200 50
#define SF_DML_HELPER_BIAS(ng, wt) ((ng) == 256 ? ((wt) == 200 ? SF_DML_HELPER_BIAS_W200 : SF_DML_HELPER_BIAS_256) : 100000)
// loop sizes per batch of reads in the bulge / 1xn minima (narrow / wide kernel; measured 2, 3, 4, 6: W=120 best at
// 3 by 0.6 %, W=200 at 6 by 3.5 %)
#define SF_HELP_NB_128 4
#define SF_HELP_NB_128G 8  // the generic merged-helper instantiation (W < 118): spill-free since the round-4 pins, it takes the longer batches — +2.2 % at W = 64 / 77, +1.6 % at W = 31, +0.7-1.3 % at W = 100; W = 120 and the generic W = 118 .. 128 kernel stay at 4 (59.0 against 59.8 ms; W = 128 -1.2 % with 8)
#define SF_HELP_NB_256 6
#define SF_INF16 30000
#define SF_FAST_THRESH 10000
#define SF_FAST_OVF (-12000)
#define SF_FAST_MAXPARAM 2500
#define SF_FAST_MAXW 256
#define SF_FAST_BIG 60000
#ifndef SF_FAST_WAVES_PER_SIMD
#define SF_FAST_WAVES_PER_SIMD 4  // 4 workgroups of 4 waves per CU (W <= 128): at most 128 VGPRs
#endif
// (what five or six workgroups per CU would buy if a layout allowed them: profiles/r04/mfe_occupancy_bound.txt)

struct SfFastParams {
  int32_t NIN[32];   // [a]  min(max_ninio, a * ninio)
  int32_t IL[32];    // [u]  internal_loop[u]
  int32_t L1N[32];   // [n]  1 x n: internal_loop[n+1] + min(max_ninio, (n-1) * ninio)
  int32_t BUL[32];   // [n]  bulge[n]
  int32_t L23;       // internal_loop[5] + ninio
  int32_t fast_ok;   // parameter magnitudes allow int16 storage
  // int16 images of the small tables, index t*25 + a*5 + b (t = pair type 0..7)
  int16_t mmI[200], mm1n[200], mm23[200], mmM[200], mmH[200], mmExt[200];
  int16_t stack[64];
  int16_t d5[40], d3[40];
  uint8_t pair[64];  // [a*8+b]
  // Special interior loops (stack, 1-nt bulge, 1x1, 1x2, 2x1, 2x2, 2x3, 3x2) read the enclosed cell from the bulge
  // view CB = c + TerminalAU(reversed type); these images carry "- TerminalAU(inner type)" already, and the inner
  // type comes reversed from rpair — no per-candidate arithmetic on types is left in the kernel.  int16, indexed
  // like the blob's tables; int21b is corrected by its FIRST type index (the 2x1 orientation), the others by the second.
  uint8_t rpair[64];   // rtype(pair[a][b])
  int16_t stackT[64], mm23in[200];
  int16_t int11T[8 * 8 * 25], int21a[8 * 8 * 125], int21b[8 * 8 * 125], int22T[8 * 8 * 625];
  // the size tables in the layout of their LDS copy (NIN[32] | IL[32] | (BUL[u], L1N[u-1]) pairs): read with wave-uniform
  // addresses from HERE — scalar loads, off the LDS pipe — by the code that needs no guarded copy (diagonals >= 36)
  int16_t uni[128];
  // ... and its guarded copies for the short diagonals: uniG[um] = uni with the entries of loop sizes above um at 32767
  // (what the kernels used to build per wave and step in LDS); um = d - 6 = 0..30
  int16_t uniG[31][128];
};

static inline void sf_fast_build_params(const SfDevParams &D, SfFastParams &F) {
  const sf_params_blob &P = D.P;
  memset(&F, 0, sizeof F);
  for (int a = 0; a < 32; a++) F.NIN[a] = a * P.ninio < P.max_ninio ? a * P.ninio : P.max_ninio;
  for (int u = 0; u <= 30; u++) F.IL[u] = P.internal_loop[u];
  for (int n = 0; n <= 29; n++) F.L1N[n] = P.internal_loop[n + 1] + F.NIN[n > 0 ? n - 1 : 0];
  for (int n = 0; n <= 30; n++) F.BUL[n] = P.bulge[n];
  F.L23 = P.internal_loop[5] + P.ninio;
  // magnitude check over every finite entry the kernel can add up
  long long mx = 0;
  auto upd = [&mx](const int32_t *p, size_t cnt) {
    for (size_t k = 0; k < cnt; k++) {
      long long v = p[k] < 0 ? -(long long)p[k] : p[k];
      if (v < SF_INF && v > mx) mx = v;
    }
  };
  upd(&P.stack[0][0], 64); upd(P.hairpin, 31); upd(P.bulge, 31); upd(P.internal_loop, 31);
  upd(&P.mismatchI[0][0][0], 200); upd(&P.mismatchH[0][0][0], 200); upd(&P.mismatchM[0][0][0], 200);
  upd(&P.mismatch1nI[0][0][0], 200); upd(&P.mismatch23I[0][0][0], 200); upd(&P.mismatchExt[0][0][0], 200);
  upd(&P.dangle5[0][0], 40); upd(&P.dangle3[0][0], 40);
  upd(&P.int11[0][0][0][0], 1600); upd(&P.int21[0][0][0][0][0], 8000); upd(&P.int22[0][0][0][0][0][0], 40000);
  upd(&P.ninio, 1); upd(&P.max_ninio, 1); upd(&P.MLbase, 1); upd(&P.MLclosing, 1); upd(P.MLintern, 8);
  upd(&P.TerminalAU, 1); upd(P.tetra_E, SF_NSPECIAL); upd(P.tri_E, SF_NSPECIAL); upd(P.hexa_E, SF_NSPECIAL);
  for (int s = 0; s <= SF_FAST_MAXW; s++) upd(&D.hp_init[s], 1);
  F.fast_ok = (mx <= SF_FAST_MAXPARAM) ? 1 : 0;
  auto clamp16 = [](int v) { return (int16_t)(v > 32000 ? 32000 : (v < -32000 ? -32000 : v)); };
  for (int t = 0; t < 8; t++) {
    for (int a = 0; a < 5; a++) {
      for (int b = 0; b < 5; b++) {
        const int k = t * 25 + a * 5 + b;
        F.mmI[k] = clamp16(P.mismatchI[t][a][b]);
        F.mm1n[k] = clamp16(P.mismatch1nI[t][a][b]);
        F.mm23[k] = clamp16(P.mismatch23I[t][a][b]);
        F.mmM[k] = clamp16(P.mismatchM[t][a][b]);
        F.mmH[k] = clamp16(P.mismatchH[t][a][b]);
        F.mmExt[k] = clamp16(P.mismatchExt[t][a][b]);
      }
      F.d5[t * 5 + a] = clamp16(P.dangle5[t][a]);
      F.d3[t * 5 + a] = clamp16(P.dangle3[t][a]);
    }
    for (int u = 0; u < 8; u++) F.stack[t * 8 + u] = clamp16(P.stack[t][u]);
  }
  for (int a = 0; a < 8; a++)
    for (int b = 0; b < 8; b++) {
      const int t = D.pair[a][b];
      F.pair[a * 8 + b] = (uint8_t)t;
      F.rpair[a * 8 + b] = (uint8_t)((t && t < 7) ? (((t - 1) ^ 1) + 1) : t);
    }
  auto tau = [&P](int t) { return t > 2 ? P.TerminalAU : 0; };
  auto off16 = [&clamp16](int32_t v, int t) { return v >= SF_INF ? (int16_t)32000 : clamp16(v - t); };
  for (int t = 0; t < 8; t++)
    for (int u = 0; u < 8; u++) {
      F.stackT[t * 8 + u] = off16(P.stack[t][u], tau(u));
      for (int a = 0; a < 5; a++)
        for (int b = 0; b < 5; b++) {
          F.int11T[((t * 8 + u) * 5 + a) * 5 + b] = off16(P.int11[t][u][a][b], tau(u));
          for (int c = 0; c < 5; c++) {
            F.int21a[(((t * 8 + u) * 5 + a) * 5 + b) * 5 + c] = off16(P.int21[t][u][a][b][c], tau(u));
            F.int21b[(((t * 8 + u) * 5 + a) * 5 + b) * 5 + c] = off16(P.int21[t][u][a][b][c], tau(t));
            for (int e = 0; e < 5; e++)
              F.int22T[((((t * 8 + u) * 5 + a) * 5 + b) * 5 + c) * 5 + e] = off16(P.int22[t][u][a][b][c][e], tau(u));
          }
        }
    }
  for (int t = 0; t < 8; t++)
    for (int a = 0; a < 5; a++)
      for (int b = 0; b < 5; b++) F.mm23in[t * 25 + a * 5 + b] = off16(P.mismatch23I[t][a][b], tau(t));
  for (int x = 0; x < 32; x++) {
    F.uni[x] = (int16_t)(F.NIN[x] < 32000 ? F.NIN[x] : 32000);
    F.uni[32 + x] = (int16_t)(F.IL[x] < 32000 ? F.IL[x] : 32000);
    F.uni[64 + 2 * x] = (int16_t)(F.BUL[x] < 32000 ? F.BUL[x] : 32000);
    F.uni[64 + 2 * x + 1] = x >= 4 ? (int16_t)(F.L1N[x - 1] < 32000 ? F.L1N[x - 1] : 32000) : (int16_t)32767;
  }
  for (int um = 0; um <= 30; um++)
    for (int e = 0; e < 128; e++) {
      const int u = e < 32 ? e + 4 : (e < 64 ? e - 32 : (e - 64) >> 1);  // NIN[u-4], IL[u], (BUL[u], L1N[u-1])
      F.uniG[um][e] = u <= um ? F.uni[e] : (int16_t)32767;
    }
}

// Byte offsets of the rolling rows for one window width (device memory, rebuilt by the host when the width changes): the row of
// loop size u for a cell whose diagonal d-2 sits in ring slot s is row (s - u) mod NR — offset * (W-4) * 2 bytes in the
// generic-loop view, * 4 in the interleaved bulge / 1xn view.  The cell code used to derive every row on the scalar unit
// (compare, select, add, multiply: ~5 instructions per row, a third of the scalar instructions of a fold); read from here with
// wave-uniform addresses they are scalar loads of 8-16 dwords.  By slot and size class ([slot][u >> 1], odd and even u apart:
// the narrow kernel uses the odd sizes' rows — an even size's row is the next one — and a few even ones), so that the rows one
// pass needs are runs of consecutive dwords.
struct SfFastRows {
  int32_t ci_odd[SF_FAST_NR][16], ci_even[SF_FAST_NR][16], bn_odd[SF_FAST_NR][16], bn_even[SF_FAST_NR][16];
};
static inline void sf_fast_build_rows(int W, SfFastRows &R) {
  for (int sl = 0; sl < SF_FAST_NR; sl++)
    for (int u = 0; u < 32; u++) {
      const int row = ((sl - u) % SF_FAST_NR + SF_FAST_NR) % SF_FAST_NR;
      if (u & 1) { R.ci_odd[sl][u >> 1] = row * (W - 4) * 2; R.bn_odd[sl][u >> 1] = row * (W - 4) * 4; }
      else { R.ci_even[sl][u >> 1] = row * (W - 4) * 2; R.bn_even[sl][u >> 1] = row * (W - 4) * 4; }
    }
}

// LDS carve (bytes); every piece a multiple of 4
struct SfFastLayout {
  int tri;  // int16 entries of the fML triangle (diagonals >= 4)
  int off_ci, off_c1n, off_cb, off_dml, off_list, off_next, off_tab, off_red, off_flag, off_S;
  int off_hc;     // (hc layouts) constraint characters | partners | enclosing pairs | int16 pseudo-energies
  int total;
};
// int16 entries: mismatch23 rows 0..6 (175), five more mismatch tables rows 1..6 (150 each), stack (64; holds
// stackT), d5, d3 (40 each), pair (64 bytes), one pad, the size tables (4 x 32: the size-tested code of the first four steps
// reads them here), reversed pair types (64 bytes),
// mismatch23 minus the terminal penalty of its type, rows 0..6 (175), one pad
#define SF_FAST_TAB_OLD (175 + 5 * 150 + 64 + 40 + 40 + 32 + 1 + 4 * 32)
#define SF_FAST_TAB_BYTES ((SF_FAST_TAB_OLD + 32 + 175 + 1) * 2)
// hc: the instantiation for constrained folds (sf_fold_constrained) also carries the window's constraint — characters,
// bracket partners, enclosing pairs: a byte each — and its Deigan pseudo-energies (int16) for positions 0..W+1
static inline __host__ __device__ SfFastLayout sf_fast_layout(int W, bool hc = false) {
  SfFastLayout L;
  int tri = (W - 4) * W - (W * (W - 1) / 2 - 6);  // sum_{d=4}^{W-1} (W-d)
  if (W > 128) tri = ((W + 3) / 2 - 3) * (W - 3);  // FOLD: (H-3) rows of S = W-3 entries (same area)
  else if (W > 4) tri += (W - 4 + (W & 1)) >> 1;   // triangle: every diagonal starts at an even index (see FBASE)
  if (tri < 0) tri = 0;
  tri = (tri + 1) & ~1;
  L.tri = tri;
  int o = tri * 2;
  const int RW = W - 4;  // a diagonal d >= 4 has at most W-4 cells
  const int roll = ((SF_FAST_NR * RW + 1) & ~1) * 2;
  // (W <= 128: CI has a row NR that mirrors its row 0, so that "the row after row r" exists for every r and the generic-loop
  // recurrence addresses the rows of two neighbouring loop sizes from ONE per-lane base)
  // (ring row 0 only ever holds the diagonals NR, 2 NR, ...: at most W - NR cells, so that is all the mirror rows keep)
  const int mirror = (W <= 128 && W > SF_FAST_NR) ? ((W - SF_FAST_NR + 1) & ~1) : 0;  // entries
  L.off_ci = o; o += roll + mirror * 2;  // the exterior pass reuses this area for f5[] and the mismatchExt table
  L.off_c1n = o; o += roll;
  L.off_cb = o; o += roll;
  L.off_dml = o;  // (end of the rolling tables; the rolling rows of multiloop-split minima that used to follow are gone)
  // the interleaved bulge / 1xn table gets row NR = a copy of its row 0 as well ("the row after row r" exists for every r: the
  // merged helper relies on it too); W < SF_HELP_MERGE_MAXW (merged helper): each of the two helper waves a 128-byte list of cells
  // (W <= 128: the same mirror row for every narrow instantiation — the bulge / 1xn minima address the rows of two loop sizes
  // from one base as well)
  o += mirror * 4;
  L.off_list = o;
  if (W < SF_HELP_MERGE_MAXW && SF_HELP_MERGE) o += 2 * 128;
  // (What follows the rolling tables matters: on the short diagonals the straight-line cell code reads candidates of
  // loop sizes that do not exist yet — up to 23 words past the end of a row, past the end of the last row into
  // whatever comes next — and relies on finding energy-sized values there (they get a weight of 32767 and saturate).
  // The mirror row and the parameter tables are that; the cell list is never in reach where it is written (W >= 64),
  // and is filled with "no structure" where it is not; the two control words below sit behind the tables.)
  L.off_tab = o; o += SF_FAST_TAB_BYTES;
  L.off_red = o;  // (unused)
  L.off_flag = o; o += 4;
  L.off_next = o; o += 4;  // index of the workgroup's next fold (dynamic distribution)
  L.off_S = o; o += (W + 2 + 3) & ~3;
  L.off_hc = o;
  if (hc) o += ((5 * (W + 2) + 3) & ~3);
  L.total = o;
  return L;
}

static inline bool sf_fast_w_supported(int W) { return W >= 16 && W <= SF_FAST_MAXW; }
static inline int sf_fast_threads(int W) { return W <= 128 ? 256 : 512; }  // two diagonal groups of 128 / 256

struct SfFastCtx {
  int16_t *fML, *CI, *C1N, *CB, *DMLr;
  const int16_t *tI, *t1n, *t23, *tM, *tH, *tStack, *tD5, *tD3;
  const int16_t *t23in;    // mismatch23 - TerminalAU(its type): the inner side of 2x3 loops
  const uint8_t *tRPair;   // reversed pair type of two bases
  const int16_t *tE;  // mismatchExt image (only where cg_ext is set)
  int cg_ext;         // 1: the c scratch holds c[i,j] + ExtLoop(i,j) (what the exterior sweep adds up); 0: c[i,j]
  const uint8_t *tPair, *S;
  const SfDevParams *D;
  const SfFastParams *F;
  const SfFastRows *R;
  int16_t *cg;
  int W, TAU, MLbase, MLclosing, MLintern;
  int fold;     // 1: the fML area is the folded rectangle (W > 128), 0: the triangle
  int bn_dup;   // 1: BN has a row NR that mirrors row 0 (merged helper)
  int maxd;     // largest allowed j - i of a base pair (max_bp_span - 1)
  SfHc8 hc;     // the fold's hard constraint (hc.c null: none)
  const int16_t *sc;  // its Deigan pseudo-energies, 1-based, or null
  const int16_t *uNIN;  // LDS copy of SfFastParams::uni (the size-tested code of diagonals < 12 reads it)
  int16_t *BN;  // sf_mfe_fast_kernel: the bulge and 1xn rolling tables interleaved, entry x = (CB[x], C1N[x]) in one
                // 32-bit word (CB / C1N above stay null there)
};

#define SF_TIDX(t, a, b) ((t)*25 + (a)*5 + (b))
// c scratch in device memory: row i (i <= W-4) holds the cells (i, j), j = i+4..W (W: the local window width);
// small enough (13.6 kB per workgroup at W=120) that the whole grid's scratch stays in L2 between the fill and
// the exterior pass, which reads it row by row
#define SF_CGIDX(i, j) (((i)-1) * (W - 3) - (((i)-1) * (i)) / 2 + ((j) - (i)-4))
// (even: a workgroup's slice stays 4-byte aligned.  The slice stride decides how the workgroups' tables fall on the L2's
// channels and lines, and with it how often a partially written line is evicted before its row is complete: at W = 120,
// L2 -> fabric writes per fold 4.1-4.3 kB at 6794 entries, 3.3 kB at 6796 or 6812, 4.0 at 6820, 5.7-7.7 at 6832..6896, and
// 9-16 kB whenever the stride is a whole number of 128-byte lines (6848, 6912, 6976) — profiles/r03/mfe_scratch_stride.txt)
// Which slice of the scratch a workgroup takes.  Workgroups go to the eight XCDs round robin (workgroup b runs on XCD b mod 8)
// and every XCD has its own L2: with slice b for workgroup b an XCD's L2 held 128 chunks of 13.6 kB scattered over 14 MB, and
// however the chunk stride was chosen some of its sets overflowed — 5.3-6.0 kB per fold of write-backs of dirty table lines at
// W = 120 with four workgroups per CU, 0.3 kB with two.  With the slices of one XCD's workgroups ADJACENT (one contiguous
// 1.75 MB range per L2) the same launch writes 1.25 kB per fold, 0.53 kB at W = 100 (was 5.2); same kernel time.
__host__ __device__ inline unsigned sf_fast_scratch_slot(const unsigned block, const unsigned grid) {
  return (block & 7u) * ((grid + 7u) >> 3) + (block >> 3);
}
// Per-workgroup stride of the scratch, in entries.  Which strides keep the tables of the resident workgroups in the L2 is an
// empirical matter (profiles/r03/mfe_scratch_stride.txt, profiles/r04/mfe_scratch_placement.txt): at W = 120, with the tables
// of one XCD's workgroups adjacent (sf_fast_scratch_slot), 7168 entries = 14 kB write 0.78 kB per fold to the fabric, the
// table's own 6796 entries 1.25 kB, 7104 / 7232 1.7 kB.
#define SF_CG_ENTRIES(W) ((W) == 120 ? 7168 : (((((W)-4) * ((W)-3)) / 2 + 8 + 3) & ~1))

// Size-dependent terms (loop initiation, asymmetry) are the same for every lane: they are read from small LDS
// tables with a wave-uniform address (a broadcast read).  (Keeping them spread over the lanes of a VGPR and
// fetching with v_readlane is NOT safe here: under the 128-VGPR cap such a register can be spilled and
// reloaded while only some lanes are active, and the inactive lanes' entries are then lost — observed on
// MI355X as wrong minima in a variant of this kernel.)
#define SF_UNI(tab, k) ((int)(tab)[k])

// One stretch of a multiloop split (see sf_fast_dml): len terms, term t = pa[t * sa] + pb[t * sb] with
// sa = AP ? S : -(S-1), sb = BP ? S : -(S-1).  Batches of eight terms whose addresses are base + compile-time offset
// (a negative stride is addressed from the batch's last term, so every offset is >= 0).  Full batches advance by a
// constant; the final batch is placed to end exactly at the stretch's last term (it may overlap the one before: a
// minimum does not mind seeing a term twice).  A stretch shorter than eight terms is ONE batch whose surplus
// terms repeat the last one, so that all of its reads are in flight together.  len is wave-uniform.
#define SF_SPLIT_NB 8  // terms per batch = LDS read pairs in flight
template <bool AP, bool BP>
__device__ __forceinline__ void sf_fast_split_stretch(const int16_t *pa, const int16_t *pb, const int len, const int S,
                                                      int &dec, int &dec2) {
  constexpr int NB = SF_SPLIT_NB;
  const int sa = AP ? S : -(S - 1), sb = BP ? S : -(S - 1);
  int a[NB], b[NB];
#define SF_SPLIT_REDUCE()                           \
  _Pragma("unroll") for (int k = 0; k < NB; k += 2) { \
    dec = sfd_min(dec, a[k] + b[k]);                \
    dec2 = sfd_min(dec2, a[k + 1] + b[k + 1]);      \
  }
  if (len >= NB) {
    if (!AP) pa += (NB - 1) * sa;
    if (!BP) pb += (NB - 1) * sb;
    const int16_t *pal = pa + (len - NB) * sa, *pbl = pb + (len - NB) * sb;
    for (int nb = (len - 1) / NB; nb > 0; --nb) {
#pragma unroll
      for (int k = 0; k < NB; k++) {
        a[k] = pa[AP ? k * S : (NB - 1 - k) * (S - 1)];
        b[k] = pb[BP ? k * S : (NB - 1 - k) * (S - 1)];
      }
      SF_SPLIT_REDUCE()
      pa += NB * sa;
      pb += NB * sb;
    }
#pragma unroll
    for (int k = 0; k < NB; k++) {
      a[k] = pal[AP ? k * S : (NB - 1 - k) * (S - 1)];
      b[k] = pbl[BP ? k * S : (NB - 1 - k) * (S - 1)];
    }
    SF_SPLIT_REDUCE()
  } else {
#pragma unroll
    for (int k = 0; k < NB; k++) {
      const int t = k < NB - 1 ? sfd_min(k, len - 1) : len - 1;
      a[k] = pa[t * sa];
      b[k] = pb[t * sb];
    }
    SF_SPLIT_REDUCE()
  }
#undef SF_SPLIT_REDUCE
}

// (CB, C1N) of the cell at column i0 of a row into the interleaved bulge / 1xn table; w points at the cell's own word.
// SHIFT: the C1N half belongs in the previous word (column 0's is never read: 1xn candidates start at column 2).
template <bool SHIFT>
__device__ __forceinline__ void sf_fast_publish_bn(int16_t *w, const int i0, const uint32_t bn) {
  if (SHIFT) {
    w[0] = (int16_t)sf_lo(bn);
    if (i0 > 0) w[-1] = (int16_t)sf_hi(bn);
  } else {
    sf_stw(w, bn);
  }
}

// Entry idx of an int16 table inside the parameter block (wave-uniform base F): the byte offset — the table's place in the block
// included — is formed in 32 bits, so that the load is "scalar base + 32-bit vector offset" (global_load saddr).  From
// F->tab[idx] the compiler builds a 64-bit address per lane (v_mad_u64_u32, v_lshl_add_u64, add / addc with carry: ~5
// half-rate vector instructions per gather, four gathers per cell pass).
// (SF_EMUL: plain indexing.  The real address path — wide kernel only — runs under tests/test_gpu_parity.py::
// test_mfe_energy_parity_both_kernels (widths 129 ... 256) and tools/gpu_wsweep_full.py)
#ifdef SF_EMUL
#define SF_GATHER16(F, field, idx) ((int)(F)->field[idx])
#else
// (the offset passes through an empty asm: otherwise the compiler splits the table's constant back off into a 64-bit add)
__device__ __forceinline__ int sf_gather16_at(const void *base, unsigned off) {
  asm("" : "+v"(off));
  return *(const int16_t *)((const char *)base + (size_t)off);
}
// (the wide kernel only: +2.7 % at W = 200, +1 % at W = 160; the narrow kernel measured 2-3 % SLOWER with it — fewer vector
// instructions but the gathers' addresses are ready later and its waits grow — and keeps the plain form)
#define SF_GATHER16(F, field, idx) \
  (FOLD ? sf_gather16_at((F), (unsigned)offsetof(SfFastParams, field) + ((unsigned)(idx) << 1)) : (int)(F)->field[idx])
#endif

#ifdef SF_EMUL
// TEST BUILD ONLY: every row of the interleaved bulge / 1xn view a cell reads must be one of the ring's NR rows or the mirror row
// behind them (the kernel records the view's base per workgroup; fibers of one emulated workgroup run in one thread)
static thread_local const char *sf_emul_bn_base = nullptr, *sf_emul_ci_base = nullptr;
static inline void sf_emul_check_bn_row(const char *row_ptr, long lane_bytes, long row_bytes, bool fold) {
  if (!sf_emul_bn_base) return;
  const long row = (row_ptr - sf_emul_bn_base - lane_bytes) / row_bytes;
  if (row < 0 || row > (fold ? SF_FAST_NR - 1 : SF_FAST_NR)) {
    fprintf(stderr, "sf_mfe_fast (emulation): bulge / 1xn row %ld outside the ring and its mirror row\n", row);
    abort();
  }
}
static inline void sf_emul_check_ci_row(const int16_t *row_ptr, long lane_entries, long row_entries, bool fold) {
  if (!sf_emul_ci_base) return;
  const long row = ((const char *)row_ptr - sf_emul_ci_base - 2 * lane_entries) / (2 * row_entries);
  if (row < 0 || row > (fold ? SF_FAST_NR - 1 : SF_FAST_NR)) {
    fprintf(stderr, "sf_mfe_fast (emulation): generic-loop row %ld outside the ring and its mirror row\n", row);
    abort();
  }
}
#endif

// One anti-diagonal for one thread.  H: this parity's per-size minima of the generic candidates of the
// enclosed cell (i+1, j-1) on entry, of (i, j) on exit.  slot2 = (d-2) mod NR, slotd = d mod NR.
// G ("guarded"): d < 36, the loop-size limit d-6 is below MAXLOOP and every size is tested against it;
// !G: all sizes 0..30 exist, the candidate code is one straight-line block the scheduler can pipeline.
// SEC: which sections of the cell this call runs.  Normally all of them, in this order; on the long diagonals,
// where the cells of a wave group fit one wave and the group's other wave would idle, the kernel gives
// SF_SEC_HELP to that wave and SF_SEC_P1 | SF_SEC_DML, then (after a barrier) SF_SEC_FIN to the first (see the kernel).
//   SF_SEC_P1    generic-loop recurrence (updates HP)
//   SF_SEC_HELP  special loops + bulge / 1xn minima -> eh (the minimum over those candidates of a pairable cell)
//   SF_SEC_DML   multiloop split -> dec
//   SF_SEC_C0    hairpin and generic minima -> e0 (needs HP)
//   SF_SEC_FIN   c = min(e0, eh, multiloop closing); publishes the cell (needs dec)
//   SF_SEC_PRE   (split steps) the terms the publish step adds to c — they depend on the sequence only — are looked up
//                BEFORE the exchange barrier, in the shadow of the cell's other LDS waits, and handed to the SF_SEC_FIN
//                call in `pub` (SF_SEC_POST): the finish after the barrier, which nothing can overlap, loses its three
//                dependent LDS round trips
enum { SF_SEC_P1 = 1, SF_SEC_HELP = 2, SF_SEC_DML = 4, SF_SEC_FIN = 8, SF_SEC_C0 = 16, SF_SEC_ALL = 31, SF_SEC_PRE = 32,
       SF_SEC_POST = 64 };
// UNP (the narrow kernels' steps from d0 = 36 on): the per-size minima of the generic-loop recurrence one int16 per register
// instead of two.  On MI355X every packed (VOP3P) instruction and the v_lshl_or that packs two 16-bit reads issue at half the rate
// of the 16-bit VOP2 forms v_min_i16 / v_add_u16 (profiles/r04/mfe_issue_rates.json), so a size pair costs 5 half-rate + 1
// full-rate vector instructions packed and 7 full-rate ones unpacked: 10.6 against 7.4 ns of a SIMD's vector time.  27
// registers instead of 14: affordable because the unsplit and the split steps are loops of their own (the kernel converts the
// state once, where the guarded steps end).  No saturation is needed: every size exists on these diagonals (no guarded weights),
// and INF16 + a parameter < 32767.  The guarded steps (v_add_i16 clamp: half-rate again) and the wide kernel measured slower
// with it and stay packed (profiles/r05/EXPERIMENTS.md).
struct SfHU {
  short v[27];  // v[x]: minimum over the generic candidates of total size x + 4
};
struct SfPub {
  uint32_t a;  // (mismatchI, mismatch1nI) of the reversed pair: what CI and C1N add to c
  uint32_t b;  // (MLstem + TerminalAU + MLintern, ExtLoop + TerminalAU): what fML and the scratch add to c
  int tau;     // TerminalAU of the pair: what CB adds
  int type;    // the cell's pair type (constraint applied): the finish after the barrier does not look it up again — two
               // dependent LDS round trips (nucleotides, then the pair table) at the head of a phase every other wave waits for
};
// UCAP (G code only): compile-time bound on the loop sizes that can exist — d <= 7: 1, d <= 11: 5 — so that the unrolled
// size tests above it (a scalar compare + branch each, ~110 of them) disappear from the first four steps of a fold, which
// cost as much as full steps before (profiles/r03/mfe_step_profile.txt).
// TBLK: the rolling rows' offsets come from the SfFastRows table (scalar loads) instead of the ring arithmetic — the kernel
// decides per instantiation (measured: +4 % at W = 120 / 200, +3 % at W = 128; the merged-helper instantiation lost 1-2 % while it
// spilled and gains 1-3 % since the round-4 pins — W = 64 +1.0, 77 +1.8, 100 +1.4, 117 +3.1 %; the generic wide instantiation, which
// still spills, keeps the arithmetic)
// @section cell_setup
template <bool G, int WT, int SEC, bool CH = false, bool FOLD = false, int UCAP = SFD_MAXLOOP, int TBLK = 1, bool MGH = false,
          bool UNP = false>
__device__ __forceinline__ void sf_fast_cell(const SfFastCtx &X, const int d, const int i, const bool valid,
                                             const int slot2, const int slotd, uint32_t (&HP)[14], SfHU &HU, int &ovf,
                                             const bool final_fml, const int fnb, int &fpart, int &dec, int &eh, int &e0,
                                             int &dprev, SfPub &pub, const int dml_lo = SFD_TURN + 1,
                                             const int dml_hi = 1 << 20) {
  // size tables: NIN[32] asymmetry | IL[32] loop initiation | 32 pairs (bulge[u], 1xn term of total size u — 32767 for u < 4)
  // third table: 32-bit pairs (bulge[u], 1xn term of total size u — 32767 for u < 4, where no 1xn loop exists)
  // The size tables are the same for every lane: they are read from DEVICE memory with wave-uniform addresses — scalar loads
  // through the scalar cache, off the LDS pipe (round 3: their LDS copy cost 56 broadcast reads per cell pass, a quarter of a
  // pass's LDS instructions; +1.5 % at W = 120).  CH (12 <= d < 36): the guarded copy for this diagonal's largest loop size —
  // sizes above it cost 32767 (round 2 built that copy per wave and step in LDS).
  // (G, the first four steps: every use sits behind its own size test — a scalar load there would be waited for on the
  // spot; that code keeps reading an LDS copy: measured 1 % faster at W = 200.)
  // (Fc: the parameter block through sf_const_base — its tables are addressed "one scalar base + offset" at every load)
  // (W > 128: measured 0.8 % slower with it, 1.5 % faster at W = 120 / 128 — profiles/r03/mfe_scalar_bases.txt)
  const SfFastParams *const Fc = FOLD ? X.F : sf_const_base(X.F);
  const int16_t *const ubase = G ? X.uNIN : (CH ? Fc->uniG[sfd_max(sfd_min(d - 2 - (SFD_TURN + 1), 30), 0)] : Fc->uni);
  const int16_t *const uNIN = ubase, *const uIL = ubase + 32, *const uBN = ubase + 64;
// Word x of a row holds (CB[x], C1N[x+1]) — the two left-edge candidates of a size, CB at column 1 and C1N at column
// 2, are then ONE word.  (SHIFT = false: (CB[x], C1N[x]), the layout until late in round 2.)
  constexpr bool SHIFT = true;
#define CBAT(x) X.BN[2 * (x)]
#define C1NAT(x) X.BN[2 * ((x) - (SHIFT ? 1 : 0)) + 1]
// H[x] (x = size - 4) lives in the int16 halves of HP[x/2]: 14 registers instead of 27 under the 128-VGPR cap
#define HGET(x) (((x)&1) ? ((int)HP[(x) >> 1] >> 16) : (int)(int16_t)(HP[(x) >> 1] & 0xffffu))
#define HSET(x, v)                                                                                       \
  HP[(x) >> 1] = ((x)&1) ? ((HP[(x) >> 1] & 0x0000ffffu) | ((uint32_t)(v) << 16))                         \
                         : ((HP[(x) >> 1] & 0xffff0000u) | ((uint32_t)(v)&0xffffu))
  const int W = WT ? WT : X.W, RW = W - 4;  // WT > 0: window width known at compile time (index math folds)
  if (!valid) return;
  const int j = i + d, i0 = i - 1;
  const uint8_t *S = X.S;
  const int umax = G ? sfd_min(SFD_MAXLOOP, d - 2 - (SFD_TURN + 1)) : SFD_MAXLOOP;
  // CH (short diagonals, 12 <= d < 36, straight-line code with guarded size tables): whole batches of sizes above
  // the wave-uniform limit are skipped — the kernel is close to VALU-bound, work on sizes that cannot exist is not free
  const int um = CH ? d - 2 - (SFD_TURN + 1) : SFD_MAXLOOP;
  int type = (SEC & SF_SEC_POST) ? pub.type : (d <= X.maxd ? X.tPair[S[i] * 8 + S[j]] : 0);  // max_bp_span: longer pairs do not exist
  if (!(SEC & SF_SEC_POST) && X.hc.c) {
    // hard constraint of the window (fc.hc_add_from_db, ScanFold-Scan.py:405-410): applied where the pair type is made.
    // The pairs enclosed by (i,j) keep their sequence-only types in the candidate look-ups: a pair the constraint
    // forbids has c = "none" and never wins.  A bracket pair of non-complementary bases (type 7) has no row in the int16
    // tables: such a fold goes to the exact kernel through the overflow list (the host routes those windows there anyway).
    type = sf_hc_type8(X.hc, type, i, j, d <= X.maxd);
    if (type == 7) { type = 0; ovf = 1; }
  }
  const int si1 = S[i + 1], sj1 = S[j - 1];
// @section publish_terms
  if (SEC & SF_SEC_PRE) pub.type = type;
  if ((SEC & SF_SEC_PRE) && type) {
    const int tr = X.tRPair[S[i] * 8 + S[j]];
    const int sp1 = S[i - 1], sq1 = S[j + 1];
    const int tau_in = tr > 2 ? X.TAU : 0;
    int stem, ext;
    if (i > 1 && j < W) { stem = X.tM[SF_TIDX(type, sp1, sq1)]; ext = X.tE[SF_TIDX(type, sp1, sq1)]; }
    else if (i > 1) stem = ext = X.tD5[type * 5 + sp1];
    else if (j < W) stem = ext = X.tD3[type * 5 + sq1];
    else stem = ext = 0;
    pub.a = sf_pk(X.tI[SF_TIDX(tr, sq1, sp1)], X.t1n[SF_TIDX(tr, sq1, sp1)]);
    pub.b = sf_pk(stem + tau_in + X.MLintern, ext + tau_in);
    pub.tau = tau_in;
  }
// @section cell_setup
// first entry of diagonal dd.  Triangle without diagonals 0..3: sum_{k=4}^{dd-1} (W-k); FOLD: see the file header
// (triangle: a diagonal of odd length W-dd is followed by one unused entry, so every diagonal starts at an even index and two
// neighbouring cells (2p, 2p+1) of a diagonal are ONE aligned 32-bit word — what sf_fast_dml2 reads)
#define FBASE(dd) (FOLD ? ((dd) <= (W + 3) / 2 ? ((dd)-4) * (W - 3) : (W - 1 - (dd)) * (W - 3) + (dd)-3) \
                        : (((dd)-4) * W - ((dd) * ((dd)-1) / 2 - 6) + (((dd)-4 + (W & 1)) >> 1)))
// length of diagonal dd in that layout
#define FLEN(dd) (W - (dd) + ((W - (dd)) & 1))
// row of diagonal d-2-u in the rolling tables
// (a v_readlane lane table for these offsets measured +1 % at W=120, -2 % at W=200, and is unsafe wherever the
// build spills registers — see SF_UNI — so the scalar unit keeps computing them)
#define ROW(u) ((slot2 - (u) < 0 ? slot2 - (u) + SF_FAST_NR : slot2 - (u)) * RW)
  // TBL (the all-sizes code): the rows' byte offsets come from SfFastRows — scalar loads of consecutive entries
  constexpr bool TBL = !G && (TBLK != 0) && SF_FAST_ROWTAB;
  const SfFastRows *const Rc = TBL ? sf_const_base(X.R) : X.R;  // (the base is pinned by a volatile asm: only where it is used)
  const int32_t *const rci_o = Rc->ci_odd[slot2], *const rci_e = Rc->ci_even[slot2];
  const int32_t *const rbn_o = Rc->bn_odd[slot2], *const rbn_e = Rc->bn_even[slot2];
#define ROWB_CI(u) (TBL ? (((u)&1) ? rci_o[(u) >> 1] : rci_e[(u) >> 1]) : 2 * ROW(u))
#define ROWB_BN(u) (TBL ? (((u)&1) ? rbn_o[(u) >> 1] : rbn_e[(u) >> 1]) : 4 * ROW(u))
#define CIROW(u) ((const int16_t *)((const char *)X.CI + ROWB_CI(u)))

// @section generic_recurrence
  if (SEC & SF_SEC_P1) {
  // ---- pass 1 (every cell): per-size minima of the generic interior candidates ----
  if (UNP) {
    static_assert(!UNP || (!G && !CH && !FOLD), "the unpacked recurrence exists for the all-sizes code of the narrow kernel only");
    const int32_t *const nin = Fc->NIN;  // wave-uniform: scalar loads of whole runs of the table
    // every candidate read of a batch is issued before the first is used: a pass is SF_UNP_PB-pair batches = that many dependent
    // LDS round trips — under load the round trip, not the arithmetic, is what a main wave's step consists of
    short r30a = 0, r30b = 0;
    {
      const int16_t *row = CIROW(30) + i0;
      r30a = row[3]; r30b = row[29];
    }
    constexpr int PB = SF_UNP_PB;
#pragma unroll
    for (int pb = 12; pb >= 1; pb -= PB) {  // descending: v[x - 2] is still the enclosed cell's
      short a3[PB], au[PB], b3[PB], bu[PB];
#pragma unroll
      for (int k = 0; k < PB; k++) {
        if (pb - k >= 1) {
          const int u = 2 * (pb - k) + 4;
          const int16_t *rb = CIROW(u + 1) + i0, *ra = rb + RW;  // row u is the row after row u + 1 (mirror row at the ring's seam)
#ifdef SF_EMUL
          sf_emul_check_ci_row(ra, i0, RW, FOLD); sf_emul_check_ci_row(rb, i0, RW, FOLD);
#endif
          a3[k] = ra[3]; au[k] = ra[u - 1];  // size u:     u1 = 2, u2 = 2
          b3[k] = rb[3]; bu[k] = rb[u];      // size u + 1
        }
      }
      if (pb == 12) {
        const short e = (short)(sf_opaque16(sfd_min16(r30a, r30b)) + (short)nin[26]);
        HU.v[26] = sf_opaque16(sfd_min16(e, HU.v[24]));
      }
#pragma unroll
      for (int k = 0; k < PB; k++) {
        if (pb - k >= 1) {
          const int x = 2 * (pb - k);  // size u = x + 4
          const short ea = (short)(sf_opaque16(sfd_min16(a3[k], au[k])) + (short)nin[x]);
          const short eb = (short)(sf_opaque16(sfd_min16(b3[k], bu[k])) + (short)nin[x + 1]);
          HU.v[x + 1] = sf_opaque16(sfd_min16(eb, HU.v[x - 1]));
          HU.v[x] = sf_opaque16(sfd_min16(ea, HU.v[x - 2]));
        }
      }
    }
    {
      const int16_t *row = CIROW(5) + i0;
      HU.v[1] = (short)(sfd_min16((short)row[3], (short)row[4]) + (short)nin[1]);
      HU.v[0] = (short)((short)(CIROW(4) + i0)[3] + (short)nin[0]);
    }
  } else {
  if (G) {
#pragma unroll
    for (int u = 30; u >= 6; --u) {
      if (u <= UCAP && u <= umax) {
        const int16_t *row = X.CI + ROW(u) + i0;
        const int e = sfd_min(row[3], row[u - 1]) + SF_UNI(uNIN, u - 4);  // u1 = 2 and u2 = 2
        HSET(u - 4, sfd_min(e, HGET(u - 6)));
      }
    }
  } else {
    // all sizes exist.  H[x] and H[x+1] (x even) share a register, so two sizes are updated by ONE packed
    // min / saturating add / min (v_pk_*_i16): the four candidates are packed pairwise (one v_perm each), the
    // asymmetry terms of both sizes are one aligned 32-bit read of the int16 table.  Descending order, so
    // HP[p-1] still holds the enclosed cell's minima.  Size 30 (x = 26) has no partner.
    if (!CH || um >= 30) {
      const int16_t *row = CIROW(30) + i0;
      const int e = sfd_min(row[3], row[29]) + SF_UNI(uNIN, 26);
      HSET(26, sfd_min(e, HGET(24)));
    }
#pragma unroll
    for (int pb = 12; pb >= 1; pb -= 3) {  // batches of three pairs = 12 candidate reads + 3 term reads in flight
      if (CH && um < 2 * pb) continue;     // sizes 2pb .. 2pb+5; skipped registers keep "none" (never written so far)
      uint32_t e1[3], e2[3], nn[3];
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const int u = 2 * (pb - k) + 4;
        // (!FOLD: row u is the row after row u+1 — the mirror row when that is the ring's last — at a compile-time offset)
        const int16_t *rb = CIROW(u + 1) + i0, *ra = FOLD ? CIROW(u) + i0 : rb + RW;
#ifdef SF_EMUL
        sf_emul_check_ci_row(ra, i0, RW, FOLD); sf_emul_check_ci_row(rb, i0, RW, FOLD);
#endif
        e1[k] = sf_pk(ra[3], rb[3]);          // u1 = 2
        e2[k] = sf_pk(ra[u - 1], rb[u]);      // u2 = 2
        nn[k] = sf_ldw(uNIN + (u - 4));
      }
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const int pp = pb - k;
        HP[pp] = sf_pkmin(sf_pkadd(sf_pkmin(e1[k], e2[k]), nn[k]), HP[pp - 1]);
      }
    }
  }
  // (CH below d = 12: sizes 4 / 5 may not exist yet — their guarded weight is 32767 and the sum must not wrap in the int16 store)
  if (!G || (UCAP >= 5 && umax >= 5)) {
    const int16_t *row = CIROW(5) + i0;
    const int e5 = sfd_min(row[3], row[4]) + SF_UNI(uNIN, 1);
    HSET(1, CH ? sfd_min(e5, 32767) : e5);
  }
  if (!G || (UCAP >= 4 && umax >= 4)) {
    const int e4 = (CIROW(4) + i0)[3] + SF_UNI(uNIN, 0);
    HSET(0, CH ? sfd_min(e4, 32767) : e4);
  }
  }

  }

// @section special_loops
  // ---- special loops, bulges, 1 x n loops (pairable cells): eh ----
  if (SEC & SF_SEC_HELP) {
    eh = SF_FAST_BIG;
    // The lane's byte offset into a row of the interleaved table, pinned: a row's address is then ONE add of the row's scalar
    // byte offset (the compiler otherwise rebuilds (row + lane) * 4 + table base — three vector instructions — for every row,
    // because the two-word reads have no room for the base in their offset fields).
    // (SF_EMUL: plain pointer arithmetic, with sf_emul_check_bn_row on every row.  The pinned LDS-address form below runs under
    // every -m gpu parity test of this kernel: test_mfe_energy_parity_both_kernels (15 widths), test_config2_all_energies_equal_oracle,
    // test_poisoned_lds_slack_does_not_move_an_energy (12 widths x 5 patterns), and the every-width sweep tools/gpu_wsweep_full.py)
#ifdef SF_EMUL
    const int bno = 4 * i0;
#define BNROWB(b) ((const int16_t *)((const char *)X.BN + (bno + (b))))
#else
    // (the table's own LDS offset goes into the pinned value too — the low half of a flat LDS address is the LDS offset — so
    // that nothing but the row's scalar offset is left to add)
    unsigned bno = (unsigned)(uintptr_t)(const void *)X.BN + 4u * (unsigned)i0;
    if (!G) SF_PIN(bno);
#define BNROWB(b) ((const int16_t *)(const __attribute__((address_space(3))) int16_t *)(uintptr_t)(bno + (unsigned)(b)))
#endif
    if (type && (!G || umax >= 0)) {
      const int TAU = X.TAU;
      const int tau_out = type > 2 ? TAU : 0;
      const int16_t *st = X.tStack + type * 8;
      // (every table below already carries "- TerminalAU(inner pair)": CB holds c + that term, see SfFastParams)
      const uint8_t *RP = X.tRPair;
      const unsigned tq = (unsigned)type * 8u;
      {  // stack (+ the Deigan pseudo-energies of its four nucleotides: fc.sc_add_SHAPE_deigan, ScanFold.py:533-539)
        const int t2r = RP[si1 * 8 + sj1];
        const int sc4 = X.sc ? X.sc[i] + X.sc[i + 1] + X.sc[j - 1] + X.sc[j] : 0;
        eh = sfd_min(eh, BNROWB(ROWB_BN(0))[2 * 1] + st[t2r] + sc4);
      }
      if (!G || umax >= 1) {  // one-nucleotide bulges keep the stack
        const int b1 = SF_UNI(uBN, 2);
        const int16_t *row = BNROWB(ROWB_BN(1));
        const int ta = RP[si1 * 8 + S[j - 2]];  // (i+1, j-2)
        eh = sfd_min(eh, row[2 * 1] + b1 + st[ta]);
        const int tb = RP[S[i + 2] * 8 + sj1];  // (i+2, j-1)
        eh = sfd_min(eh, row[2 * 2] + b1 + st[tb]);
      }
      if (!G || (UCAP >= 2 && umax >= 2)) {  // 1 x 1: (i+2, j-2)
        const unsigned t2r = RP[S[i + 2] * 8 + S[j - 2]];
        eh = sfd_min(eh, BNROWB(ROWB_BN(2))[2 * 2] + SF_GATHER16(Fc, int11T, ((tq + t2r) * 5u + si1) * 5u + sj1));
      }
      if (!G || (UCAP >= 3 && umax >= 3)) {  // 1 x 2 and 2 x 1
        const int16_t *row = BNROWB(ROWB_BN(3));
        const unsigned ta = RP[S[i + 2] * 8 + S[j - 3]];  // (i+2, j-3), sq1 = S[j-2]
        eh = sfd_min(eh, row[2 * 2] + SF_GATHER16(Fc, int21a, (((tq + ta) * 5u + si1) * 5u + S[j - 2]) * 5u + sj1));
        const unsigned tb = RP[S[i + 3] * 8 + S[j - 2]];  // (i+3, j-2), sp1 = S[i+2]
        eh = sfd_min(eh, row[2 * 3] + SF_GATHER16(Fc, int21b, (((tb * 8u + type) * 5u + sj1) * 5u + si1) * 5u + S[i + 2]));
      }
      if (!G || (UCAP >= 4 && umax >= 4)) {  // 2 x 2: (i+3, j-3)
        const unsigned t2r = RP[S[i + 3] * 8 + S[j - 3]];
        eh = sfd_min(eh, BNROWB(ROWB_BN(4))[2 * 3] +
                             SF_GATHER16(Fc, int22T, ((((tq + t2r) * 5u + si1) * 5u + S[i + 2]) * 5u + S[j - 2]) * 5u + sj1));
      }
      if (!G || (UCAP >= 5 && umax >= 5)) {  // 2 x 3 and 3 x 2
        const int16_t *row = BNROWB(ROWB_BN(5));
        const int m23 = X.t23[SF_TIDX(type, si1, sj1)] + Fc->L23;
        const int ta = RP[S[i + 3] * 8 + S[j - 4]];  // (i+3, j-4); sp1 = S[i+2], sq1 = S[j-3]
        eh = sfd_min(eh, row[2 * 3] + m23 + X.t23in[SF_TIDX(ta, S[j - 3], S[i + 2])]);
        const int tb = RP[S[i + 4] * 8 + S[j - 3]];  // (i+4, j-3); sp1 = S[i+3], sq1 = S[j-2]
        eh = sfd_min(eh, row[2 * 4] + m23 + X.t23in[SF_TIDX(tb, S[j - 2], S[i + 3])]);
      }
// @section bulge_1xn
      // bulges (size u >= 2) and 1 x n loops (total size u >= 4), one rolling row per u
      int gb = SF_FAST_BIG, g1 = SF_FAST_BIG;
      if (G) {
#pragma unroll
        for (int u = 2; u <= 30; ++u) {
          if (u <= UCAP && u <= umax) {
            const int rw = ROW(u) + i0;
            gb = sfd_min(gb, sfd_min(CBAT(rw + 1), CBAT(rw + 1 + u)) + SF_UNI(uBN, 2 * u));
            if (u >= 4) g1 = sfd_min(g1, sfd_min(C1NAT(rw + 2), C1NAT(rw + u)) + SF_UNI(uBN, 2 * u + 1));
          }
        }
      } else {
        // The candidates of size u are CB[1], CB[1+u] and C1N[2], C1N[u] of row u: with the two tables interleaved
        // they are the low / high halves of the word pairs (1, 2) and (u, u+1) — two two-word reads; one
        // bit-select each puts (bulge, 1xn) candidates side by side, and the rest is packed: min, saturating
        // add of the (bulge[u], 1xn[u]) weights (one uniform read), min.  Batches of SF_HELP_NB sizes.
        // (gfx950: a packed VOP3P result read by the very next instruction costs one s_nop — the per-size chain min -> add ->
        // min(acc) has two of them per size.  STAGED (the W = 120 instantiation): the sizes of a batch go through each stage
        // together, two accumulators take turns, and a fence keeps the vector stages in that order: same vector instructions,
        // 30 fewer idle issue slots per cell pass, +1 %.  The other instantiations measured 1-2 % SLOWER with it — their
        // longer live ranges spill — and keep the chain.)
        // (re-measured after the spill pins freed the generic narrow kernels' registers: still 0.4-3 % slower there)
        constexpr bool STAGED = (WT == 120);
        uint32_t acc = sf_pk(32767, 32767), accb = acc;
        // (TBLK == 2: the merged-helper instantiation of the generic narrow kernel, W < 118)
        constexpr int SF_HELP_NB = FOLD ? SF_HELP_NB_256 : (TBLK == 2 ? SF_HELP_NB_128G : SF_HELP_NB_128);
#pragma unroll
        for (int ub = 2; ub <= 30; ub += SF_HELP_NB) {
          if (CH && ub > um) continue;
          uint32_t w0[SF_HELP_NB], w1[SF_HELP_NB], w2[SF_HELP_NB], w3[SF_HELP_NB], wt[SF_HELP_NB], t[SF_HELP_NB];
#pragma unroll
          for (int k = 0; k < SF_HELP_NB; k++) {
            const int u = ub + k;
            if (u <= 30) {
              // (!FOLD: sizes (u, u+1), u even, share a base — row u is the row after row u+1, the mirror row at the ring's seam)
              const int16_t *tp = (FOLD || MGH || u == 30) ? BNROWB(ROWB_BN(u)) : BNROWB(ROWB_BN(u | 1)) + ((u & 1) ? 0 : 2 * RW);
#ifdef SF_EMUL
              sf_emul_check_bn_row((const char *)tp, 4 * i0, 4 * RW, FOLD);
#endif
              w0[k] = sf_ldw(tp + 2 * 1);
              if (!SHIFT) w1[k] = sf_ldw(tp + 2 * 2);
              w2[k] = sf_ldw(tp + 2 * (SHIFT ? u - 1 : u)); w3[k] = sf_ldw(tp + 2 * (u + 1));
              wt[k] = sf_ldw(uBN + 2 * u);
            }
          }
          if (STAGED) {
#pragma unroll
            for (int k = 0; k < SF_HELP_NB; k++) {
              const int u = ub + k;
              if (u <= 30) {
                const uint32_t x = SHIFT ? w0[k] : ((w0[k] & 0xffffu) | (w1[k] & 0xffff0000u));  // (CB[1], C1N[2])
                const uint32_t y = (w3[k] & 0xffffu) | (w2[k] & 0xffff0000u);  // (CB[1+u], C1N[u])
                t[k] = sf_pkmin(x, y);
              }
            }
            SF_VALU_FENCE();
#pragma unroll
            for (int k = 0; k < SF_HELP_NB; k++)
              if (ub + k <= 30) t[k] = sf_pkadd(t[k], wt[k]);
            SF_VALU_FENCE();
#pragma unroll
            for (int k = 0; k + 1 < SF_HELP_NB; k += 2)
              if (ub + k + 1 <= 30) t[k] = sf_pkmin(t[k], t[k + 1]);
            SF_VALU_FENCE();
#pragma unroll
            for (int k = 0; k < SF_HELP_NB; k += 2)
              if (ub + k <= 30) { if (k & 2) accb = sf_pkmin(accb, t[k]); else acc = sf_pkmin(acc, t[k]); }
          } else {
#pragma unroll
            for (int k = 0; k < SF_HELP_NB; k++) {
              const int u = ub + k;
              if (u <= 30) {
                const uint32_t x = SHIFT ? w0[k] : ((w0[k] & 0xffffu) | (w1[k] & 0xffff0000u));  // (CB[1], C1N[2])
                const uint32_t y = (w3[k] & 0xffffu) | (w2[k] & 0xffff0000u);  // (CB[1+u], C1N[u])
                acc = sf_pkmin(acc, sf_pkadd(sf_pkmin(x, y), wt[k]));
              }
            }
          }
        }
        acc = sf_pkmin(acc, accb);
        gb = sf_lo(acc); g1 = sf_hi(acc);
      }
      eh = sfd_min(eh, gb + tau_out);
      eh = sfd_min(eh, g1 + X.t1n[SF_TIDX(type, si1, sj1)]);
    }
  }

// @section multiloop_split_1cell
  // ---- multiloop split: before the barrier when the cell is finished by a later call, else after publishing c
  // (fewer values live across it) ----
  auto multiloop_split = [&]() {
    dec = SF_FAST_BIG;
  if (FOLD) {
    // min over m = 4 .. d-5 of fML[i, i+m] + fML[i+m+1, j] in the folded rectangle T (rows of S = W-3 entries):
    //   fML[i, i+m]   = T[FBASE(m) + i0]            moves by +S per m while m <= H, by -(S-1) per m after that;
    //   fML[i+m+1, j] = T[FBASE(d-1-m) + i0+m+1]    moves by +S per m while d-1-m > H, by -(S-1) per m after that
    // (the lane's own shift by one per m cancels against the fold).  Inside a stretch of m in which neither operand
    // changes sides every address of a batch of eight terms is the batch base plus a compile-time offset: one
    // pointer bump per operand and batch instead of one per term (14 instead of 29 vector instructions per batch).
    const int S_ = W - 3, H_ = (W + 3) / 2;
    const int mend = d - SFD_TURN - 2;
    const int bB = d - 1 - H_;  // first m whose second operand lies in the left part
    int dec2 = SF_FAST_BIG;
    const int16_t *T = X.fML + i0;
    // (a split step shares the terms between the waves of a cell: only m = dml_lo .. min(dml_hi, mend), wave-uniform)
    const int lo = dml_lo, hi = sfd_min(mend, dml_hi);
    {  // stretch 1, m = 4 .. bB-1: both operands step by +S (second operand in the right part)
      const int m0 = sfd_max(SFD_TURN + 1, lo), m1 = sfd_min(bB - 1, hi);
      if (m1 >= m0) sf_fast_split_stretch<true, true>(T + (m0 - 4) * S_, T + (W - d + m0) * S_ + d - 3, m1 - m0 + 1, S_, dec, dec2);
    }
    {  // stretch 2, m = max(4, bB) .. min(mend, H): first operand +S, second -(S-1) (both in the left part)
      const int m0 = sfd_max(sfd_max(SFD_TURN + 1, bB), lo), m1 = sfd_min(sfd_min(mend, H_), hi);
      if (m1 >= m0) sf_fast_split_stretch<true, false>(T + (m0 - 4) * S_, T + (d - 5 - m0) * S_ + m0 + 1, m1 - m0 + 1, S_, dec, dec2);
    }
    {  // stretch 3, m = H+1 .. mend: both operands step by -(S-1) (first operand in the right part)
      const int m0 = sfd_max(H_ + 1, lo), m1 = hi;
      if (m1 >= m0) sf_fast_split_stretch<false, false>(T + (W - 1 - m0) * S_ + m0 - 3, T + (d - 5 - m0) * S_ + m0 + 1, m1 - m0 + 1, S_, dec, dec2);
    }
    dec = sfd_min(dec, dec2);
  } else {
    // fML[i, i+m] = fML_tri[FBASE(m) + i0], fML[i+m+1, j] = fML_tri[FBASE(d-m-1) + i0+m+1].  From one term to the next the first
    // offset grows by the length of diagonal m, the second shrinks by the length of diagonal d-m-2 less one — lengths
    // that fall / rise by two every second diagonal (FLEN).  Inside a batch of eight terms (m0 even) each address is the
    // previous one plus ONE of two uniform byte steps (they alternate: a single v_add with a scalar operand), the rest of
    // the step sequence is a compile-time load offset:
    //   first operand:  a_k = a_0 + k L0 - e_k - 2 pa (k >> 1),  L0 = FLEN(m0), pa = (W - m0) & 1, e = 0 0 0 2 4 8 12 18 (24)
    //   second operand: b_k = b_0 - k (M - 1) - 2 g_k + q (k & 1),  M = W - (d-1-m0), q = M & 1, 2 g = 0 2 4 8 12 18 24 32 (40)
    // (a split step shares the terms between the waves of a cell: m = dml_lo .. min(dml_hi, d-5), wave-uniform; dml_lo even)
    int m = dml_lo;
    const int mend = sfd_min(d - SFD_TURN - 2, dml_hi);
    const char *pa = (const char *)(X.fML + i0 + FBASE(m)) - 36;                      // FBASE(4) = 0
    const char *pb = (const char *)(X.fML + i0 + 1 + FBASE(d - m - 1) + m) - 64;
    const int par_a = (W - m) & 1, M0 = W - d + 1 + m, par_b = M0 & 1;
    int sa0 = 2 * FLEN(m), sa1 = sa0 - 4 * par_a;               // after an even / odd term of the batch
    int sb0 = 2 * (M0 - 1 - par_b), sb1 = 2 * (M0 - 1 + par_b);
    int dec2 = SF_FAST_BIG;
    // (Software-pipelining these batches on the split steps' main waves — the reads of batch t+1 in flight while batch t
    // is reduced, 16 more registers — measured 0.6 % slower at W = 120, 3-4 % at W = 100 / 128 in round 3, as the pipelined
    // folded layout had in round 2: the split is bound by instruction issue under contention, not by the LDS latency.)
    for (; m + 7 <= mend; m += 8) {
      int a[8], b[8];
#pragma unroll
      for (int k = 0; k < 8; k++) {
        constexpr int E[8] = {0, 0, 0, 2, 4, 8, 12, 18}, G2[8] = {0, 2, 4, 8, 12, 18, 24, 32};
        a[k] = *(const int16_t *)(pa + (36 - 2 * E[k]));
        b[k] = *(const int16_t *)(pb + (64 - 2 * G2[k]));
        pa += (k & 1) ? (k < 7 ? sa1 : sa1 - 48) : sa0;  // the last step also rebases for the next batch
        pb -= (k & 1) ? (k < 7 ? sb1 : sb1 + 80) : sb0;
      }
      sa0 -= 16; sa1 -= 16; sb0 += 16; sb1 += 16;
#pragma unroll
      for (int k = 0; k < 8; k += 2) {
        dec = sfd_min(dec, a[k] + b[k]);
        dec2 = sfd_min(dec2, a[k + 1] + b[k + 1]);
      }
    }
    for (; m <= mend; m++) {
      dec = sfd_min(dec, *(const int16_t *)(pa + 36) + *(const int16_t *)(pb + 64));
      pa += 2 * FLEN(m);
      pb -= 2 * (FLEN(d - m - 2) - 1);
    }
    dec = sfd_min(dec, dec2);
  }
    };

// @section hairpin
  // ---- hairpin, generic minima, multiloop closing (pairable cells): e0 ----
  if (SEC & SF_SEC_C0) {
    e0 = SF_FAST_BIG;
    if (type) {
      const int TAU = X.TAU;
      int e;
      if (G && d <= 7) {
        // hairpins of 3, 4 and 6 nucleotides may be tabulated special loops (sfd_hairpin's rules).  The key list is
        // walked in blocks of eight wave-uniform entries without an early exit — scalar loads, one compare and one
        // select per key — last to first, so that the FIRST matching entry wins as in sfd_special_hairpin.  (The
        // per-key loop with its exit test cost ~2 k cycles in each of the first two steps of every fold.)
        const int size = d - 1;
        e = X.D->hp_init[size] + (size == 3 ? (type > 2 ? TAU : 0) : (int)X.tH[SF_TIDX(type, si1, sj1)]);
        if (size == 3 || size == 4 || size == 6) {
          const uint32_t *keys = size == 4 ? X.D->tetra_key : (size == 6 ? X.D->hexa_key : X.D->tri_key);
          const int32_t *en = size == 4 ? X.D->P.tetra_E : (size == 6 ? X.D->P.hexa_E : X.D->P.tri_E);
          const int n = size == 4 ? X.D->P.n_tetra : (size == 6 ? X.D->P.n_hexa : X.D->P.n_tri);
          const uint32_t key = sfd_loop_key(S, i, size + 2);
          for (int k0 = ((n + 7) & ~7) - 8; k0 >= 0; k0 -= 8) {  // (the arrays hold SF_NSPECIAL = 40 entries; unused keys never match)
#pragma unroll
            for (int k = 7; k >= 0; --k) e = keys[k0 + k] == key ? (int)en[k0 + k] : e;
          }
        }
      } else e = X.D->hp_init[d - 1] + X.tH[SF_TIDX(type, si1, sj1)];
// @section generic_minima
      if (!G || umax >= 0) {
        int gg = SF_FAST_BIG;
        if (UNP) {
          // one size per full-rate add / min; two accumulators
          const int32_t *const il = Fc->IL;
          short g0 = (short)(HU.v[2] + (short)il[6]), g1 = (short)(HU.v[3] + (short)il[7]);
#pragma unroll
          for (int x = 4; x <= 26; x += 2) {
            g0 = sfd_min16(g0, sf_opaque16((short)(HU.v[x] + (short)il[x + 4])));
            if (x + 1 <= 26) g1 = sfd_min16(g1, sf_opaque16((short)(HU.v[x + 1] + (short)il[x + 5])));
          }
          gg = sfd_min16(g0, g1);
        } else if (G) {
#pragma unroll
          for (int u = 6; u <= 30; ++u)
            if (u <= UCAP && u <= umax) gg = sfd_min(gg, HGET(u - 4) + SF_UNI(uIL, u));
        } else {
          // generic minima plus loop initiation, two sizes per packed add / min (size 31 does not exist: its
          // half of HP[13] stays INF)
          uint32_t ggp = sf_pk(32767, 32767);
#pragma unroll
          for (int pp = 1; pp <= 13; pp++)
            if (!CH || um >= 2 * pp + 4) ggp = sf_pkmin(ggp, sf_pkadd(HP[pp], sf_ldw(uIL + 2 * pp + 4)));
          gg = sfd_min(sf_lo(ggp), sf_hi(ggp));
        }
        e = sfd_min(e, gg + X.tI[SF_TIDX(type, si1, sj1)]);
      }
// @section multiloop_closing
      // multiloop closed by (i,j)
      {
        const int tr = X.tRPair[S[i] * 8 + S[j]];  // type != 0 here: the reversed type of the cell's own pair
        const int dml = dprev;  // multiloop split of (i+1, j-1): this thread's previous cell
        e = sfd_min(e, dml + X.tM[SF_TIDX(tr, sj1, si1)] + (tr > 2 ? TAU : 0) + X.MLintern + X.MLclosing);
      }
      e0 = e;
    }
  }
// @section finish_publish
  if ((SEC & SF_SEC_DML) && !(SEC & SF_SEC_FIN)) multiloop_split();

  if (!(SEC & SF_SEC_FIN)) return;
  // ---- c[i,j] of a pairable cell ----
  int c = SF_INF16;
  if (type) {
    c = sfd_min(e0, eh);
    if (c < SF_FAST_OVF) ovf = 1;
  }

  // ---- publish the cell ----
  // ---- publish the cell ----
  const int rbd = slotd * RW + i0;
  int f = SF_FAST_BIG, cx = SF_INF16;
  if (SEC & SF_SEC_POST) {
    // (no test of the pair type: a cell that cannot pair has c = "none" and publish terms that are all zero — the stores below then
    // publish exactly "none" — and the divergent branch was a few dozen cycles of a phase every other wave of the workgroup waits for)
    X.CI[rbd] = (int16_t)(c + sf_lo(pub.a));
    if (!FOLD && slotd == 0) X.CI[SF_FAST_NR * RW + i0] = (int16_t)(c + sf_lo(pub.a));
    const uint32_t bn = sf_pk(c + pub.tau, c + sf_hi(pub.a));  // (CB, C1N)
    sf_fast_publish_bn<SHIFT>(X.BN + 2 * rbd, i0, bn);
    if (X.bn_dup && slotd == 0) sf_fast_publish_bn<SHIFT>(X.BN + 2 * (SF_FAST_NR * RW + i0), i0, bn);
    f = c + sf_lo(pub.b);
    cx = sfd_min(c + sf_hi(pub.b), SF_INF16);
  } else if (type) {
    const int tr = X.tRPair[S[i] * 8 + S[j]];
    const int sp1 = S[i - 1], sq1 = S[j + 1];
    const int tau_in = tr > 2 ? X.TAU : 0;
    X.CI[rbd] = (int16_t)(c + X.tI[SF_TIDX(tr, sq1, sp1)]);
    if (!FOLD && slotd == 0) X.CI[SF_FAST_NR * RW + i0] = X.CI[rbd];
    const uint32_t bn = sf_pk(c + tau_in, c + X.t1n[SF_TIDX(tr, sq1, sp1)]);  // (CB, C1N)
    sf_fast_publish_bn<SHIFT>(X.BN + 2 * rbd, i0, bn);
    if (X.bn_dup && slotd == 0) sf_fast_publish_bn<SHIFT>(X.BN + 2 * (SF_FAST_NR * RW + i0), i0, bn);
    // E_MLstem and ExtLoop of (type, S[i-1], S[j+1]) differ in the mismatch table only; at the sequence ends both
    // are a dangle
    int stem, ext;
    if (i > 1 && j < W) { stem = X.tM[SF_TIDX(type, sp1, sq1)]; ext = X.tE[SF_TIDX(type, sp1, sq1)]; }
    else if (i > 1) stem = ext = X.tD5[type * 5 + sp1];
    else if (j < W) stem = ext = X.tD3[type * 5 + sq1];
    else stem = ext = 0;
    f = c + stem + tau_in + X.MLintern;
    cx = sfd_min(c + ext + tau_in, SF_INF16);
  } else {
    X.CI[rbd] = SF_INF16; sf_fast_publish_bn<SHIFT>(X.BN + 2 * rbd, i0, sf_pk(SF_INF16, SF_INF16));
    if (!FOLD && slotd == 0) X.CI[SF_FAST_NR * RW + i0] = SF_INF16;
    if (X.bn_dup && slotd == 0) sf_fast_publish_bn<SHIFT>(X.BN + 2 * (SF_FAST_NR * RW + i0), i0, sf_pk(SF_INF16, SF_INF16));
  }
  // the scratch (row i, column j: the exterior sweep reads rows coalesced) takes c + ExtLoop, the only form the
  // sweep needs; the traceback (native windows only) subtracts the term again (sf_fast_c).  (Storing only the cells
  // that can pair — 3 of 8 — and masking the rest in the sweeps was measured: -5 % speed, and MORE memory traffic,
  // 16 kB against 7.5 kB written per fold: partially written lines are fetched first.  profiles/r02/mfe_scratch_traffic.txt)
  X.cg[SF_CGIDX(i, j)] = (int16_t)cx;
  // fML[i,j]: the two neighbours on diagonal d-1 are final only for the even-diagonal group, which gets their
  // minimum (+MLbase) in fnb from its own fix-up of the previous step (see the kernel)
  if (final_fml && d > SFD_TURN + 1) f = sfd_min(f, fnb);
  if ((SEC & SF_SEC_DML) && (SEC & SF_SEC_FIN)) multiloop_split();
  f = sfd_min(f, dec);
  dprev = dec > SF_FAST_THRESH ? SF_INF16 : dec;
  fpart = f;
  if (final_fml && f < SF_FAST_OVF) ovf = 1;
  // final on the even diagonal; provisional (neighbour term still missing) on the odd one
  X.fML[FBASE(d) + i0] = (int16_t)(f > SF_FAST_THRESH ? SF_INF16 : f);
#undef ROW
#undef ROWB_CI
#undef ROWB_BN
#undef CIROW
#undef BNROWB
}

// @section wave_min
// minimum over the 64 lanes of a wave, returned in every lane.  DPP row operations + one readlane: no LDS
// round trips (the generic __shfl_xor butterfly lowers to ds_bpermute, ~6 dependent LDS-crossbar trips).
// (SF_EMUL: a plain loop over the lanes.  The DPP sequence runs in the trailing sweep of every fold whose structure is not wanted:
// tests/test_gpu_parity.py::test_mfe_energy_parity_both_kernels, test_config2_all_energies_equal_oracle)
__device__ __forceinline__ int sf_wave_min(int v) {
#ifdef SF_EMUL
  return sfemul_wave_min(v);
#else
  v = sfd_min(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
  v = sfd_min(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
  v = sfd_min(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xF, 0xF, false));  // row_half_mirror
  v = sfd_min(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xF, 0xF, false));  // row_mirror
  v = sfd_min(v, __builtin_amdgcn_update_dpp(v, v, 0x142, 0xA, 0xF, false));  // row_bcast:15 -> rows 1,3
  v = sfd_min(v, __builtin_amdgcn_update_dpp(v, v, 0x143, 0xC, 0xF, false));  // row_bcast:31 -> rows 2,3
  return __builtin_amdgcn_readlane(v, 63);
#endif
}

// the 32-bit value of the next lane (lane l gets lane l+1's; the last lane's result is unspecified)
// (SF_EMUL: __shfl_down.  The DPP form runs in sf_fast_dml2 — the split steps of every narrow-kernel fold: the same GPU tests)
__device__ __forceinline__ uint32_t sf_wave_next(uint32_t v) {
#ifdef SF_EMUL
  return __shfl_down(v, 1);
#else
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xF, 0xF, true);  // wave_shl:1, bound_ctrl: no "old" value to set up (a v_mov per use)
#endif
}

// @section multiloop_split_2cell
// Multiloop split of a WHOLE diagonal by one wave (split steps of the narrow kernel: the diagonal's n = W-d <= 62 cells are
// all in this wave): dec[i] = min_{m=4}^{d-5} fML[i, i+m] + fML[i+m+1, j] for every cell, returned to the lane that owns the cell.
// The one-cell-per-lane form keeps (W-d)/64 of the lanes busy and issues two 16-bit reads and ~4 vector instructions per
// term and cell.  Here a lane takes TWO neighbouring cells (2p, 2p+1): their first operands are one aligned word of diagonal
// m (every diagonal starts at an even index, FBASE), their second operands one word of diagonal d-1-m — aligned for odd m;
// for even m they straddle two words, the lane's own and the next lane's (sf_wave_next, a DPP move: lane p+1 reads the word
// that follows) — and sums / minima are packed int16 (saturating: INF + INF stays "none", sums of real energies are exact
// below the overflow threshold that sends a fold to the int32 kernel anyway).  The n/2 pairs need at most 31 lanes (+ one
// that only feeds its word to its neighbour), so the wave's 64 lanes form 2, 4 or 8 chunks (W-d <= 62 / 30 / 14) that share
// the terms: chunk c takes T8 terms from m = 4 + c S (the last chunks clamped to end at the last term; a minimum does not
// mind a term twice), and a butterfly over the chunks joins them.  Batches of eight terms starting at even m with the address
// arithmetic of the one-cell loop (two alternating steps + compile-time offsets, here per lane).  d odd: the term m = d-5
// (even, so that the batches can stop at an odd m) is done by every chunk on its own.
// LGC: the chunk width as a compile-time constant (the caller runs the steps of each width as a loop of their own), 0 = from d
template <int WT, int LGC = 0>
__device__ __forceinline__ int sf_fast_dml2(const SfFastCtx &X, const int d, const int lane, const int i_own, const bool valid) {
  constexpr bool FOLD = false;
  const int W = WT ? WT : X.W;
  const int n = W - d, npairs = (n + 1) >> 1;
  const int lg = LGC ? LGC : (npairs <= 7 ? 3 : (npairs <= 15 ? 4 : 5));  // lanes per chunk = 1 << lg (pairs + at least one feeder lane)
  const int NC = 64 >> lg;
  const int q = lane & ((1 << lg) - 1), c = lane >> lg;
  const int elast = (d & 1) ? d - 6 : d - 5;            // last term of the batched range (odd)
  const int N = elast - 3;                              // terms 4 .. elast (an even number)
  const int S = (((N + NC - 1) >> (6 - lg)) + 1) & ~1;  // a chunk's share, even (NC = 64 >> lg: a shift, not a division by a run-time value)
  const int T8 = (S + 7) & ~7;                          // what it runs: whole batches
  const int ms = sfd_min(4 + c * S, elast - T8 + 1);    // per lane (per chunk), even
  const int ys = d - 1 - ms;
  const int par_a = W & 1, M0 = W - ys, par_b = (W - d + 1) & 1;
  const char *pa = (const char *)(X.fML + FBASE(ms) + 2 * q) - 36;
  const char *pb = (const char *)(X.fML + FBASE(ys) + 2 * q + ms + 1) - 64;
  int sa0 = 2 * (W - ms + par_a), sa1 = sa0 - 4 * par_a;
  int sb0 = 2 * (M0 - 1 - par_b), sb1 = 2 * (M0 - 1 + par_b);
  uint32_t acc0 = sf_pk(32767, 32767), acc1 = acc0;
  for (int t = 0; t < T8; t += 8) {
    uint32_t wa[8], wb[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      constexpr int E[8] = {0, 0, 0, 2, 4, 8, 12, 18}, G2[8] = {0, 2, 4, 8, 12, 18, 24, 32};
      wa[k] = sf_ldw((const int16_t *)(pa + (36 - 2 * E[k])));
      wb[k] = sf_ldw((const int16_t *)(pb + (64 - 2 * G2[k] - ((k & 1) ? 0 : 2))));  // even m: the word its first half ends
      pa += (k & 1) ? (k < 7 ? sa1 : sa1 - 48) : sa0;
      pb -= (k & 1) ? (k < 7 ? sb1 : sb1 + 80) : sb0;
    }
    sa0 -= 16; sa1 -= 16; sb0 += 16; sb1 += 16;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      uint32_t b = wb[k];
      if (!(k & 1)) b = (sf_wave_next(b) << 16) | (b >> 16);
      if (k & 1) acc1 = sf_pkmin(acc1, sf_pkadd(wa[k], b)); else acc0 = sf_pkmin(acc0, sf_pkadd(wa[k], b));
    }
  }
  if (d & 1) {  // m = d-5: first operands on diagonal d-5, second on diagonal 4 from index 2q + d-4 (odd)
    const uint32_t a = sf_ldw(X.fML + FBASE(d - 5) + 2 * q);
    const uint32_t w = sf_ldw(X.fML + 2 * q + d - 5);
    acc0 = sf_pkmin(acc0, sf_pkadd(a, (sf_wave_next(w) << 16) | (w >> 16)));
  }
  // the chunks' partial minima meet by butterfly: lanes l ^ 32, l ^ 16, l ^ 8 — on the vector ALU (sf_pkmin_xor*: v_permlane32_swap /
  // v_permlane16_swap / a DPP row rotate, gfx950), not through the LDS crossbar: __shfl_xor is a ds_bpermute, and each of the one
  // to three of them was a dependent LDS round trip in the middle of the main wave's step
  uint32_t acc = sf_pkmin(acc0, acc1);
  acc = sf_pkmin_xor32(acc);
  if (lg <= 4) acc = sf_pkmin_xor16(acc);
  if (lg <= 3) acc = sf_pkmin_xor8(acc);
  // the cell's owner fetches its pair's result (every chunk holds it now) and takes its half
  const int i0 = valid ? i_own - 1 : 0;
  const uint32_t r = __shfl(acc, i0 >> 1);
  return (i0 & 1) ? sf_hi(r) : sf_lo(r);
}

// @section traceback
// c[i,j] from the scratch (which may hold c + ExtLoop, see SfFastCtx::cg_ext)
__device__ __forceinline__ int sf_fast_c(const SfFastCtx &X, const int16_t *tExt, const int i, const int j) {
  const int W = X.W;
  const int v = X.cg[SF_CGIDX(i, j)];
  if (!X.cg_ext || v >= SF_INF16) return v;
  const uint8_t *S = X.S;
  const int type = X.tPair[S[i] * 8 + S[j]];
  int ext;
  if (i > 1 && j < W) ext = tExt[SF_TIDX(type, S[i - 1], S[j + 1])];
  else if (i > 1) ext = X.tD5[type * 5 + S[i - 1]];
  else if (j < W) ext = X.tD3[type * 5 + S[j + 1]];
  else ext = 0;
  return v - ext - (type > 2 ? X.TAU : 0);
}

// Wave-cooperative traceback over the tables the fill left behind (fML triangle in LDS, c in device memory,
// f5 in LDS).  Same order of alternatives as sf_mfe_full_kernel / the oracle; candidate
// tests are spread over the 64 lanes and the first hit in that order is taken with ballot + ffs.
// Every lane executes the same control flow on the same values, so stack and string writes are redundant
// identical stores.  Returns nonzero if some table value has no decomposition.
__device__ inline int sf_fast_traceback(const SfFastCtx &X, const int32_t *f5s, const int16_t *tExt, const int lane,
                                        int16_t *stI, int16_t *stJ, int16_t *stM, char *dbL) {
  const int W = X.W;
  const bool FOLD = X.fold != 0;
  const uint8_t *S = X.S;
  const SfDevParams *D = X.D;
#define TC(i, j) sf_fast_c(X, tExt, (i), (j))
#define TF(i, j) (((j) - (i) < SFD_TURN + 1) ? SF_INF16 : (int)X.fML[FBASE((j) - (i)) + (i)-1])
#define TPAIR(i, j) (((j) - (i)) <= X.maxd ? (int)X.tPair[S[i] * 8 + S[j]] : 0)
  for (int x = lane; x < W; x += 64) dbL[x] = '.';
  int sp = 0, bad = 0;
  stI[0] = 1; stJ[0] = (int16_t)W; stM[0] = 0; sp = 1;
  while (sp > 0 && !bad) {
    --sp;
    int i = stI[sp], j = stJ[sp];
    const int ml = stM[sp];
    bool have_pair = false;
    if (ml == 0) {
      while (j > 0 && f5s[j] == f5s[j - 1]) j--;
      if (j < SFD_TURN + 2) continue;
      const int fij = f5s[j];
      int found = -1;
      for (int base = j - SFD_TURN - 1; base >= 1 && found < 0; base -= 64) {
        const int k = base - lane;
        bool ok = false;
        if (k >= 1) {
          const int type = TPAIR(k, j);
          if (type) {
            int ext;
            if (k > 1 && j < W) ext = tExt[SF_TIDX(type, S[k - 1], S[j + 1])];
            else if (k > 1) ext = X.tD5[type * 5 + S[k - 1]];
            else if (j < W) ext = X.tD3[type * 5 + S[j + 1]];
            else ext = 0;
            ok = (fij == f5s[k - 1] + TC(k, j) + ext + (type > 2 ? X.TAU : 0));
          }
        }
        const unsigned long long m = __ballot(ok);
        if (m) found = base - (__ffsll(m) - 1);
      }
      if (found < 0) { bad = 1; break; }
      stI[sp] = 1; stJ[sp] = (int16_t)(found - 1); stM[sp] = 0; sp++;
      i = found;
      have_pair = true;
    } else {
      if (j - i < SFD_TURN + 1) { bad = 1; break; }
      while (j - i > SFD_TURN + 1 && TF(i, j - 1) < SF_INF16 && TF(i, j) == TF(i, j - 1) + X.MLbase) j--;
      while (j - i > SFD_TURN + 1 && TF(i + 1, j) < SF_INF16 && TF(i, j) == TF(i + 1, j) + X.MLbase) i++;
      const int fij = TF(i, j);
      const int type = TPAIR(i, j);
      int stem = 0;
      if (type) {
        if (i > 1 && j < W) stem = X.tM[SF_TIDX(type, S[i - 1], S[j + 1])];
        else if (i > 1) stem = X.tD5[type * 5 + S[i - 1]];
        else if (j < W) stem = X.tD3[type * 5 + S[j + 1]];
        stem += (type > 2 ? X.TAU : 0) + X.MLintern;
      }
      if (type && fij == TC(i, j) + stem) {
        have_pair = true;
      } else {
        int found = -1;
        for (int base = i + SFD_TURN + 1; base <= j - SFD_TURN - 2 && found < 0; base += 64) {
          const int k = base + lane;
          bool ok = false;
          if (k <= j - SFD_TURN - 2) {
            const int a = TF(i, k), b = TF(k + 1, j);
            ok = a < SF_INF16 && b < SF_INF16 && fij == a + b;
          }
          const unsigned long long m = __ballot(ok);
          if (m) found = base + (__ffsll(m) - 1);
        }
        if (found < 0) { bad = 1; break; }
        stI[sp] = (int16_t)i; stJ[sp] = (int16_t)found; stM[sp] = 1; sp++;
        stI[sp] = (int16_t)(found + 1); stJ[sp] = (int16_t)j; stM[sp] = 1; sp++;
      }
    }
    while (have_pair) {
      dbL[i - 1] = '(';
      dbL[j - 1] = ')';
      const int type = TPAIR(i, j);
      const int cij = TC(i, j);
      if (cij == sfd_hairpin(D, S, i, j, type)) break;
      const int d = j - i;
      const int umax = sfd_min(SFD_MAXLOOP, d - 2 - (SFD_TURN + 1));
      int found = -1;
      if (umax >= 0) {
        // candidates in the order u1 ascending (p ascending), u2 ascending (q descending): t = u1*31 + u2
        const int tmax = umax * 31 + umax;
        for (int base = 0; base <= tmax && found < 0; base += 64) {
          const int t = base + lane;
          const int u1 = t / 31, u2 = t - u1 * 31;
          bool ok = false;
          if (u1 + u2 <= umax) {
            const int p = i + 1 + u1, q = j - 1 - u2;
            const int t2 = TPAIR(p, q);
            if (t2)
              ok = (cij == sfd_intloop(D, u1, u2, type, sfd_rtype(t2), S[i + 1], S[j - 1], S[p - 1], S[q + 1]) + TC(p, q) +
                           ((X.sc && (u1 | u2) == 0) ? X.sc[i] + X.sc[i + 1] + X.sc[j - 1] + X.sc[j] : 0));
          }
          const unsigned long long m = __ballot(ok);
          if (m) found = base + (__ffsll(m) - 1);
        }
      }
      if (found >= 0) {
        const int u1 = found / 31, u2 = found - u1 * 31;
        i = i + 1 + u1;
        j = j - 1 - u2;
        continue;
      }
      const int tr = sfd_rtype(type);
      const int mm = X.MLclosing + X.tM[SF_TIDX(tr, S[j - 1], S[i + 1])] + (tr > 2 ? X.TAU : 0) + X.MLintern;
      int fk = -1;
      for (int base = i + 1 + SFD_TURN + 1; base <= j - 1 - SFD_TURN - 2 && fk < 0; base += 64) {
        const int k = base + lane;
        bool ok = false;
        if (k <= j - 1 - SFD_TURN - 2) {
          const int a = TF(i + 1, k), b = TF(k + 1, j - 1);
          ok = a < SF_INF16 && b < SF_INF16 && cij == a + b + mm;
        }
        const unsigned long long m = __ballot(ok);
        if (m) fk = base + (__ffsll(m) - 1);
      }
      if (fk < 0) { bad = 1; break; }
      stI[sp] = (int16_t)(i + 1); stJ[sp] = (int16_t)fk; stM[sp] = 1; sp++;
      stI[sp] = (int16_t)(fk + 1); stJ[sp] = (int16_t)(j - 1); stM[sp] = 1; sp++;
      break;
    }
  }
#undef TC
#undef TF
#undef TPAIR
  return bad;
}


// @section exterior_sweep_native
// NG = threads per diagonal group.  The workgroup has two groups: group 0 handles the even diagonals, group 1
// the odd ones.  c[.,.] of diagonal d+1 does not depend on diagonal d (an enclosed pair spans at most d-1, the
// multiloop split of d+1 reads fML spans <= d-3), only fML[d+1] needs its two neighbours on d — so the pair
// (d, d+1) is computed concurrently by the two groups, then group 1 adds the neighbour term after one barrier.
// Exterior loop f5[j] = min(f5[j-1], min_i f5[i-1] + c[i,j] + ExtLoop(i,j)), the energy / overflow record and
// (for native windows) the traceback: ONE wave, after the fill.  No workgroup barrier inside.
// NQ = columns per lane (ceil(W/64)).  etab: c[i,j] + ExtLoop(i,j), laid out like the c scratch, prepared in LDS
// by the whole workgroup (sf_fast_ext_table) — or null, then the terms are looked up here from the scratch.
template <int NQ>
__device__ __forceinline__ void sf_fast_exterior(const SfFastCtx &X, const int W, const int lane, const int seq,
                                                 int32_t *f5s, const int16_t *tExt, const int16_t *etab,
                                                 const int32_t *flag,
                                                 int16_t *stack_area, char *dbL, int32_t *__restrict__ out,
                                                 int *__restrict__ ovf_cnt, int *__restrict__ ovf_list,
                                                 const int trace_stride, char *__restrict__ db_out,
                                                 int *__restrict__ status) {
  const uint8_t *S = X.S;
  const uint8_t *tPair = X.tPair;
  // Lane l owns the columns j = l+1, l+65, ...: P[q] = min over the closing rows i seen so far of
  // f5[i-1] + c[i,j] + ExtLoop(i,j).  Rows are visited in ascending order; f5[i-1] = min(f5[i-2], P of column
  // i-1) is final by then (column i-1 only has rows <= i-5), so one step is: fetch that P from its lane
  // (v_readlane), one coalesced row of c (fetched a few rows ahead), one table look-up and a min per lane.
  // No reduction across lanes anywhere.
  int P[NQ], sj[NQ], sj1[NQ];
#pragma unroll
  for (int q = 0; q < NQ; q++) {
    const int j = lane + 64 * q + 1;
    P[q] = SF_FAST_BIG * 2;
    sj[q] = j <= W ? S[j] : 0;
    sj1[q] = j <= W ? S[j + 1] : 0;
  }
  int f5prev = 0;  // f5[i-1] while row i is processed
  const bool want_trace = db_out && (seq % trace_stride) == 0;  // only the traceback reads f5s[]
  if (lane == 0) f5s[0] = 0;
  auto column_min = [&](const int jf) -> int {  // P of column jf, from the lane that owns it
    const int l = (jf - 1) & 63, q = (jf - 1) >> 6;
    int v = SF_LANE_READ(P[0], l);
#pragma unroll
    for (int qq = 1; qq < NQ; qq++) {
      const int vq = SF_LANE_READ(P[qq], l);
      if (q == qq) v = vq;
    }
    return v;
  };
  constexpr int PF = 8;  // rows of c in flight
  int cb[PF][NQ];
  auto load_row = [&](const int i, int(&dst)[NQ]) {
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      const int j = lane + 64 * q + 1;
      dst[q] = (i <= W - SFD_TURN - 1 && j <= W && i + SFD_TURN + 1 <= j) ? (int)(etab ? etab : X.cg)[SF_CGIDX(i, j)] : SF_INF16;
    }
  };
#pragma unroll
  for (int k = 0; k < PF; k++) load_row(1 + k, cb[k]);
#pragma unroll 1
  for (int i0 = 1; i0 <= W - SFD_TURN - 1; i0 += PF) {
#pragma unroll
    for (int k = 0; k < PF; k++) {
      const int i = i0 + k;
      if (i <= W - SFD_TURN - 1) {
        if (i >= 2) {  // f5[i-1]
          f5prev = sfd_min(f5prev, column_min(i - 1));
          if (want_trace && lane == 0) f5s[i - 1] = f5prev;
        }
        if (etab || X.cg_ext) {
#pragma unroll
          for (int q = 0; q < NQ; q++) P[q] = sfd_min(P[q], f5prev + cb[k][q]);
        } else {
        const int si = S[i], sim1 = S[i - 1];
#pragma unroll
        for (int q = 0; q < NQ; q++) {
          const int j = lane + 64 * q + 1;
          if (j <= W && i + SFD_TURN + 1 <= j) {
            const int type = j - i <= X.maxd ? tPair[si * 8 + sj[q]] : 0;
            if (type) {
              int ext;
              if (i > 1 && j < W) ext = tExt[SF_TIDX(type, sim1, sj1[q])];
              else if (i > 1) ext = X.tD5[type * 5 + sim1];
              else if (j < W) ext = X.tD3[type * 5 + sj1[q]];
              else ext = 0;
              P[q] = sfd_min(P[q], f5prev + cb[k][q] + ext + (type > 2 ? X.TAU : 0));
            }
          }
        }
        }
        load_row(i + PF, cb[k]);
      }
    }
  }
  // the last columns: every row has been seen
  for (int jf = sfd_max(W - SFD_TURN - 1, 1); jf <= W; jf++) {
    f5prev = sfd_min(f5prev, column_min(jf));
    if (want_trace && lane == 0) f5s[jf] = f5prev;
  }
  SF_WAVE_SYNC();  // f5s[] was written by lane 0, the traceback reads it from every lane
  const int over = flag[0] || f5prev < SF_FAST_OVF;
  if (lane == 0) {
    out[seq] = f5prev;
    if (over) {
      const int k = atomicAdd(ovf_cnt, 1);
      ovf_list[k] = seq;
    }
  }
  // ---- traceback for the sequences whose structure is wanted (native windows) ----
  if (db_out && !over && (seq % trace_stride) == 0) {
    int16_t *stI = stack_area;
    int16_t *stJ = stI + W + 8;
    int16_t *stM = stJ + W + 8;
    const int bad = sf_fast_traceback(X, f5s, tExt, lane, stI, stJ, stM, dbL);
    char *dst = db_out + (size_t)(seq / trace_stride) * (W + 1);
    for (int x = lane; x <= W; x += 64) dst[x] = x < W ? dbL[x] : 0;
    if (bad && lane == 0) atomicOr(status, 1);
  }
}

// @section trailing_sweep
// Trailing exterior sweep.  A fold whose structure is not wanted does not run the 5' -> 3' sweep at its end (one wave
// working for ~50 k cycles while the other three wait: 8 % of a fold's residency) and does not hand its scratch to the
// next fold either (round 2: two scratch tables per workgroup, 28 MB for the grid against 32 MB of L2, 65 GB of
// L2 <-> fabric traffic per cfg3 launch).  It runs the SAME recurrence from the other end,
//   f3[i] = min(f3[i+1], min_j c[i,j] + ExtLoop(i,j) + f3[j+1]),   f3[k > W-4] = 0,   MFE = f3[1] (= f5[W]),
// whose rows are needed in DESCENDING order — and row i of the scratch is complete as soon as diagonal W-i is, so the
// sweep trails the fill inside the same fold: the helper wave of the odd diagonal group sweeps up to SF_DEFER_ROWS rows in
// every step of the long-diagonal phase, in the slack it has there (loads issued before its own work, consumed after
// it), and only the rows the last step completes are left for the end of the fold.  One scratch per workgroup.
// State: lane l holds f3[j+1] of its columns j = l+1, l+65, ...; f3 of the row above the next one.
#define SF_DEFER 1
#define SF_DEFER_256 1  // the wide kernel trails too (four columns per lane)
#define SF_DEFER_W200 1
#define SF_DEFER_ROWS_256 3
template <int NQ>
struct SfTrail {
  int F[NQ];   // lane l: f3[j+1] for the columns j = l+1+64q (0 until row j+1 has been swept)
  int f3n;     // f3[row+1]
  int row;     // next row to sweep (descending); 0 = done / nothing pending
};
template <int NQ>
__device__ __forceinline__ void sf_trail_load(const int16_t *cgp, const int W, const int lane, const int i, int (&c)[NQ]) {
#pragma unroll
  for (int q = 0; q < NQ; q++) {
    const int j = lane + 1 + 64 * q;
    c[q] = (i >= 1 && j <= W && i + SFD_TURN + 1 <= j) ? (int)cgp[SF_CGIDX(i, j)] : SF_INF16;
  }
}
// Row i (wave-uniform; rows below 1 do not exist).  (The rows of a step do not depend on each other — a row reads f3[j+1]
// for j >= i+4 only — but computing their wave minima side by side changed nothing: the sweep costs issue slots, not
// latency.  Sweeping fewer rows per step in the first half of the split phase and more in the second, where the stamps
// show the helper waves waiting longer, measured 0.5-2 % slower.  profiles/r03/mfe_trailing_sweep.txt)
template <int NQ>
__device__ __forceinline__ void sf_trail_row(SfTrail<NQ> &T, const int lane, const int i, const int (&c)[NQ]) {
  if (i < 1) return;
  int m = c[0] + T.F[0];
#pragma unroll
  for (int q = 1; q < NQ; q++) m = sfd_min(m, c[q] + T.F[q]);
  T.f3n = sfd_min(T.f3n, sf_wave_min(m));  // f3[i] ...
  if (i >= 2) {                            // ... which is f3[j+1] of column j = i-1
    const int l = (i - 2) & 63, h = (i - 2) >> 6;
#pragma unroll
    for (int q = 0; q < NQ; q++) T.F[q] = (lane == l && h == q) ? T.f3n : T.F[q];
  }
}
template <int NQ, int NR>
__device__ __forceinline__ void sf_trail_rows(SfTrail<NQ> &T, const int lane, const int nrows, const int (&c)[NR][NQ]) {
#pragma unroll
  for (int k = 0; k < NR; k++)
    if (k < nrows) sf_trail_row<NQ>(T, lane, T.row - k, c[k]);
  T.row -= nrows;
}
// the energy / overflow record of a fold whose sweep is complete
template <int NQ>
__device__ __forceinline__ void sf_trail_result(const SfTrail<NQ> &T, const int lane, const int seq, const int over,
                                                int32_t *__restrict__ out, int *__restrict__ ovf_cnt,
                                                int *__restrict__ ovf_list) {
  if (lane == 0) {
    out[seq] = T.f3n;
    if (over || T.f3n < SF_FAST_OVF) {
      const int k = atomicAdd(ovf_cnt, 1);
      ovf_list[k] = seq;
    }
  }
}


// @section kernel_prologue
// the poison build's pattern for int16 entry x (see PZ below)
__device__ __forceinline__ int16_t sf_poison16(const int poison, const int x) {
  const int k = poison == 5 ? 1 + (int)(((uint32_t)x * 2654435761u) >> 30) : poison;
  return (int16_t)(k == 1 ? -32768 : (k == 2 ? -28000 : (k == 3 ? 0 : 32767)));
}

// MG: the merged helper (narrow kernel, W < SF_HELP_MERGE_MAXW; the launcher picks the instantiation)
// HC: every fold has its own hard constraint (cons_rows + seq * W, W characters) and / or Deigan pseudo-energies
// (sc_rows + seq * W, dcal/mol per nucleotide): the constrained native windows of `-c` / `--react` (sf_fold_constrained)
// PZ: the poison build (SCANFOLD_MFE_POISON, tests only): before every fold each LDS byte the fold has not written itself —
// the whole fML area, the rolling tables with their mirror rows and cell lists, the constraint area — is filled with an
// adversarial int16 pattern (poison = 1: -32768, 2: -28000, 3: 0, 4: 32767, 5: one of them per entry); the fold's own
// initialisation then runs as in the product.  Energies and structures must not move (tests/test_gpu_paths.py,
// tests/test_emul_kernels.py).
template <int NG, int WT, bool MG = false, bool HC = false, bool PZ = false>
__global__ __launch_bounds__(2 * NG, SF_FAST_WAVES_PER_SIMD) void sf_mfe_fast_kernel(const uint8_t *__restrict__ seqs, int n, int Wrt,
                                                             const SfDevParams *__restrict__ D,
                                                             const SfFastParams *__restrict__ F,
                                                             const SfFastRows *__restrict__ Rows,
                                                             int16_t *__restrict__ cg_all, int32_t *__restrict__ out,
                                                             int *__restrict__ ovf_cnt, int *__restrict__ ovf_list,
                                                             int trace_stride, char *__restrict__ db_out,
                                                             int *__restrict__ status, int *__restrict__ work_ctr,
                                                             const char *__restrict__ cons_rows,
                                                             const int32_t *__restrict__ sc_rows, int poison) {
  constexpr int NT = 2 * NG;
  constexpr bool FOLD = (NG == 256);  // W > 128: fML in the folded rectangle (see the file header)
  const int W = WT ? WT : Wrt;
  SF_DYN_SMEM(smem);
  const SfFastLayout Lo = sf_fast_layout(W, HC);
  SfFastCtx X;
  char *const hcC = (char *)(smem + Lo.off_hc);
  uint8_t *const hcP = (uint8_t *)hcC + (W + 2), *const hcE = hcP + (W + 2);
  int16_t *const scS = (int16_t *)(smem + Lo.off_hc + (((3 * (W + 2)) + 1) & ~1));
  X.hc.c = (HC && cons_rows) ? hcC : nullptr; X.hc.partner = hcP; X.hc.encl = hcE;
  X.sc = (HC && sc_rows) ? scS : nullptr;
  X.fML = (int16_t *)smem;
  X.CI = (int16_t *)(smem + Lo.off_ci);
  X.C1N = nullptr; X.CB = nullptr;
  X.BN = (int16_t *)(smem + Lo.off_c1n);  // spans the two areas
#ifdef SF_EMUL
  sf_emul_bn_base = (const char *)X.BN; sf_emul_ci_base = (const char *)X.CI;
#endif
  X.DMLr = nullptr;  // (the multiloop split of the enclosed cell is carried in a register: same thread, two diagonals earlier)
  int16_t *tab = (int16_t *)(smem + Lo.off_tab);
  // Pair types are 1..6, so only those rows of a [type][5][5] table exist here and the pointers are biased by
  // one row.  mismatch23 keeps its (zero) row 0: it is also indexed with the type of an enclosed cell that may
  // not pair (the sum stays >= INF through the cell's table entry).
  X.t23 = tab; X.tI = tab + 175 - 25; X.t1n = tab + 325 - 25; X.tM = tab + 475 - 25; X.tH = tab + 625 - 25;
  X.tE = tab + 775 - 25; X.cg_ext = 1;
  X.tStack = tab + 925; X.tD5 = tab + 989; X.tD3 = tab + 1029;
  uint8_t *tPair = (uint8_t *)(tab + 1069);
  X.tPair = tPair;
  uint8_t *tRPair = (uint8_t *)(tab + SF_FAST_TAB_OLD);
  int16_t *t23in = tab + SF_FAST_TAB_OLD + 32;
  X.tRPair = tRPair; X.t23in = t23in;
  (void)Lo.off_red;
  int32_t *flag = (int32_t *)(smem + Lo.off_flag);
  uint8_t *S = (uint8_t *)(smem + Lo.off_S);
  X.S = S;
  X.D = D; X.F = F; X.R = Rows; X.W = W; X.fold = FOLD; X.maxd = D->max_pair_dist;
  constexpr bool MERGE = MG && SF_HELP_MERGE && (NG == 128);
  // the split steps' generic-loop recurrence with its state unpacked (SfHU): the W = 120 instantiation
  // (every narrow instantiation: +2-3 % at W = 64 .. 117 on top of the loops by kind, W = 128 +-0; the wide kernel measured 0.5-1.5 %
  // SLOWER with it — W = 200: 55.1 -> 55.6 ms per 65 536 folds, the generic wide instantiation at W = 136 .. 250 — and stays packed)
  constexpr bool UNPK = SF_FAST_UNPACK && (NG == 128) && SF_FAST_DML2;
  constexpr bool BYKIND = SF_LOOPS_BY_KIND != 0;  // the steps of a fold as one loop per kind of step (see the loops)
  // ... and the split steps of the merged-helper instantiations once more by the chunk width of their two-cells-per-lane multiloop
  // split, the width a compile-time constant in each (W = 120: 56.2 -> 55.4 ms per 262 144 folds, W = 100 +1.4 %; the instantiation
  // without the merged helper, W = 118 .. 128, measured 1.2 % slower with it)
  constexpr bool LGLOOPS = BYKIND && MG && (NG == 128) && SF_FAST_DML2;
  // rolling-row offsets from SfFastRows (see sf_fast_cell): 1 = yes, 2 = yes + the long read batches of the generic merged-helper
  // instantiation, 0 = no (the generic wide kernel)
  constexpr int TBLK = WT > 0 ? 1 : (NG == 128 ? (MG ? 2 : 1) : 0);
  X.bn_dup = (NG == 128);
  // (W >= SF_HELP_MERGE_MAXW with the merged helper — the W = 120 instantiation: its two 128-byte cell lists do not fit the 40 960 B
  // of four workgroups per CU, so they take the place of the LDS copy of the size tables, which only the first four steps of a fold
  // read — long before the first list is built — and which is copied again for every fold)
  uint8_t *const cell_list = (MG && W >= SF_HELP_MERGE_MAXW) ? (uint8_t *)((int16_t *)(smem + Lo.off_tab) + 1069 + 32 + 1)
                                                            : (uint8_t *)(smem + Lo.off_list);
  X.TAU = D->P.TerminalAU; X.MLbase = D->P.MLbase; X.MLclosing = D->P.MLclosing; X.MLintern = D->P.MLintern[1];
  // exterior pass aliases (the rolling CI area is dead by then)
  int32_t *f5s = (int32_t *)(smem + Lo.off_ci);
  int16_t *tExt = (int16_t *)(smem + Lo.off_ci + (((W + 1) * 4 + 3) & ~3));

  const int tid = threadIdx.x;
  X.cg = cg_all + (size_t)sf_fast_scratch_slot(blockIdx.x, gridDim.x) * SF_CG_ENTRIES(W);  // c + ExtLoop by (row i, column j), triangular
  // parameter tables -> LDS, once per workgroup
  for (int x = tid; x < 175; x += NT) {
    tab[x] = F->mm23[x];
    if (x < 150) {
      tab[175 + x] = F->mmI[25 + x]; tab[325 + x] = F->mm1n[25 + x]; tab[475 + x] = F->mmM[25 + x];
      tab[625 + x] = F->mmH[25 + x]; tab[775 + x] = F->mmExt[25 + x];
    }
  }
  for (int x = tid; x < 175; x += NT) t23in[x] = F->mm23in[x];
  for (int x = tid; x < 64; x += NT) { tab[925 + x] = F->stackT[x]; tPair[x] = F->pair[x]; tRPair[x] = F->rpair[x]; }
  for (int x = tid; x < 40; x += NT) { tab[989 + x] = F->d5[x]; tab[1029 + x] = F->d3[x]; }

  {
    int16_t *uni = tab + 1069 + 32 + 1;  // after the 64-byte pair table, at an even index (read as 32-bit pairs)
    X.uNIN = uni;
    for (int x = tid; x < 128; x += NT) uni[x] = F->uni[x];
  }
  // group and centre-based mapping inside the group: v = (tg + OFF) mod NG, cell i = v - d/2
  const int grp = SF_WAVE_UNIFORM(tid / NG);  // a wave lies in one group: keep d, row slots, loop limits scalar
  const int tg = tid - grp * NG;
  // CENTRE: the lane (of the group) around which the cells of the long diagonals gather.  NG = 128: lane 32, so the
  // last 62 cells of a diagonal lie in the group's first wave; NG = 256: lane 128, so the last 126 lie in its two
  // middle waves — the other waves of the group then work as helpers (see `split` below).
  constexpr int CENTRE = (NG == 256) ? 128 : 32;
  constexpr bool SHARE = SF_DML_HELPER_BIAS(NG, WT) < 10000;  // the helper waves take part of the multiloop split
  constexpr bool DML2 = (NG == 128) && SF_FAST_DML2;    // split steps: the main wave splits two cells per lane (sf_fast_dml2)
  const int OFFs = ((W + 1) >> 1) - CENTRE;                  // signed: v = tg + OFFs (mod NG)
  const int OFF = (NG > 64) ? (OFFs + NG) & (NG - 1) : 0;
  const int v = (tg + OFF) & (NG - 1);
  // first even diagonal from which the cells of both groups lie in the main lanes: tg in [0, 63] (NG = 128, W >= 64
  // so that nothing wraps) or [64, 191] (NG = 256)
  constexpr int MAIN_LO = (NG == 256) ? 64 : 0, MAIN_HI = (NG == 256) ? 191 : 63;
  const bool can_split = SF_FAST_SPLIT && ((NG == 128 && W >= 64) || (NG == 256 && SF_FAST_SPLIT_256));
  const int split_d0 = can_split ? ((sfd_max(sfd_max(SFD_MAXLOOP + 6, 2 * (MAIN_LO - 1 + OFFs)), 2 * (W - OFFs - MAIN_HI)) + 1) & ~1) : 1 << 30;
  // NARROW (wide kernel, round 4): from the even diagonal where the cells of a diagonal fit ONE wave (lanes 96..159 of the group,
  // <= 62 cells: d0 >= 138 at W = 200) the two main + two helper waves of a group are twice what the work needs — both mains
  // run the whole cell code on <= 31 cells each.  The cells move 32 lanes down into the group's wave 1 (main), wave 0 mirrors it
  // (helper), waves 2 and 3 only keep the barriers; the per-thread state that follows a centre (the recurrence's 14 registers,
  // the enclosed cell's split minimum, the even group's neighbour term) changes lanes ONCE, through the unused ends of 32 rows of
  // the interleaved ring (a row holds <= 94 cells by then: dwords 100..163 of each row are free; needs W - 4 >= 164).
  constexpr bool NARROW = (NG == 256) && SF_FAST_NARROW_256;
  const int narrow_d0 = (NARROW && can_split && W >= 168)
                            ? sfd_max(split_d0, (sfd_max(2 * (96 - 1 + OFFs), 2 * (W - OFFs - 159)) + 1) & ~1) : 1 << 30;

  // trailing exterior sweep (see SfTrail): wave 3 (the helper of the odd group) sweeps the rows of this fold's scratch
  // that are already complete
  // (round 2, with the sweep deferred to the next fold: the generic wide kernel gained 4-5 % — W = 136 / 160 / 256:
  // 1.35 -> 1.43 M, 1.09 -> 1.15 M, 327 -> 342 k; in the W = 200 instantiation the sweeper's state spills ~50 B/lane:
  // with four rows in flight it loses what it saves, with two it is +0.8 %, 894 -> 900 k)
  constexpr bool DEFER = SF_DEFER && (NG == 128 || (SF_DEFER_256 && (WT != 200 || SF_DEFER_W200)));
  constexpr int NQ = NG / 64;  // columns per lane of the sweeper wave
  const bool sweeper = DEFER && SF_WAVE_UNIFORM(tid >> 6) == (NG == 128 ? 3 : 7);  // a helper wave of the odd group
  const int n_split_steps = split_d0 < W ? (W - split_d0 + 1) / 2 : 0;
  // rows per step: what finishes the sweep inside the split phase, but no more than DROWS in flight (their loads live
  // in registers across the wave's own work: 4 x 2 columns in the narrow kernel, 2 x 4 in the wide one); what is
  // left over is swept at the end of the fold
  constexpr int DROWS = (NG == 128 || WT != 200) ? 4 : SF_DEFER_ROWS_256;
  const int defer_need = n_split_steps > 0 ? (W - SFD_TURN - 1 + n_split_steps - 1) / n_split_steps : 0;
  const int defer_rows = sfd_min(defer_need, DROWS);
  const bool defer_on = DEFER && n_split_steps >= 8 && defer_need <= 4;
  SfTrail<NQ> T;
#pragma unroll
  for (int q = 0; q < NQ; q++) T.F[q] = 0;
  T.f3n = 0; T.row = 0;

  // Folds are handed out dynamically: a workgroup's first fold is its block index, every further one comes from a
  // device-wide counter (zeroed by the host; grid size + its value).  The folds cost the same number of instructions, but
  // the workgroups do not run at the same speed — with a static deal (fold b, b + grid, ...) the slowest of 1024
  // took 897 ms for its 2 947 folds at cfg3 while the mean took 772 ms, and the launch lasts as long as the slowest.
  // The index of the NEXT fold is requested at the start of a fold (one atomic by thread 0), parked in LDS two steps
  // later, and read by everybody one step after that: its latency is never waited for.
  int *const next_slot = (int *)(smem + Lo.off_next);
// @section fold_prologue
  int seq = blockIdx.x;
  while (seq < n) {
    const uint8_t *src = seqs + (size_t)seq * W;
    // the per-thread addresses of this prologue are recomputed per fold: hoisted out of the fold loop they live in VGPRs through
    // the whole fold, i.e. in spill slots (13 dwords per lane = 13.6 kB per workgroup of private memory competing with the
    // scratch tables for the L2)
    // (round 4 left the generic wide instantiation out — 107 -> 16 spilled registers with the pins but 1.3-2.5 % slower; with the
    // loops by kind of round 5 the speed is the same either way (W = 136 .. 250: +-0.3 %) and the pins take its private segment
    // from 504 to 160 B, its scratch instructions from 232 to 104: every instantiation has them now)
    constexpr bool PINF = true;
    int tidf = tid;
    if (PINF) SF_PIN(tidf);
    __syncthreads();
    if (PZ) {
      for (int x = tid; x < Lo.off_tab / 2; x += NT) ((int16_t *)smem)[x] = sf_poison16(poison, x);
      if (HC) for (int x = tid; x < (Lo.total - Lo.off_hc) / 2; x += NT) ((int16_t *)(smem + Lo.off_hc))[x] = sf_poison16(poison, x);
      __syncthreads();
    }
    for (int x = tidf; x < W; x += NT) S[x + 1] = sf_encode_nt(src[x]);
    if (MG && W >= SF_HELP_MERGE_MAXW) for (int x = tidf; x < 128; x += NT) ((int16_t *)(smem + Lo.off_tab) + 1069 + 32 + 1)[x] = F->uni[x];
    // The rolling tables (with their mirror rows and the cell lists) start every fold as "no structure".  The straight-line
    // cell code of the short diagonals reads candidates of loop sizes that do not exist yet — rows no diagonal of this fold
    // has written, up to 23 words past the end of a row — and charges them 32 767; the sum only stays "none" if what it
    // finds is energy-sized.  Until round 4 the tables were initialised once per workgroup and later folds found the
    // previous fold's energies there — as good, unless that fold had left the int16 range.  24 dword stores per thread
    // and fold (0.05 % of a fold) make a fold's result a function of its own sequence only.
    for (int x = tidf; x < (Lo.off_tab - Lo.off_ci) / 4; x += NT) ((uint32_t *)X.CI)[x] = sf_pk(SF_INF16, SF_INF16);
    int fetched = 0;
    if (tid == 0) {
      S[0] = 0; S[W + 1] = 0; flag[0] = 0;
      fetched = (int)gridDim.x + atomicAdd(work_ctr, 1);
    }
    __syncthreads();
    if (HC) {
      // the fold's constraint: copied by everybody, parsed by one thread (unbalanced brackets: reported, nothing folded)
      if (cons_rows)
        for (int x = tid; x < W; x += NT) hcC[x + 1] = cons_rows[(size_t)seq * W + x];
      if (sc_rows)
        for (int x = tid; x < W; x += NT) scS[x + 1] = (int16_t)sfd_max(sfd_min(sc_rows[(size_t)seq * W + x], 30000), -30000);
      __syncthreads();
      if (cons_rows && tid == 0) flag[0] = sf_hc_parse8(W, hcC, hcP, hcE) ? 2 : 0;
      __syncthreads();
      if (cons_rows && flag[0] == 2) {
        if (tid == 0) { atomicOr(status, 2); out[seq] = 0; *next_slot = fetched; }  // (thread 0 holds the next fold's index)
        __syncthreads();
        seq = *next_slot;
        continue;
      }
    }
    int next_seq = n;  // (set in the third step; W >= 16 has at least five)
    // a fold whose structure is wanted keeps the 5' -> 3' sweep at its end (the traceback reads f5[])
    const bool trail_this = defer_on && !(db_out && (seq % trace_stride) == 0);
    if (sweeper) {
#pragma unroll
      for (int q = 0; q < NQ; q++) T.F[q] = 0;
      T.f3n = 0; T.row = trail_this ? W - SFD_TURN - 1 : 0;
    }
    int ovf = 0;
    uint32_t H[14];  // packed int16 pairs, see HGET/HSET
    int dprev = SF_INF16;   // multiloop split of this thread's previous cell (diagonals 2, 3: none)
    int fnb = SF_FAST_BIG;  // even group: min of the two fML neighbours of the next cell, + MLbase
#pragma unroll
    for (int k = 0; k < 14; k++) H[k] = (uint32_t)SF_INF16 | ((uint32_t)SF_INF16 << 16);

    // this thread's diagonal in the step that starts at the even diagonal d0 is d0 + grp
    int slot2 = (SFD_TURN + 1 + grp - 2) % SF_FAST_NR, slotd = (SFD_TURN + 1 + grp) % SF_FAST_NR;
// @section cell_list
    // Merged helper: the cells of the diagonals dd0 and dd0+1 that can pair, as a helper wave sees them (lane -> mirror
    // cell; i is the same on both diagonals since (dd0 + 1) >> 1 == dd0 >> 1).  Entry = i, +128 for the odd diagonal,
    // even diagonal first; byte 127 of the buffer = the number of entries (<= 124).  Built by wave 3 one step ahead
    // (two buffers, by step parity), read by wave 1 (entries 0..63) and wave 3 (entries 64.., rarely any).
    auto build_list = [&](const int dd0) {
      const int lane = tid & 63;
      const int d1 = dd0 + 1;
      const int iH = ((v - 64) & (NG - 1)) - (dd0 >> 1);
      const bool vA = (dd0 < W) && (iH >= 1) && (iH + dd0 <= W), vB = (d1 < W) && (iH >= 1) && (iH + d1 <= W);
      const int tA = (vA && dd0 <= X.maxd) ? X.tPair[S[iH] * 8 + S[iH + dd0]] : 0;
      const int tB = (vB && d1 <= X.maxd) ? X.tPair[S[iH] * 8 + S[iH + d1]] : 0;
      const unsigned long long mA = __ballot(tA != 0), mB = __ballot(tB != 0);
      const int cA = __popcll(mA);
      uint8_t *list = cell_list + (((dd0 >> 1) & 1) << 7);
      const unsigned long long below = (1ull << lane) - 1ull;
      if (tA) list[__popcll(mA & below)] = (uint8_t)iH;
      if (tB) list[cA + __popcll(mB & below)] = (uint8_t)(iH | 128);
      if (lane == 0) list[127] = (uint8_t)(cA + __popcll(mB));
    };
// @section step_control
    // One step = the diagonals d0 (even group) and d0 + 1 (odd group).  The body is instantiated once per kind of step (PH).
    SfHU HU;
    auto step = [&](const int d0, auto phase_tag) {
      // PH: the kind of step this instantiation of the body is for (BYKIND: one loop per kind, see below) — 3: d0 < 12 (size-tested
      // code), 4: 12 <= d0 < 36 (guarded code), 1: the unsplit steps from d0 = 36 on, 2: the split steps, 5: the wide kernel's split
      // steps from the diagonal on where a diagonal's cells fit one wave (NARROW); 0: any (one loop)
      constexpr int PH = decltype(phase_tag)::value;
      constexpr bool P2 = (PH == 2 || PH >= 5);  // a split step (5: the wide kernel's NARROW phase, 6 / 7: the narrow kernel's split steps
                                                 // whose two-cells-per-lane multiloop split runs in chunks of 16 / 8 lanes — loops of their own as well)
      constexpr int LGC = PH == 6 ? 4 : (PH == 7 ? 3 : (PH == 2 && LGLOOPS ? 5 : 0));
      constexpr bool DO_G = (PH == 0 || PH == 3), DO_CH = (PH == 0 || PH == 4);
      const int d = d0 + grp;
      if (d0 == SFD_TURN + 1 + 2 && tid == 0) *next_slot = fetched;            // the barriers of this step publish it
      if (d0 == SFD_TURN + 1 + 4) next_seq = SF_WAVE_UNIFORM(*next_slot);
      // Long diagonals (d0 >= split_d0): the cells of a group fit its first wave, so the second wave — which would
      // idle — mirrors it (same lane -> same cell) and takes the special loops and the bulge / 1xn minima of
      // those cells, while the first does the generic-loop recurrence and the multiloop split; the partial
      // result crosses in LDS (in the C1N entry the cell will publish, unread until the next step) at a barrier,
      // then the first wave finishes the cell.  The dependent chain of such a step is ~45 % shorter.
      const bool split = P2 || (!BYKIND && d0 >= split_d0);  // (BYKIND: the split steps are the last loop)
      // helper lanes mirror a main lane 64 away: NG = 128: wave 1 -> wave 0; NG = 256: wave 0 -> wave 1, wave 3 -> wave 2
      const bool narrow = NARROW && (PH == 5 || (PH == 0 && d0 >= narrow_d0));
      if (NARROW && d0 == narrow_d0) {
        // the one-time move of the centres' state: old owners tg = 96..159 -> new owners tg - 32 = 64..127
        uint32_t *const xa = (uint32_t *)X.BN;  // dword = cell of the interleaved ring; row r, cell 100 + lane
        const int RWc = W - 4;
        if (tg >= 96 && tg < 160) {
#pragma unroll
          for (int k = 0; k < 14; k++) xa[(grp * 16 + k) * RWc + 100 + (tg - 96)] = H[k];
          xa[(grp * 16 + 14) * RWc + 100 + (tg - 96)] = (uint32_t)dprev;
          xa[(grp * 16 + 15) * RWc + 100 + (tg - 96)] = (uint32_t)fnb;
        }
        __syncthreads();
        if (tg >= 64 && tg < 128) {
#pragma unroll
          for (int k = 0; k < 14; k++) H[k] = xa[(grp * 16 + k) * RWc + 100 + (tg - 64)];
          dprev = (int)xa[(grp * 16 + 14) * RWc + 100 + (tg - 64)];
          fnb = (int)xa[(grp * 16 + 15) * RWc + 100 + (tg - 64)];
        }
        __syncthreads();
      }
      const bool helper = split && (NG == 256 ? (narrow ? tg < 64 : (tg < 64 || tg >= 192)) : tg >= 64);
      const int vh = NG == 256 ? (narrow ? v + 96 : (tg < 64 ? v + 64 : v - 64)) : v - 64;
      // (narrow: lanes 64..127 take the cells of lanes 96..159, lanes 0..63 mirror them; lanes >= 128 have no cell)
      const int vm = (NARROW && narrow) ? ((v + 32) & (NG - 1)) : v;
      // (narrow: waves 2 and 3 of the group would idle — they mirror the main wave too and take most of the multiloop split, the
      // part of a cell that keeps growing with d: 40 % of the terms each, the main wave the first fifth; their partial minima
      // cross in the dwords the centres' state moved through)
      // (the W = 200 instantiation only: +4 % there; the generic wide instantiation measured 2-6 % SLOWER with it at W = 168 / 256 and
      // leaves those waves without cells)
      constexpr bool DML4 = NARROW && (WT == 200);
      const bool dmlw = DML4 && narrow && tg >= 128;
      const int vd = tg >= 192 ? v - 96 : v - 32;
      const int i = (NARROW && !DML4 && narrow && tg >= 128) ? -1
                    : ((dmlw ? (vd & (NG - 1)) : (helper ? (vh & (NG - 1)) : vm)) - (d >> 1));
      // split step: the multiloop split (d-8 terms, the part of a cell that grows with d) is shared with the helper
      // wave: terms m >= dml_cut are the helper's (wave-uniform; both waves of a cell compute the same cut)
      const int dml_terms = d - 2 * SFD_TURN - 2;  // m = 4 .. d-5
      const int dml_cut = d - SFD_TURN - 1 - sfd_max(sfd_min(dml_terms / 2 - SF_DML_HELPER_BIAS(NG, WT), dml_terms), 0);
      const bool valid = (d < W) && (i >= 1) && (i + d <= W);
      int fpart = SF_FAST_BIG;
      int dec = SF_FAST_BIG, eh = SF_FAST_BIG, e0 = SF_FAST_BIG;
      SfPub pub;
      pub.a = pub.b = 0; pub.tau = 0; pub.type = 0;
      // trailing sweep: the rows i >= W-d0+1 are complete (diagonals < d0 are); this step's rows are requested now, used
      // after the wave's own work
      int dc[DROWS][NQ];
      const bool sweep_now = defer_on && sweeper && split && T.row > 0;
      const int sweep_rows = sweep_now ? sfd_max(sfd_min(defer_rows, T.row - (W - d0)), 0) : 0;
      if (sweep_now) {
#pragma unroll
        for (int k = 0; k < DROWS; k++)
          if (k < sweep_rows) sf_trail_load<NQ>(X.cg, W, tid & 63, T.row - k, dc[k]);
      }
      if (__ballot(valid)) {
        if (DO_G && d0 < 8) sf_fast_cell<true, WT, SF_SEC_ALL, false, FOLD, 1>(X, d, i, valid, slot2, slotd, H, HU, ovf, grp == 0, fnb, fpart, dec, eh, e0, dprev, pub);
        else if (DO_G && d0 < SF_FAST_TINY_D0) sf_fast_cell<true, WT, SF_SEC_ALL, false, FOLD, 5>(X, d, i, valid, slot2, slotd, H, HU, ovf, grp == 0, fnb, fpart, dec, eh, e0, dprev, pub);
        else if (DO_CH && d0 < SF_FAST_CHUNK_D0) sf_fast_cell<false, WT, SF_SEC_ALL, true, FOLD, SFD_MAXLOOP, TBLK>(X, d, i, valid, slot2, slotd, H, HU, ovf, grp == 0, fnb, fpart, dec, eh, e0, dprev, pub);
        else if (!P2 && !split) sf_fast_cell<false, WT, SF_SEC_ALL, false, FOLD, SFD_MAXLOOP, TBLK, false, UNPK && PH == 1>(X, d, i, valid, slot2, slotd, H, HU, ovf, grp == 0, fnb, fpart, dec, eh, e0, dprev, pub);
        else if (DML4 && narrow) {
          // terms m = 4 .. d-5: main [4, cA), wave 2 [cA, cB), wave 3 [cB, d-5]
          const int nmain = dml_terms / 5, cA = SFD_TURN + 1 + nmain, cB = cA + (dml_terms - nmain + 1) / 2;
          uint32_t *const xa = (uint32_t *)X.BN;
          if (dmlw) {
            const bool wB = tg >= 192;
            sf_fast_cell<false, WT, SF_SEC_DML, false, FOLD, SFD_MAXLOOP, TBLK>(X, d, i, valid, slot2, slotd, H, HU, ovf, false, fnb, fpart, dec, eh, e0, dprev, pub, wB ? cB : cA, wB ? (1 << 20) : cB - 1);
            if (valid) xa[(grp * 16 + (wB ? 1 : 0)) * (W - 4) + 100 + (tg & 63)] = (uint32_t)sfd_min(dec, 32000);
          } else if (!helper) {
            sf_fast_cell<false, WT, SF_SEC_P1 | SF_SEC_C0 | SF_SEC_DML | SF_SEC_PRE, false, FOLD, SFD_MAXLOOP, TBLK>(X, d, i, valid, slot2, slotd, H, HU, ovf, grp == 0, fnb, fpart, dec, eh, e0, dprev, pub, SFD_TURN + 1, cA - 1);
          } else {
            sf_fast_cell<false, WT, SF_SEC_HELP, false, FOLD, SFD_MAXLOOP, TBLK>(X, d, i, valid, slot2, slotd, H, HU, ovf, grp == 0, fnb, fpart, dec, eh, e0, dprev, pub);
            if (valid) X.BN[2 * (slotd * (W - 4) + i - 1)] = (int16_t)sfd_min(eh, 32000);
          }
        }
        else if (!helper) {
          if (SHARE) sf_fast_cell<false, WT, SF_SEC_P1 | SF_SEC_C0 | SF_SEC_DML | SF_SEC_PRE, false, FOLD, SFD_MAXLOOP, TBLK>(X, d, i, valid, slot2, slotd, H, HU, ovf, grp == 0, fnb, fpart, dec, eh, e0, dprev, pub, SFD_TURN + 1, dml_cut - 1);
          else if (DML2) {
            dec = sf_fast_dml2<WT, LGC>(X, d, tid & 63, i, valid);
            sf_fast_cell<false, WT, SF_SEC_P1 | SF_SEC_C0 | SF_SEC_PRE, false, FOLD, SFD_MAXLOOP, TBLK, false, UNPK && P2>(X, d, i, valid, slot2, slotd, H, HU, ovf, grp == 0, fnb, fpart, dec, eh, e0, dprev, pub);
          } else sf_fast_cell<false, WT, SF_SEC_P1 | SF_SEC_C0 | SF_SEC_DML | SF_SEC_PRE, false, FOLD, SFD_MAXLOOP, TBLK>(X, d, i, valid, slot2, slotd, H, HU, ovf, grp == 0, fnb, fpart, dec, eh, e0, dprev, pub);
        } else if (MERGE) {
          // Merged helper (W <= 128).  ONE helper wave serves both diagonals of the step: it works on the list of the
          // cells of d0 and d0+1 that can pair (build_list above; 3 of 8 cells, so ordinary sequences fit wave 1's 64
          // lanes; entries 64.. go to wave 3).  A lane's cell may lie on either diagonal: the rolling rows of d0+1 are
          // the rows of d0 plus one (the extra table row makes that true at the ring's seam), so the lane-dependent
          // part is a pointer offset and everything scalar stays scalar.
          const int lane = tid & 63;
          const uint8_t *list = cell_list + (((d0 >> 1) & 1) << 7);
          const int cT = SF_WAVE_UNIFORM((int)list[127]);
          const int e = (grp << 6) + lane;  // wave 1: entries 0..63, wave 3: 64..127
          if (SF_WAVE_UNIFORM(cT > (grp << 6))) {
            const bool vC = e < cT;
            const int ent = vC ? list[e] : 1;
            const int g = ent >> 7, iC = ent & 127;
            const int s2 = slot2 - grp < 0 ? slot2 - grp + SF_FAST_NR : slot2 - grp;      // the even group's rows
            const int sd0 = slotd - grp < 0 ? slotd - grp + SF_FAST_NR : slotd - grp;
            const int sd1 = sd0 + 1 >= SF_FAST_NR ? 0 : sd0 + 1;
            SfFastCtx Xh = X;
            Xh.BN = X.BN + 2 * g * (W - 4);
            sf_fast_cell<false, WT, SF_SEC_HELP, false, FOLD, SFD_MAXLOOP, TBLK, true>(Xh, d0 + g, iC, vC, s2, sd0, H, HU, ovf, false, fnb, fpart, dec, eh, e0, dprev, pub);
            if (vC) X.BN[2 * ((g ? sd1 : sd0) * (W - 4) + iC - 1)] = (int16_t)sfd_min(eh, 32000);
          }
        } else {
          // the helper's results -> the cell's OWN entries of the row it is about to publish (unread until the next
          // step, and rewritten by nobody but the cell's own lane): eh in the CB half of its word, its part of the
          // multiloop split in its CI entry
          if (SHARE) {
            sf_fast_cell<false, WT, SF_SEC_HELP | SF_SEC_DML, false, FOLD, SFD_MAXLOOP, TBLK>(X, d, i, valid, slot2, slotd, H, HU, ovf, grp == 0, fnb, fpart, dec, eh, e0, dprev, pub, dml_cut);
            if (valid) {
              X.BN[2 * (slotd * (W - 4) + i - 1)] = (int16_t)sfd_min(eh, 32000);
              X.CI[slotd * (W - 4) + i - 1] = (int16_t)sfd_min(dec, 32000);
            }
          } else {
            sf_fast_cell<false, WT, SF_SEC_HELP, false, FOLD, SFD_MAXLOOP, TBLK>(X, d, i, valid, slot2, slotd, H, HU, ovf, grp == 0, fnb, fpart, dec, eh, e0, dprev, pub);
            if (valid) X.BN[2 * (slotd * (W - 4) + i - 1)] = (int16_t)sfd_min(eh, 32000);
          }
        }
      }
      if (sweep_now && sweep_rows > 0) sf_trail_rows<NQ, DROWS>(T, tid & 63, sweep_rows, dc);
      // merged helper: wave 3 lists the next step's cells (the barriers of this step order the list before its readers)
      if (MERGE && grp == 1 && tg >= 64 && d0 + 2 >= split_d0 && d0 + 2 < W) build_list(d0 + 2);
// @section exchange
      if (split) {
        __syncthreads();
        if (!helper && !dmlw && __ballot(valid)) {
          if (valid) {
            eh = X.BN[2 * (slotd * (W - 4) + i - 1)];
            if (DML4 && narrow) {
              const uint32_t *const xa = (const uint32_t *)X.BN;
              dec = sfd_min(dec, sfd_min((int)xa[(grp * 16) * (W - 4) + 100 + (tg & 63)], (int)xa[(grp * 16 + 1) * (W - 4) + 100 + (tg & 63)]));
            } else if (SHARE) dec = sfd_min(dec, (int)X.CI[slotd * (W - 4) + i - 1]);
          }
          sf_fast_cell<false, WT, SF_SEC_FIN | SF_SEC_POST, false, FOLD, SFD_MAXLOOP, TBLK>(X, d, i, valid, slot2, slotd, H, HU, ovf, grp == 0, fnb, fpart, dec, eh, e0, dprev, pub);
        }
      }
      __syncthreads();
// @section fml_fixup
      // fML on the odd diagonal d0+1 = its provisional value (written by the odd group) min the two neighbours on
      // the even diagonal d0, now final.  The EVEN group does this: the only early reader of fML[d0+1] is the even
      // group's next cell (i-1, j+1), which needs the cells i-1 and i of it — so every even lane finishes BOTH
      // (storing only its own, i) and keeps their minimum in a register.  No second barrier: nothing else reads
      // these entries before several later barriers (the multiloop split of d reads spans <= d-5).
      // (Round 5 folded this into the even group's NEXT finish — reads issued before the multiloop split, used after it, no separate
      // phase: bit-exact and 3.5 % SLOWER at every width.  The step stamps say why: with the even group no longer trailing the odd one
      // by these ~800 cycles, all four waves of a workgroup issue the same LDS bursts at the same time and every phase of every wave
      // got ~8 % longer — profiles/r05/mfe_step_profile_lazy_fixup_rejected.txt against mfe_step_profile_start_of_round.txt.)
      if (grp == 0 && valid && !helper && !dmlw) {
        // (branch-free: the five entries the two cells x = i and x = i - 1 need are read at once — a cell that does not exist reads
        // a clamped or neighbouring entry and is dropped by a select — one LDS round trip where the two tested cells took two, in a
        // phase in which every other wave of the workgroup waits)
        const int d1 = d0 + 1;
        const int16_t *pd = X.fML + FBASE(d1), *pe = X.fML + FBASE(d0);
        const bool ex0 = i + d1 <= W, ex1 = i >= 2;  // cell (i, i + d1) / (i - 1, i + d0) exists
        const int ia = i - 1, ib = ex1 ? i - 2 : 0;
        const int p0 = pd[ia], p1 = pd[ib], q0 = pe[i], q1 = pe[ia], q2 = pe[ib];
        const int v0 = sfd_min(p0, sfd_min(q0, q1) + X.MLbase), v1 = sfd_min(p1, sfd_min(q1, q2) + X.MLbase);
        if ((ex0 && v0 < SF_FAST_OVF) || (ex1 && v1 < SF_FAST_OVF)) ovf = 1;
        const int g0 = (ex0 && v0 <= SF_FAST_THRESH) ? v0 : SF_INF16, g1 = (ex1 && v1 <= SF_FAST_THRESH) ? v1 : SF_INF16;
        if (ex0) X.fML[FBASE(d1) + ia] = (int16_t)g0;
        fnb = sfd_min(g0, g1) + X.MLbase;
      }
      slot2 += 2; if (slot2 >= SF_FAST_NR) slot2 -= SF_FAST_NR;
      slotd += 2; if (slotd >= SF_FAST_NR) slotd -= SF_FAST_NR;
    };
    {
      int d0 = SFD_TURN + 1;
      // BYKIND: the steps of a fold as four loops, one per kind of step — d0 < 12 (size-tested code), 12 <= d0 < 36 (guarded code),
      // the unsplit steps from d0 = 36 on, the split steps — instead of one loop with the kinds as branches: the same code, but every
      // loop gets its own register allocation and schedule (W = 120, per 262 144 folds: one loop 59.0 ms; two loops, the split steps
      // apart, 59.8 packed / 57.5 with the split steps' state unpacked; three loops 56.6; four 56.2).
      // UNPK: the state is unpacked at d0 = 36 — the first diagonal pair on which every loop size exists (the guarded code before it
      // needs the saturating packed adds; unpacked with v_add_i16 clamp it measured slower: 58.8 ms).
      if constexpr (BYKIND) {
        for (; d0 < SF_FAST_TINY_D0 && d0 < W; d0 += 2) step(d0, std::integral_constant<int, 3>{});
        for (; d0 < SF_FAST_CHUNK_D0 && d0 < W; d0 += 2) step(d0, std::integral_constant<int, 4>{});
        if constexpr (UNPK) {
#pragma unroll
          for (int x = 0; x < 27; x++) HU.v[x] = (short)((x & 1) ? (H[x >> 1] >> 16) : (H[x >> 1] & 0xffffu));
        }
        for (; d0 < split_d0 && d0 < W; d0 += 2) step(d0, std::integral_constant<int, 1>{});
        if constexpr (LGLOOPS) {
          // (chunks of 32 lanes while a diagonal has more than 30 cells, 16 down to 15 cells, then 8: d0 >= W - 30 / W - 14)
          for (; d0 < W - 30 && d0 < W; d0 += 2) step(d0, std::integral_constant<int, 2>{});
          for (; d0 < W - 14 && d0 < W; d0 += 2) step(d0, std::integral_constant<int, 6>{});
          for (; d0 < W; d0 += 2) step(d0, std::integral_constant<int, 7>{});
        }
        for (; d0 < (NARROW ? narrow_d0 : W) && d0 < W; d0 += 2) step(d0, std::integral_constant<int, 2>{});
        if constexpr (NARROW)
          for (; d0 < W; d0 += 2) step(d0, std::integral_constant<int, 5>{});
      } else {
        for (; d0 < W; d0 += 2) step(d0, std::integral_constant<int, 0>{});
      }
    }

// @section fold_epilogue
    // ---- exterior loop f5[j] = min(f5[j-1], min_i f5[i-1] + c[i,j] + ExtLoop(i,j)) : wave 0 only ----
    // (sf_fast_exterior: lane = column, rows of c + ExtLoop stream from the scratch a few rows ahead)
    if (ovf) flag[0] = 1;
    if (trail_this) {
      // the rows the last steps completed, the energy / overflow record
      // (issuing these rows' loads before the barrier measured 3 % slower at W = 120)
      __syncthreads();  // every thread's overflow flag is in
      if (sweeper) {
        while (T.row >= 1) {
          int tc[4][NQ];
#pragma unroll
          for (int k = 0; k < 4; k++) sf_trail_load<NQ>(X.cg, W, tid & 63, T.row - k, tc[k]);
          sf_trail_rows<NQ, 4>(T, tid & 63, sfd_min(4, T.row), tc);
        }
        sf_trail_result<NQ>(T, tid & 63, seq, flag[0], out, ovf_cnt, ovf_list);
      }
    } else {
    // (as tidf above: one fold in r + 1 comes here, and what its inlined sweep and traceback derive from the thread index need
    // not live through the others: with this pin the narrow instantiations and W = 200 have no or 2 spilled registers — 45-53 and
    // 12 before — and run 2-3.5 % faster (W = 64 / 128 / 200).  Not at W = 120: the allocation it gets there is 1.8 % slower —
    // 59.8 against 58.8 ms per 262 144 folds, same hot blocks with five more s_waitcnt — and the first pin alone already leaves
    // it 6 cold spill slots; profiles/r04/mfe_scratch_placement.txt)
    int tidn = tid;
    if (PINF && WT != 120) SF_PIN(tidn);
    for (int x = tidn; x < 200; x += NT) tExt[x] = F->mmExt[x];
    __syncthreads();
    // (the scratch already holds c + ExtLoop: the sweep is one add and one min per cell; the mismatchExt table
    // above is for the traceback)
    int16_t *etab = nullptr;
    if (tid < 64)
      sf_fast_exterior<NG / 64>(X, W, tidn, seq, f5s, tExt, etab, flag, (int16_t *)(smem + Lo.off_cb),
                      (char *)(smem + Lo.off_cb + ((3 * (W + 8) * 2 + 3) & ~3)), out, ovf_cnt, ovf_list, trace_stride,
                      db_out, status);
    }
    seq = next_seq;
  }
#undef FBASE
#undef FLEN
}

template <int NG, int WT, bool MG = false, bool HC = false, bool PZ = false>
static inline hipError_t sf_fast_configure_one() {
  return hipFuncSetAttribute((const void *)sf_mfe_fast_kernel<NG, WT, MG, HC, PZ>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}
static inline hipError_t sf_fast_configure() {
  hipError_t e;
  if ((e = sf_fast_configure_one<128, 0>()) != hipSuccess) return e;
  if ((e = sf_fast_configure_one<128, 0, true>()) != hipSuccess) return e;
  if ((e = sf_fast_configure_one<128, 120, SF_MG120 != 0>()) != hipSuccess) return e;
  if ((e = sf_fast_configure_one<256, 200>()) != hipSuccess) return e;
  if ((e = sf_fast_configure_one<256, 0>()) != hipSuccess) return e;
  // the instantiations for constrained folds (generic widths only)
  if ((e = sf_fast_configure_one<128, 0, false, true>()) != hipSuccess) return e;
  if ((e = sf_fast_configure_one<128, 0, true, true>()) != hipSuccess) return e;
  if ((e = sf_fast_configure_one<256, 0, false, true>()) != hipSuccess) return e;
  // the poison builds (SCANFOLD_MFE_POISON: tests only)
  if ((e = sf_fast_configure_one<128, 0, false, false, true>()) != hipSuccess) return e;
  if ((e = sf_fast_configure_one<128, 0, true, false, true>()) != hipSuccess) return e;
  if ((e = sf_fast_configure_one<128, 120, SF_MG120 != 0, false, true>()) != hipSuccess) return e;
  if ((e = sf_fast_configure_one<256, 200, false, false, true>()) != hipSuccess) return e;
  return sf_fast_configure_one<256, 0, false, false, true>();
}

// grid / LDS / scratch for n folds of W nt on a chip with n_cu CUs
static inline void sf_fast_geometry(int W, int n_cu, int n, int *grid, int *threads, size_t *lds, size_t *scratch,
                                    bool hc = false) {
  const SfFastLayout L = sf_fast_layout(W, hc);
  const int nt = sf_fast_threads(W);
  int per_cu = (160 * 1024) / L.total;
  // The kernel is compiled for SF_FAST_WAVES_PER_SIMD waves per SIMD (128 VGPRs each): more workgroups than that
  // cannot be resident however little LDS they need, and a persistent grid larger than what is resident runs its
  // surplus workgroups as a second round (measured at W=100: 5 per CU 3.31 M folds/s, 4 per CU 4.12 M).
  const int by_waves = (4 * SF_FAST_WAVES_PER_SIMD) / (nt / 64);
  if (per_cu > by_waves) per_cu = by_waves;
  if (per_cu < 1) per_cu = 1;
  long long gsz = (long long)n_cu * per_cu;
  if (gsz > n) gsz = n;
  *grid = (int)gsz;
  *threads = nt;
  *lds = (size_t)L.total;
  *scratch = (size_t)((gsz + 7) & ~7LL) * SF_CG_ENTRIES(W) * sizeof(int16_t);  // whole groups of eight: sf_fast_scratch_slot
}

// W = 120 is ScanFold's default window (ScanFold-Scan.py:37) and W = 200 is BASELINE config 5: they get
// instantiations with the width folded in (less scalar index arithmetic, fewer spills); any other width runs
// the generic instantiation
template <bool PZ, typename... A>
static inline void sf_fast_launch_pz(int grid, int threads, size_t lds, hipStream_t st, const uint8_t *seqs, int n, int W,
                                     A... args) {
  if (threads == 256 && W == 120) SF_LAUNCH((sf_mfe_fast_kernel<128, 120, SF_MG120 != 0, false, PZ>), grid, 256, lds, st, seqs, n, W, args...);
  else if (threads == 256 && W < SF_HELP_MERGE_MAXW && SF_HELP_MERGE) SF_LAUNCH((sf_mfe_fast_kernel<128, 0, true, false, PZ>), grid, 256, lds, st, seqs, n, W, args...);
  else if (threads == 256) SF_LAUNCH((sf_mfe_fast_kernel<128, 0, false, false, PZ>), grid, 256, lds, st, seqs, n, W, args...);
  else if (W == 200) SF_LAUNCH((sf_mfe_fast_kernel<256, 200, false, false, PZ>), grid, 512, lds, st, seqs, n, W, args...);
  else SF_LAUNCH((sf_mfe_fast_kernel<256, 0, false, false, PZ>), grid, 512, lds, st, seqs, n, W, args...);
}
// poison != 0: the poison builds (tests only); the last kernel argument
template <typename... A>
static inline void sf_fast_launch(int poison, int grid, int threads, size_t lds, hipStream_t st, const uint8_t *seqs, int n, int W,
                                  A... args) {
  if (poison) sf_fast_launch_pz<true>(grid, threads, lds, st, seqs, n, W, args..., poison);
  else sf_fast_launch_pz<false>(grid, threads, lds, st, seqs, n, W, args..., 0);
}
// constrained folds (per-fold hard constraint / Deigan pseudo-energies; every fold traced)
template <typename... A>
static inline void sf_fast_launch_hc(int grid, int threads, size_t lds, hipStream_t st, const uint8_t *seqs, int n, int W,
                                     A... args) {
  if (threads == 256 && W < SF_HELP_MERGE_MAXW && SF_HELP_MERGE) SF_LAUNCH((sf_mfe_fast_kernel<128, 0, true, true>), grid, 256, lds, st, seqs, n, W, args..., 0);
  else if (threads == 256) SF_LAUNCH((sf_mfe_fast_kernel<128, 0, false, true>), grid, 256, lds, st, seqs, n, W, args..., 0);
  else SF_LAUNCH((sf_mfe_fast_kernel<256, 0, false, true>), grid, 512, lds, st, seqs, n, W, args..., 0);
}
