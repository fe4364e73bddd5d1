// sf_pf_lds.hip.h — McCaskill partition function with every table resident in LDS (W <= 120).
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
//  * a thread slot owns a centre s = i+j (cell (s-j, j) of column j), so the generic interior-loop sums are
//    carried in registers from the enclosed / enclosing cell exactly as in sf_pf_fast.hip.h.  Centres s and
//    s+128 are never live in the same column (a column has at most W-4 <= 128 cells), so 128 slots cover all
//    of them and every wave has work in every column.
//  * one workgroup = 4 teams of 128 threads working on the SAME cells, split by role (wave-uniform):
//      team 0  generic interior-loop sums (the register recurrence), small special loops, hairpin / exterior
//      team 1  bulges                      team 2  1xn loops, then the cell's final sum and all table writes
//      team 3  multiloop sum of the cell (inside: closing term, plus the hairpin; outside: stem term)
//      inside, teams 0 and 1 also share the second O(W) sum of a column (qm of column j-1).  Outside, that second sum (R1 of
//      column l-1) is a matrix product and evaluated for eight columns at a time (r1_block below: a fixed row per thread, two of
//      the block's columns per team, the rows of qm read across lanes); team 1 takes the tail of team 3's multiloop sums in the
//      late columns and six of team 0's special loops in the early ones (shares from per-team cycle stamps, tools/dev/abl.py
//      "pfstamps")
//    partial sums meet in LDS; two barriers per column.  Between the passes two waves walk the exterior sums while the other six
//    park the inside state (shared runs) and build the outside pass's tables.
//  * qb/ob are stored column-major (a column's rows are consecutive: the lanes of a wave read consecutive
//    doubles for every interior-loop candidate), qm diagonal-major (multiloop sums run over the offset from the
//    thread's own row).  Interior-loop candidates come in two families:
//      B: the column is one of the last three, the row runs with the loop size — read from three small
//         rolling buffers that hold qb (ob) pre-multiplied by the pair's own mismatch / terminal weights;
//      A: the row is fixed per thread, the column runs with the loop size — qb (ob) times a weight taken
//         from a 25x25 table indexed by (nucleotides at the column: wave-uniform row) x (nucleotides at the
//         thread's row: 25 consecutive doubles, so the gather is bank-conflict free).  Column offsets and
//         weight rows are wave-uniform and come from per-column lane tables (v_readlane).
// FP64 throughout; sums are re-associated with respect to the oracle (agreement ~1e-12 relative).
#pragma once
#include "sf_energy.h"
#include "sf_pf.hip.h"

#define SF_PFL_NT 512
#define SF_PFL_SLOTS 128
#define SF_PFL_PAD 32
#define SF_PFL_LDS_LIMIT (160 * 1024)
#define SF_PFL_NZP 9  // partial-sum vectors of a column: four teams, four parts of qm / R1, team 1's share of the multiloop sum
// Outside pass, R1 (sf_pf_lds_kernel): columns per block of the blocked evaluation; the share of team 3's multiloop sum that team 1
// takes (blocks of eight terms: (l - MLS0) / MLS1)
#ifndef SF_PFL_RB
#define SF_PFL_RB 8
#endif
#ifndef SF_PFL_LSP
#define SF_PFL_LSP 64  // outside pass: team 1 takes six of team 0's special loops in the columns l <= LSP
#endif
#ifndef SF_PFL_MLS0
#define SF_PFL_MLS0 30
#define SF_PFL_MLS1 12
#endif
// A per-column table of wave-uniform values, one entry per lane, read back with v_readlane (no LDS round trip
// per use).  The table is filled outside divergent control flow.  The CPU emulation keeps it as a plain array.
// (SF_EMUL cannot see the hazard the pinned load guards against — stale entries in inactive lanes; the v_readlane path runs under
// tests/test_gpu_parity.py::test_traceback_and_partition_function_parity, test_scan_step_one_shares_inside_tables and the every-width sweep)
#ifdef SF_EMUL
#define SF_LANE_TABLE(name, L, expr) int name[64]; for (int L = 0; L < 64; L++) name[L] = (expr)
#define SF_LANE_TABLE_DECL(name) int name[64]
#define SF_LANE_TABLE_SET(name, L, expr) for (int L = 0; L < 64; L++) name[L] = (expr)
#define SF_LANE_TABLE_LOAD(name, L, expr) for (int L = 0; L < 64; L++) name[L] = (expr)
#define SF_LANE_TABLE_PIN(name) (void)0
#define SF_LANE_GET(name, idx) name[idx]
#else
// (the empty asm pins the load here, with every lane active: the compiler must not sink it into the divergent
// region where the entries are read back, or inactive lanes would hold stale values)
#define SF_LANE_TABLE(name, L, expr) int name; { const int L = threadIdx.x & 63; name = (expr); asm volatile("" : "+v"(name)); }
#define SF_LANE_TABLE_DECL(name) int name
#define SF_LANE_TABLE_SET(name, L, expr) { const int L = threadIdx.x & 63; name = (expr); asm volatile("" : "+v"(name)); }
// (the two halves apart: the load issued where every lane is active, the pin — still outside divergent control flow — later, so
// that independent work sits between the read and the wait for it)
#define SF_LANE_TABLE_LOAD(name, L, expr) { const int L = threadIdx.x & 63; name = (expr); }
#define SF_LANE_TABLE_PIN(name) asm volatile("" : "+v"(name))
#define SF_LANE_GET(name, idx) __builtin_amdgcn_readlane(name, idx)
#endif
// Sizes UHI down to ULO in batches of NB: PRE (fills ca[t], pa[t]: the lane-table entries of size u — a v_readlane each, whose
// scalar result the vector ALU may not use in the next instruction: read for the whole batch first, they need no hazard no-ops),
// LOAD (fills qa[t], fa[t], da[t], wa[t] for size u) for a whole batch, then USE for the same sizes in the same order.  The arithmetic
// and its order are those of the plain loop; the fences (SF_SCHED_FENCE, sf_launch.h) keep the compiler from re-interleaving the
// phases (it then waits for nearly every LDS read where it is issued).
#define SF_PFL_BATCHES(UHI, ULO, NB, PRE, LOAD, USE)                              \
  _Pragma("unroll") for (int ub_ = (UHI); ub_ >= (ULO); ub_ -= (NB)) {            \
    double qa[NB], fa[NB], da[NB], wa[NB];                                        \
    int ca[NB], pa[NB];                                                           \
    (void)fa; (void)pa; (void)ca;                                                           \
    SF_SCHED_FENCE();                                                             \
    _Pragma("unroll") for (int t = 0; t < (NB); t++) {                            \
      const int u = ub_ - t;                                                      \
      ca[t] = 0; pa[t] = 0;                                                       \
      if (u >= (ULO)) PRE                                                         \
    }                                                                             \
    SF_SCHED_FENCE();                                                             \
    _Pragma("unroll") for (int t = 0; t < (NB); t++) {                            \
      const int u = ub_ - t;                                                      \
      if (u >= (ULO)) LOAD                                                        \
    }                                                                             \
    SF_SCHED_FENCE();                                                             \
    _Pragma("unroll") for (int t = 0; t < (NB); t++) {                            \
      const int u = ub_ - t;                                                      \
      if (u >= (ULO)) USE                                                         \
    }                                                                             \
  }
// packed neighbour codes: bits 0-9 code*25 (row offset into a 25x25 weight table), 12-16 code, 20-22 nucleotide
#define SF_PK_ROW(p) ((p) & 0x3ff)
#define SF_PK_CODE(p) (((p) >> 12) & 31)
#define SF_PK_NT(p) ((p) >> 20)

// hc: the instantiation for constrained folds also holds the window's constraint (characters, bracket partners, enclosing
// pairs: a byte each for positions 0..W+1)
__host__ __device__ inline size_t sf_pfl_lds_bytes(int W, bool hc = false) {
  const size_t NC = (size_t)(W - 4) * (W - 3) / 2, RP = W + 2 * SF_PFL_PAD, VW = W + 8;
  const size_t dbl = 2 * NC + 12 * RP + 3 * 625 + 2 * VW + 8 + 6 * VW + SF_PFL_NZP * VW + (W + 2) + (W + 3) + 16 + (W + 8);
  return dbl * sizeof(double) + 2 * (size_t)(W + 2) * sizeof(int) + (size_t)(W + 8) + 64 + (hc ? (size_t)((3 * (W + 2) + 7) & ~7) : 0);
}
// doubles per workgroup of the shared-inside state: qb, qm, derived buffers, qm1, 27 registers per centre slot
#define SF_PFL_SHARE_DOUBLES(W) ((size_t)((W)-4) * ((W)-3) + 12 * ((W) + 2 * SF_PFL_PAD) + 2 * ((W) + 8) + 8 + SF_PFL_SLOTS * 27)
static inline bool sf_pfl_supported(int W) {
  return W >= 16 && W <= 128 && 2 * W - 4 < 2 * SF_PFL_SLOTS && sf_pfl_lds_bytes(W) <= SF_PFL_LDS_LIMIT;  // (exterior sweeps: two columns per lane)
}

// v of lane l (l wave-uniform), 32-bit
__device__ __forceinline__ int sf_lane_read_i32(const int v, const int l) {
#ifdef SF_EMUL
  return __shfl(v, l);
#else
  return __builtin_amdgcn_readlane(v, l);
#endif
}
// v of lane l (l wave-uniform), 64-bit
__device__ __forceinline__ double sf_lane_read_f64(const double v, const int l) {
#ifdef SF_EMUL
  return __shfl(v, l);
#else
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b & 0xffffffffu), l);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), l);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
#endif
}


// HC: fold k has its own hard constraint, W characters at cons_rows + k * row_stride * W (fc.hc_add_from_db, ScanFold-Scan.py:
// 405-410; sf_fold_constrained): applied where a cell's own pair type is made, inside and outside.  Unbalanced brackets set
// bit 1 of *status.  The caller keeps windows with a bracket pair of non-complementary bases (type 7) away from this kernel
// (its mismatch-weight tables are built from sequence-only pair types).
template <int WT, bool SH, bool HC = false>
__global__ __launch_bounds__(SF_PFL_NT) void sf_pf_lds_kernel(const uint8_t *__restrict__ seqs, int n, int row_stride,
                                                              int Wrt, const SfDevParams *__restrict__ D,
                                                              const SfDevParamsPF *__restrict__ X,
                                                              double *__restrict__ ens_dG,
                                                              double *__restrict__ mean_bp_dist,
                                                              char *__restrict__ centroid,
                                                              double *__restrict__ centroid_dist,
                                                              const uint8_t *__restrict__ tr, int L, int win0,
                                                              int step, int run_len, double *__restrict__ share,
                                                              const char *__restrict__ cons_rows,
                                                              int *__restrict__ status) {
  // SH (native windows of one transcript, `step` nucleotides apart, sf_scan): a workgroup takes RUNS of run_len
  // consecutive windows.  The inside tables of window w+1 are those of window w shifted by `step` rows and columns
  // plus `step` new columns — provided the ends of a window are treated like any other position
  // (neighbours from the transcript): the entries that then differ from a stand-alone fold (first row / last column
  // of qm, qm1 and the derived buffers) are never read by anything that reaches the outputs.  After the inside pass
  // the workgroup dumps qb, qm, the derived buffers, qm1 and team 0's recurrence registers to its slice of `share`;
  // the next window reloads them shifted and runs the column loop for its last `step` columns only.
  SF_DYN_SMEM(smem);
  const int W = WT ? WT : Wrt;
  const int tid = threadIdx.x;
  const int c = tid & (SF_PFL_SLOTS - 1);            // centre slot: centres c and c + 128
  const int team = SF_WAVE_UNIFORM(tid >> 7);        // role, the same for all lanes of a wave
  const int W1 = W + 1;
  const int NC = ((W - 4) * (W - 3)) >> 1, RP = W + 2 * SF_PFL_PAD, VW = W + 8;
  double *QB = (double *)smem;    // qb, then ob: column-major, column j holds rows 1..j-4
  double *QM = QB + NC;           // qm: diagonal-major, diagonal d >= 4 holds rows 1..W-d
  double *DER = QM + NC;          // [3 kinds][4 column slots][RP rows, row r at r + PAD]
  double *FAC = DER + 12 * RP;    // [3][25][25] family-A weights
  double *QM1 = FAC + 3 * 625;    // [2][VW]
  double *RV = QM1 + 2 * VW + 8;  // R0[2], R1[2], R01[2], each VW, row r at r + 8 (rows <= 0 stay 0)
  double *ZP = RV + 6 * VW;       // [4 teams][VW] partial sums of the current column, [4..7] = the parts of qm / R1, [8] = team 1's part of the multiloop sum
  double *q5 = ZP + SF_PFL_NZP * VW;  // [W+2]
  double *q3 = q5 + (W + 2);      // [W+3]
  double *red = q3 + (W + 3);     // [16]
  // size weights and MLbase^a: the same for every lane — scalar loads from the parameter block (they used to be LDS copies read
  // with broadcast reads: a third of the size loops' LDS instructions)
  const double *const WN = X->ninio, *const WB = X->bulge, *const WIL = X->internal_loop, *const WIL1N = X->il1n;
  const double *const MLB = X->mlbase_pow;
  double *ZS = red + 16;          // [W+8] zeros, never written: the "column" beyond the window's end in the outside pass
  int *FWD = (int *)(ZS + (W + 8));  // [W+2]  packed code of (S[x], S[x+1])
  int *BWD = FWD + (W + 2);           // [W+2]  packed code of (S[x], S[x-1])
  uint8_t *S = (uint8_t *)(BWD + (W + 2));  // [W+8]
  uint8_t *PT8 = S + (W + 8);               // [64] pair type of two nucleotide codes
  char *hcC = (char *)(PT8 + 64);           // (HC) [W+2] constraint characters, then partners, then enclosing pairs
  uint8_t *hcP = (uint8_t *)hcC + (W + 2), *hcE = hcP + (W + 2);
  SfHc8 hc;
  hc.c = (HC && cons_rows) ? hcC : nullptr; hc.partner = hcP; hc.encl = hcE;
#define COFF(j) ((((j)-5) * ((j)-4)) >> 1)
#define DOFF(d) (((d)-4) * W - ((((d) * ((d)-1)) >> 1) - 6))
#define QBC(i, j) QB[COFF(j) + (i)-1]
#define QMD(d, i) QM[DOFF(d) + (i)-1]
#define DERP(kind, col) (DER + ((kind)*4 + ((col)&3)) * RP + SF_PFL_PAD)
#define PAIR(a, b) PT8[(a)*8 + (b)]
// pair type of the cell itself: none beyond RNA.md().max_bp_span
#define OWN(a, b) sf_hc_type8(hc, (((b) - (a)) <= maxd ? PAIR(S[a], S[b]) : 0), (a), (b), ((b) - (a)) <= maxd)
  const double xTAU = X->TermAU;
  const double xMLbase = X->MLbase;
  const int maxd = D->max_pair_dist;
  // speculative (discarded or zero-weighted) reads below may land anywhere in the tables: keep them finite
  for (int x = tid; x < 2 * NC; x += SF_PFL_NT) QB[x] = 0.0;
  for (int x = tid; x < W + 8; x += SF_PFL_NT) ZS[x] = 0.0;
  if (tid < 64) PT8[tid] = (uint8_t)D->pair[tid >> 3][tid & 7];

  const bool shared = SH;  // (a template parameter: the stand-alone instantiation carries none of the extra state)
  const int SV_QM = NC, SV_DER = 2 * NC, SV_QM1 = SV_DER + 12 * RP, SV_H = SV_QM1 + 2 * VW + 8;
  // (slices of the workgroups of one XCD — block b runs on XCD b mod 8 — adjacent: the state a workgroup parks, 102 kB at W = 120, is
  // read back by the same workgroup one window later; 32 workgroups per XCD x 102 kB = 3.3 MB stay in that XCD's 4-MB L2)
  double *sv = SH ? share + (size_t)((blockIdx.x & 7u) * ((gridDim.x + 7u) >> 3) + (blockIdx.x >> 3)) * SF_PFL_SHARE_DOUBLES(W) : nullptr;
  if (!shared) run_len = 1;
  for (int fold0 = blockIdx.x * run_len; fold0 < n; fold0 += gridDim.x * run_len)
  for (int fold = fold0; fold < fold0 + run_len && fold < n; fold++) {
    const bool resume = SH && fold > fold0;                          // the previous window's state is in sv
    const bool keep = SH && fold + 1 < fold0 + run_len && fold + 1 < n;  // the next window will want this one's
    const uint8_t *src = seqs + (size_t)fold * row_stride * W;
    const int pos = (win0 + fold) * step;  // window start in the transcript
    const bool nbL = SH && pos > 0, nbR = SH && pos + W < L;
    __syncthreads();
    for (int x = tid; x < W; x += SF_PFL_NT) S[x + 1] = sf_encode_nt(src[x]);
    if (tid == 0) { S[0] = nbL ? sf_encode_nt(tr[pos - 1]) : 0; S[W + 1] = nbR ? sf_encode_nt(tr[pos + W]) : 0; }
    if (HC && cons_rows) {
      for (int x = tid; x < W; x += SF_PFL_NT) hcC[x + 1] = cons_rows[(size_t)fold * row_stride * W + x];
      __syncthreads();
      if (tid == 0 && sf_hc_parse8(W, hcC, hcP, hcE)) {  // unbalanced: reported; the fold runs with whatever matched
        if (status) atomicOr(status, 2);
      }
    }
    for (int x = tid; x < 12 * RP; x += SF_PFL_NT) DER[x] = 0.0;
    for (int x = tid; x < (8 + SF_PFL_NZP) * VW + 8; x += SF_PFL_NT) QM1[x] = 0.0;  // QM1, the six R vectors, the partial sums
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
    if (resume) {
      const int wv = tid >> 6, ln = tid & 63;
      // k = step.  qb: new (i,j) = old (i+k,j+k), columns 5..W-k;  qm: new (d,i) = old (d,i+k)
      const int k = step, j0 = W - k + 1;  // j0: first column to compute
      // (four columns / diagonals of a wave at a time, their loads in flight together: one dependent global load per
      // trip made this reload ~7 % of a resumed window)
      constexpr int NWV = SF_PFL_NT / 64, RB = 4;
      for (int j0b = 5 + wv; j0b <= W - k; j0b += RB * NWV) {
        double v[RB][2];
#pragma unroll
        for (int b = 0; b < RB; b++)
#pragma unroll
          for (int h = 0; h < 2; h++) {
            const int jj = j0b + b * NWV, ii = 1 + ln + 64 * h;
            v[b][h] = (jj <= W - k && ii <= jj - 4) ? sv[COFF(jj + k) + ii + k - 1] : 0.0;
          }
#pragma unroll
        for (int b = 0; b < RB; b++)
#pragma unroll
          for (int h = 0; h < 2; h++) {
            const int jj = j0b + b * NWV, ii = 1 + ln + 64 * h;
            if (jj <= W - k && ii <= jj - 4) QBC(ii, jj) = v[b][h];
          }
      }
      // qm never left the LDS (the outside pass does not write it): shifted in place, diagonal by diagonal.  A diagonal belongs to
      // one wave, which reads a whole batch before it writes it (the wave-level sync only matters to the CPU emulation, whose
      // lanes run one after the other in no fixed order; in lockstep the reads of an instruction precede the writes of a later one)
      for (int d0b = 4 + wv; d0b <= W - k - 1; d0b += RB * NWV) {
        double v[RB][2];
#pragma unroll
        for (int b = 0; b < RB; b++)
#pragma unroll
          for (int h = 0; h < 2; h++) {
            const int dd = d0b + b * NWV, ii = 1 + ln + 64 * h;
            v[b][h] = (dd <= W - k - 1 && ii <= W - k - dd) ? QM[DOFF(dd) + ii + k - 1] : 0.0;
          }
        SF_WAVE_SYNC();
#pragma unroll
        for (int b = 0; b < RB; b++)
#pragma unroll
          for (int h = 0; h < 2; h++) {
            const int dd = d0b + b * NWV, ii = 1 + ln + 64 * h;
            if (dd <= W - k - 1 && ii <= W - k - dd) QMD(dd, ii) = v[b][h];
          }
        SF_WAVE_SYNC();
      }
      // derived buffers of the last three old columns (= new columns j0-3 .. j0-1), rows shifted; qm1 of new column j0-1
      for (int x = tid; x < 9 * W; x += SF_PFL_NT) {
        const int kc = x / W, ii = x - kc * W + 1, kind = kc / 3, cn = j0 - 3 + (kc - kind * 3);
        if (ii + k <= W) DERP(kind, cn)[ii] = sv[SV_DER + (kind * 4 + ((cn + k) & 3)) * RP + SF_PFL_PAD + ii + k];
      }
      for (int ii = tid + 1; ii + k <= W + 1; ii += SF_PFL_NT) QM1[((j0 - 1) & 1) * VW + ii] = sv[SV_QM1 + (W & 1) * VW + ii + k];
      if (team == 0) {  // recurrence registers of the cell this thread's cell (i, j0) encloses: old cell (i+1+k, W)
        const int sW = (j0 <= c - 1) ? c : c + SF_PFL_SLOTS, iW = sW - j0;
        if (iW >= 1 && iW + 1 + k <= W) {
          const double *hp = sv + SV_H + ((iW + 1 + k + W) & (SF_PFL_SLOTS - 1)) * 27;
#pragma unroll
          for (int u = 0; u < 27; u++) H[u] = hp[u];
        }
      }
      __syncthreads();
    }
    // (Round 5, after the MFE kernels gained 5-8 % from running each KIND of step as a loop of its own: the same idea here — every
    // team its own copy of the column loops, roles as compile-time constants, team 0's 27 recurrence registers live in its loop
    // only — is bit-identical, compiles to 197 instead of 217 VGPRs, and changes nothing that matters: the cfg3 scan's partition
    // functions 65.4 -> 67.6 ms (W = 120 instantiation), 53.5 -> 51.3 ms at W = 100.  Neither version spills; not kept.)
    for (int j = resume ? W - step + 1 : SFD_TURN + 2; j <= W + 1; j++) {
      // j = W+1 only finishes qm of column W
      const int s = (j <= c - 1) ? c : c + SF_PFL_SLOTS;
      const int i = s - j, d = j - i;
      const bool valid = (i >= 1) && (d >= SFD_TURN + 1) && (j <= W);
      double *qm1c = QM1 + (j & 1) * VW, *qm1p = QM1 + ((j & 1) ^ 1) * VW;
      // lane tables, entry L: column max(j-L, 5)
      SF_LANE_TABLE(tpk, L, FWD[sfd_min(sfd_max(j - L, 5), W)]);
      SF_LANE_TABLE(tcol, L, COFF(sfd_min(sfd_max(j - L, 5), W)));
      const int umax = sfd_min(SFD_MAXLOOP, d - 2 - (SFD_TURN + 1));
      // qm of column j-1, row i:  qm[i,j-1] = sum_{a>=0} MLbase^a qm1[i+a,j-1] + sum_{a>=5} qm[i,i+a-1] qm1[i+a,j-1].
      // The a >= 5 part runs in blocks of eight terms; team 1 takes the first half of the blocks (plus the
      // a < 5 terms), team 0 — the lightest here, tools/gpu_pf_stamp.py — the rest (a share for team 3 was measured
      // slower: its own multiloop sum grows with the same columns); the overshoot of the last block reads rows > j-5 of qm1, which are 0.
      const int dq = j - 1 - i;
      const bool qvalid = (i >= 1) && (dq >= SFD_TURN + 1);
      auto qm_part = [&](const int part) -> double {
        const int amax = dq - SFD_TURN - 1;
        const int nb = (amax - 4 + 7) >> 3, nbh = nb >> 1;  // blocks of eight terms starting at a = 5
        double m = 0.0, m2 = 0.0;
        if (part == 0) {
          m = qm1p[i];
          for (int a = 1; a <= sfd_min(amax, 4); a++) m += MLB[a] * qm1p[i + a];
        }
        const int b0 = part ? nbh : 0, b1 = part ? nb : nbh;
        const int a0 = 5 + 8 * b0;
        const double *qmr = QM + i - 1, *q1 = qm1p + i;
        int off = DOFF(a0 - 1), st = W - (a0 - 1);  // DOFF(a-1) and its increment
        for (int a = a0; a < 5 + 8 * b1; a += 8) {
          double t0 = 0.0, t1 = 0.0;
          double mb[8], qv[8], rv[8];
#pragma unroll
          for (int t = 0; t < 8; t++) {
            mb[t] = MLB[a + t]; qv[t] = qmr[off]; rv[t] = q1[a + t];
            off += st--;
          }
          SF_SCHED_FENCE();
#pragma unroll
          for (int t = 0; t < 8; t += 2) {
            t0 += (mb[t] + qv[t]) * rv[t];
            t1 += (mb[t + 1] + qv[t + 1]) * rv[t + 1];
          }
          m += t0;
          m2 += t1;
        }
        return m + m2;
      };
      double wml = 0.0;      // team 2: multiloop-stem weight of (i,j), fetched from device memory ahead of its use after the barrier
      if (team == 0) {
        if (valid) {
          const int type = OWN(i, j);
          const int si1 = S[i + 1], sj1 = S[j - 1];
          if (d < SFD_TURN + 3) {  // first cell of this centre
#pragma unroll
            for (int u = 0; u < 27; u++) H[u] = 0.0;
          }
          const double *dI3 = DERP(0, j - 3) + i;
          const double *fI = FAC + SF_PK_CODE(BWD[i + 3]);
          const double *qbA = QB + i + 2;  // row i+3
          // Straight-line: family-A values are loaded speculatively (column clamped to an existing one) and
          // dropped by a select when the inner span would be < TURN+1; family-B values are 0 there by
          // themselves.  Sizes beyond umax therefore stay exactly 0 and need no separate bookkeeping.
          SF_PFL_BATCHES(30, 6, 9, { ca[t] = SF_LANE_GET(tcol, u - 1); pa[t] = SF_PK_ROW(SF_LANE_GET(tpk, u - 1)); }, {
            qa[t] = qbA[ca[t]]; fa[t] = fI[pa[t]];
            da[t] = dI3[u - 1]; wa[t] = WN[u - 4];
          }, {
            const double a = qa[t] * fa[t];
            H[u - 4] = H[u - 6] + ((u <= umax ? a : 0.0) + da[t]) * wa[t];
          })
          {
            const double a = qbA[SF_LANE_GET(tcol, 4)] * fI[SF_PK_ROW(SF_LANE_GET(tpk, 4))];
            H[1] = ((umax >= 5 ? a : 0.0) + dI3[4]) * WN[1];
            H[0] = dI3[3] * WN[0];
          }
          // Small special loops, branch-free: every weight (device-memory gathers) and every inner qb is fetched
          // unconditionally — so the fetches overlap — with columns clamped to existing ones; an inner cell
          // that does not exist (span < TURN+1) is dropped by a select.  (The hairpin term is team 3's.)
          const int s2 = S[i + 2], s3 = S[i + 3], s4 = S[i + 4], t2 = S[j - 2], t3 = S[j - 3], t4 = S[j - 4];
          const int c1 = sfd_max(j - 1, 5), c2 = sfd_max(j - 2, 5), c3 = sfd_max(j - 3, 5), c4 = sfd_max(j - 4, 5);
          const double w00 = X->stack[type][sfd_rtype(PAIR(si1, sj1))];
          const double w01 = X->stack[type][sfd_rtype(PAIR(si1, t2))], w10 = X->stack[type][sfd_rtype(PAIR(s2, sj1))];
          const double w11 = X->int11[type][sfd_rtype(PAIR(s2, t2))][si1][sj1];
          const double w12 = X->int21[type][sfd_rtype(PAIR(s2, t3))][si1][t2][sj1];
          const double w21 = X->int21[sfd_rtype(PAIR(s3, t2))][type][sj1][si1][s2];
          const double w22 = X->int22[type][sfd_rtype(PAIR(s3, t3))][si1][s2][t2][sj1];
          const double w23o = X->mismatch23I[type][si1][sj1];
          const double w23 = X->mismatch23I[sfd_rtype(PAIR(s3, t4))][t3][s2], w32 = X->mismatch23I[sfd_rtype(PAIR(s4, t3))][t2][s3];
          const double wI = X->mismatchI[type][si1][sj1];
          const double q00 = QBC(i + 1, c1), q01 = QBC(i + 1, c2), q10 = QBC(i + 2, c1), q11 = QBC(i + 2, c2);
          const double q12 = QBC(i + 2, c3), q21 = QBC(i + 3, c2), q22 = QBC(i + 3, c3), q23 = QBC(i + 3, c4), q32 = QBC(i + 4, c3);
          double z = (umax >= 0 ? q00 : 0.0) * w00;
          z += ((umax >= 1 ? q01 : 0.0) * w01 + (umax >= 1 ? q10 : 0.0) * w10) * WB[1];
          z += (umax >= 2 ? q11 : 0.0) * w11;
          z += (umax >= 3 ? q12 : 0.0) * w12 + (umax >= 3 ? q21 : 0.0) * w21;
          z += (umax >= 4 ? q22 : 0.0) * w22;
          z += (WIL[5] * WN[1] * w23o) * ((umax >= 5 ? q23 : 0.0) * w23 + (umax >= 5 ? q32 : 0.0) * w32);
          {
            double gg = 0.0, gg2 = 0.0;
#pragma unroll
            for (int u = 6; u <= 30; u += 2) {
              gg += H[u - 4] * WIL[u];
              if (u + 1 <= 30) gg2 += H[u - 3] * WIL[u + 1];
            }
            z += (gg + gg2) * wI;
          }
          ZP[i] = z;
        }
        if (qvalid) ZP[5 * VW + i] = qm_part(1);
      } else if (team == 1) {
        if (valid) {
          const int type = OWN(i, j);
          const int sp = S[i + 1];  // row of the u1 = 0 bulge candidates
          const double *dB1 = DERP(2, j - 1) + i + 1;
          const double *qbA = QB + i;  // row i+1
          double gb = 0.0, gb2 = 0.0;
          SF_PFL_BATCHES(30, 2, 10, { ca[t] = SF_LANE_GET(tcol, 33 - u); pa[t] = SF_PK_NT(SF_LANE_GET(tpk, 33 - u)); }, {
            const int u_ = 32 - u;  // ascending sizes 2..30, as the sums were always accumulated
            qa[t] = qbA[ca[t]]; da[t] = dB1[u_]; wa[t] = WB[u_]; fa[t] = 0.0;
          }, {
            const int u_ = 32 - u;
            const double ta_ = (sp * pa[t] == 6) ? 1.0 : xTAU;  // C-G / G-C: no terminal penalty
            const double ab = qa[t] * ta_;
            const double tt = ((u_ <= umax ? ab : 0.0) + da[t]) * wa[t];
            if (u_ & 1) gb2 += tt; else gb += tt;
          })
          ZP[VW + i] = (gb + gb2) * (type > 2 ? xTAU : 1.0);
        }
        if (qvalid) ZP[4 * VW + i] = qm_part(0);
      } else if (team == 2) {
        if (valid) {
          const int type = OWN(i, j);
          if (type) wml = sfx_mlstem(X, type, (i > 1 || nbL) ? S[i - 1] : -1, (j < W || nbR) ? S[j + 1] : -1);
        }
        if (valid) {
          const double *d1N2 = DERP(1, j - 2) + i;
          const double *f1N = FAC + 625 + SF_PK_CODE(BWD[i + 2]);
          const double *qbA = QB + i + 1;  // row i+2
          double g1 = 0.0, g2 = 0.0;
          SF_PFL_BATCHES(30, 4, 9, { ca[t] = SF_LANE_GET(tcol, 34 - u); pa[t] = SF_PK_ROW(SF_LANE_GET(tpk, 34 - u)); }, {
            const int u_ = 34 - u;  // ascending sizes 4..30
            qa[t] = qbA[ca[t]]; fa[t] = f1N[pa[t]];
            da[t] = d1N2[u_]; wa[t] = WIL1N[u_];
          }, {
            const int u_ = 34 - u;
            const double an = qa[t] * fa[t];
            const double tt = ((u_ <= umax ? an : 0.0) + da[t]) * wa[t];
            if (u_ & 1) g2 += tt; else g1 += tt;
          })
          // mismatch1nI[type][S[i+1]][S[j-1]]: the family-A table read with the roles of row and column swapped
          ZP[2 * VW + i] = (g1 + g2) * FAC[625 + SF_PK_ROW(FWD[i]) + SF_PK_CODE(BWD[j])];
        }
      } else {
        if (valid) {
          const int type = OWN(i, j);
          const int si1 = S[i + 1], sj1 = S[j - 1];
          // sum_a qm[i+1,i+a-1] qm1[i+a,j-1], a = 6..d-5; eight terms per trip, the overshoot reads rows of
          // qm1 that are still 0 (rows > j-5 of column j-1)
          double ml = 0.0, ml1 = 0.0;
          const double *qmr = QM + i, *q1 = qm1p + i;
          int off = 0, st = W - 4;  // DOFF(a-2) and its increment, a = 6
          for (int a = SFD_TURN + 3; a <= d - SFD_TURN - 2; a += 8) {
            double t0 = 0.0, t1 = 0.0;
            double qv[8], rv[8];
#pragma unroll
            for (int t = 0; t < 8; t++) {
              qv[t] = qmr[off]; rv[t] = q1[a + t];
              off += st--;
            }
            SF_SCHED_FENCE();
#pragma unroll
            for (int t = 0; t < 8; t += 2) {
              t0 += qv[t] * rv[t];
              t1 += qv[t + 1] * rv[t + 1];
            }
            ml += t0;
            ml1 += t1;
          }
          ZP[3 * VW + i] = (ml + ml1) * X->MLclosing * sfx_mlstem(X, sfd_rtype(type), sj1, si1) +
                           (type ? sfx_hairpin(D, X, S, i, j, type) : 0.0);
        }
      }
      __syncthreads();
      if (team == 2 && qvalid) QMD(dq, i) = ZP[5 * VW + i] + ZP[4 * VW + i];
      if (team == 2 && valid) {
        const int type = OWN(i, j);
        const int tr = sfd_rtype(type);
        const double qbij = type ? (ZP[i] + ZP[VW + i]) + (ZP[2 * VW + i] + ZP[3 * VW + i]) : 0.0;
        QBC(i, j) = qbij;
        const int fa = SF_PK_ROW(FWD[j]) + SF_PK_CODE(BWD[i]);  // (S[j], S[j+1]) x (S[i], S[i-1]): LDS, not device memory
        DERP(0, j)[i] = type ? qbij * FAC[fa] : 0.0;
        DERP(1, j)[i] = type ? qbij * FAC[625 + fa] : 0.0;
        DERP(2, j)[i] = (type && tr > 2) ? qbij * xTAU : qbij;
        double m1 = qm1p[i] * xMLbase;
        if (type) m1 += qbij * wml;
        qm1c[i] = m1;
      }
      __syncthreads();
    }

    // ================= exterior =================
    // q5[j] = q5[j-1] + sum_i q5[i-1] qb[i,j] ExtLoop(i,j) and its mirror image q3, each by ONE wave as a sweep
    // without any reduction across lanes (the block-wide sum + barrier per column this replaces took 15 % of a
    // fold): wave 0 walks the rows i upwards with a lane per column j, P[j] += q5[i-1] qb[i,j] w(i,j), where
    // q5[i-1] = q5[i-2] + P[i-1] is final by then (column i-1 only has rows <= i-5) and comes from its lane by
    // v_readlane; wave 1 walks the columns downwards with a lane per row, q3[j+1] from the lane of row j+1.
    // w(i,j) from a 30 x 30 table over (S[j], S[j+1] or "none") x (S[i], S[i-1] or "none"), built in the partial-sum area (idle
    // between the passes) where that holds it: the other six waves then park the inside state, build the outside pass's weight
    // tables in FAC and clear the derived buffers WHILE the two sweeps run (one after the other these were 7 % of a fold).
    const bool ovl = SF_PFL_NZP * VW >= 900;
    double *const EXT = ovl ? ZP : FAC;  // (FAC: its inside-orientation tables are dead, the outside ones are built after the sweeps)
    auto outside_tables = [&](const int t0, const int tn) {
      // family-A weights, outside orientation: row = nucleotides at the enclosing pair's column l' (S[l'],
      // S[l'-1]), entry = its row k' (S[k'], S[k'+1]); third table = multiloop closing weight of (k', l')
      for (int e = t0; e < 625; e += tn) {
        const int b = e / 25, f = e - b * 25;
        const int a = f / 5, a1 = f - a * 5, bs = b / 5, b1 = b - bs * 5;
        const int t = D->pair[a][bs];
        FAC[e] = t ? X->mismatchI[t][a1][b1] : 0.0;
        FAC[625 + e] = t ? X->mismatch1nI[t][a1][b1] : 0.0;
        FAC[1250 + e] = t ? X->MLclosing * sfx_mlstem(X, sfd_rtype(t), b1, a1) : 0.0;
      }
    };
    for (int e = tid; e < 900; e += SF_PFL_NT) {
      const int cf = e / 30, cb = e - cf * 30;
      const int sj = cf / 6, s3 = cf - sj * 6, si = cb / 6, s5 = cb - si * 6;
      const int t = (si < 5 && sj < 5) ? D->pair[si][sj] : 0;
      EXT[e] = t ? sfx_extloop(X, t, s5 < 5 ? s5 : -1, s3 < 5 ? s3 : -1) : 0.0;
    }
    __syncthreads();
    if (tid >= 128) {
      const int ht = tid - 128;
      constexpr int HN = SF_PFL_NT - 128;
      if (keep) {  // the inside state for the next window of the run (nothing below reads sv)
        // (qb — about to be overwritten by the outside values —, the derived buffers, qm1; qm stays where it is: 54 of the 156 kB)
        for (int x = ht; x < NC; x += HN) sv[x] = QB[x];
        for (int x = SV_DER + ht; x < 2 * NC + 12 * RP + 2 * VW + 8; x += HN)
          sv[x] = x < SV_QM1 ? DER[x - SV_DER] : QM1[x - SV_QM1];
      }
      if (ovl) outside_tables(ht, HN);
      if (centroid)
        for (int x = ht; x <= W; x += HN) centroid[(size_t)fold * W1 + x] = (x < W) ? '.' : 0;
      // the derived buffers of columns W+1 .. W+3 are read as zeros (the inside values they hold were saved above if the next
      // window wants them: an entry is cleared by the thread that saved it); a column beyond the window's end in qb is the zero
      // strip: no test per loop size for either
      for (int x = ht; x < 12 * RP; x += HN) DER[x] = 0.0;
    } else {
      if (keep) {  // (team 0 = these two waves: the recurrence registers)
#pragma unroll
        for (int u = 0; u < 27; u++) sv[SV_H + c * 27 + u] = H[u];
      }
      constexpr int NQ = 2;  // sf_pfl_supported: W <= 128
      const int lane = tid & 63;
      const bool fwd = tid < 64;
      double P[NQ];
      int codeF[NQ], codeB[NQ];  // this lane's columns: table row; this lane's rows: table entry
#pragma unroll
      for (int q = 0; q < NQ; q++) {
        const int x = lane + 64 * q + 1;
        P[q] = 0.0;
        codeF[q] = x > W ? 0 : (S[x] * 6 + (x < W ? S[x + 1] : 5)) * 30;
        codeB[q] = x > W ? 0 : S[x] * 6 + (x > 1 ? S[x - 1] : 5);
      }
      auto owner_sum = [&](const int x) -> double {  // P of column / row x, from the lane that owns it
        const int l = (x - 1) & 63, q = (x - 1) >> 6;
        const double v0 = sf_lane_read_f64(P[0], l), v1 = sf_lane_read_f64(P[1], l);
        return q ? v1 : v0;
      };
      // A step of either walk is ONE dependent chain of a single wave (~8 cycles per instruction, six waves waiting for it): the
      // steps are written for as few instructions as possible — which of a lane's two registers a step reads (the owner of the
      // finished sum, the step's neighbour code) is a compile-time constant per stretch of the walk, q5 / q3 stay in registers (a
      // lane per entry) until the walk is over, and the half of the columns / rows a step cannot reach is left out.
      using I0 = std::integral_constant<int, 0>;
      using I1 = std::integral_constant<int, 1>;
      double qr[NQ] = {0.0, 0.0};  // q5[lane + 64 h] (forward), q3[lane + 64 h + 1] (backward)
      // (a step's table values are fetched during the step before: the reads do not depend on the walk's sums)
      if (fwd) {
        int cj[NQ];  // COFF(j) - 1 of this lane's columns
#pragma unroll
        for (int q = 0; q < NQ; q++) cj[q] = COFF(sfd_max(sfd_min(lane + 64 * q + 1, W), 5)) - 1;
        const int codeBp = codeB[0] | (codeB[1] << 16);  // both rows of a lane in one register: one v_readlane per step
        double qprev = 1.0;  // q5[i-1] while row i is processed
        if (lane == 0) qr[0] = 1.0;
        double qv[NQ] = {0.0, 0.0}, ev[NQ] = {0.0, 0.0};
        auto fetch = [&](const int i, auto Q0) {  // qb[i, j] (0 where the cell does not exist) and the weight of (i, j) for this lane's columns
          constexpr int QLO = decltype(Q0)::value;
          const int cb = (sf_lane_read_i32(codeBp, (i - 1) & 63) >> (((i - 1) >> 6) << 4)) & 0xffff;
#pragma unroll
          for (int q = QLO; q < NQ; q++) {
            const int j = lane + 64 * q + 1;
            const bool ok = j <= W && i + SFD_TURN + 1 <= j && j - i <= maxd;
            const double qb = QB[cj[q] + sfd_min(i, W - SFD_TURN - 1)];
            qv[q] = ok ? qb : 0.0;
            ev[q] = EXT[codeF[q] + cb];
          }
        };
        auto step = [&](const int i, auto OHH, auto XHH, auto Q0) {
          constexpr int OH = decltype(OHH)::value, XH = decltype(XHH)::value, QLO = decltype(Q0)::value;
          if (i >= 2) {
            qprev += sf_lane_read_f64(P[OH], (i - 2) & 63);
            qr[XH] = (lane == ((i - 1) & 63)) ? qprev : qr[XH];
          }
          if (i > W - SFD_TURN - 1) return;
          double t[NQ];
#pragma unroll
          for (int q = QLO; q < NQ; q++) t[q] = qprev * qv[q] * ev[q];
          if (i + 1 <= W - SFD_TURN - 1) fetch(i + 1, Q0);
#pragma unroll
          for (int q = QLO; q < NQ; q++) P[q] += t[q];
        };
        fetch(1, I0{});
        for (int i = 1; i <= sfd_min(W, 64); i++) step(i, I0{}, I0{}, I0{});
        if (W >= 65) step(65, I0{}, I1{}, I1{});
        for (int i = 66; i <= W; i++) step(i, I1{}, I1{}, I1{});
        qprev += owner_sum(W);
#pragma unroll
        for (int q = 0; q < NQ; q++)
          if (lane + 64 * q < W) q5[lane + 64 * q] = qr[q];
        if (lane == 0) q5[W] = qprev;
      } else {
        const int codeFp = (codeF[0] / 30) | ((codeF[1] / 30) << 16);
        double qnext = 1.0;  // q3[j+1] while column j is processed
        double qv[NQ] = {0.0, 0.0}, ev[NQ] = {0.0, 0.0};
        auto fetch = [&](const int j, auto Q1) {
          constexpr int QHI = decltype(Q1)::value;
          const int cf = ((sf_lane_read_i32(codeFp, (j - 1) & 63) >> (((j - 1) >> 6) << 4)) & 0xffff) * 30;
          const int cq = COFF(sfd_max(j, 5)) - 1;
#pragma unroll
          for (int q = 0; q <= QHI; q++) {
            const int i = lane + 64 * q + 1;
            const bool ok = i + SFD_TURN + 1 <= j && j - i <= maxd;
            const double qb = QB[cq + sfd_min(i, W)];
            qv[q] = ok ? qb : 0.0;
            ev[q] = EXT[cf + codeB[q]];
          }
        };
        auto step = [&](const int j, auto OHH, auto Q1) {
          constexpr int OH = decltype(OHH)::value, QHI = decltype(Q1)::value;
          if (j < W) {
            qnext += sf_lane_read_f64(P[OH], j & 63);
            qr[OH] = (lane == (j & 63)) ? qnext : qr[OH];
          }
          if (j < SFD_TURN + 2) return;
          double t[NQ];
#pragma unroll
          for (int q = 0; q <= QHI; q++) t[q] = qv[q] * ev[q] * qnext;
          if (j - 1 >= SFD_TURN + 2) fetch(j - 1, Q1);
#pragma unroll
          for (int q = 0; q <= QHI; q++) P[q] += t[q];
        };
        fetch(W, I1{});
        for (int j = W; j >= 65; j--) step(j, I1{}, I1{});
        if (W >= 64) step(64, I1{}, I0{});
        for (int j = sfd_min(W, 63); j >= 1; j--) step(j, I0{}, I0{});
        qnext += owner_sum(1);
#pragma unroll
        for (int q = 0; q < NQ; q++) {
          const int x = lane + 64 * q + 1;
          if (x >= 2 && x <= W) q3[x] = qr[q];
        }
        if (lane == 0) { q3[W + 1] = 1.0; q3[1] = qnext; }
      }
    }
    __syncthreads();
    if (!ovl) {
      outside_tables(tid, SF_PFL_NT);
      __syncthreads();
    }
    const double Z = q5[W];
    const int ZOFF = (int)(ZS - QB);

    // ================= outside: columns l descending =================
#pragma unroll
    for (int u = 0; u < 27; u++) H[u] = 0.0;
    double mbd = 0.0, cd = 0.0;
    double racc[2] = {0.0, 0.0};  // the block sums of R1 for this team's two columns of the block, row kf
    int tbw[2];  // lane table: byte offset of the weight-table row of column m = lane + 64 h + 1 (its nucleotides: wave-uniform per term)
#pragma unroll
    for (int h = 0; h < 2; h++) tbw[h] = SF_PK_ROW(BWD[sfd_min((tid & 63) + 64 * h + 1, W)]) * (int)sizeof(double);
    SF_LANE_TABLE_PIN(tbw[0]);
    SF_LANE_TABLE_PIN(tbw[1]);
    // lane tables, entry L: column min(l+L, W).  Those of column l-1 are fetched while column l's results are written (after the
    // first barrier): a column does not start with an LDS round trip every wave waits for
    SF_LANE_TABLE_DECL(tpk);
    SF_LANE_TABLE_DECL(tcol);
    SF_LANE_TABLE_SET(tpk, L, BWD[sfd_min(W + L, W)]);
    SF_LANE_TABLE_SET(tcol, L, W + L <= W ? COFF(W + L) : ZOFF);
    for (int l = W; l >= SFD_TURN + 2; l--) {
      const int s = (l <= c - 1) ? c : c + SF_PFL_SLOTS;
      const int k = s - l, d = l - k;
      const bool valid = (k >= 1) && (d >= SFD_TURN + 1);
      const bool inner = (k > 1) && (l < W);
      const double qbkl = (valid && team != 0) ? QBC(k, l) : 0.0;  // qb[k,l]: replaced by ob[k,l] at the end of this column
      const double *R1c = RV + (2 + (l & 1)) * VW, *R01c = RV + (4 + (l & 1)) * VW, *R0c = RV + (l & 1) * VW;
      double *R0n = RV + ((l & 1) ^ 1) * VW, *R1n = RV + (2 + ((l & 1) ^ 1)) * VW, *R01n = RV + (4 + ((l & 1) ^ 1)) * VW;
      const bool r3 = k - 3 >= 1, r2 = k - 2 >= 1;  // the row exists
      // carried across the barrier by teams 1-3 (looked up again there, they were three dependent LDS round trips at the head of a
      // phase every wave waits for): the cell's pair type; the cell's weights from the outside-orientation tables, R0 of this column
      int typeC = 0;
      double fc0 = 0.0, fc1 = 0.0;
      // R1 of the next column l-1, row kr: R1[kr] = sum_{m >= l+5} w(kr,m) qm[l,m-1], w(kr,m) = ob[kr,m] x the closing weight of (kr,m)
      // — the product of a matrix that is final once column m is (w) with the rows of qm, which the inside pass left complete.  It
      // used to be evaluated column by column (~30 instructions per term: the largest single item of the outside pass); now it is
      // evaluated for a block of RB = 8 columns (l = r0, r0-1, .., r0-7) when column r0 starts: a thread owns the FIXED row kf (its
      // index in the team + 1), fetches w(kf,m), m > r0, once and adds it to the sums of two of the block's columns — each team has
      // its two, no exchange —, each sum with its own row of qm, held a column per lane and read with v_readlane (entries the row
      // does not have, m < r+5, are zeros).  What a block cannot contain — the terms m = l+5 .. r0 of its last three columns, one to
      // three per row — is added when the column comes.  The owner of a column writes R1 before the barrier.
      const int rblk = (W - l) % SF_PFL_RB, r0 = l + rblk;
      const int nblk = sfd_max(l - 10 + 7, 0) >> 3;  // blocks of eight terms of the longest row's multiloop sum
      const int mlblk1 = sfd_min(sfd_max(l - SF_PFL_MLS0, 0) / SF_PFL_MLS1, nblk >> 1);
      const int mlsplit = 6 + 8 * (nblk - mlblk1);  // team 3: a < mlsplit, team 1: the rest
      auto r1_block = [&](const int kr, const int jb) {
        const int ln = tid & 63;
        double g[2][2];  // qm[l-jb-jj, c], column c = lane + 64 h (0 where the entry does not exist)
#pragma unroll
        for (int jj = 0; jj < 2; jj++)
#pragma unroll
          for (int h = 0; h < 2; h++) {
            const int r = l - jb - jj, cq = ln + 64 * h, dd = cq - r;
            g[jj][h] = (r >= 1 && dd >= SFD_TURN + 1 && cq <= W) ? QMD(dd, r) : 0.0;
          }
        racc[0] = racc[1] = 0.0;
        const double *fW = FAC + 1250 + SF_PK_CODE(FWD[sfd_min(kr, W)]);
        const double *qp = QB + kr - 1;
        // four columns m per trip, their reads issued together (one m per trip was two dependent LDS round trips per term).  The
        // columns m-1 < 64 and >= 64 as two loops: which half of the rows of qm a term reads is then a compile-time constant (as a
        // test per term it compiled to ~30 branches per trip).
        auto half = [&](auto HH, const int mlo, const int mhi) {
          constexpr int h = decltype(HH)::value;
          if (mlo > mhi) return;
          // (whole trips without any clamping, the last — partial — one with; the row of a term's weight in its table comes from
          // a lane table, tbw: as an LDS read it was a round trip ahead of the read of the weight)
          int m = mlo, coff = COFF(mlo);
          for (; m + 3 <= mhi; m += 4) {
            double q[4], f[4];
#pragma unroll
            for (int t = 0; t < 4; t++) {
              q[t] = qp[coff];
              coff += m + t - 4;  // COFF(x+1) - COFF(x) = x - 4
              f[t] = *(const double *)((const char *)fW + sf_lane_read_i32(tbw[h], m + t - 1 - 64 * h));
            }
            // (the trip's eight values of qm into scalar registers first, then the arithmetic: read where they are used, every
            // v_readlane pair was followed by a hazard no-op before the multiply-add that takes it)
            double gv[4][2];
#pragma unroll
            for (int t = 0; t < 4; t++) {
              gv[t][0] = sf_lane_read_f64(g[0][h], m + t - 1 - 64 * h);
              gv[t][1] = sf_lane_read_f64(g[1][h], m + t - 1 - 64 * h);
            }
            SF_SCHED_FENCE();
#pragma unroll
            for (int t = 0; t < 4; t++) {
              const double w = q[t] * f[t];
              racc[0] += w * gv[t][0];
              racc[1] += w * gv[t][1];
            }
          }
          if (m <= mhi) {
            double q[4], f[4];
#pragma unroll
            for (int t = 0; t < 4; t++) {
              const int mt = sfd_min(m + t, mhi);
              q[t] = qp[COFF(mt)];
              f[t] = *(const double *)((const char *)fW + sf_lane_read_i32(tbw[h], mt - 1 - 64 * h));
            }
#pragma unroll
            for (int t = 0; t < 4; t++) {
              const double w = (m + t <= mhi) ? q[t] * f[t] : 0.0;
              const int cq = sfd_min(m + t, mhi) - 1 - 64 * h;
              racc[0] += w * sf_lane_read_f64(g[0][h], cq);
              racc[1] += w * sf_lane_read_f64(g[1][h], cq);
            }
          }
        };
        half(std::integral_constant<int, 0>{}, l + 1, sfd_min(W, 64));
        half(std::integral_constant<int, 1>{}, sfd_max(l + 1, 65), W);
      };
      {
        static_assert(SF_PFL_RB == 8, "four teams x two columns of a block");
        const int kf = (tid & (SF_PFL_SLOTS - 1)) + 1;
        // (the columns with terms left over go to the teams with the lighter roles)
        const int jb = team == 0 ? 4 : (team == 1 ? 6 : (team == 2 ? 2 : 0));
        if (rblk == 0) r1_block(kf, jb);
        if ((rblk >> 1) == (jb >> 1)) {
          double r1v = (rblk & 1) ? racc[1] : racc[0];
          if (rblk >= 5) {
            // the one to three terms m = l+5 .. r0 the block could not contain: their reads issued together, the row of the weight
            // from the lane table (as a loop with an LDS read of the neighbour code per term: two dependent round trips per term in
            // the columns where this team is the last to arrive)
            const double *fW = FAC + 1250 + SF_PK_CODE(FWD[sfd_min(kf, W)]);
            double q[3], f[3], gq[3];
#pragma unroll
            for (int t = 0; t < 3; t++) {
              const int mt = sfd_min(l + SFD_TURN + 2 + t, r0);
              const int o0 = sf_lane_read_i32(tbw[0], (mt - 1) & 63), o1 = sf_lane_read_i32(tbw[1], (mt - 1) & 63);
              q[t] = QBC(kf, mt);
              f[t] = *(const double *)((const char *)fW + (((mt - 1) >> 6) ? o1 : o0));
              gq[t] = QMD(mt - 1 - l, l);
            }
            double r1 = 0.0;
#pragma unroll
            for (int t = 0; t < 3; t++) r1 += (l + SFD_TURN + 2 + t <= r0) ? q[t] * f[t] * gq[t] : 0.0;
            r1v += r1;
          }
          if (kf <= l - SFD_TURN - 1) R1n[kf] = r1v;
        }
      }
      // (k,l) as a stem of a multiloop closed by (i,m), i < k, m > l: the terms a = k - i in [alo, ahi), eight closers per trip;
      // the overshoot reads rows <= 0 of R1 / R01, which are 0
      auto ml_part = [&](const int alo, const int ahi) -> double {
        double ms = 0.0, ms2 = 0.0;
        const double *qmr = QM + k, *r1p = R1c + k, *r01p = R01c + k;
        int off = DOFF(alo - 2) - alo, st = W - alo + 1;  // DOFF(a-2) - a and its increment
        for (int a = alo; a <= k - 1 && a < ahi; a += 8) {
          double t0 = 0.0, t1 = 0.0;
          double mb[8], rv[8], qv[8], sv8[8];
#pragma unroll
          for (int t = 0; t < 8; t++) {
            mb[t] = MLB[a + t - 1]; rv[t] = r1p[-a - t]; qv[t] = qmr[off]; sv8[t] = r01p[-a - t];
            off += st--;
          }
          SF_SCHED_FENCE();
#pragma unroll
          for (int t = 0; t < 8; t += 2) {
            t0 += mb[t] * rv[t] + qv[t] * sv8[t];
            t1 += mb[t + 1] * rv[t + 1] + qv[t + 1] * sv8[t + 1];
          }
          ms += t0;
          ms2 += t1;
        }
        return ms + ms2;
      };
      // The special loops 1x1 .. 2x3 with (k,l) as the inner pair (six of the nine candidates): team 0's in the late columns (handing
      // them to team 2 there: measured, no gain), team 1's in the early ones (l <= LSP), where team 0's own role is the longest of the four (per-team stamps: 5.5 against 3.2 - 4.7 k
      // cycles per column) and team 1 has neither long rows nor a share of the multiloop sums.
      const bool spx = l <= SF_PFL_LSP;
      auto sp_tail = [&](const int type) -> double {
        const int rt = sfd_rtype(type);
        const int sp1 = S[k - 1], sq1 = S[l + 1];
        const int k2 = sfd_max(k - 2, 1), k3 = sfd_max(k - 3, 1), k4 = sfd_max(k - 4, 1);
        const int l2 = sfd_min(l + 2, W), l3 = sfd_min(l + 3, W), l4 = sfd_min(l + 4, W);
        const bool e2k = k - 2 >= 1, e3k = k - 3 >= 1, e4k = k - 4 >= 1;
        const bool e2l = l + 2 <= W, e3l = l + 3 <= W, e4l = l + 4 <= W;
        const int sk2 = S[k2], sk3 = S[k3], sl2 = S[l2], sl3 = S[l3];  // neighbours towards the loop
        const double w11 = X->int11[PAIR(S[k2], S[l2])][rt][sp1][sq1];
        const double w12 = X->int21[PAIR(S[k2], S[l3])][rt][sp1][sq1][sl2];  // u1 = 1, u2 = 2
        const double w21 = X->int21[rt][PAIR(S[k3], S[l2])][sq1][sk2][sp1];  // u1 = 2, u2 = 1
        const double w22 = X->int22[PAIR(S[k3], S[l3])][rt][sk2][sp1][sq1][sl2];
        const double w23o = X->mismatch23I[rt][sq1][sp1];
        const double w23 = X->mismatch23I[PAIR(S[k3], S[l4])][sk2][sl3], w32 = X->mismatch23I[PAIR(S[k4], S[l3])][sk3][sl2];
        const double q11 = QBC(k2, l2), q12 = QBC(k2, l3), q21 = QBC(k3, l2), q22 = QBC(k3, l3), q23 = QBC(k3, l4), q32 = QBC(k4, l3);
        double t = ((e2k && e2l) ? q11 : 0.0) * w11;
        t += ((e2k && e3l) ? q12 : 0.0) * w12 + ((e3k && e2l) ? q21 : 0.0) * w21;
        t += ((e3k && e3l) ? q22 : 0.0) * w22;
        t += (WIL[5] * WN[1] * w23o) * (((e3k && e4l) ? q23 : 0.0) * w23 + ((e4k && e3l) ? q32 : 0.0) * w32);
        return t;
      };
      if (team == 0) {
        if (valid) {
          // (the cell's weights first: a dozen gathers from device memory, in flight while the recurrence below runs — they used to
          // be fetched after it, an L2 round trip in the longest team's column)
          const int type = OWN(k, l);
          // exterior term, then the small special loops with (k,l) as the INNER pair — branch-free as in the
          // inside pass: rows / columns clamped to existing ones, enclosing pairs that do not exist dropped by
          // a select, all weights fetched up front
          const int rt = sfd_rtype(type);
          const int sp1 = S[k - 1], sq1 = S[l + 1];
          const int k1 = sfd_max(k - 1, 1), k2 = sfd_max(k - 2, 1);
          const int l1 = sfd_min(l + 1, W), l2 = sfd_min(l + 2, W);
          const bool e1 = inner, e2k = k - 2 >= 1, e2l = l + 2 <= W;
          const double w00 = X->stack[PAIR(S[k1], S[l1])][rt];
          const double w01 = X->stack[PAIR(S[k1], S[l2])][rt], w10 = X->stack[PAIR(S[k2], S[l1])][rt];
          const double tail0 = spx ? 0.0 : sp_tail(type);
          const double wI = X->mismatchI[rt][sq1][sp1];
          const double wx = sfx_extloop(X, type, k > 1 ? sp1 : -1, l < W ? sq1 : -1);
          const double *dI3 = DERP(0, l + 3) + k;
          const int kr3 = r3 ? k - 3 : 1;  // row for speculative reads
          const double *fI = FAC + SF_PK_CODE(FWD[kr3]);
          const double *qbA = QB + kr3 - 1;
          if (!inner) {
#pragma unroll
            for (int u = 0; u < 27; u++) H[u] = 0.0;
          } else {
            // (reads of a batch of sizes are issued together, then used: left to itself the compiler waits for almost
            // every LDS read on the spot — ~40 round trips in this block alone, and the column is nothing but such chains)
            SF_PFL_BATCHES(30, 6, 9, { ca[t] = SF_LANE_GET(tcol, u - 1); pa[t] = SF_PK_ROW(SF_LANE_GET(tpk, u - 1)); }, {
              qa[t] = qbA[ca[t]]; fa[t] = fI[pa[t]];
              da[t] = dI3[1 - u]; wa[t] = WN[u - 4];
            }, {
              const double a = qa[t] * fa[t];
              const double e1 = r3 ? a : 0.0;  // u1 = 2
              H[u - 4] = H[u - 6] + (e1 + da[t]) * wa[t];  // da: u2 = 2
            })
            {
              const double a = qbA[SF_LANE_GET(tcol, 4)] * fI[SF_PK_ROW(SF_LANE_GET(tpk, 4))];
              const double e1 = r3 ? a : 0.0;
              H[1] = (e1 + dI3[-4]) * WN[1];
            }
            H[0] = dI3[-3] * WN[0];
          }
          const double q00 = QBC(k1, l1), q01 = QBC(k1, l2), q10 = QBC(k2, l1);
          double o = q5[k - 1] * q3[l + 1] * wx;
          {
            double oi = q00 * w00;
            oi += ((e2l ? q01 : 0.0) * w01 + (e2k ? q10 : 0.0) * w10) * WB[1];
            oi += tail0;
            double gg = 0.0, gg2 = 0.0;
#pragma unroll
            for (int u = 6; u <= 30; u += 2) {
              gg += H[u - 4] * WIL[u];
              if (u + 1 <= 30) gg2 += H[u - 3] * WIL[u + 1];
            }
            oi += (gg + gg2) * wI;
            if (e1) o += oi;
          }
          if (!type) o = 0.0;
          ZP[k] = o;
        }
      } else if (team == 1) {
        if (valid) {
          const int type = OWN(k, l);
          typeC = type;
          fc0 = FAC[1250 + SF_PK_ROW(BWD[l]) + SF_PK_CODE(FWD[k])];
          fc1 = R0c[k];
          const int sp1 = S[k - 1];
          const double wst = mlblk1 ? sfx_mlstem(X, type, sp1, S[l + 1]) : 0.0;  // (device memory: in flight during the sums)
          const double tail1 = spx ? sp_tail(type) : 0.0;
          const double *dB1 = DERP(2, l + 1) + k - 1;
          const double *qbA = QB + (k > 1 ? k - 2 : 0);  // row k-1 (row 1 for speculative reads)
          double gb = 0.0, gb2 = 0.0;
          SF_PFL_BATCHES(30, 2, 10, { ca[t] = SF_LANE_GET(tcol, 33 - u); pa[t] = SF_PK_NT(SF_LANE_GET(tpk, 33 - u)); }, {
            const int u_ = 32 - u;  // ascending sizes 2..30, as the sums were always accumulated
            qa[t] = qbA[ca[t]]; da[t] = dB1[-u_]; wa[t] = WB[u_]; fa[t] = 0.0;
          }, {
            const int u_ = 32 - u;
            const double ab = qa[t] * ((sp1 * pa[t] == 6) ? 1.0 : xTAU);
            const double tt = (ab + da[t]) * wa[t];  // u1 = 0, u2 = 0
            if (u_ & 1) gb2 += tt; else gb += tt;
          })
          ZP[VW + k] = (gb + gb2) * (type > 2 ? xTAU : 1.0) + tail1;  // rtype(type) > 2 <=> type > 2
          // (wave-uniform: the late columns only)
          ZP[8 * VW + k] = mlblk1 ? ml_part(mlsplit, W) * wst : 0.0;
        }

      } else if (team == 2) {
        if (valid) {
          const int type = OWN(k, l);
          typeC = type;
          {
            const int fa = SF_PK_ROW(BWD[l]) + SF_PK_CODE(FWD[k]);  // (S[l], S[l-1]) x (S[k], S[k+1])
            fc0 = FAC[fa];
            fc1 = FAC[625 + fa];
          }
          const int rt = sfd_rtype(type);
          const int sp1 = S[k - 1], sq1 = S[l + 1];
          const double w1n = X->mismatch1nI[rt][sq1][sp1];  // device memory: issued before the long sums below
          const double *d1N2 = DERP(1, l + 2) + k;
          const int kr2 = r2 ? k - 2 : 1;
          const double *f1N = FAC + 625 + SF_PK_CODE(FWD[kr2]);
          const double *qbA = QB + kr2 - 1;
          double g1 = 0.0, g2 = 0.0;
          SF_PFL_BATCHES(30, 4, 9, { ca[t] = SF_LANE_GET(tcol, 34 - u); pa[t] = SF_PK_ROW(SF_LANE_GET(tpk, 34 - u)); }, {
            const int u_ = 34 - u;  // ascending sizes 4..30
            qa[t] = qbA[ca[t]]; fa[t] = f1N[pa[t]];
            da[t] = d1N2[-u_]; wa[t] = WIL1N[u_];
          }, {
            const int u_ = 34 - u;
            const double an = qa[t] * fa[t];
            const double n1 = r2 ? an : 0.0;  // u1 = 1
            const double tt = (n1 + da[t]) * wa[t];  // da: u2 = 1
            if (u_ & 1) g2 += tt; else g1 += tt;
          })
          ZP[2 * VW + k] = (g1 + g2) * w1n;
        }
      } else {
        if (valid) {
          const int type = OWN(k, l);
          typeC = type;
          const int sp1 = S[k - 1], sq1 = S[l + 1];
          const double wst = sfx_mlstem(X, type, sp1, sq1);  // (device memory: in flight during the sums)
          double ms = 0.0;
          for (int a = 1; a <= sfd_min(k - 1, 5); a++) ms += MLB[a - 1] * R1c[k - a];
          ms += ml_part(6, mlsplit);
          ZP[3 * VW + k] = ms * wst;
        }
      }
      __syncthreads();
      SF_LANE_TABLE_LOAD(tpk, L, BWD[sfd_min(l - 1 + L, W)]);
      SF_LANE_TABLE_LOAD(tcol, L, l - 1 + L <= W ? COFF(l - 1 + L) : ZOFF);
      // The column's results, shared out: team 2 writes the tables the next column's loops read, team 1 the multiloop vectors,
      // team 3 the pair probability and what hangs on it (each forms the cell's sum from the partial sums itself; qb[k,l],
      // which team 2 overwrites here, was read before the barrier).
      if (team != 0 && valid) {
        const int type = typeC;
        double o = 0.0;
        if (type && qbkl != 0.0) o = inner ? (ZP[k] + ZP[VW + k]) + (ZP[2 * VW + k] + (ZP[3 * VW + k] + ZP[8 * VW + k])) : ZP[k];
        if (team == 2) {
          QBC(k, l) = o;
          DERP(0, l)[k] = type ? o * fc0 : 0.0;
          DERP(1, l)[k] = type ? o * fc1 : 0.0;
          DERP(2, l)[k] = (type > 2) ? o * xTAU : o;
        } else if (team == 1) {
          const double w = type ? o * fc0 : 0.0;
          const double r0 = w + xMLbase * fc1;
          const double r1 = R1n[k];  // (written before the barrier by the thread that owns row k)
          R0n[k] = r0;
          R01n[k] = r0 + r1;
        } else {
          const double p = o * qbkl / Z;
          mbd += p * (1.0 - p);
          if (p > 0.5) {
            cd += 1.0 - p;
            if (centroid) { centroid[(size_t)fold * W1 + k - 1] = '('; centroid[(size_t)fold * W1 + l - 1] = ')'; }
          } else cd += p;
        }
      }
      SF_LANE_TABLE_PIN(tpk);
      SF_LANE_TABLE_PIN(tcol);
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
#undef PAIR
#undef OWN
}

static inline hipError_t sf_pfl_configure() {
  hipError_t e = hipFuncSetAttribute((const void *)sf_pf_lds_kernel<120, false>, hipFuncAttributeMaxDynamicSharedMemorySize, SF_PFL_LDS_LIMIT);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute((const void *)sf_pf_lds_kernel<120, true>, hipFuncAttributeMaxDynamicSharedMemorySize, SF_PFL_LDS_LIMIT);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute((const void *)sf_pf_lds_kernel<0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, SF_PFL_LDS_LIMIT);
  if (e != hipSuccess) return e;
  e = hipFuncSetAttribute((const void *)sf_pf_lds_kernel<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, SF_PFL_LDS_LIMIT);
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute((const void *)sf_pf_lds_kernel<0, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, SF_PFL_LDS_LIMIT);
}

// constrained folds: stand-alone folds only (a window's constraint slice changes which cells exist, so consecutive windows
// do not share their inside tables)
template <typename... A>
static inline void sf_pf_lds_launch_hc(int grid, int W, hipStream_t st, A... args) {
  SF_LAUNCH((sf_pf_lds_kernel<0, false, true>), grid, SF_PFL_NT, sf_pfl_lds_bytes(W, true), st, args...);
}
template <typename... A>
static inline void sf_pf_lds_launch(int grid, int W, bool shared, hipStream_t st, A... args) {
  const size_t lds = sf_pfl_lds_bytes(W);
  if (W == 120 && shared) SF_LAUNCH((sf_pf_lds_kernel<120, true>), grid, SF_PFL_NT, lds, st, args...);
  else if (W == 120) SF_LAUNCH((sf_pf_lds_kernel<120, false>), grid, SF_PFL_NT, lds, st, args...);
  else if (shared) SF_LAUNCH((sf_pf_lds_kernel<0, true>), grid, SF_PFL_NT, lds, st, args...);
  else SF_LAUNCH((sf_pf_lds_kernel<0, false>), grid, SF_PFL_NT, lds, st, args...);
}
