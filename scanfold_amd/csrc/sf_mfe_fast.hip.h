// sf_mfe_fast.hip.h — the hot kernel: batched Zuker MFE fill (energy only) with LDS-resident int16 tables.
//
// Replaces energies(seq_list) -> rna_folder -> RNA.fold(seq) (ScanFold-Scan.py:244-246,253-262;
// ScanFoldFunctions.py:774-789,805-814): r+1 folds per window whose structures the caller throws away.
//
// One workgroup folds one sequence at a time (persistent grid, sequences dealt round-robin).  Thread t owns
// cell (i = t+1, j = i+d) of anti-diagonal d; one barrier per diagonal.  LDS holds, per workgroup:
//   fML   int16, full triangle, diagonal-major      (the O(W^3) multiloop split reads every diagonal)
//   CI, C1N, CB  int16, rolling window of SF_FAST_NR diagonals — an interior loop reaches at most MAXLOOP+2
//         diagonals inwards, so older diagonals of c are dead.  They hold c pre-added with the inner
//         pair's own terms so the 496-candidate search is one LDS read + add + min per candidate:
//           CI  = c + mismatchI [rtype][S[j+1]][S[i-1]]   generic loops
//           C1N = c + mismatch1nI[rtype][S[j+1]][S[i-1]]  1 x n loops
//           CB  = c + TerminalAU(rtype)                   bulges (and, minus that term, the few special loops)
//   DML   rolling 3 diagonals of min_k fML[i,k]+fML[k+1,j]
// The size-dependent part of a candidate (loop initiation + asymmetry) is the same for every thread of a
// diagonal, so it comes from scalar loads (SfFastParams).  c itself is streamed to a device scratch table
// (int16, 2 B per cell, coalesced) for the exterior-loop pass at the end.
//
// int16 is exact while |energy| < 12000 dcal/mol; a fold that leaves that range (a >120 kcal/mol helix) is
// appended to an overflow list and redone by the int32 kernel (sf_mfe_full.hip.h), so results never depend
// on the storage width.  INF is SF_INF16; any sum above SF_FAST_THRESH means "no structure" — parameter
// sets whose entries exceed SF_FAST_MAXPARAM in magnitude are routed to the int32 kernel entirely.
#pragma once
#include "sf_energy.h"

#define SF_FAST_NR 33
#define SF_INF16 30000
#define SF_FAST_THRESH 10000
#define SF_FAST_OVF (-12000)
#define SF_FAST_MAXPARAM 2500
#define SF_FAST_MAXW 256

struct SfFastParams {
  int32_t LT[31][32];  // [u][u1] generic interior: internal_loop[u] + min(max_ninio, |u - 2*u1| * ninio)
  int32_t L1N[32];     // [n] 1 x n: internal_loop[n+1] + min(max_ninio, (n-1) * ninio)
  int32_t BUL[32];     // [n] bulge[n]
  int32_t L23;         // internal_loop[5] + ninio
  int32_t fast_ok;     // parameter magnitudes allow int16 storage
};

static inline void sf_fast_build_params(const SfDevParams &D, SfFastParams &F) {
  const sf_params_blob &P = D.P;
  memset(&F, 0, sizeof F);
  for (int u = 0; u <= 30; u++)
    for (int u1 = 0; u1 <= u && u1 < 32; u1++) {
      const int a = u - 2 * u1 < 0 ? 2 * u1 - u : u - 2 * u1;
      const int nin = a * P.ninio < P.max_ninio ? a * P.ninio : P.max_ninio;
      F.LT[u][u1] = P.internal_loop[u] + nin;
    }
  for (int n = 0; n <= 29; n++) {
    const int nin = (n - 1) * P.ninio < P.max_ninio ? (n - 1) * P.ninio : P.max_ninio;
    F.L1N[n] = P.internal_loop[n + 1] + nin;
  }
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
}

// LDS carve (bytes); every piece a multiple of 4
struct SfFastLayout {
  int tri;      // int16 entries of the fML triangle (diagonals >= 4)
  int off_ci, off_c1n, off_cb, off_dml, off_f5, off_red, off_flag, off_S;
  int total;
};
static inline __host__ __device__ SfFastLayout sf_fast_layout(int W) {
  SfFastLayout L;
  const int nd = W - 4;  // diagonals 4..W-1
  int tri = nd > 0 ? nd * W - (W * (W - 1) / 2 - 6) + 0 : 0;  // sum_{d=4}^{W-1} (W-d)
  if (tri < 0) tri = 0;
  tri = (tri + 1) & ~1;
  L.tri = tri;
  int o = tri * 2;
  const int roll = ((SF_FAST_NR * W + 1) & ~1) * 2;
  L.off_ci = o; o += roll;
  L.off_c1n = o; o += roll;
  L.off_cb = o; o += roll;
  L.off_dml = o; o += ((3 * W + 1) & ~1) * 2;
  L.off_f5 = o; o += (W + 1) * 4;
  L.off_red = o; o += 8 * 4;
  L.off_flag = o; o += 4;
  L.off_S = o; o += (W + 2 + 3) & ~3;
  L.total = o;
  return L;
}

static inline bool sf_fast_w_supported(int W) { return W >= 8 && W <= SF_FAST_MAXW; }
static inline int sf_fast_threads(int W) { return W <= 128 ? 128 : 256; }

template <int NT>
__global__ __launch_bounds__(NT) void sf_mfe_fast_kernel(const uint8_t *__restrict__ seqs, int n, int W,
                                                         const SfDevParams *__restrict__ D,
                                                         const SfFastParams *__restrict__ F,
                                                         int16_t *__restrict__ cg_all, int32_t *__restrict__ out,
                                                         int *__restrict__ ovf_cnt, int *__restrict__ ovf_list) {
  SF_DYN_SMEM(smem);
  const SfFastLayout Lo = sf_fast_layout(W);
  int16_t *fML = (int16_t *)smem;
  int16_t *CI = (int16_t *)(smem + Lo.off_ci);
  int16_t *C1N = (int16_t *)(smem + Lo.off_c1n);
  int16_t *CB = (int16_t *)(smem + Lo.off_cb);
  int16_t *DMLr = (int16_t *)(smem + Lo.off_dml);
  int32_t *f5s = (int32_t *)(smem + Lo.off_f5);
  int32_t *red = (int32_t *)(smem + Lo.off_red);
  int32_t *flag = (int32_t *)(smem + Lo.off_flag);
  uint8_t *S = (uint8_t *)(smem + Lo.off_S);

  const int tid = threadIdx.x;
  const sf_params_blob &P = D->P;
  int16_t *cg = cg_all + (size_t)blockIdx.x * W * W;  // c[d][i0] for the exterior pass
  const int TAU = P.TerminalAU;
  const int MLbase = P.MLbase, MLclosing = P.MLclosing;
// fML triangle without diagonals 0..3: base(d) = sum_{k=4}^{d-1} (W-k)
#define FBASE(d) (((d)-4) * W - ((d) * ((d)-1) / 2 - 6))
#define RB(dd) (((dd) % SF_FAST_NR) * W)

  for (int seq = blockIdx.x; seq < n; seq += gridDim.x) {
    const uint8_t *src = seqs + (size_t)seq * W;
    __syncthreads();
    for (int x = tid; x < W; x += NT) S[x + 1] = sf_encode_nt(src[x]);
    if (tid == 0) { S[0] = 0; S[W + 1] = 0; f5s[0] = 0; flag[0] = 0; }
    for (int x = tid; x < 3 * W; x += NT) DMLr[x] = SF_INF16;  // diagonals 2,3 have no multiloop split
    __syncthreads();
    int ovf = 0;

    for (int d = SFD_TURN + 1; d < W; d++) {
      const int i = tid + 1, j = i + d, i0 = tid;
      if (j <= W) {
        const int type = D->pair[S[i]][S[j]];
        int c = SF_INF16;
        if (type) {
          int e = sfd_hairpin(D, S, i, j, type);
          const int umax = sfd_min(SFD_MAXLOOP, d - 2 - (SFD_TURN + 1));
          const int si1 = S[i + 1], sj1 = S[j - 1];
          if (umax >= 0) {
            const int tau_out = type > 2 ? TAU : 0;
            {  // stack
              const int t2r = sfd_rtype(D->pair[si1][sj1]);
              const int cc = CB[RB(d - 2) + i0 + 1] - (t2r > 2 ? TAU : 0);
              e = sfd_min(e, cc + P.stack[type][t2r]);
            }
            if (umax >= 1) {  // bulges of one nucleotide keep the stack
              const int b1 = F->BUL[1];
              const int ta = sfd_rtype(D->pair[si1][S[j - 2]]);  // (i+1, j-2)
              const int ca = CB[RB(d - 3) + i0 + 1] - (ta > 2 ? TAU : 0);
              e = sfd_min(e, ca + b1 + P.stack[type][ta]);
              const int tb = sfd_rtype(D->pair[S[i + 2]][sj1]);  // (i+2, j-1)
              const int cb = CB[RB(d - 3) + i0 + 2] - (tb > 2 ? TAU : 0);
              e = sfd_min(e, cb + b1 + P.stack[type][tb]);
            }
            // longer bulges: c + TerminalAU(inner) is pre-added
            for (int nn = 2; nn <= umax; nn++) {
              const int add = F->BUL[nn] + tau_out;
              const int rb = RB(d - 2 - nn);
              const int v1 = CB[rb + i0 + 1];       // u1 = 0, u2 = nn : (i+1, j-1-nn)
              const int v2 = CB[rb + i0 + 1 + nn];  // u1 = nn, u2 = 0 : (i+1+nn, j-1)
              e = sfd_min(e, sfd_min(v1, v2) + add);
            }
            if (umax >= 2) {  // 1 x 1
              const int t2r = sfd_rtype(D->pair[S[i + 2]][S[j - 2]]);
              const int cc = CB[RB(d - 4) + i0 + 2] - (t2r > 2 ? TAU : 0);
              e = sfd_min(e, cc + P.int11[type][t2r][si1][sj1]);
            }
            if (umax >= 3) {  // 1 x 2 and 2 x 1
              const int ta = sfd_rtype(D->pair[S[i + 2]][S[j - 3]]);  // u1=1,u2=2: (i+2, j-3), sq1 = S[j-2]
              const int ca = CB[RB(d - 5) + i0 + 2] - (ta > 2 ? TAU : 0);
              e = sfd_min(e, ca + P.int21[type][ta][si1][S[j - 2]][sj1]);
              const int tb = sfd_rtype(D->pair[S[i + 3]][S[j - 2]]);  // u1=2,u2=1: (i+3, j-2), sp1 = S[i+2]
              const int cb = CB[RB(d - 5) + i0 + 3] - (tb > 2 ? TAU : 0);
              e = sfd_min(e, cb + P.int21[tb][type][sj1][si1][S[i + 2]]);
            }
            if (umax >= 4) {
              {  // 2 x 2: (i+3, j-3)
                const int t2r = sfd_rtype(D->pair[S[i + 3]][S[j - 3]]);
                const int cc = CB[RB(d - 6) + i0 + 3] - (t2r > 2 ? TAU : 0);
                e = sfd_min(e, cc + P.int22[type][t2r][si1][S[i + 2]][S[j - 2]][sj1]);
              }
              // 1 x n and n x 1, n >= 3 (total size n+1 <= umax)
              const int m1 = P.mismatch1nI[type][si1][sj1];
              for (int nn = 3; nn + 1 <= umax; nn++) {
                const int add = F->L1N[nn] + m1;
                const int rb = RB(d - 3 - nn);
                const int v1 = C1N[rb + i0 + 2];       // u1 = 1, u2 = nn : (i+2, j-1-nn)
                const int v2 = C1N[rb + i0 + 1 + nn];  // u1 = nn, u2 = 1 : (i+1+nn, j-2)
                e = sfd_min(e, sfd_min(v1, v2) + add);
              }
            }
            if (umax >= 5) {  // 2 x 3 and 3 x 2
              const int m23 = P.mismatch23I[type][si1][sj1] + F->L23;
              const int ta = sfd_rtype(D->pair[S[i + 3]][S[j - 4]]);  // u1=2,u2=3: (i+3, j-4); sp1=S[i+2], sq1=S[j-3]
              const int ca = CB[RB(d - 7) + i0 + 3] - (ta > 2 ? TAU : 0);
              e = sfd_min(e, ca + m23 + P.mismatch23I[ta][S[j - 3]][S[i + 2]]);
              const int tb = sfd_rtype(D->pair[S[i + 4]][S[j - 3]]);  // u1=3,u2=2: (i+4, j-3); sp1=S[i+3], sq1=S[j-2]
              const int cb = CB[RB(d - 7) + i0 + 4] - (tb > 2 ? TAU : 0);
              e = sfd_min(e, cb + m23 + P.mismatch23I[tb][S[j - 2]][S[i + 3]]);
            }
            if (umax >= 6) {  // generic loops: both sides >= 2, not 2x2 / 2x3 / 3x2
              int g = SF_INF16 * 2;
              for (int u = 6; u <= umax; u++) {
                const int rb = RB(d - 2 - u) + i0 + 1;
                const int32_t *lt = F->LT[u];
                for (int u1 = 2; u1 <= u - 2; u1++) g = sfd_min(g, CI[rb + u1] + lt[u1]);
              }
              e = sfd_min(e, g + P.mismatchI[type][si1][sj1]);
            }
          }
          // multiloop closed by (i,j)
          const int dml = DMLr[((d - 2) % 3) * W + i0 + 1];
          e = sfd_min(e, dml + sfd_mlstem(D, sfd_rtype(type), sj1, si1) + MLclosing);
          c = e;
          if (c < SF_FAST_OVF) ovf = 1;
        }
        // publish the cell
        const int rbd = RB(d) + i0;
        int f = SF_INF16 * 2;
        if (type) {
          const int tr = sfd_rtype(type);
          const int sp1 = S[i - 1], sq1 = S[j + 1];
          CI[rbd] = (int16_t)(c + P.mismatchI[tr][sq1][sp1]);
          C1N[rbd] = (int16_t)(c + P.mismatch1nI[tr][sq1][sp1]);
          CB[rbd] = (int16_t)(c + (tr > 2 ? TAU : 0));
          f = c + sfd_mlstem(D, type, i > 1 ? sp1 : -1, j < W ? sq1 : -1);
        } else {
          CI[rbd] = SF_INF16; C1N[rbd] = SF_INF16; CB[rbd] = SF_INF16;
        }
        cg[d * W + i0] = (int16_t)c;
        // fML[i,j]
        if (d > SFD_TURN + 1) {
          const int fb = FBASE(d - 1);
          f = sfd_min(f, sfd_min(fML[fb + i0 + 1], fML[fb + i0]) + MLbase);
        }
        int dec = SF_INF16 * 2;
        for (int m = SFD_TURN + 1; m <= d - SFD_TURN - 2; m++)
          dec = sfd_min(dec, fML[FBASE(m) + i0] + fML[FBASE(d - m - 1) + i0 + m + 1]);
        f = sfd_min(f, dec);
        if (f < SF_FAST_OVF) ovf = 1;
        DMLr[(d % 3) * W + i0] = (int16_t)(dec > SF_FAST_THRESH ? SF_INF16 : dec);
        fML[FBASE(d) + i0] = (int16_t)(f > SF_FAST_THRESH ? SF_INF16 : f);
      }
      __syncthreads();
    }

    // exterior loop (same recurrence as sf_mfe_full_kernel)
    if (ovf) flag[0] = 1;
    for (int j = 1; j <= W; j++) {
      int v = SF_INF16 * 4;
      const int i = tid + 1;
      if (i + SFD_TURN + 1 <= j) {
        const int type = D->pair[S[i]][S[j]];
        if (type) v = f5s[i - 1] + cg[(j - i) * W + i - 1] + sfd_extloop(D, type, i > 1 ? S[i - 1] : -1, j < W ? S[j + 1] : -1);
      }
      v = sf_block_min(v, red);
      if (tid == 0) f5s[j] = sfd_min(f5s[j - 1], v);
      __syncthreads();
    }
    if (tid == 0) {
      out[seq] = f5s[W];
      if (flag[0] || f5s[W] < SF_FAST_OVF) {
        const int k = atomicAdd(ovf_cnt, 1);
        ovf_list[k] = seq;
      }
    }
  }
#undef FBASE
#undef RB
}

static inline hipError_t sf_fast_configure() {
  hipError_t e = hipFuncSetAttribute((const void *)sf_mfe_fast_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute((const void *)sf_mfe_fast_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

// grid / LDS / scratch for n folds of W nt on a chip with n_cu CUs
static inline void sf_fast_geometry(int W, int n_cu, int n, int *grid, int *threads, size_t *lds, size_t *scratch) {
  const SfFastLayout L = sf_fast_layout(W);
  const int nt = sf_fast_threads(W);
  int per_cu = (160 * 1024) / L.total;
  const int by_waves = 32 / (nt / 64);
  if (per_cu > by_waves) per_cu = by_waves;
  if (per_cu > 8) per_cu = 8;
  if (per_cu < 1) per_cu = 1;
  long long gsz = (long long)n_cu * per_cu;
  if (gsz > n) gsz = n;
  *grid = (int)gsz;
  *threads = nt;
  *lds = (size_t)L.total;
  *scratch = (size_t)gsz * W * W * sizeof(int16_t);
}

template <typename... A>
static inline void sf_fast_launch(int grid, int threads, size_t lds, hipStream_t st, A... args) {
  if (threads == 128) SF_LAUNCH(sf_mfe_fast_kernel<128>, grid, 128, lds, st, args...);
  else SF_LAUNCH(sf_mfe_fast_kernel<256>, grid, 256, lds, st, args...);
}
