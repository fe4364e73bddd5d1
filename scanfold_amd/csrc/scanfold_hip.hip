// scanfold_hip.hip — host side of libscanfold_hip.so: the C ABI of include/scanfold_hip.h over the HIP kernels.
//
// The reference drives this path from Python with a 12-process pool created per call
// (ScanFold-Scan.py:73-77,256,274); here one process owns one GPU and every call is a few batched launches
// on one HIP stream.  No CPU compute path exists in this file: without a GPU sf_init fails.
#include <math.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <string>
#include <vector>

#include "sf_launch.h"
#include "sf_energy.h"
#include "sf_mfe_full.hip.h"
#include "sf_mfe_fast.hip.h"
#include "sf_pf.hip.h"
#include "sf_pf_fast.hip.h"
#include "sf_pf_lds.hip.h"
#include "sf_shuffle.hip.h"
#include "sf_tabulate.hip.h"

namespace {

struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
};

struct Ctx {
  bool init = false;
  bool have_params = false;
  int dev = 0;
  int n_cu = 0;
  std::string dev_name;
  hipStream_t stream = nullptr;
  SfDevParams *dP = nullptr;  // the resident model the launches use: one of the two slots below
  SfDevParamsPF *dX = nullptr;
  SfFastParams *dF = nullptr;
  // Two models stay resident (a scan with -t folds the native windows at T and the shuffles at 37 C, chunk after chunk:
  // ScanFold-Scan.py:70-71 with F8 of SURVEY.md): loading a set that is already in a slot switches the pointers above, nothing
  // is rebuilt, copied or waited for.
  struct ModelSlot {
    SfDevParams *dP = nullptr;
    SfDevParamsPF *dX = nullptr;
    SfFastParams *dF = nullptr;
    uint64_t key = 0;                 // FNV-1a of the blobs it was built from: finds the candidate ...
    std::vector<unsigned char> src;   // ... and their bytes decide (1 blob, or 3 for a rescaled set)
    bool valid = false;
    int fast_ok = 0;
    int span = 0;          // the max_bp_span its max_pair_dist field was written for
    double temperature = 37.0;
    uint64_t used = 0;     // load counter value of its last use
  } slot[2];
  uint64_t loads = 0;
  int cur = 0;
  double temperature = 37.0;
  DevBuf full_scratch, pf_scratch, pf_share, fast_scratch, seqs, energies, db, cen, dbl, status, transcript, ovf, cons, sc;
  // the rolling-row offsets (SfFastRows, 8.7 kB) of every width that has been launched: one device table per width, written
  // once and never again, so launches of different widths on different caller streams (sf_mfe_device) cannot see each other's
  std::map<int, SfFastRows *> fast_rows;
  DevBuf tab_in, tab_partner, tab_counts, tab_out;  // sf_tabulate_pairs
  int64_t tab_groups = -1;
  std::string last_hip_error;
  // profiling of the dominant kernel
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;  // recorded only while profiling is on (sf_prof_reset)
  bool prof_on = false;
  double prof_ms = 0.0;
  int64_t prof_launches = 0, prof_folds = 0;
  int force_full = 0;
  int fast_ok = 0;
  int max_bp_span = 0;  // RNA.md().max_bp_span; <= 0: no limit
  int pf_kernel = 0;  // 0: LDS-resident kernel where it fits; 1: device-memory tables (SCANFOLD_PF_KERNEL=global)
  int pf_blocks_per_cu = 4;  // 256 VGPRs per thread: 2 waves per SIMD
  int pf_share_inside = 1;   // sf_scan, step 1: consecutive native windows share their inside tables (SCANFOLD_PF_SHARE=0: off)
} g;

#define HIPCHK(call)                                                              \
  do {                                                                            \
    hipError_t e_ = (call);                                                       \
    if (e_ != hipSuccess) {                                                       \
      char b_[512];                                                               \
      snprintf(b_, sizeof b_, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      g.last_hip_error = b_;                                                      \
      return SF_ERR_HIP;                                                          \
    }                                                                             \
  } while (0)

int ensure(DevBuf &b, size_t need) {
  if (need <= b.cap) return SF_OK;
  if (b.p) HIPCHK(hipFree(b.p));
  b.p = nullptr;
  b.cap = 0;
  size_t cap = need + need / 8 + 256;
  HIPCHK(hipMalloc(&b.p, cap));
  b.cap = cap;
  return SF_OK;
}

int block_threads(int W) {
  int t = ((W + 63) / 64) * 64;
  return t < 64 ? 64 : t;
}

// Every entry point that touches the device: the library must be initialised, and HIP's current device is PER THREAD —
// hipSetDevice in sf_init binds only the thread that called it, so a caller's helper thread (scan.py runs the engine
// from one) would otherwise allocate and copy on device 0 while g.stream and the kernels belong to g.dev.
#define SF_ENTER()                          \
  do {                                      \
    if (!g.init) return SF_ERR_NOT_INIT;    \
    HIPCHK(hipSetDevice(g.dev));            \
  } while (0)

int check_ready() {
  SF_ENTER();
  if (!g.have_params) return SF_ERR_NO_PARAMS;
  return SF_OK;
}

double smooth_term(double x) {  // ViennaRNA's SMOOTH() with pf_smooth=1
  const double SCALE = 10.0;
  if (x / SCALE < -1.2283697) return 0.0;
  if (x / SCALE > 0.8660254) return x;
  const double s = sin(x / SCALE - 0.34242663) + 1.0;
  return SCALE * 0.38490018 * s * s;
}

uint32_t pack_key(const char *s, int len) {
  uint32_t k = 0;
  for (int i = 0; i < len; i++) k = (k << 3) | sf_encode_nt((uint8_t)s[i]);
  return k;
}

// P37 / PdH (both or neither): the 37 C free energies and the enthalpies `P` was rescaled from.  With them every Boltzmann
// weight comes from the un-truncated double dG(T) = dH - (dH - dG37) (T + K0) / (37 + K0), as ViennaRNA's
// get_boltzmann_factors computes it (md.pf_smooth, its default) [EXT]; the MFE tables stay the truncated integers of P.
void build_dev_params(const sf_params_blob &P, SfDevParams &D, SfDevParamsPF &X, const sf_params_blob *P37 = nullptr,
                      const sf_params_blob *PdH = nullptr) {
  memset(&D, 0, sizeof D);
  D.P = P;
  for (int s = 0; s <= SF_MAX_W + 1; s++)
    D.hp_init[s] = (s <= 30) ? P.hairpin[s] : P.hairpin[30] + (int)(P.lxc * log(s / 30.));
  for (int k = 0; k < SF_NSPECIAL; k++) {
    D.tetra_key[k] = k < P.n_tetra ? pack_key(P.tetra_seq[k], 6) : 0xFFFFFFFFu;
    D.tri_key[k] = k < P.n_tri ? pack_key(P.tri_seq[k], 5) : 0xFFFFFFFFu;
    D.hexa_key[k] = k < P.n_hexa ? pack_key(P.hexa_seq[k], 8) : 0xFFFFFFFFu;
  }
  D.pair[2][3] = 1; D.pair[3][2] = 2; D.pair[3][4] = 3; D.pair[4][3] = 4; D.pair[1][4] = 5; D.pair[4][1] = 6;

  memset(&X, 0, sizeof X);
  const double kT = (P.temperature + 273.15) * 1.98717;
  X.kT = kT;
  const double tempf = (P.temperature + 273.15) / (37.0 + 273.15);
  auto ex = [&](const int32_t &ref) -> double {  // the energy behind field `ref` of P
    if (!P37 || !PdH) return (double)ref;
    const size_t off = (size_t)((const char *)&ref - (const char *)&P);
    const int32_t g37 = *(const int32_t *)((const char *)P37 + off), dh = *(const int32_t *)((const char *)PdH + off);
    if (g37 >= SF_INF || g37 <= -SF_INF) return (double)ref;
    return (double)dh - ((double)dh - (double)g37) * tempf;
  };
  auto bw = [kT](double e) { return exp(-e * 10.0 / kT); };
  auto bws = [kT](double e) { return exp(smooth_term(-e) * 10.0 / kT); };
  for (int a = 0; a < 8; a++)
    for (int b = 0; b < 8; b++) X.stack[a][b] = bw(ex(P.stack[a][b]));
  for (int i = 0; i <= 30; i++) {
    X.bulge[i] = bw(ex(P.bulge[i]));
    X.internal_loop[i] = bw(ex(P.internal_loop[i]));
    X.ninio[i] = bw(std::min((double)P.max_ninio, i * ex(P.ninio)));
  }
  for (int s = 0; s <= SF_MAX_W + 1; s++)
    X.hp_init[s] = (s <= 30) ? bw(ex(P.hairpin[s])) : bw(ex(P.hairpin[30])) * exp(-(P.lxc * log(s / 30.)) * 10. / kT);
  for (int t = 0; t < 8; t++)
    for (int a = 0; a < 5; a++) {
      X.dangle5[t][a] = bws(ex(P.dangle5[t][a]));
      X.dangle3[t][a] = bws(ex(P.dangle3[t][a]));
      for (int b = 0; b < 5; b++) {
        X.mismatchI[t][a][b] = bw(ex(P.mismatchI[t][a][b]));
        X.mismatchH[t][a][b] = bw(ex(P.mismatchH[t][a][b]));
        X.mismatch1nI[t][a][b] = bw(ex(P.mismatch1nI[t][a][b]));
        X.mismatch23I[t][a][b] = bw(ex(P.mismatch23I[t][a][b]));
        X.mismatchM[t][a][b] = bws(ex(P.mismatchM[t][a][b]));
        X.mismatchExt[t][a][b] = bws(ex(P.mismatchExt[t][a][b]));
      }
    }
  for (int a = 0; a < 8; a++)
    for (int b = 0; b < 8; b++)
      for (int c = 0; c < 5; c++)
        for (int d = 0; d < 5; d++) {
          X.int11[a][b][c][d] = bw(ex(P.int11[a][b][c][d]));
          for (int e = 0; e < 5; e++) {
            X.int21[a][b][c][d][e] = bw(ex(P.int21[a][b][c][d][e]));
            for (int f = 0; f < 5; f++) X.int22[a][b][c][d][e][f] = bw(ex(P.int22[a][b][c][d][e][f]));
          }
        }
  X.MLbase = bw(ex(P.MLbase));
  X.MLclosing = bw(ex(P.MLclosing));
  for (int t = 0; t < 8; t++) X.MLintern[t] = bw(ex(P.MLintern[t]));
  X.TermAU = bw(ex(P.TerminalAU));
  for (int k = 0; k < SF_NSPECIAL; k++) {
    X.tetra[k] = bw(ex(P.tetra_E[k]));
    X.tri[k] = bw(ex(P.tri_E[k]));
    X.hexa[k] = bw(ex(P.hexa_E[k]));
  }
  for (int u = 2; u <= SF_MAXLOOP; u++) X.il1n[u] = X.internal_loop[u] * X.ninio[u - 2];
  X.mlbase_pow[0] = 1.0;
  for (int k = 1; k <= SF_MAX_W + 1; k++) X.mlbase_pow[k] = X.mlbase_pow[k - 1] * X.MLbase;
  // MFE model: dangle / multiloop / exterior mismatch terms are stored as min(0, x), as ViennaRNA's get_scaled_params
  // does ("must be <= 0") [EXT]; the Boltzmann weights above come from the unclamped values through SMOOTH().
  for (int t = 0; t < 8; t++)
    for (int a = 0; a < 5; a++) {
      if (D.P.dangle5[t][a] > 0) D.P.dangle5[t][a] = 0;
      if (D.P.dangle3[t][a] > 0) D.P.dangle3[t][a] = 0;
      for (int b = 0; b < 5; b++) {
        if (D.P.mismatchM[t][a][b] > 0) D.P.mismatchM[t][a][b] = 0;
        if (D.P.mismatchExt[t][a][b] > 0) D.P.mismatchExt[t][a][b] = 0;
      }
    }
}

// sf_pf_fast_kernel is compiled for two waves per SIMD (256 VGPRs): eight waves per CU = four workgroups of 128 threads
// (W <= 128) or two of 256.  A grid larger than that only runs its surplus workgroups as a second round — on a second set of
// 2-MB table slices (W = 200: 2.1 GB instead of 1.07 GB of scratch).
static int pf_fast_blocks_per_cu(int W) { return W > 128 ? g.pf_blocks_per_cu / 2 : g.pf_blocks_per_cu; }
int max_resident_blocks() { return g.n_cu * 4; }

// FULL kernel over n items; see sf_mfe_full_kernel for the indexing arguments
int launch_full(const uint8_t *d_seqs, const int *d_idx, const int *d_count, int n, int row_stride, int mfe_stride,
                int W, int32_t *d_mfe, char *d_db, int db_stride, hipStream_t st, const char *d_cons = nullptr,
                const int32_t *d_sc = nullptr) {
  if (n <= 0) return SF_OK;
  int grid = n < max_resident_blocks() ? n : max_resident_blocks();
  int rc = ensure(g.full_scratch, (size_t)grid * SF_FULL_SCRATCH_INTS(W) * sizeof(int32_t));
  if (rc) return rc;
  SF_LAUNCH(sf_mfe_full_kernel, grid, block_threads(W), 0, st, d_seqs, d_idx, d_count, n, row_stride, mfe_stride, W,
            (const SfDevParams *)g.dP, (int32_t *)g.full_scratch.p, d_mfe, d_db, db_stride, (int *)g.status.p, d_cons, d_sc);
  HIPCHK(hipGetLastError());
  return SF_OK;
}

// d_tr != null: the n rows are the native windows of transcript d_tr (length L) that start at win0, win0+1, ...
// (sf_scan with step 1): consecutive windows share their inside tables (sf_pf_lds.hip.h).
int launch_pf(const uint8_t *d_seqs, int n, int row_stride, int W, double *d_dG, double *d_mbd, char *d_cen,
              double *d_cd, hipStream_t st, const uint8_t *d_tr = nullptr, int L = 0, int win0 = 0, int step = 1) {
  if (n <= 0) return SF_OK;
  int grid = n < max_resident_blocks() ? n : max_resident_blocks();
  if (sf_pfl_supported(W) && !g.force_full && g.pf_kernel == 0) {
    // every table of a fold in the LDS of one CU: one workgroup per CU
    grid = n < g.n_cu ? n : g.n_cu;
    int run_len = 1;
    double *share = nullptr;
    if (d_tr && g.pf_share_inside && n >= 2 && step >= 1 && step <= W / 4) {
      // run length: the makespan of ceil(runs / CUs) runs per workgroup; a resumed window costs the outside pass
      // (~0.6 of a full fold) plus its share of the inside pass
      const double resumed = 0.6 + 0.4 * step / (double)(W - 4);
      double best = 1e300;
      for (int t = 1; t <= 128; t++) {
        const int runs = (n + t - 1) / t, per = (runs + g.n_cu - 1) / g.n_cu;
        const double cost = per * (1.0 + resumed * (t - 1));
        if (cost < best) { best = cost; run_len = t; }
      }
      if (run_len > 1) {
        grid = (n + run_len - 1) / run_len < g.n_cu ? (n + run_len - 1) / run_len : g.n_cu;
        int rc = ensure(g.pf_share, (size_t)((grid + 7) & ~7) * SF_PFL_SHARE_DOUBLES(W) * sizeof(double));  // whole groups of eight slices: see sv in the kernel
        if (rc) return rc;
        share = (double *)g.pf_share.p;
      }
    }
    sf_pf_lds_launch(grid, W, share != nullptr, st, d_seqs, n, row_stride, W, (const SfDevParams *)g.dP, (const SfDevParamsPF *)g.dX,
                     d_dG, d_mbd, d_cen, d_cd, d_tr, L, win0, step, run_len, share, (const char *)nullptr, (int *)nullptr);
  } else if (W >= 16 && W <= SF_PFF_MAXW && !g.force_full) {
    const int pf_blocks = g.n_cu * pf_fast_blocks_per_cu(W);
    grid = n < pf_blocks ? n : pf_blocks;
    int rc = ensure(g.pf_scratch, (size_t)grid * SF_PFF_SCRATCH_DOUBLES(W) * sizeof(double));
    if (rc) return rc;
    sf_pf_fast_launch(grid, W, st, d_seqs, n, row_stride, W, (const SfDevParams *)g.dP, (const SfDevParamsPF *)g.dX,
                      (double *)g.pf_scratch.p, d_dG, d_mbd, d_cen, d_cd);
  } else {
    int rc = ensure(g.pf_scratch, (size_t)grid * SF_PF_SCRATCH_DOUBLES(W) * sizeof(double));
    if (rc) return rc;
    SF_LAUNCH(sf_pf_kernel, grid, block_threads(W), 0, st, d_seqs, n, row_stride, W, (const SfDevParams *)g.dP,
              (const SfDevParamsPF *)g.dX, (double *)g.pf_scratch.p, d_dG, d_mbd, d_cen, d_cd, (const char *)nullptr,
              (int *)nullptr);
  }
  HIPCHK(hipGetLastError());
  return SF_OK;
}

// energies of n rows: LDS-resident int16 kernel, then the exact int32 kernel on the rows it flagged.
// If d_db, every row that is a multiple of trace_stride also gets its structure, row/trace_stride-th string.
// HIP events around the dominant kernel, only while profiling is on; a failed call never leaks a pair
struct ProfPair {
  hipEvent_t e0 = nullptr, e1 = nullptr;
  bool armed = false;
  int begin(hipStream_t st) {
    if (!g.prof_on) return SF_OK;
    if (g.ev.size() >= 1024) {  // bounded: fold what has accumulated into the running sum
      int rc = prof_drain();
      if (rc) return rc;
    }
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess || hipEventRecord(e0, st) != hipSuccess) {
      drop();
      g.last_hip_error = "hipEventCreate/Record failed in launch_mfe";
      return SF_ERR_HIP;
    }
    armed = true;
    return SF_OK;
  }
  int end(hipStream_t st) {
    if (!armed) return SF_OK;
    if (hipEventRecord(e1, st) != hipSuccess) {
      drop();
      g.last_hip_error = "hipEventRecord failed in launch_mfe";
      return SF_ERR_HIP;
    }
    g.ev.push_back({e0, e1});
    armed = false;
    e0 = e1 = nullptr;
    return SF_OK;
  }
  void drop() {
    if (e0) hipEventDestroy(e0);
    if (e1) hipEventDestroy(e1);
    e0 = e1 = nullptr;
    armed = false;
  }
  ~ProfPair() { drop(); }
  static int prof_drain() {
    for (auto &e : g.ev) {
      float t = 0.f;
      hipError_t err = hipEventSynchronize(e.second);
      if (err == hipSuccess) err = hipEventElapsedTime(&t, e.first, e.second);
      hipEventDestroy(e.first);
      hipEventDestroy(e.second);
      if (err == hipSuccess) g.prof_ms += t;
    }
    g.ev.clear();
    return SF_OK;
  }
};

// SCANFOLD_MFE_POISON=1..5 (tests only): the LDS kernel's poison build — every byte a fold has not written itself holds an
// adversarial pattern (sf_mfe_fast.hip.h, PZ).  Read at every launch so that a test can switch it inside one process.
static int mfe_poison() {
  const char *e = getenv("SCANFOLD_MFE_POISON");
  const int v = e ? atoi(e) : 0;
  if (v < 1 || v > 5) return 0;
  static bool said = false;
  if (!said) {  // a test hook in the product library: never silent
    fprintf(stderr, "scanfold_hip: SCANFOLD_MFE_POISON=%d — TEST MODE: the MFE kernel refills its LDS slack with an adversarial "
                    "pattern before every fold (slower; results must not change). Unset it for production runs.\n", v);
    said = true;
  }
  return v;
}

// d_cons / d_sc: every fold has its own hard constraint / Deigan pseudo-energies (row k of each; trace_stride 1): the
// constrained native windows of sf_fold_constrained, on the LDS kernel where it applies (W <= 250), else the general one
int launch_mfe(const uint8_t *d_seqs, int n, int W, int32_t *d_out, hipStream_t st, int trace_stride = 1,
               char *d_db = nullptr, const char *d_cons = nullptr, const int32_t *d_sc = nullptr) {
  if (n <= 0) return SF_OK;
  ProfPair prof;
  int rc = SF_OK;
  const bool hc = d_cons || d_sc;
  if (g.force_full || !g.fast_ok || !sf_fast_w_supported(W) || (hc && (W > 250 || trace_stride != 1))) {
    if ((rc = prof.begin(st))) return rc;
    rc = launch_full(d_seqs, nullptr, nullptr, n, 1, 1, W, d_out, d_db, d_db ? trace_stride : 0, st, d_cons, d_sc);
    if (rc) return rc;
    if ((rc = prof.end(st))) return rc;
  } else {
    rc = ensure(g.ovf, sizeof(int) * ((size_t)n + 2));
    if (rc) return rc;
    // [overflow count][work counter][overflow list]: the counter (+ grid size) hands out the folds beyond each workgroup's first
    int *d_cnt = (int *)g.ovf.p, *d_work = d_cnt + 1, *d_list = d_cnt + 2;
    int grid = 0, threads = 0;
    size_t lds = 0, scratch_bytes = 0;
    sf_fast_geometry(W, g.n_cu, n, &grid, &threads, &lds, &scratch_bytes, hc);
    HIPCHK(hipMemsetAsync(d_cnt, 0, 2 * sizeof(int), st));
    rc = ensure(g.fast_scratch, scratch_bytes);
    if (rc) return rc;
    // the rolling rows' offsets for this width (SfFastRows): built on first use, immutable afterwards
    const SfFastRows *d_rows = nullptr;
    {
      auto it = g.fast_rows.find(W);
      if (it == g.fast_rows.end()) {
        static SfFastRows host_rows;
        SfFastRows *d = nullptr;
        sf_fast_build_rows(W, host_rows);
        HIPCHK(hipMalloc((void **)&d, sizeof(SfFastRows)));
        if (hipMemcpy(d, &host_rows, sizeof(SfFastRows), hipMemcpyHostToDevice) != hipSuccess) {  // synchronous: host_rows is reused
          hipFree(d);
          g.last_hip_error = "hipMemcpy of the rolling-row offset table failed";
          return SF_ERR_HIP;
        }
        it = g.fast_rows.emplace(W, d).first;
      }
      d_rows = it->second;
    }
    if ((rc = prof.begin(st))) return rc;
    if (hc)
      sf_fast_launch_hc(grid, threads, lds, st, d_seqs, n, W, (const SfDevParams *)g.dP, (const SfFastParams *)g.dF,
                        d_rows, (int16_t *)g.fast_scratch.p, d_out, d_cnt, d_list, trace_stride, d_db, (int *)g.status.p, d_work,
                        d_cons, d_sc);
    else
      sf_fast_launch(mfe_poison(), grid, threads, lds, st, d_seqs, n, W, (const SfDevParams *)g.dP, (const SfFastParams *)g.dF,
                     d_rows, (int16_t *)g.fast_scratch.p, d_out, d_cnt, d_list, trace_stride, d_db, (int *)g.status.p,
                     d_work, (const char *)nullptr, (const int32_t *)nullptr);
    HIPCHK(hipGetLastError());
    if ((rc = prof.end(st))) return rc;
    // folds that left the int16 range (or hold a forced pair of non-complementary bases) are redone exactly
    rc = launch_full(d_seqs, d_list, d_cnt, n, 1, 1, W, d_out, d_db, d_db ? trace_stride : 0, st, d_cons, d_sc);
  }
  if (g.prof_on) {
    g.prof_launches++;
    g.prof_folds += n;
  }
  return rc;
}

// Read the sticky traceback status word after everything queued on `st` (nullptr: the whole device) has finished,
// and clear it.  Non-zero means a traceback found no decomposition: SF_ERR_INTERNAL.
int read_status(hipStream_t st, bool whole_device) {
  int v = 0;
  if (whole_device) HIPCHK(hipDeviceSynchronize());
  else HIPCHK(hipStreamSynchronize(st));
  HIPCHK(hipMemcpy(&v, g.status.p, sizeof(int), hipMemcpyDeviceToHost));
  if (v) HIPCHK(hipMemset(g.status.p, 0, sizeof(int)));
  if (v & 2) return SF_ERR_CONSTRAINT;  // unbalanced brackets in a window's constraint string
  if (v & (SF_TAB_ST_UNBALANCED | SF_TAB_ST_TOO_MANY)) return SF_ERR_TABLE;
  return v ? SF_ERR_INTERNAL : SF_OK;
}

}  // namespace

extern "C" {

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
const char *sf_last_hip_error(void) { return g.last_hip_error.c_str(); }

int sf_init(int device_ordinal) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return SF_ERR_NO_DEVICE;
  if (device_ordinal < 0 || device_ordinal >= ndev) return SF_ERR_BAD_ARG;
  if (g.init) {
    if (g.dev == device_ordinal) return SF_OK;
    sf_shutdown();
  }
  HIPCHK(hipSetDevice(device_ordinal));
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, device_ordinal));
  g.n_cu = prop.multiProcessorCount;
  char nm[256];
  snprintf(nm, sizeof nm, "%s (%s), %d CUs", prop.name[0] ? prop.name : "AMD GPU", prop.gcnArchName, g.n_cu);
  g.dev_name = nm;
  g.dev = device_ordinal;
  HIPCHK(hipStreamCreate(&g.stream));
  for (auto &m : g.slot) {
    HIPCHK(hipMalloc((void **)&m.dP, sizeof(SfDevParams)));
    HIPCHK(hipMalloc((void **)&m.dX, sizeof(SfDevParamsPF)));
    HIPCHK(hipMalloc((void **)&m.dF, sizeof(SfFastParams)));
    m.valid = false;
  }
  g.cur = 0;
  g.dP = g.slot[0].dP; g.dX = g.slot[0].dX; g.dF = g.slot[0].dF;
  {  // the sticky device status word every traceback ORs into (read and cleared by read_status)
    int rc = ensure(g.status, sizeof(int));
    if (rc) return rc;
    HIPCHK(hipMemset(g.status.p, 0, sizeof(int)));
  }
  HIPCHK(sf_fast_configure());
  HIPCHK(sf_pfl_configure());
  if (const char *pk = getenv("SCANFOLD_PF_KERNEL")) g.pf_kernel = (strcmp(pk, "global") == 0);
  if (const char *ps = getenv("SCANFOLD_PF_SHARE")) g.pf_share_inside = atoi(ps) != 0;
  g.force_full = 0;  // (sf_set_kernel_mode(1) selects the general kernels)
  g.init = true;
  g.have_params = false;
  return SF_OK;
}

int sf_shutdown(void) {
  if (!g.init) return SF_OK;
  hipDeviceSynchronize();
  DevBuf *bufs[] = {&g.full_scratch, &g.pf_scratch, &g.pf_share, &g.fast_scratch, &g.seqs, &g.energies, &g.db, &g.cen,
                    &g.dbl, &g.status, &g.transcript, &g.ovf, &g.cons, &g.sc, &g.tab_in, &g.tab_partner, &g.tab_counts,
                    &g.tab_out};
  for (DevBuf *b : bufs) {
    if (b->p) hipFree(b->p);
    b->p = nullptr;
    b->cap = 0;
  }
  for (auto &e : g.ev) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
  g.ev.clear();
  for (auto &m : g.slot) {
    if (m.dP) hipFree(m.dP);
    if (m.dX) hipFree(m.dX);
    if (m.dF) hipFree(m.dF);
    m.dP = nullptr; m.dX = nullptr; m.dF = nullptr; m.valid = false;
  }
  g.dP = nullptr; g.dX = nullptr; g.dF = nullptr;
  if (g.stream) hipStreamDestroy(g.stream);
  g.stream = nullptr;
  g.init = false;
  for (auto &kv : g.fast_rows) hipFree(kv.second);
  g.fast_rows.clear();
  g.have_params = false;
  return SF_OK;
}

int sf_device_name(char *buf, size_t n) {
  SF_ENTER();
  if (!buf || n == 0) return SF_ERR_BAD_ARG;
  snprintf(buf, n, "%s", g.dev_name.c_str());
  return SF_OK;
}

int sf_params_load(const void *blob, size_t nbytes, double temperature_c) {
  return sf_params_load_rescaled(blob, nbytes, temperature_c, nullptr, nullptr);
}

int sf_params_load_rescaled(const void *blob, size_t nbytes, double temperature_c, const void *blob_37c,
                            const void *blob_enthalpy) {
  SF_ENTER();
  if (!blob || nbytes != sizeof(sf_params_blob)) return SF_ERR_BAD_PARAMS;
  if ((blob_37c == nullptr) != (blob_enthalpy == nullptr)) return SF_ERR_BAD_ARG;
  static sf_params_blob P, P37, PdH;
  memcpy(&P, blob, sizeof P);
  if (P.magic != SF_PARAMS_MAGIC || P.version != SF_PARAMS_VERSION) return SF_ERR_BAD_PARAMS;
  if (blob_37c) {
    memcpy(&P37, blob_37c, sizeof P37);
    memcpy(&PdH, blob_enthalpy, sizeof PdH);
    if (P37.magic != SF_PARAMS_MAGIC || P37.version != SF_PARAMS_VERSION) return SF_ERR_BAD_PARAMS;
    if (PdH.magic != SF_PARAMS_MAGIC || PdH.version != SF_PARAMS_VERSION) return SF_ERR_BAD_PARAMS;
    if (fabs(P37.temperature - 37.0) > 1e-9) return SF_ERR_TEMPERATURE;  // the record the rescale starts from
  }
  if (fabs(P.temperature - temperature_c) > 1e-9) return SF_ERR_TEMPERATURE;
  // the set's identity: FNV-1a over the bytes handed in
  uint64_t key = 1469598103934665603ull;
  auto mix = [&key](const void *p, size_t n) {
    const unsigned char *b = (const unsigned char *)p;
    for (size_t k = 0; k < n; k++) { key ^= b[k]; key *= 1099511628211ull; }
  };
  mix(&P, sizeof P);
  if (blob_37c) { mix(&P37, sizeof P37); mix(&PdH, sizeof PdH); }
  const int32_t md = g.max_bp_span > 0 ? g.max_bp_span - 1 : 0x7fffffff;
  g.loads++;
  int hit = -1;
  // (the hash only finds the candidate: a slot is this set iff the bytes it was built from are these bytes)
  const size_t nb = sizeof(sf_params_blob);
  for (int k = 0; k < 2; k++) {
    auto &m = g.slot[k];
    if (!m.valid || m.key != key || m.src.size() != (blob_37c ? 3 : 1) * nb) continue;
    if (memcmp(m.src.data(), &P, nb) != 0) continue;
    if (blob_37c && (memcmp(m.src.data() + nb, &P37, nb) != 0 || memcmp(m.src.data() + 2 * nb, &PdH, nb) != 0)) continue;
    hit = k;
  }
  if (hit < 0) {
    // the slot not in use (or the older one) is rebuilt; queued work may still read it
    const int k = !g.slot[0].valid ? 0 : (!g.slot[1].valid ? 1 : (g.have_params ? 1 - g.cur : (g.slot[0].used <= g.slot[1].used ? 0 : 1)));
    static SfDevParams D;
    static SfDevParamsPF X;
    static SfFastParams F;
    build_dev_params(P, D, X, blob_37c ? &P37 : nullptr, blob_37c ? &PdH : nullptr);
    D.max_pair_dist = md;
    sf_fast_build_params(D, F);
    HIPCHK(hipDeviceSynchronize());  // *_dev work queued on the callers' own streams may still read that slot's old tables
    g.slot[k].valid = false;
    HIPCHK(hipMemcpy(g.slot[k].dP, &D, sizeof D, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(g.slot[k].dX, &X, sizeof X, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(g.slot[k].dF, &F, sizeof F, hipMemcpyHostToDevice));
    g.slot[k].src.assign((const unsigned char *)&P, (const unsigned char *)&P + nb);
    if (blob_37c) {
      g.slot[k].src.insert(g.slot[k].src.end(), (const unsigned char *)&P37, (const unsigned char *)&P37 + nb);
      g.slot[k].src.insert(g.slot[k].src.end(), (const unsigned char *)&PdH, (const unsigned char *)&PdH + nb);
    }
    g.slot[k].key = key; g.slot[k].valid = true; g.slot[k].fast_ok = F.fast_ok; g.slot[k].span = g.max_bp_span;
    g.slot[k].temperature = temperature_c;
    hit = k;
  } else if (g.slot[hit].span != g.max_bp_span) {  // sf_set_max_bp_span was called while the other model was the resident one
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy((char *)g.slot[hit].dP + offsetof(SfDevParams, max_pair_dist), &md, sizeof md, hipMemcpyHostToDevice));
    g.slot[hit].span = g.max_bp_span;
  }
  g.slot[hit].used = g.loads;
  g.cur = hit;
  g.dP = g.slot[hit].dP; g.dX = g.slot[hit].dX; g.dF = g.slot[hit].dF;
  g.fast_ok = g.slot[hit].fast_ok;
  g.temperature = temperature_c;
  g.have_params = true;
  return SF_OK;
}

int sf_mfe_batch_dev(const uint8_t *d_seqs, int n, int W, int32_t *d_out, void *stream) {
  int rc = check_ready();
  if (rc) return rc;
  if (n < 0 || W < 1 || W > SF_MAX_W || (n > 0 && (!d_seqs || !d_out))) return SF_ERR_BAD_ARG;
  return launch_mfe(d_seqs, n, W, d_out, stream ? (hipStream_t)stream : g.stream);
}

int sf_mfe_batch(const uint8_t *seqs, int n, int W, int32_t *out) {
  int rc = check_ready();
  if (rc) return rc;
  if (n < 0 || W < 1 || W > SF_MAX_W || (n > 0 && (!seqs || !out))) return SF_ERR_BAD_ARG;
  if (n == 0) return SF_OK;
  if ((rc = ensure(g.seqs, (size_t)n * W))) return rc;
  if ((rc = ensure(g.energies, (size_t)n * sizeof(int32_t)))) return rc;
  HIPCHK(hipMemcpyAsync(g.seqs.p, seqs, (size_t)n * W, hipMemcpyHostToDevice, g.stream));
  if ((rc = launch_mfe((const uint8_t *)g.seqs.p, n, W, (int32_t *)g.energies.p, g.stream))) return rc;
  HIPCHK(hipMemcpyAsync(out, g.energies.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, g.stream));
  HIPCHK(hipStreamSynchronize(g.stream));
  return SF_OK;
}

int sf_mfe_trace_batch(const uint8_t *seqs, int n, int W, int32_t *mfe_out, char *db_out) {
  int rc = check_ready();
  if (rc) return rc;
  if (n < 0 || W < 1 || W > SF_MAX_W || (n > 0 && (!seqs || !db_out))) return SF_ERR_BAD_ARG;
  if (n == 0) return SF_OK;
  if ((rc = ensure(g.seqs, (size_t)n * W))) return rc;
  if ((rc = ensure(g.energies, (size_t)n * sizeof(int32_t)))) return rc;
  if ((rc = ensure(g.db, (size_t)n * (W + 1)))) return rc;
  HIPCHK(hipMemcpyAsync(g.seqs.p, seqs, (size_t)n * W, hipMemcpyHostToDevice, g.stream));
  if ((rc = launch_mfe((const uint8_t *)g.seqs.p, n, W, (int32_t *)g.energies.p, g.stream, 1, (char *)g.db.p)))
    return rc;
  if (mfe_out)
    HIPCHK(hipMemcpyAsync(mfe_out, g.energies.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, g.stream));
  HIPCHK(hipMemcpyAsync(db_out, g.db.p, (size_t)n * (W + 1), hipMemcpyDeviceToHost, g.stream));
  return read_status(g.stream, false);
}

int sf_pf_batch(const uint8_t *seqs, int n, int W, double *ens_dG, double *mbd, char *centroid, double *cdist) {
  int rc = check_ready();
  if (rc) return rc;
  if (n < 0 || W < 1 || W > SF_MAX_W || (n > 0 && !seqs)) return SF_ERR_BAD_ARG;
  if (n == 0) return SF_OK;
  if ((rc = ensure(g.seqs, (size_t)n * W))) return rc;
  if ((rc = ensure(g.dbl, (size_t)n * 3 * sizeof(double)))) return rc;
  if ((rc = ensure(g.cen, (size_t)n * (W + 1)))) return rc;
  double *d_dG = (double *)g.dbl.p, *d_mbd = d_dG + n, *d_cd = d_mbd + n;
  HIPCHK(hipMemcpyAsync(g.seqs.p, seqs, (size_t)n * W, hipMemcpyHostToDevice, g.stream));
  if ((rc = launch_pf((const uint8_t *)g.seqs.p, n, 1, W, d_dG, d_mbd, (char *)g.cen.p, d_cd, g.stream))) return rc;
  if (ens_dG) HIPCHK(hipMemcpyAsync(ens_dG, d_dG, n * sizeof(double), hipMemcpyDeviceToHost, g.stream));
  if (mbd) HIPCHK(hipMemcpyAsync(mbd, d_mbd, n * sizeof(double), hipMemcpyDeviceToHost, g.stream));
  if (cdist) HIPCHK(hipMemcpyAsync(cdist, d_cd, n * sizeof(double), hipMemcpyDeviceToHost, g.stream));
  if (centroid) HIPCHK(hipMemcpyAsync(centroid, g.cen.p, (size_t)n * (W + 1), hipMemcpyDeviceToHost, g.stream));
  HIPCHK(hipStreamSynchronize(g.stream));
  return SF_OK;
}

// Host-side look at the constraint rows before anything is launched.  Returns SF_ERR_CONSTRAINT for unbalanced brackets (ViennaRNA
// aborts there); *noncanonical = some bracket pair joins two bases that cannot pair (a "type 7" pair: only the general
// kernels carry its table rows).
static int scan_constraints(const uint8_t *seqs, const char *cons, int n, int W, bool *noncanonical) {
  static const uint8_t can[5][5] = {{0, 0, 0, 0, 0}, {0, 0, 0, 0, 1}, {0, 0, 0, 1, 0}, {0, 0, 1, 0, 1}, {0, 1, 0, 1, 0}};  // N A C G U
  std::vector<int> stack((size_t)W + 1);
  *noncanonical = false;
  for (int k = 0; k < n; k++) {
    const char *c = cons + (size_t)k * W;
    const uint8_t *s = seqs + (size_t)k * W;
    int sp = 0;
    for (int i = 0; i < W; i++) {
      if (c[i] == '(') stack[sp++] = i;
      else if (c[i] == ')') {
        if (sp == 0) return SF_ERR_CONSTRAINT;
        const int o = stack[--sp];
        if (!can[sf_encode_nt(s[o])][sf_encode_nt(s[i])]) *noncanonical = true;
      }
    }
    if (sp) return SF_ERR_CONSTRAINT;
  }
  return SF_OK;
}

// fc.hc_add_from_db(window_constraints) / fc.sc_add_SHAPE_deigan(...) followed by fc.mfe(), fc.pf(), fc.centroid(),
// fc.mean_bp_distance() on n windows (ScanFold-Scan.py:405-418; ScanFold.py:508-544).  Only native windows come here — the
// reference folds its shuffles unconstrained.  The window's constraint is applied where a kernel makes a
// cell's pair type: in the LDS kernels of the hot path (sf_mfe_fast_kernel / sf_pf_lds_kernel, HC instantiations) where
// they apply, else — W beyond their range, a bracket pair of non-complementary bases, sf_set_kernel_mode(1) — in the general
// int32 / FP64 kernels.
int sf_fold_constrained(const uint8_t *seqs, int n, int W, const char *cons, const int32_t *sc_stack_dcal, unsigned flags,
                        int32_t *mfe_out, char *db_out, double *ens_dG, double *mbd, char *centroid, double *cdist) {
  int rc = check_ready();
  if (rc) return rc;
  if (n < 0 || W < 1 || W > SF_MAX_W || (n > 0 && !seqs)) return SF_ERR_BAD_ARG;
  if (n == 0) return SF_OK;
  const bool want_mfe = !(flags & SF_FOLD_NO_MFE), want_pf = !(flags & SF_FOLD_NO_PF);
  bool noncanonical = false;
  if (cons && (rc = scan_constraints(seqs, cons, n, W, &noncanonical))) return rc;
  if ((rc = ensure(g.seqs, (size_t)n * W))) return rc;
  HIPCHK(hipMemcpyAsync(g.seqs.p, seqs, (size_t)n * W, hipMemcpyHostToDevice, g.stream));
  const char *d_cons = nullptr;
  const int32_t *d_sc = nullptr;
  if (cons) {
    if ((rc = ensure(g.cons, (size_t)n * W))) return rc;
    HIPCHK(hipMemcpyAsync(g.cons.p, cons, (size_t)n * W, hipMemcpyHostToDevice, g.stream));
    d_cons = (const char *)g.cons.p;
  }
  if (sc_stack_dcal) {
    if ((rc = ensure(g.sc, (size_t)n * W * sizeof(int32_t)))) return rc;
    HIPCHK(hipMemcpyAsync(g.sc.p, sc_stack_dcal, (size_t)n * W * sizeof(int32_t), hipMemcpyHostToDevice, g.stream));
    d_sc = (const int32_t *)g.sc.p;
  }
  if (want_mfe) {
    if ((rc = ensure(g.energies, (size_t)n * sizeof(int32_t)))) return rc;
    if ((rc = ensure(g.db, (size_t)n * (W + 1)))) return rc;
    // (the LDS kernel with the per-fold constraint where it applies; every fold traced: the structure is the point)
    if (noncanonical || !(d_cons || d_sc))
      rc = (d_cons || d_sc) ? launch_full((const uint8_t *)g.seqs.p, nullptr, nullptr, n, 1, 1, W, (int32_t *)g.energies.p,
                                          (char *)g.db.p, 0, g.stream, d_cons, d_sc)
                            : launch_mfe((const uint8_t *)g.seqs.p, n, W, (int32_t *)g.energies.p, g.stream, 1, (char *)g.db.p);
    else
      rc = launch_mfe((const uint8_t *)g.seqs.p, n, W, (int32_t *)g.energies.p, g.stream, 1, (char *)g.db.p, d_cons, d_sc);
    if (rc) return rc;
    if (mfe_out)
      HIPCHK(hipMemcpyAsync(mfe_out, g.energies.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, g.stream));
    if (db_out) HIPCHK(hipMemcpyAsync(db_out, g.db.p, (size_t)n * (W + 1), hipMemcpyDeviceToHost, g.stream));
  }
  if (want_pf) {
    int grid = n < max_resident_blocks() ? n : max_resident_blocks();
    if ((rc = ensure(g.dbl, (size_t)n * 3 * sizeof(double)))) return rc;
    if ((rc = ensure(g.cen, (size_t)n * (W + 1)))) return rc;
    double *d_dG = (double *)g.dbl.p, *d_mbd = d_dG + n, *d_cd = d_mbd + n;
    if (!d_cons) {  // (SHAPE data alone do not touch the partition function: the plain kernels)
      if ((rc = launch_pf((const uint8_t *)g.seqs.p, n, 1, W, d_dG, d_mbd, (char *)g.cen.p, d_cd, g.stream))) return rc;
    } else if (!noncanonical && !g.force_full && g.pf_kernel == 0 && sf_pfl_supported(W) &&
               sf_pfl_lds_bytes(W, true) <= SF_PFL_LDS_LIMIT) {
      grid = n < g.n_cu ? n : g.n_cu;  // every table of a fold in the LDS of one CU
      sf_pf_lds_launch_hc(grid, W, g.stream, (const uint8_t *)g.seqs.p, n, 1, W, (const SfDevParams *)g.dP,
                          (const SfDevParamsPF *)g.dX, d_dG, d_mbd, (char *)g.cen.p, d_cd, (const uint8_t *)nullptr, 0, 0, 1, 1,
                          (double *)nullptr, d_cons, (int *)g.status.p);
    } else if (!noncanonical && !g.force_full && W >= 16 && W <= SF_PFF_HC_MAXW) {
      // the constrained instantiation of the device-table kernel (120 < W <= 250, or SCANFOLD_PF_KERNEL=global)
      const int pf_blocks = g.n_cu * pf_fast_blocks_per_cu(W);
      grid = n < pf_blocks ? n : pf_blocks;
      if ((rc = ensure(g.pf_scratch, (size_t)grid * SF_PFF_SCRATCH_DOUBLES(W) * sizeof(double)))) return rc;
      sf_pf_fast_launch_hc(grid, W, g.stream, (const uint8_t *)g.seqs.p, n, 1, W, (const SfDevParams *)g.dP,
                           (const SfDevParamsPF *)g.dX, (double *)g.pf_scratch.p, d_dG, d_mbd, (char *)g.cen.p, d_cd, d_cons,
                           (int *)g.status.p);
    } else {
      if ((rc = ensure(g.pf_scratch, (size_t)grid * SF_PF_SCRATCH_DOUBLES(W) * sizeof(double)))) return rc;
      SF_LAUNCH(sf_pf_kernel, grid, block_threads(W), 0, g.stream, (const uint8_t *)g.seqs.p, n, 1, W,
                (const SfDevParams *)g.dP, (const SfDevParamsPF *)g.dX, (double *)g.pf_scratch.p, d_dG, d_mbd,
                (char *)g.cen.p, d_cd, d_cons, (int *)g.status.p);
    }
    HIPCHK(hipGetLastError());
    if (ens_dG) HIPCHK(hipMemcpyAsync(ens_dG, d_dG, n * sizeof(double), hipMemcpyDeviceToHost, g.stream));
    if (mbd) HIPCHK(hipMemcpyAsync(mbd, d_mbd, n * sizeof(double), hipMemcpyDeviceToHost, g.stream));
    if (cdist) HIPCHK(hipMemcpyAsync(cdist, d_cd, n * sizeof(double), hipMemcpyDeviceToHost, g.stream));
    if (centroid) HIPCHK(hipMemcpyAsync(centroid, g.cen.p, (size_t)n * (W + 1), hipMemcpyDeviceToHost, g.stream));
  }
  return read_status(g.stream, false);
}

static int check_scan_args(int L, int W, int step, int win_begin, int n_win, int r, int kind) {
  if (L < 1 || W < 1 || W > SF_MAX_W || step < 1 || win_begin < 0 || n_win < 0 || r < 0) return SF_ERR_BAD_ARG;
  if (kind != SF_SHUFFLE_MONO && kind != SF_SHUFFLE_DI) return SF_ERR_BAD_ARG;
  if (n_win > 0 && (long long)(win_begin + n_win - 1) * step + W > L) return SF_ERR_BAD_ARG;
  return SF_OK;
}

static int launch_shuffle(const uint8_t *d_tr, int L, int W, int step, int win_begin, int n_win, int r, int kind,
                          uint64_t seed, uint8_t *d_seqs, hipStream_t st) {
  const long long total = (long long)n_win * (r + 1);
  if (total <= 0) return SF_OK;
  const int grid = (int)((total + SF_SHUF_BLOCK - 1) / SF_SHUF_BLOCK);
  const size_t lds = (((size_t)SF_SHUF_BLOCK * W + 3) & ~(size_t)3) * 2 + SF_SHUF_BLOCK * 25 * sizeof(uint16_t);
  SF_LAUNCH(sf_shuffle_kernel, grid, SF_SHUF_BLOCK, lds, st, d_tr, L, W, step, win_begin, n_win, r, kind, seed, d_seqs);
  HIPCHK(hipGetLastError());
  return SF_OK;
}

int sf_shuffle_windows(const uint8_t *transcript, int L, int W, int step, int win_begin, int n_win, int r, int kind,
                       uint64_t seed, uint8_t *seqs_out) {
  SF_ENTER();
  int rc = check_scan_args(L, W, step, win_begin, n_win, r, kind);
  if (rc) return rc;
  if (!transcript || (n_win > 0 && !seqs_out)) return SF_ERR_BAD_ARG;
  if (n_win == 0) return SF_OK;
  const size_t nb = (size_t)n_win * (r + 1) * W;
  if ((rc = ensure(g.transcript, (size_t)L))) return rc;
  if ((rc = ensure(g.seqs, nb))) return rc;
  HIPCHK(hipMemcpyAsync(g.transcript.p, transcript, (size_t)L, hipMemcpyHostToDevice, g.stream));
  if ((rc = launch_shuffle((const uint8_t *)g.transcript.p, L, W, step, win_begin, n_win, r, kind, seed,
                           (uint8_t *)g.seqs.p, g.stream)))
    return rc;
  HIPCHK(hipMemcpyAsync(seqs_out, g.seqs.p, nb, hipMemcpyDeviceToHost, g.stream));
  HIPCHK(hipStreamSynchronize(g.stream));
  return SF_OK;
}

int sf_scan_dev(const uint8_t *d_tr, int L, int W, int step, int win_begin, int n_win, int r, int kind, uint64_t seed,
                unsigned flags, int32_t *d_energies, char *d_structure, char *d_centroid, double *d_ens_div,
                double *d_ens_dG, void *stream) {
  int rc = check_ready();
  if (rc) return rc;
  if ((rc = check_scan_args(L, W, step, win_begin, n_win, r, kind))) return rc;
  if (!d_tr || (n_win > 0 && !d_energies)) return SF_ERR_BAD_ARG;
  if (n_win == 0) return SF_OK;
  hipStream_t st = stream ? (hipStream_t)stream : g.stream;
  // windows are processed in chunks so the materialised shuffles stay below ~1 GiB
  const size_t row_bytes = (size_t)(r + 1) * W;
  int chunk = (int)(((size_t)1 << 30) / row_bytes);
  if (chunk < 1) chunk = 1;
  if (chunk > n_win) chunk = n_win;
  if ((rc = ensure(g.seqs, (size_t)chunk * row_bytes))) return rc;
  uint8_t *d_seqs = (uint8_t *)g.seqs.p;
  for (int w0 = 0; w0 < n_win; w0 += chunk) {
    const int nw = (n_win - w0 < chunk) ? n_win - w0 : chunk;
    if ((rc = launch_shuffle(d_tr, L, W, step, win_begin + w0, nw, r, kind, seed, d_seqs, st))) return rc;
    char *dbp = (!(flags & SF_SCAN_NO_TRACE) && d_structure) ? d_structure + (size_t)w0 * (W + 1) : nullptr;
    // r+1 energies per window; the native row (every (r+1)-th) also gets its structure in the same launch
    if ((rc = launch_mfe(d_seqs, nw * (r + 1), W, d_energies + (size_t)w0 * (r + 1), st, r + 1, dbp))) return rc;
    if (!(flags & SF_SCAN_NO_PF)) {
      if ((rc = launch_pf(d_seqs, nw, r + 1, W, d_ens_dG ? d_ens_dG + w0 : nullptr, d_ens_div ? d_ens_div + w0 : nullptr,
                          d_centroid ? d_centroid + (size_t)w0 * (W + 1) : nullptr, nullptr, st,
                          d_tr, L, win_begin + w0, step)))
        return rc;
    }
  }
  return SF_OK;
}

int sf_scan(const uint8_t *transcript, int L, int W, int step, int win_begin, int n_win, int r, int kind, uint64_t seed,
            unsigned flags, int32_t *energies, char *structure, char *centroid, double *ens_div, double *ens_dG) {
  int rc = check_ready();
  if (rc) return rc;
  if ((rc = check_scan_args(L, W, step, win_begin, n_win, r, kind))) return rc;
  if (!transcript || (n_win > 0 && !energies)) return SF_ERR_BAD_ARG;
  if (n_win == 0) return SF_OK;
  const size_t ne = (size_t)n_win * (r + 1);
  if ((rc = ensure(g.transcript, (size_t)L))) return rc;
  if ((rc = ensure(g.energies, ne * sizeof(int32_t)))) return rc;
  if ((rc = ensure(g.db, (size_t)n_win * (W + 1)))) return rc;
  if ((rc = ensure(g.cen, (size_t)n_win * (W + 1)))) return rc;
  if ((rc = ensure(g.dbl, (size_t)n_win * 2 * sizeof(double)))) return rc;
  double *d_div = (double *)g.dbl.p, *d_dG = d_div + n_win;
  HIPCHK(hipMemcpyAsync(g.transcript.p, transcript, (size_t)L, hipMemcpyHostToDevice, g.stream));
  rc = sf_scan_dev((const uint8_t *)g.transcript.p, L, W, step, win_begin, n_win, r, kind, seed, flags,
                   (int32_t *)g.energies.p, structure ? (char *)g.db.p : nullptr, centroid ? (char *)g.cen.p : nullptr,
                   ens_div ? d_div : nullptr, ens_dG ? d_dG : nullptr, g.stream);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(energies, g.energies.p, ne * sizeof(int32_t), hipMemcpyDeviceToHost, g.stream));
  if (structure && !(flags & SF_SCAN_NO_TRACE))
    HIPCHK(hipMemcpyAsync(structure, g.db.p, (size_t)n_win * (W + 1), hipMemcpyDeviceToHost, g.stream));
  if (!(flags & SF_SCAN_NO_PF)) {
    if (centroid) HIPCHK(hipMemcpyAsync(centroid, g.cen.p, (size_t)n_win * (W + 1), hipMemcpyDeviceToHost, g.stream));
    if (ens_div) HIPCHK(hipMemcpyAsync(ens_div, d_div, n_win * sizeof(double), hipMemcpyDeviceToHost, g.stream));
    if (ens_dG) HIPCHK(hipMemcpyAsync(ens_dG, d_dG, n_win * sizeof(double), hipMemcpyDeviceToHost, g.stream));
  }
  return read_status(g.stream, false);
}

// output arrays of sf_tabulate_pairs inside g.tab_out: four int32 arrays, then three double arrays, n entries each
struct TabOut {
  int32_t *gk, *gj, *gcount, *gfirst;
  double *sz, *sm, *se;
};
static TabOut tab_out_views(int64_t n) {
  TabOut o;
  char *p = (char *)g.tab_out.p;
  const size_t n8 = ((size_t)n + 1) & ~(size_t)1;  // keeps the doubles 8-byte aligned
  o.gk = (int32_t *)p;
  o.gj = o.gk + n8;
  o.gcount = o.gj + n8;
  o.gfirst = o.gcount + n8;
  o.sz = (double *)(o.gfirst + n8);
  o.sm = o.sz + n8;
  o.se = o.sm + n8;
  return o;
}

int sf_tabulate_pairs(const char *structures, int row_stride, int structures_on_device, int n_win, int W,
                      const int32_t *starts, const double *z, const double *mfe, const double *ed, int64_t *n_groups) {
  SF_ENTER();
  g.tab_groups = -1;
  if (n_win < 1 || W < 1 || W > SF_MAX_W || row_stride < W || !structures || !starts || !z || !mfe || !ed || !n_groups)
    return SF_ERR_BAD_ARG;
  for (int w = 1; w < n_win; w++)
    if (starts[w] <= starts[w - 1]) return SF_ERR_TABLE;
  const int64_t span = (int64_t)starts[n_win - 1] + W - starts[0];  // coordinates starts[0] .. last start + W - 1
  if (span > 0x7fffff00 || (int64_t)n_win * W > 0x7fffff00) return SF_ERR_BAD_ARG;
  const int lo = starts[0], n_coords = (int)span;
  hipStream_t st = g.stream;
  // inputs: [starts int32 n (padded to 8 bytes)] [z] [mfe] [ed] [structures, host case]
  const size_t off_z = (((size_t)n_win * 4) + 7) & ~(size_t)7;
  const size_t off_s = off_z + 3 * (size_t)n_win * 8;
  int rc = ensure(g.tab_in, off_s + (structures_on_device ? 0 : (size_t)n_win * row_stride));
  if (rc) return rc;
  char *in = (char *)g.tab_in.p;
  HIPCHK(hipMemcpyAsync(in, starts, (size_t)n_win * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(in + off_z, z, (size_t)n_win * 8, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(in + off_z + (size_t)n_win * 8, mfe, (size_t)n_win * 8, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(in + off_z + 2 * (size_t)n_win * 8, ed, (size_t)n_win * 8, hipMemcpyHostToDevice, st));
  const char *d_structs = structures;
  if (!structures_on_device) {
    HIPCHK(hipMemcpyAsync(in + off_s, structures, (size_t)n_win * row_stride, hipMemcpyHostToDevice, st));
    d_structs = in + off_s;
  } else {
    HIPCHK(hipDeviceSynchronize());  // the table may have been written on the caller's stream
  }
  const int32_t *d_starts = (const int32_t *)in;
  const double *d_z = (const double *)(in + off_z), *d_m = d_z + n_win, *d_e = d_m + n_win;
  rc = ensure(g.tab_partner, (size_t)n_win * W * sizeof(int16_t));
  if (rc) return rc;
  rc = ensure(g.tab_counts, ((size_t)n_coords + 1) * sizeof(int32_t));
  if (rc) return rc;
  int16_t *d_partner = (int16_t *)g.tab_partner.p;
  int32_t *d_counts = (int32_t *)g.tab_counts.p;
  int *d_status = (int *)g.status.p;
  SF_LAUNCH(sf_tab_partner_kernel, (n_win + 63) / 64, 64, (size_t)64 * (W / 2 + 1) * sizeof(int16_t), st, d_structs,
            row_stride, n_win, W, d_partner, d_status);
  SF_LAUNCH(sf_tab_groups_kernel<false>, n_coords, 64, 0, st, (const int16_t *)d_partner, d_starts, n_win, W, lo, d_counts,
            d_z, d_m, d_e, (int32_t *)nullptr, (int32_t *)nullptr, (int32_t *)nullptr, (int32_t *)nullptr,
            (double *)nullptr, (double *)nullptr, (double *)nullptr, d_status);
  SF_LAUNCH(sf_tab_scan_kernel, 1, 256, 0, st, d_counts, n_coords);
  HIPCHK(hipGetLastError());
  int32_t total = 0;
  HIPCHK(hipMemcpyAsync(&total, d_counts + n_coords, sizeof total, hipMemcpyDeviceToHost, st));
  rc = read_status(st, false);
  if (rc) return rc;
  const size_t n8 = ((size_t)total + 1) & ~(size_t)1;
  rc = ensure(g.tab_out, n8 * (4 * sizeof(int32_t) + 3 * sizeof(double)) + 16);
  if (rc) return rc;
  TabOut o = tab_out_views(total);
  SF_LAUNCH(sf_tab_groups_kernel<true>, n_coords, 64, 0, st, (const int16_t *)d_partner, d_starts, n_win, W, lo, d_counts,
            d_z, d_m, d_e, o.gk, o.gj, o.gcount, o.gfirst, o.sz, o.sm, o.se, d_status);
  HIPCHK(hipGetLastError());
  rc = read_status(st, false);
  if (rc) return rc;
  g.tab_groups = total;
  *n_groups = total;
  return SF_OK;
}

int sf_tabulate_fetch(int32_t *group_k, int32_t *group_j, int32_t *group_windows, int32_t *group_first_window,
                      double *group_sum_z, double *group_sum_mfe, double *group_sum_ed) {
  SF_ENTER();
  if (g.tab_groups < 0) return SF_ERR_BAD_ARG;  // no finished sf_tabulate_pairs
  const int64_t n = g.tab_groups;
  if (n == 0) return SF_OK;
  TabOut o = tab_out_views(n);
  hipStream_t st = g.stream;
  if (group_k) HIPCHK(hipMemcpyAsync(group_k, o.gk, (size_t)n * 4, hipMemcpyDeviceToHost, st));
  if (group_j) HIPCHK(hipMemcpyAsync(group_j, o.gj, (size_t)n * 4, hipMemcpyDeviceToHost, st));
  if (group_windows) HIPCHK(hipMemcpyAsync(group_windows, o.gcount, (size_t)n * 4, hipMemcpyDeviceToHost, st));
  if (group_first_window) HIPCHK(hipMemcpyAsync(group_first_window, o.gfirst, (size_t)n * 4, hipMemcpyDeviceToHost, st));
  if (group_sum_z) HIPCHK(hipMemcpyAsync(group_sum_z, o.sz, (size_t)n * 8, hipMemcpyDeviceToHost, st));
  if (group_sum_mfe) HIPCHK(hipMemcpyAsync(group_sum_mfe, o.sm, (size_t)n * 8, hipMemcpyDeviceToHost, st));
  if (group_sum_ed) HIPCHK(hipMemcpyAsync(group_sum_ed, o.se, (size_t)n * 8, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return SF_OK;
}

int sf_last_status(void) {
  SF_ENTER();
  return read_status(nullptr, true);
}

int sf_set_max_bp_span(int span) {
  SF_ENTER();
  g.max_bp_span = span > 0 ? span : 0;
  if (g.have_params) {  // patch the field of the resident model
    const int32_t md = g.max_bp_span > 0 ? g.max_bp_span - 1 : 0x7fffffff;
    HIPCHK(hipStreamSynchronize(g.stream));
    HIPCHK(hipMemcpy((char *)g.dP + offsetof(SfDevParams, max_pair_dist), &md, sizeof md, hipMemcpyHostToDevice));
    g.slot[g.cur].span = g.max_bp_span;  // (the other resident model is patched when it is switched to)
  }
  return SF_OK;
}

int sf_set_kernel_mode(int mode) {
  SF_ENTER();
  if (mode < 0 || mode > 1) return SF_ERR_BAD_ARG;
  g.force_full = (mode == 1);
  return SF_OK;
}


int sf_prof_reset(void) {
  SF_ENTER();
  HIPCHK(hipDeviceSynchronize());
  for (auto &e : g.ev) { hipEventDestroy(e.first); hipEventDestroy(e.second); }
  g.ev.clear();
  g.prof_ms = 0.0;
  g.prof_launches = 0;
  g.prof_folds = 0;
  g.prof_on = true;  // from now on launch_mfe brackets the dominant kernel with two events
  return SF_OK;
}

int sf_prof_get(double *ms, int64_t *launches, int64_t *folds) {
  SF_ENTER();
  HIPCHK(hipDeviceSynchronize());
  ProfPair::prof_drain();
  if (ms) *ms = g.prof_ms;
  if (launches) *launches = g.prof_launches;
  if (folds) *folds = g.prof_folds;
  return SF_OK;
}

int sf_prof_stop(void) {
  SF_ENTER();
  g.prof_on = false;
  return SF_OK;
}

}  // extern "C"
