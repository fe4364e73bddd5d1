/*
 * scanfold_hip.h — C ABI of libscanfold_hip.so, the MI355X (gfx950) engine behind ScanFold-Scan's hot path.
 *
 * The reference (moss-lab/ScanFold) has no FFI of its own: its seam is the set of Python call sites into
 * the ViennaRNA module `RNA` and into its own helper functions.  Each entry point below names the
 * reference call site(s) it replaces.  Host Python (scanfold_amd/) keeps the reference's function names
 * and binds these symbols with ctypes; INTEGRATION.md shows the stub a ScanFold maintainer would add.
 *
 * Conventions
 *   - return 0 on success, a negative sf_status otherwise; nothing throws across the ABI;
 *     sf_strerror() maps a status to a static string.
 *   - all buffers are caller-allocated and caller-freed; the library owns only device scratch.
 *   - sequences are rows of W bytes: ASCII (A,C,G,U/T, any case; anything else = non-pairing N) or the
 *     codes 0..4 (N,A,C,G,U) — both are accepted in the same buffer.
 *   - energies are int32 dcal/mol (1 = 0.01 kcal/mol), exactly ViennaRNA's internal unit; the float the
 *     reference sees is (float)dcal / 100 (SURVEY.md §8b "Return-value fidelity").
 *   - "_dev" variants take DEVICE pointers and a hipStream_t (as void*; NULL = the library's stream) and
 *     return without synchronising; everything else takes host pointers and returns when results are
 *     on the host.
 *   - one process drives one device (one process per GPU); the library is not re-entrant.
 */
#ifndef SCANFOLD_HIP_H
#define SCANFOLD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum sf_status {
  SF_OK = 0,
  SF_ERR_NOT_INIT = -1,     /* sf_init not called */
  SF_ERR_NO_PARAMS = -2,    /* sf_params_load not called */
  SF_ERR_BAD_ARG = -3,      /* NULL pointer, negative size, W out of range, ... */
  SF_ERR_BAD_PARAMS = -4,   /* blob size / magic / version mismatch */
  SF_ERR_TEMPERATURE = -5,  /* requested temperature differs from the one the blob is valid at */
  SF_ERR_HIP = -6,          /* a HIP runtime call failed; sf_last_hip_error() has the text */
  SF_ERR_NO_DEVICE = -7,    /* no usable GPU */
  SF_ERR_INTERNAL = -8,     /* a traceback found no decomposition (would indicate a kernel bug) */
  SF_ERR_CONSTRAINT = -9,   /* unbalanced brackets in a window's constraint string (ViennaRNA aborts there) */
  SF_ERR_TABLE = -10        /* sf_tabulate_pairs: an unbalanced structure string, or window starts not ascending */
} sf_status;

#define SF_MAX_W 400 /* longest window the kernels accept */
#define SF_SHUFFLE_MONO 0 /* uniform permutation: randomizer(), ScanFold-Scan.py:248-250 */
#define SF_SHUFFLE_DI 1   /* Altschul-Erikson dinucleotide shuffle: dinuclShuffle(), ScanFold-Scan.py:187-209 */

const char *sf_strerror(int status);
const char *sf_last_hip_error(void);

/* Lifetime.  Replaces nothing in the reference (which forks a 12-process pool per call,
 * ScanFold-Scan.py:73-77,256): selects the device and allocates scratch. */
int sf_init(int device_ordinal);
int sf_shutdown(void);
int sf_device_name(char *buf, size_t n); /* e.g. "AMD Instinct MI355X (gfx950), 256 CUs" */

/* Folding model.  Replaces RNA.md() + md.temperature (ScanFold-Scan.py:70-71;
 * ScanFoldFunctions.py:776-777): `blob` is a struct sf_params_blob (include/sf_params_blob.h).
 * temperature_c must equal blob->temperature (free energies are not rescaled on the device yet).
 * The library keeps the last TWO distinct sets resident: loading one of them again only switches which one the next launches
 * use (a scan with -t alternates between the T model of the native folds and the 37 C model of the shuffles). */
int sf_params_load(const void *blob, size_t nbytes, double temperature_c);

/* md.temperature = T with T != 37 (ScanFold-Scan.py:70-71; ScanFoldFunctions.py:776-777): `blob` holds the free
 * energies rescaled to T and truncated to integers (what the MFE recursions use); blob_37c and blob_enthalpy are the
 * same struct holding the 37 C free energies and the enthalpies it was rescaled from.  The Boltzmann weights of the
 * partition function are then built from the un-truncated dG(T) = dH - (dH - dG37) (T + 273.15) / 310.15, as ViennaRNA's
 * get_boltzmann_factors does [EXT].  Both NULL: exactly sf_params_load. */
int sf_params_load_rescaled(const void *blob, size_t nbytes, double temperature_c, const void *blob_37c,
                            const void *blob_enthalpy);

/* energies(seq_list) / rna_folder(frag): MFE of n sequences of W nt, no structure
 * (ScanFold-Scan.py:244-246,253-262; ScanFoldFunctions.py:774-789,805-814).  mfe_dcal_out[n]. */
int sf_mfe_batch(const uint8_t *seqs, int n, int W, int32_t *mfe_dcal_out);
int sf_mfe_batch_dev(const uint8_t *d_seqs, int n, int W, int32_t *d_mfe_dcal_out, void *stream);

/* RNA.fold(seq) / fc.mfe(): MFE and dot-bracket structure (ScanFold-Scan.py:245,385,394;
 * ScanFoldFunctions.py:787).  db_out is n rows of W+1 bytes (NUL terminated). */
int sf_mfe_trace_batch(const uint8_t *seqs, int n, int W, int32_t *mfe_dcal_out, char *db_out);

/* fc.pf() / RNA.pf_fold / fc.centroid() / fc.mean_bp_distance() (ScanFold-Scan.py:383-384,388-389,
 * 395,400-401).  Any output pointer may be NULL.  ensemble_dG in kcal/mol; centroid_out n rows of W+1. */
int sf_pf_batch(const uint8_t *seqs, int n, int W, double *ensemble_dG, double *mean_bp_dist, char *centroid_out,
                double *centroid_dist);

/* fc.hc_add_from_db(window_constraints) and fc.sc_add_SHAPE_deigan(window_reactivities, m, b) followed by fc.mfe(),
 * fc.pf(), fc.centroid(), fc.mean_bp_distance() (ScanFold-Scan.py:405-418; ScanFold.py:508-544) for n windows of W nt.
 *   cons           n rows of W characters, or NULL: '.', 'x', '|', '<', '>', '(' and ')' with ViennaRNA's non-enforcing
 *                  default semantics ('x' unpaired; '<' / '>' may pair downstream / upstream only; a bracket pair may
 *                  pair with each other only — type 7 if not complementary — and nothing may cross it; '|' and '.'
 *                  change nothing).  Unbalanced brackets in a row: SF_ERR_CONSTRAINT.
 *   sc_stack_dcal  n rows of W int32 (dcal/mol), or NULL: the Deigan pseudo-energy of each nucleotide, added to every
 *                  stacked pair it takes part in — MFE and traceback only, as the reference adds SHAPE data after its
 *                  partition function call.
 * Outputs as sf_mfe_trace_batch / sf_pf_batch; any may be NULL.  flags: SF_FOLD_NO_PF, SF_FOLD_NO_MFE.
 * These folds run on the kernels of the hot path with the constraint applied where a cell's pair type is made: the MFE on the
 * LDS kernel (W <= 250); the partition function on the LDS kernel (W <= 120: what fits the LDS of one CU) and above that on the
 * device-table kernel sf_pf_fast_kernel<..., HC> (W <= 250).  Wider windows, and any batch that holds a bracket pair of
 * non-complementary bases (type 7), go to the general int32 / FP64 kernels; same results either way.  Shuffles are folded
 * unconstrained, as upstream. */
#define SF_FOLD_NO_PF 1u
#define SF_FOLD_NO_MFE 2u
int sf_fold_constrained(const uint8_t *seqs, int n, int W, const char *cons, const int32_t *sc_stack_dcal,
                        unsigned flags, int32_t *mfe_dcal_out, char *db_out, double *ensemble_dG, double *mean_bp_dist,
                        char *centroid_out, double *centroid_dist);

/* scramble(text, r, type) (ScanFold-Scan.py:266-282; ScanFoldFunctions.py:834-851) for n_win windows at
 * once: window w is transcript[(win_begin+w)*step .. +W).  Output: n_win*(r+1) rows of W codes (0..4); row 0
 * of each window is the native window itself, rows 1..r its shuffles.  The generator is counter-based
 * (Philox4x32-10 keyed by seed, window index and shuffle index), so a window's shuffles do not depend on
 * how windows are batched or sharded over GPUs.  The reference's generator (unseeded Mersenne Twister
 * state of the parent / forked workers, SURVEY.md F5,Q11) is not reproducible even against itself. */
int sf_shuffle_windows(const uint8_t *transcript, int L, int W, int step, int win_begin, int n_win, int r,
                       int shuffle_kind, uint64_t seed, uint8_t *seqs_out);

/* The whole per-window hot loop of ScanFold-Scan.py:355-449 for windows [win_begin, win_begin+n_win):
 *   energies[n_win*(r+1)]  dcal/mol; [w*(r+1)+0] = native window, +k = k-th shuffle     (lines 419-423)
 *   structure[n_win*(W+1)] native MFE structure                                          (line 385)
 *   centroid[n_win*(W+1)], ens_div[n_win] (= mean_bp_distance), ens_dG[n_win]            (lines 383-389)
 * z-score / p-score / rounding / TSV stay in host Python (numpy) so their floating-point results are
 * bit-equal to the reference's (ScanFold-Scan.py:218-242,426-433).
 * flags: bit0 = skip partition function (centroid/ens_div/ens_dG untouched), bit1 = skip traceback. */
#define SF_SCAN_NO_PF 1u
#define SF_SCAN_NO_TRACE 2u
int sf_scan(const uint8_t *transcript, int L, int W, int step, int win_begin, int n_win, int r, int shuffle_kind,
            uint64_t seed, unsigned flags, int32_t *energies, char *structure, char *centroid, double *ens_div,
            double *ens_dG);
/* Same, transcript and every output resident in device memory; asynchronous on `stream`. */
int sf_scan_dev(const uint8_t *d_transcript, int L, int W, int step, int win_begin, int n_win, int r,
                int shuffle_kind, uint64_t seed, unsigned flags, int32_t *d_energies, char *d_structure,
                char *d_centroid, double *d_ens_div, double *d_ens_dG, void *stream);

/* Outcome of the asynchronous "_dev" calls: waits for all work queued on the device, then returns SF_ERR_INTERNAL if
 * any traceback since the last report found no decomposition of a cell (never seen; it would be a kernel bug), else
 * SF_OK, and clears the flag.  The host-pointer calls (sf_scan, sf_mfe_trace_batch) report it themselves. */
int sf_last_status(void);

/* Base-pair tabulation of a scan table — the first half of the Fold stage.  Replaces the window loop that appends each
 * window's (z-score, MFE, ED) to the list of every (nucleotide, partner) pair its structure holds, and the np.sum over
 * every such list (ScanFold-Fold.py:583-682,704-760; ScanFold.py:564-677,1080-1180).
 *   structures   n_win rows of dot-bracket characters ('(' ')' anything else = unpaired), row_stride bytes apart
 *                (W + 1 for the rows sf_scan / sf_scan_dev wrote); a DEVICE pointer if structures_on_device != 0
 *                (the table sf_scan_dev left resident), else a host pointer
 *   starts       coordinate of each window's first nucleotide, strictly ascending (host)
 *   z, mfe, ed   the window metrics as the scan table prints them (host, n_win doubles each)
 * A "group" is one (nucleotide k, partner j) pair; j == k stands for "k unpaired in that window".  Groups come out
 * ordered by k, and inside one k by the first window that holds them (the reference's dict insertion order).
 * The sums add the supporting windows' values in window order with numpy's pairwise scheme: bit-equal to np.sum.
 * sf_tabulate_pairs leaves the result on the device and reports its size; sf_tabulate_fetch copies it out
 * (n_groups entries per array; any pointer may be NULL).  Needs sf_init only (no energy parameters). */
int sf_tabulate_pairs(const char *structures, int row_stride, int structures_on_device, int n_win, int W,
                      const int32_t *starts, const double *z, const double *mfe, const double *ed, int64_t *n_groups);
int sf_tabulate_fetch(int32_t *group_k, int32_t *group_j, int32_t *group_windows, int32_t *group_first_window,
                      double *group_sum_z, double *group_sum_mfe, double *group_sum_ed);

/* Maximum base-pair span of the folding model.  Replaces md.max_bp_span = args.span (ScanFold.py:214-215; the
 * Scan stage script has no such flag): base pairs (i, j) with j - i + 1 > span do not exist, in the MFE fill, the
 * partition function and the traceback.  span <= 0 removes the limit (the default).  Survives sf_params_load.
 * SURVEY.md 8(f) rank 1, first item. */
int sf_set_max_bp_span(int span);

/* Diagnostics: 0 = automatic (LDS int16 kernel with int32 fallback), 1 = always the int32 kernel.
 * Results are identical in both modes; tests use it to cross-check the kernels.  (Round 1's packed variants,
 * modes 2 and 3, were slower and have been removed.) */
int sf_set_kernel_mode(int mode);

/* Measurement support for bench.py: HIP-event time (ms) and launch count of the dominant kernel
 * (the batched MFE fill) accumulated since the last reset, measured on the stream it was launched on.
 * Events are recorded only between sf_prof_reset (turns profiling on) and sf_prof_stop; outside, no event is
 * ever created.  At most 1024 pairs are pending at a time (older ones are folded into the sum). */
int sf_prof_reset(void);
int sf_prof_get(double *mfe_kernel_ms, int64_t *mfe_kernel_launches, int64_t *mfe_folds);
int sf_prof_stop(void);

#ifdef __cplusplus
}
#endif
#endif
