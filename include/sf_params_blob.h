/*
 * sf_params_blob.h — binary layout of one nearest-neighbour energy-parameter set.
 *
 * The reference (moss-lab/ScanFold) never touches energy parameters itself: every fold is
 * delegated to the third-party ViennaRNA `RNA` module (ScanFold-Scan.py:245,382-389;
 * ScanFoldFunctions.py:786-787), which compiles in `rna_turner2004.par`.  This struct is the
 * in-memory image of such a parameter file (ViennaRNA ".par" v2.0 sections, free energies at
 * the stated temperature only) so that the HIP library and the test oracle consume the same
 * bytes.  Energies are int32 in dcal/mol (1 = 0.01 kcal/mol); SF_INF marks forbidden loops.
 *
 * Nucleotide codes: 0 = N/other (never pairs), 1 = A, 2 = C, 3 = G, 4 = U.
 * Pair types:       0 = none, 1 = CG, 2 = GC, 3 = GU, 4 = UG, 5 = AU, 6 = UA, 7 = non-standard.
 */
#ifndef SF_PARAMS_BLOB_H
#define SF_PARAMS_BLOB_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SF_PARAMS_MAGIC 0x31504653u /* "SFP1" little endian */
#define SF_PARAMS_VERSION 1u
#define SF_INF 10000000
#define SF_TURN 3
#define SF_MAXLOOP 30
#define SF_MAX_SPECIAL 40

typedef struct sf_params_blob {
  uint32_t magic;
  uint32_t version;
  double temperature; /* deg C the free energies below are valid at */
  double lxc;         /* logarithmic loop-size extrapolation coefficient (107.856) */
  int32_t stack[8][8];
  int32_t hairpin[31];
  int32_t bulge[31];
  int32_t internal_loop[31];
  int32_t mismatchI[8][5][5];
  int32_t mismatchH[8][5][5];
  int32_t mismatchM[8][5][5];
  int32_t mismatch1nI[8][5][5];
  int32_t mismatch23I[8][5][5];
  int32_t mismatchExt[8][5][5];
  int32_t dangle5[8][5];
  int32_t dangle3[8][5];
  int32_t int11[8][8][5][5];
  int32_t int21[8][8][5][5][5];
  int32_t int22[8][8][5][5][5][5];
  int32_t ninio;     /* per-nt asymmetry penalty (60) */
  int32_t max_ninio; /* cap (300) */
  int32_t MLbase;
  int32_t MLclosing;
  int32_t MLintern[8];
  int32_t TerminalAU;
  int32_t DuplexInit;
  int32_t n_tetra, n_tri, n_hexa;
  int32_t pad0;
  char tetra_seq[SF_MAX_SPECIAL][8]; /* 6 chars (closing pair included) + NUL */
  int32_t tetra_E[SF_MAX_SPECIAL];
  char tri_seq[SF_MAX_SPECIAL][8]; /* 5 chars + NUL */
  int32_t tri_E[SF_MAX_SPECIAL];
  char hexa_seq[SF_MAX_SPECIAL][12]; /* 8 chars + NUL */
  int32_t hexa_E[SF_MAX_SPECIAL];
} sf_params_blob;

#ifdef __cplusplus
}
#endif
#endif
