// sf_dev_params.h — device-resident image of the folding model (built by sf_params_load on the host).
#pragma once
#include <stdint.h>

#include "../../include/scanfold_hip.h"
#include "../../include/sf_params_blob.h"

#define SF_NSPECIAL SF_MAX_SPECIAL

struct SfDevParams {
  sf_params_blob P;                   // integer tables, dcal/mol
  int32_t hp_init[SF_MAX_W + 2];      // hairpin initiation by loop size, log-extrapolated beyond 30 on the host
  uint32_t tetra_key[SF_NSPECIAL];    // special hairpins as packed 3-bit nucleotide codes (closing pair included)
  uint32_t tri_key[SF_NSPECIAL];
  uint32_t hexa_key[SF_NSPECIAL];
  int32_t pair[8][8];                 // pair type of two nucleotide codes
  int32_t max_pair_dist;              // largest j - i a base pair may have: RNA.md().max_bp_span - 1 (ScanFold.py:214-215); INT32_MAX = no limit
};

// Boltzmann weights for the partition function at the blob's temperature.
struct SfDevParamsPF {
  double kT;  // cal/mol
  double stack[8][8], bulge[31], internal_loop[31];
  double hp_init[SF_MAX_W + 2];
  double mismatchI[8][5][5], mismatchH[8][5][5], mismatchM[8][5][5], mismatch1nI[8][5][5], mismatch23I[8][5][5],
      mismatchExt[8][5][5];
  double dangle5[8][5], dangle3[8][5];
  double int11[8][8][5][5], int21[8][8][5][5][5], int22[8][8][5][5][5][5];
  double ninio[SF_MAXLOOP + 1];
  double MLbase, MLclosing, MLintern[8], TermAU;
  double tetra[SF_NSPECIAL], tri[SF_NSPECIAL], hexa[SF_NSPECIAL];
  double mlbase_pow[SF_MAX_W + 2];  // MLbase^k
  double il1n[32];                  // internal_loop[u] * ninio[u - 2], u = 2..30: the size weight of a 1 x n loop (0 elsewhere)
};
