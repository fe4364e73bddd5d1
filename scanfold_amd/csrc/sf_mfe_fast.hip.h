// placeholder until the LDS-resident kernel lands
#pragma once
#include "sf_energy.h"
struct SfFastParams { int dummy; };
#define SF_FAST_THREADS 128
static inline bool sf_fast_supported(int) { return false; }
static inline void sf_fast_geometry(int, int, int, int *grid, size_t *lds, size_t *scratch) { *grid = 1; *lds = 0; *scratch = 16; }
static inline hipError_t sf_fast_configure() { return hipSuccess; }
static inline void sf_fast_build_params(const SfDevParams &, SfFastParams &) {}
__global__ void sf_mfe_fast_kernel(const uint8_t *, int, int, const SfFastParams *, int16_t *, int32_t *, int *, int *) {}
