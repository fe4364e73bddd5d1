// Development aid (never part of the product): ONE instantiation of sf_mfe_fast_kernel in a translation unit of its own, so that
// an ISA listing with line tables takes seconds instead of the library's 80 s:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -gline-tables-only -S --cuda-device-only -I scanfold_amd/csrc -I include \
//         [-DMFE_NG=128 -DMFE_WT=120 -DMFE_MG=true] -o /tmp/mfe_one.s tools/dev/mfe_one.hip
//   python tools/isa_sections.py /tmp/mfe_one.s
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "sf_launch.h"
#include "sf_energy.h"
#include "sf_mfe_fast.hip.h"
#ifndef MFE_NG
#define MFE_NG 128
#define MFE_WT 120
#define MFE_MG true
#endif
template __global__ void sf_mfe_fast_kernel<MFE_NG, MFE_WT, MFE_MG, false, false>(
    const uint8_t *, int, int, const SfDevParams *, const SfFastParams *, const SfFastRows *, int16_t *, int32_t *, int *, int *, int,
    char *, int *, int *, const char *, const int32_t *, int);
