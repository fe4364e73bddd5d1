// Development aid: ONE instantiation of sf_pf_lds_kernel as its own translation unit (ISA listing / resource usage in seconds):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -I scanfold_amd/csrc -I include -o /tmp/pf_one.s tools/dev/pf_one.hip
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "sf_launch.h"
#include "sf_energy.h"
#include "sf_pf.hip.h"
#include "sf_pf_lds.hip.h"
#ifndef PF_WT
#define PF_WT 120
#define PF_SH true
#endif
template __global__ void sf_pf_lds_kernel<PF_WT, PF_SH, false>(const uint8_t *, int, int, int, const SfDevParams *, const SfDevParamsPF *,
                                                               double *, double *, char *, double *, const uint8_t *, int, int, int, int,
                                                               double *, const char *, int *);
