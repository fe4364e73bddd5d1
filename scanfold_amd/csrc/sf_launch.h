// sf_launch.h — kernel-launch and dynamic-LDS spelling used by every kernel file.
// The product build is plain HIP for gfx950.  tests/emul/ compiles the same sources with g++ against a
// fiber-based stand-in (TEST ONLY, see tests/emul/hip_emul.h) by pre-including that header, which
// defines SF_EMUL and its own SF_LAUNCH / SF_DYN_SMEM.
#pragma once
#ifndef SF_EMUL
#include <hip/hip_runtime.h>
#define SF_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) char name[]
#define SF_LAUNCH(kern, grid, block, shmem, stream, ...) \
  hipLaunchKernelGGL(kern, dim3(grid), dim3(block), shmem, stream, __VA_ARGS__)
// scheduling hint: the next `n` instructions of class `mask` (0x100 = LDS read, 0x2 = VALU) form one group
#define SF_SCHED_GROUP(mask, n) __builtin_amdgcn_sched_group_barrier((mask), (n), 0)
// nothing is scheduled across this point (bounds how many loads the compiler keeps in flight = live registers)
#define SF_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// vector ALU instructions keep their order across this point; loads, LDS and scalar instructions may still move
#define SF_VALU_FENCE() __builtin_amdgcn_sched_barrier(0x7fc)
// the value must be computed by this point (keeps the arithmetic next to the loads that feed it)
#define SF_PIN(x) asm volatile("" : "+v"(x))
// A wave-uniform pointer into constant device memory whose value the optimiser has to take as given at this point: addresses
// derived from it stay "base + immediate" at their loads (s_load / global_load saddr) instead of being hoisted out of the caller's
// loops one 64-bit scalar pair per address — pairs that are then parked in the lanes of a VGPR and fetched back with two
// v_readlane before every use (8-10 % of the vector instructions of the MFE kernel's main blocks were that).
template <class T>
__device__ __forceinline__ const T *sf_const_base(const T *p) {
  const uintptr_t a = (uintptr_t)p;  // (readfirstlane: uniform by construction, but derived from threadIdx where the wave's diagonal is)
  uintptr_t b = ((uintptr_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) |
                (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a);
  asm volatile("" : "+s"(b));
  return (const T *)(const __attribute__((address_space(4))) T *)b;
}
// v of lane l (l wave-uniform, known only at run time)
#define SF_LANE_READ(v, l) __builtin_amdgcn_readlane((v), (l))
// a value that is the same in every lane of the wave: tell the compiler (keeps derived index math scalar)
#define SF_WAVE_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
// lanes of one wave exchanging data through LDS: keep the compiler from moving LDS accesses across this point
#define SF_WAVE_SYNC()                                  \
  do {                                                  \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
    __builtin_amdgcn_wave_barrier();                    \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
  } while (0)
#endif
