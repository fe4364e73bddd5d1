// sf_pk16.h — two int16 values per 32-bit register: packed minimum / saturating add (v_pk_min_i16,
// v_pk_add_i16 clamp on gfx950) and aligned pair access to int16 tables in LDS.  Used by the Zuker kernels
// (sf_mfe_fast.hip.h, sf_mfe_pk.hip.h); the CPU emulation build (tests/emul) has plain C equivalents.
#pragma once
#include <stdint.h>
#include <string.h>
#include "sf_launch.h"

#ifdef SF_EMUL
static inline uint32_t sf_pk(int lo, int hi) { return ((uint32_t)lo & 0xffffu) | ((uint32_t)hi << 16); }
static inline int sf_lo(uint32_t p) { return (int)(int16_t)(p & 0xffffu); }
static inline int sf_hi(uint32_t p) { return (int)(int16_t)(p >> 16); }
static inline int sf_sat16(int v) { return v > 32767 ? 32767 : (v < -32768 ? -32768 : v); }
static inline uint32_t sf_pkmin(uint32_t a, uint32_t b) {
  return sf_pk(sf_lo(a) < sf_lo(b) ? sf_lo(a) : sf_lo(b), sf_hi(a) < sf_hi(b) ? sf_hi(a) : sf_hi(b));
}
static inline uint32_t sf_pkadd(uint32_t a, uint32_t b) {
  return sf_pk(sf_sat16(sf_lo(a) + sf_lo(b)), sf_sat16(sf_hi(a) + sf_hi(b)));
}
static inline uint32_t sf_ldw(const int16_t *p) {  // one aligned dword (the emulation checks the alignment claim)
  if ((uintptr_t)p & 3) { fprintf(stderr, "sf_mfe_pk: misaligned dword access\n"); abort(); }
  uint32_t v; memcpy(&v, p, 4); return v;
}
static inline void sf_stw(int16_t *p, uint32_t v) {
  if ((uintptr_t)p & 3) { fprintf(stderr, "sf_mfe_pk: misaligned dword access\n"); abort(); }
  memcpy(p, &v, 4);
}
static inline void sf_store_fence() {}
#else
typedef short sf_s2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t sf_pk(int lo, int hi) { return ((uint32_t)lo & 0xffffu) | ((uint32_t)hi << 16); }
__device__ __forceinline__ int sf_lo(uint32_t p) { return (int)(int16_t)(p & 0xffffu); }
__device__ __forceinline__ int sf_hi(uint32_t p) { return (int)p >> 16; }
__device__ __forceinline__ uint32_t sf_pkmin(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(sf_s2, a), __builtin_bit_cast(sf_s2, b)));
}
__device__ __forceinline__ uint32_t sf_pkadd(uint32_t a, uint32_t b) {  // saturating
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_add_sat(__builtin_bit_cast(sf_s2, a), __builtin_bit_cast(sf_s2, b)));
}
__device__ __forceinline__ uint32_t sf_ldw(const int16_t *p) { return *(const uint32_t *)p; }  // p is 4-byte aligned
__device__ __forceinline__ void sf_stw(int16_t *p, uint32_t v) { *(uint32_t *)p = v; }
__device__ __forceinline__ void sf_store_fence() { asm volatile("" ::: "memory"); }  // keeps two b16 stores apart
#endif
// Packed-int16 minimum of a lane's value with lane l ^ 32 / l ^ 16 / l ^ 8's — one step of a butterfly all-reduce — without the LDS
// crossbar (a generic __shfl_xor lowers to ds_bpermute: an LDS round trip).  gfx950: v_permlane32_swap swaps the upper half of one
// copy with the lower half of another, so the two copies hold (lo, lo) and (hi, hi); v_permlane16_swap the same for the 16-lane
// rows; lanes 8 apart inside a row meet by a DPP row rotate.
// (SF_EMUL: __shfl_xor.  The lane swaps run in sf_fast_dml2 of every narrow-kernel fold on the GPU: tests/test_gpu_parity.py::
// test_mfe_energy_parity_both_kernels — widths with 2, 4 and 8 chunks per wave —, test_config2_all_energies_equal_oracle,
// tools/gpu_wsweep_full.py, tools/gpu_cfg3_full_check.py)
#ifdef SF_EMUL
static inline uint32_t sf_pkmin_xor32(uint32_t v) { return sf_pkmin(v, (uint32_t)__shfl_xor((int)v, 32)); }
static inline uint32_t sf_pkmin_xor16(uint32_t v) { return sf_pkmin(v, (uint32_t)__shfl_xor((int)v, 16)); }
static inline uint32_t sf_pkmin_xor8(uint32_t v) { return sf_pkmin(v, (uint32_t)__shfl_xor((int)v, 8)); }
#else
__device__ __forceinline__ uint32_t sf_pkmin_xor32(uint32_t v) {
  const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  return sf_pkmin(r[0], r[1]);
}
__device__ __forceinline__ uint32_t sf_pkmin_xor16(uint32_t v) {
  const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
  return sf_pkmin(r[0], r[1]);
}
__device__ __forceinline__ uint32_t sf_pkmin_xor8(uint32_t v) {
  return sf_pkmin(v, (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x128, 0xF, 0xF, false));  // row_ror:8
}
#endif
// An int16 value the optimiser must take as given: keeps chains of SCALAR 16-bit operations (v_min_i16, v_add_u16: full rate on
// MI355X) from being re-packed into half-rate v_pk_* instructions behind v_perm packs by the SLP vectoriser (sf_mfe_fast.hip.h, UNP)
#ifdef SF_EMUL  // (identity either way: only the optimiser is told nothing)
static inline short sf_opaque16(short x) { return x; }
#else
__device__ __forceinline__ short sf_opaque16(short x) { asm("" : "+v"(x)); return x; }
#endif
// the pair (p[0], p[1]); odd = parity of p's int16 index, known at compile time after unrolling
__device__ __forceinline__ uint32_t sf_ld2(const int16_t *p, const int odd) {
  if (odd) {
    const uint32_t d0 = sf_ldw(p - 1), d1 = sf_ldw(p + 1);
    return (d0 >> 16) | (d1 << 16);
  }
  return sf_ldw(p);
}
// the same with a wave-uniform parity only known at run time
__device__ __forceinline__ uint32_t sf_ld2r(const int16_t *p, const int odd) {
  const uint32_t d0 = sf_ldw(p - odd), d1 = sf_ldw(p - odd + 2);
  return (uint32_t)((((uint64_t)d1 << 32) | d0) >> (odd << 4));
}
__device__ __forceinline__ void sf_st2(int16_t *p, const int odd, uint32_t v) {
  if (odd) {
    p[0] = (int16_t)(v & 0xffffu);
    sf_store_fence();
    p[1] = (int16_t)(v >> 16);
  } else {
    sf_stw(p, v);
  }
}

