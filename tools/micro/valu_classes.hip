// valu_classes.hip — which vector instructions run at the full rate on MI355X and which at half?
// issue_rates.hip found two classes at four waves per SIMD: v_add_u32 1.1 ns per wave-instruction and SIMD, the packed int16 /
// min / bit-select instructions the MFE kernel lives on 1.95 ns.  This probe maps more opcodes (same method: 2000 x 64
// independent instructions per wave, 4 workgroups of 4 waves per CU, wall clock by HIP events) so that the kernel's hot loops
// can prefer the fast class where there is a choice.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/valu_classes tools/micro/valu_classes.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define R8(op) op(0) op(1) op(2) op(3) op(4) op(5) op(6) op(7)
#define REP8(x) x x x x x x x x
#define KERNEL(NAME, TEXT)                                                                                                  \
  __global__ __launch_bounds__(256) void k_##NAME(uint32_t *out, int iters) {                                                \
    const int tid = threadIdx.x;                                                                                             \
    uint32_t a0 = tid, a1 = tid * 3, a2 = tid * 5, a3 = tid * 7, a4 = tid * 11, a5 = tid * 13, a6 = tid * 17, a7 = tid * 19; \
    uint32_t b = tid * 29 + 1, c = 0x00ff00ffu;                                                                              \
    for (int it = 0; it < iters; it++)                                                                                       \
      asm volatile(REP8(R8(TEXT)) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc"); \
    out[blockIdx.x * 256 + tid] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                                                    \
  }
#define O2(name) #name " %" 
#define T_add_u32(k) "v_add_u32 %" #k ", %" #k ", %8\n"
#define T_sub_u32(k) "v_sub_u32 %" #k ", %" #k ", %8\n"
#define T_and_b32(k) "v_and_b32 %" #k ", %" #k ", %8\n"
#define T_or_b32(k) "v_or_b32 %" #k ", %" #k ", %8\n"
#define T_lshlrev(k) "v_lshlrev_b32 %" #k ", 1, %" #k "\n"
#define T_lshl_or(k) "v_lshl_or_b32 %" #k ", %" #k ", 16, %8\n"
#define T_lshl_add(k) "v_lshl_add_u32 %" #k ", %" #k ", 1, %8\n"
#define T_add3(k) "v_add3_u32 %" #k ", %" #k ", %8, %9\n"
#define T_min_i32(k) "v_min_i32 %" #k ", %" #k ", %8\n"
#define T_max_i32(k) "v_max_i32 %" #k ", %" #k ", %8\n"
#define T_min_u32(k) "v_min_u32 %" #k ", %" #k ", %8\n"
#define T_min3_i32(k) "v_min3_i32 %" #k ", %" #k ", %8, %9\n"
#define T_med3_i32(k) "v_med3_i32 %" #k ", %" #k ", %8, %9\n"
#define T_min_i16(k) "v_min_i16 %" #k ", %" #k ", %8\n"
#define T_add_u16(k) "v_add_u16 %" #k ", %" #k ", %8\n"
#define T_cndmask(k) "v_cndmask_b32 %" #k ", %" #k ", %8, vcc\n"
#define T_mad_u24(k) "v_mad_u32_u24 %" #k ", %" #k ", %8, %9\n"
#define T_mul_u24(k) "v_mul_u32_u24 %" #k ", %" #k ", %8\n"
#define T_mov(k) "v_mov_b32 %" #k ", %8\n"
#define T_add_f32(k) "v_add_f32 %" #k ", %" #k ", %8\n"
#define T_min_f32(k) "v_min_f32 %" #k ", %" #k ", %8\n"
#define T_fma_f32(k) "v_fma_f32 %" #k ", %" #k ", %8, %9\n"
#define T_pk_add_f16(k) "v_pk_add_f16 %" #k ", %" #k ", %8\n"
#define T_pk_min_f16(k) "v_pk_min_f16 %" #k ", %" #k ", %8\n"
#define T_pk_add_u16(k) "v_pk_add_u16 %" #k ", %" #k ", %8\n"
#define T_pk_add_i16(k) "v_pk_add_i16 %" #k ", %" #k ", %8\n"
#define T_pk_sub_i16(k) "v_pk_sub_i16 %" #k ", %" #k ", %8\n"
#define T_pk_min_i16(k) "v_pk_min_i16 %" #k ", %" #k ", %8\n"
#define T_pk_min_u16(k) "v_pk_min_u16 %" #k ", %" #k ", %8\n"
#define T_pk_max_i16(k) "v_pk_max_i16 %" #k ", %" #k ", %8\n"
#define T_pk_lshl(k) "v_pk_lshlrev_b16 %" #k ", %8, %" #k "\n"
#define T_pk_mad_i16(k) "v_pk_mad_i16 %" #k ", %" #k ", %8, %9\n"
#define T_bfi(k) "v_bfi_b32 %" #k ", %9, %" #k ", %8\n"
#define T_bfe(k) "v_bfe_u32 %" #k ", %" #k ", 3, 5\n"
#define T_perm(k) "v_perm_b32 %" #k ", %" #k ", %8, %9\n"
#define T_alignbit(k) "v_alignbit_b32 %" #k ", %" #k ", %8, 16\n"
#define T_sad(k) "v_sad_u32 %" #k ", %" #k ", %8, %9\n"
#define T_min_sdwa(k) "v_min_i32_sdwa %" #k ", %" #k ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n"
#define T_dot2(k) "v_dot2_i32_i16 %" #k ", %" #k ", %8, %9\n"
#define LIST(X) X(add_u32) X(sub_u32) X(and_b32) X(or_b32) X(lshlrev) X(lshl_or) X(lshl_add) X(add3) X(min_i32) X(max_i32) \
  X(min_u32) X(min3_i32) X(med3_i32) X(min_i16) X(add_u16) X(cndmask) X(mad_u24) X(mul_u24) X(mov) X(add_f32) X(min_f32)    \
  X(fma_f32) X(pk_add_f16) X(pk_min_f16) X(pk_add_u16) X(pk_add_i16) X(pk_sub_i16) X(pk_min_i16) X(pk_min_u16) X(pk_max_i16) \
  X(pk_lshl) X(pk_mad_i16) X(bfi) X(bfe) X(perm) X(alignbit) X(sad) X(min_sdwa) X(dot2)
#define DEF(n) KERNEL(n, T_##n)
LIST(DEF)
template <class K>
static void run(const char *name, K kern, int n_cu, uint32_t *d, bool first) {
  const int iters = 2000, grid = n_cu * 4;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d, 50);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, 0, d, iters);
  (void)hipEventRecord(e1);
  (void)hipDeviceSynchronize();
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%s  \"%s\": {\"launch_ms\": %.4f, \"ns_per_inst_per_simd\": %.3f}", first ? "" : ",\n", name, ms, ms * 1e6 / (iters * 64.0 * 4));
}
int main() {
  hipDeviceProp_t prop;
  (void)hipGetDeviceProperties(&prop, 0);
  uint32_t *d;
  (void)hipMalloc(&d, (size_t)prop.multiProcessorCount * 4 * 256 * 4);
  printf("{\"device\": \"%s\", \"waves_per_simd\": 4, \"kinds\": {\n", prop.gcnArchName);
  bool first = true;
#define RUN(n) run("v_" #n, k_##n, prop.multiProcessorCount, d, first); first = false;
  LIST(RUN)
  printf("\n}}\n");
  return 0;
}
