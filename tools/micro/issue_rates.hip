// issue_rates.hip — what one wave-instruction of the MFE kernel's mix costs on MI355X, by unit.
//
// The bench line's "is the vector unit / the LDS the bound" figures were computed with an ASSUMED 4 cycles per wave64 vector
// instruction; MI355X_MICROARCH.md measures 2 cycles once a SIMD has two waves to pick from (SIMD-32).  This probe measures it
// for the instructions sf_mfe_fast_kernel actually issues — v_pk_min_i16, v_pk_add_i16 clamp, v_add_u32, v_min_i32, v_bfi_b32,
// v_perm_b32, DPP moves, v_readlane, scalar adds, ds_read_u16 / ds_read_b32 / ds_read2_b32, 16-bit LDS stores — at 1, 2 and 4
// waves per SIMD (workgroups of 4 waves, one wave per SIMD; the LDS request sets how many workgroups a CU holds), and for the
// kernel's own VALU : SALU : LDS proportion interleaved in one stream.
//
// Per kind: every wave runs ITERS x 64 instructions (eight independent destination registers, eight rounds) between two
// s_memtime reads; cycles per wave-instruction and SIMD = mean wave time / (instructions x waves per SIMD); for LDS kinds
// also per CU (= / (4 x waves per SIMD)).  The whole grid is resident (256 CUs x k workgroups), so the figure includes
// whatever the shared front end and the LDS pipe cost under full load.  Wall time (HIP events) gives the clock the chip held.
//
// Output: one JSON object (stdout) — bench.py reads the committed copy under profiles/ (mfe_issue_rates.json).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/issue_rates tools/micro/issue_rates.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>

enum { K_PKMIN, K_PKADD, K_ADD, K_MIN, K_BFI, K_PERM, K_DPP, K_READLANE, K_VALU_MIX, K_SALU, K_DS_U16, K_DS_B32, K_DS_2B32,
       K_DS_W16, K_KERNEL_MIX, K_COUNT };
static const char *KNAME[K_COUNT] = {"v_pk_min_i16", "v_pk_add_i16_clamp", "v_add_u32", "v_min_i32", "v_bfi_b32", "v_perm_b32",
                                      "v_mov_b32_dpp", "v_readlane_b32", "valu_mix", "s_add_u32", "ds_read_u16", "ds_read_b32",
                                      "ds_read2_b32", "ds_write_b16", "kernel_mix_10v_6s_3lds"};
// instructions per unrolled block, by unit (vector, scalar, LDS)
static const int NV[K_COUNT] = {64, 64, 64, 64, 64, 64, 64, 64, 64, 0, 0, 0, 0, 0, 40};
static const int NS[K_COUNT] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 64, 0, 0, 0, 0, 24};
static const int NL[K_COUNT] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 64, 64, 64, 64, 12};

#define R8(op) op(0) op(1) op(2) op(3) op(4) op(5) op(6) op(7)
#define REP8(x) x x x x x x x x

template <int KIND>
__global__ __launch_bounds__(256) void probe(uint32_t *out, long long *cyc, int iters) {
  extern __shared__ uint32_t lds[];
  const int tid = threadIdx.x;
  for (int x = tid; x < 4096; x += 256) lds[x] = x * 2654435761u;
  __syncthreads();
  uint32_t a0 = tid, a1 = tid * 3, a2 = tid * 5, a3 = tid * 7, a4 = tid * 11, a5 = tid * 13, a6 = tid * 17, a7 = tid * 19;
  uint32_t b = tid * 29 + 1, c = 0x00ff00ffu;
  // LDS addresses: consecutive 16-bit / 32-bit / 64-bit-strided words of the wave's own 1 kB slice (conflict-free)
  const uint32_t ad16 = (tid & 63) * 2 + (tid >> 6) * 1024, ad32 = (tid & 63) * 4 + (tid >> 6) * 1024;
  uint32_t s0 = 1, s1 = 2, s2 = 3, s3 = 4;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
    if (KIND == K_PKMIN) {
#define OP(k) "v_pk_min_i16 %" #k ", %" #k ", %8\n"
      asm volatile(REP8(R8(OP)) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
#undef OP
    } else if (KIND == K_PKADD) {
#define OP(k) "v_pk_add_i16 %" #k ", %" #k ", %8 clamp\n"
      asm volatile(REP8(R8(OP)) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
#undef OP
    } else if (KIND == K_ADD) {
#define OP(k) "v_add_u32 %" #k ", %" #k ", %8\n"
      asm volatile(REP8(R8(OP)) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
#undef OP
    } else if (KIND == K_MIN) {
#define OP(k) "v_min_i32 %" #k ", %" #k ", %8\n"
      asm volatile(REP8(R8(OP)) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
#undef OP
    } else if (KIND == K_BFI) {
#define OP(k) "v_bfi_b32 %" #k ", %9, %" #k ", %8\n"
      asm volatile(REP8(R8(OP)) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
#undef OP
    } else if (KIND == K_PERM) {
#define OP(k) "v_perm_b32 %" #k ", %" #k ", %8, %9\n"
      asm volatile(REP8(R8(OP)) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
#undef OP
    } else if (KIND == K_DPP) {
#define OP(k) "v_mov_b32_dpp %" #k ", %8 wave_shl:1 row_mask:0xf bank_mask:0xf\n"
      asm volatile(REP8(R8(OP)) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
#undef OP
    } else if (KIND == K_READLANE) {
      asm volatile(REP8(REP8("v_readlane_b32 %0, %4, 5\n" "v_readlane_b32 %1, %4, 6\n" "v_readlane_b32 %2, %4, 7\n" "v_readlane_b32 %3, %4, 8\n")
                        "s_nop 0\n")
                   : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "v"(b));
      // (64 x 4 / ... : this kind issues 256 v_readlane per block; accounted for below)
    } else if (KIND == K_VALU_MIX) {
      // the kernel's vector mix in rough proportion: packed min / add, address adds, plain min, bit select
      asm volatile(REP8("v_pk_min_i16 %0, %0, %8\n" "v_add_u32 %1, %1, %8\n" "v_pk_add_i16 %2, %2, %8 clamp\n" "v_add_u32 %3, %3, %8\n"
                        "v_pk_min_i16 %4, %4, %8\n" "v_min_i32 %5, %5, %8\n" "v_bfi_b32 %6, %9, %6, %8\n" "v_add_u32 %7, %7, %8\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    } else if (KIND == K_SALU) {
      asm volatile(REP8(REP8("s_add_u32 %0, %0, %1\n" "s_add_u32 %1, %1, %2\n" "s_add_u32 %2, %2, %3\n" "s_add_u32 %3, %3, %0\n"
                             "s_add_u32 %0, %0, %2\n" "s_add_u32 %1, %1, %3\n" "s_add_u32 %2, %2, %0\n" "s_add_u32 %3, %3, %1\n")
                        "s_nop 0\n")
                   : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
    } else if (KIND == K_DS_U16) {
#define OP(k) "ds_read_u16 %" #k ", %8 offset:1" #k "28\n"
      asm volatile(REP8(R8(OP) "s_waitcnt lgkmcnt(0)\n")
                   : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5), "=&v"(a6), "=&v"(a7) : "v"(ad16) : "memory");
#undef OP
    } else if (KIND == K_DS_B32) {
#define OP(k) "ds_read_b32 %" #k ", %8 offset:1" #k "56\n"
      asm volatile(REP8(R8(OP) "s_waitcnt lgkmcnt(0)\n")
                   : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5), "=&v"(a6), "=&v"(a7) : "v"(ad32) : "memory");
#undef OP
    } else if (KIND == K_DS_2B32) {
      uint64_t w0, w1, w2, w3, w4, w5, w6, w7;
#define OP(k) "ds_read2_b32 %" #k ", %8 offset0:1" #k "0 offset1:1" #k "7\n"
      asm volatile(REP8(R8(OP) "s_waitcnt lgkmcnt(0)\n")
                   : "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(w3), "=&v"(w4), "=&v"(w5), "=&v"(w6), "=&v"(w7) : "v"(ad32) : "memory");
#undef OP
      a0 += (uint32_t)(w0 ^ w1 ^ w2 ^ w3 ^ w4 ^ w5 ^ w6 ^ w7);
    } else if (KIND == K_DS_W16) {
#define OP(k) "ds_write_b16 %8, %" #k " offset:1" #k "28\n"
      asm volatile(REP8(R8(OP) "s_waitcnt lgkmcnt(0)\n")
                   : : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7), "v"(ad16) : "memory");
#undef OP
    } else if (KIND == K_KERNEL_MIX) {
      // 10 vector : 6 scalar : 3 LDS per group (sf_mfe_fast_kernel<128,120>: 109 k : 69 k : 33 k per fold), four groups,
      // one wait per block — the kernel waits ~20 times per ~900 instructions
      uint32_t l0, l1, l2;
#define GRP(o)                                                                                                              \
  "ds_read_u16 %8, %17 offset:" #o "00\n" "v_pk_min_i16 %0, %0, %15\n" "s_add_u32 %11, %11, %12\n" "v_add_u32 %1, %1, %15\n"  \
  "v_pk_add_i16 %2, %2, %15 clamp\n" "s_add_u32 %12, %12, %13\n" "ds_read_b32 %9, %18 offset:" #o "56\n" "v_add_u32 %3, %3, %15\n" \
  "s_add_u32 %13, %13, %14\n" "v_pk_min_i16 %4, %4, %15\n" "v_min_i32 %5, %5, %15\n" "s_add_u32 %14, %14, %11\n"                \
  "ds_read_u16 %10, %17 offset:" #o "28\n" "v_bfi_b32 %6, %16, %6, %15\n" "s_add_u32 %11, %11, %13\n" "v_add_u32 %7, %7, %15\n"  \
  "v_pk_min_i16 %0, %0, %15\n" "s_add_u32 %12, %12, %14\n" "v_add_u32 %1, %1, %15\n"
      asm volatile(GRP(1) GRP(2) GRP(3) GRP(4) "s_waitcnt lgkmcnt(0)\n"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=&v"(l0), "=&v"(l1), "=&v"(l2),
                     "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3)
                   : "v"(b), "v"(c), "v"(ad16), "v"(ad32)
                   : "memory", "scc");
#undef GRP
      a7 += l0 + l1 + l2;
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 256 + tid] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ s0 ^ s1 ^ s2 ^ s3;
  if ((tid & 63) == 0) cyc[blockIdx.x * 4 + (tid >> 6)] = t1 - t0;
}

template <int KIND>
static void run(int n_cu, uint32_t *d_out, long long *d_cyc, bool first) {
  const int iters = 2000;
  (void)hipFuncSetAttribute((const void *)probe<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  // v_readlane block issues 8 x 8 x 4 = 256 instructions, the scalar block 8 x 8 x 8 = 512
  const int nv = KIND == K_READLANE ? 256 : NV[KIND], ns = KIND == K_SALU ? 512 : NS[KIND], nl = NL[KIND];
  printf("%s  \"%s\": {\"valu_per_block\": %d, \"salu_per_block\": %d, \"lds_per_block\": %d, \"by_waves_per_simd\": {", first ? "" : ",\n",
         KNAME[KIND], nv, ns, nl);
  const int ks[3] = {1, 2, 4};
  for (int q = 0; q < 3; q++) {
    const int k = ks[q];
    const size_t lds = (size_t)(160 * 1024 / k) - (k == 1 ? 0 : 1280);  // exactly k workgroups per CU
    const int grid = n_cu * k;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    probe<KIND><<<grid, 256, lds>>>(d_out, d_cyc, 50);  // warm-up
    (void)hipEventRecord(e0);
    probe<KIND><<<grid, 256, lds>>>(d_out, d_cyc, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(grid * 4);
    (void)hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double mean = 0; long long mx = 0;
    for (long long v : h) { mean += (double)v; if (v > mx) mx = v; }
    mean /= (double)h.size();
    const double blocks = (double)iters;
    const double tot = (double)(nv + ns + nl);
    printf("%s\"%d\": {\"wave_ticks_mean\": %.0f, \"wave_ticks_max\": %lld, \"launch_ms\": %.4f, \"ticks_per_inst_per_simd\": %.3f",
           q ? ", " : "", k, mean, mx, ms, mean / (blocks * tot * k));
    if (nl) printf(", \"ticks_per_lds_inst_per_cu\": %.3f", mean / (blocks * nl * k * 4));
    if (nv && (ns || nl)) printf(", \"ticks_per_valu_inst_per_simd\": %.3f", mean / (blocks * nv * k));
    printf(", \"ticks_per_us\": %.1f}", mean / (ms * 1e3));
  }
  printf("}}");
}

int main() {
  hipDeviceProp_t prop;
  (void)hipGetDeviceProperties(&prop, 0);
  const int n_cu = prop.multiProcessorCount;
  uint32_t *d_out; long long *d_cyc;
  (void)hipMalloc(&d_out, (size_t)n_cu * 4 * 256 * 4);
  (void)hipMalloc(&d_cyc, (size_t)n_cu * 4 * 4 * 8);
  printf("{\"device\": \"%s\", \"n_cu\": %d, \"clock_khz\": %d, \"what\": \"ticks (s_memtime) per wave-instruction; by_waves_per_simd = "
         "workgroups of four waves per CU\", \"kinds\": {\n", prop.gcnArchName, n_cu, prop.clockRate);
  run<K_PKMIN>(n_cu, d_out, d_cyc, true);
  run<K_PKADD>(n_cu, d_out, d_cyc, false);
  run<K_ADD>(n_cu, d_out, d_cyc, false);
  run<K_MIN>(n_cu, d_out, d_cyc, false);
  run<K_BFI>(n_cu, d_out, d_cyc, false);
  run<K_PERM>(n_cu, d_out, d_cyc, false);
  run<K_DPP>(n_cu, d_out, d_cyc, false);
  run<K_READLANE>(n_cu, d_out, d_cyc, false);
  run<K_VALU_MIX>(n_cu, d_out, d_cyc, false);
  run<K_SALU>(n_cu, d_out, d_cyc, false);
  run<K_DS_U16>(n_cu, d_out, d_cyc, false);
  run<K_DS_B32>(n_cu, d_out, d_cyc, false);
  run<K_DS_2B32>(n_cu, d_out, d_cyc, false);
  run<K_DS_W16>(n_cu, d_out, d_cyc, false);
  run<K_KERNEL_MIX>(n_cu, d_out, d_cyc, false);
  printf("\n}}\n");
  return 0;
}
