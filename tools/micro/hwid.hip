// Where do the waves of a workgroup land?  (block, wave) -> (XCC, SE, CU, SIMD), for a persistent grid shaped like
// the MFE kernel's (1024 workgroups of 256 threads, 40 kB of LDS each = 4 workgroups per CU).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <map>
#include <vector>
__global__ void k(uint32_t *out, int spin) {
  extern __shared__ char smem[];
  smem[threadIdx.x] = 0;
  const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
  const uint32_t xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
  // stay resident long enough for the whole grid to be placed
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  while ((int64_t)(__builtin_amdgcn_s_memtime() - t0) < spin) {}
  if ((threadIdx.x & 63) == 0) {
    out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw;
    out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc;
  }
}
int main() {
  const int nb = 1024;
  uint32_t *d; (void)hipMalloc(&d, nb * 4 * 2 * 4);
  (void)hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  k<<<nb, 256, 40900>>>(d, 2000000);
  std::vector<uint32_t> h(nb * 8);
  (void)hipMemcpy(h.data(), d, nb * 8 * 4, hipMemcpyDeviceToHost);
  // SIMD of wave w of a block; which blocks share a CU
  int simd_eq_wave = 0, total = 0;
  std::map<uint32_t, std::vector<int>> cu_blocks;
  for (int b = 0; b < nb; b++) {
    for (int w = 0; w < 4; w++) {
      const uint32_t hw = h[(b * 4 + w) * 2], xcc = h[(b * 4 + w) * 2 + 1] & 0xf;
      const int simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, se = (hw >> 13) & 7;
      if (w == 0) cu_blocks[(xcc << 16) | (se << 8) | cu].push_back(b);
      simd_eq_wave += (simd == w);
      total++;
      if (b < 6) printf("block %d wave %d: xcc %u se %d cu %d simd %d\n", b, w, xcc, se, cu, simd);
    }
  }
  printf("waves with simd == wave index: %d of %d\n", simd_eq_wave, total);
  int shown = 0;
  for (auto &kv : cu_blocks) {
    if (shown++ < 4) { printf("CU %06x hosts blocks:", kv.first); for (int b : kv.second) printf(" %d", b); printf("\n"); }
  }
  printf("distinct CUs seen: %zu\n", cu_blocks.size());
  // SIMD of wave 0 across the blocks of one CU
  for (auto &kv : cu_blocks) { printf("first CU: wave-0 SIMDs:"); for (int b : kv.second) printf(" %u", (h[(b * 4) * 2] >> 4) & 3); printf("\n"); break; }
  return 0;
}
