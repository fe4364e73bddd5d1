// Placement probe: a persistent grid shaped like the W=120 MFE kernel's (1024 workgroups of 256 threads, 40 960 B of
// LDS each = 4 per CU).  Every wave reports (XCC, SE, CU, SIMD).  Built as a shared library so that a Python script can
// run it in the same process right after other kernels (tools/gpu_placement.py): the question is whether the four
// workgroups of a CU get their waves on the SIMDs in the same or in rotated order, and whether that depends on what ran
// on the CU before.   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/micro/libplace_probe.so tools/micro/place_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
__global__ void probe_kernel(uint32_t *out, int spin) {
  extern __shared__ char smem[];
  smem[threadIdx.x] = 0;
  const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
  const uint32_t xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
  uint64_t t0 = __builtin_amdgcn_s_memtime();
  while ((int64_t)(__builtin_amdgcn_s_memtime() - t0) < spin) {}    // stay resident until the whole grid is placed
  if ((threadIdx.x & 63) == 0) {
    out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = hw;
    out[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = xcc;
  }
}
extern "C" int place_probe(uint32_t *host_out /* 1024 * 4 * 2 words */) {
  static uint32_t *d = nullptr;
  const int nb = 1024;
  if (!d) {
    if (hipMalloc(&d, nb * 8 * 4) != hipSuccess) return 1;
    if (hipFuncSetAttribute((const void *)probe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return 2;
  }
  probe_kernel<<<nb, 256, 40960>>>(d, 400000);
  if (hipDeviceSynchronize() != hipSuccess) return 3;
  return hipMemcpy(host_out, d, nb * 8 * 4, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 4;
}
