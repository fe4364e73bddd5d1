// Microbenchmark: throughput of ds_read_b32 at 4-byte-aligned vs 2-byte-aligned (odd int16 index) addresses.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
struct __attribute__((packed, aligned(2))) u32a2 { uint32_t v; };
__global__ void k(uint32_t *out, int off, int iters) {
  __shared__ int16_t tab[8192];
  for (int x = threadIdx.x; x < 8192; x += blockDim.x) tab[x] = (int16_t)x;
  __syncthreads();
  uint32_t acc = 0;
  const int16_t *p = tab + 2 * (threadIdx.x & 63) + off;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 16; u++) acc += ((const u32a2 *)(p + 232 * u + ((it & 3) << 1)))->v;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main() {
  uint32_t *d; hipMalloc(&d, 1024 * 256 * 4);
  for (int off = 0; off < 2; off++) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<<<1024, 256>>>(d, off, 100);
    hipEventRecord(a);
    k<<<1024, 256>>>(d, off, 2000);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double reads = 1024.0 * 4 * 2000 * 16;  // wave-level ds_read_b32 instructions
    printf("offset %d int16: %.3f ms, %.2f cycles per wave-read per CU at 2.4 GHz\n", off, ms, ms * 1e-3 * 2.4e9 / (reads / 256));
  }
  return 0;
}
