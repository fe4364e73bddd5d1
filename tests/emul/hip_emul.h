/*
 * hip_emul.h — TEST-ONLY CPU emulation of the small HIP subset the kernels in scanfold_amd/csrc use.
 *
 * Purpose: this container has no GPU, and a gpurun round-trip takes minutes; compiling the very same
 * kernel sources with g++ against this header lets tests/ (and AddressSanitizer/UBSan, which are not
 * available on the GPU pool) exercise the kernel logic on the CPU.  The resulting library
 * (tests/emul/libscanfold_emul.so) is loaded ONLY by tests/test_emul_*.py; the product loader
 * (scanfold_amd/_lib.py) never looks at it and fails loudly when libscanfold_hip.so is missing.
 *
 * Model: one workgroup at a time; its threads are ucontext fibers scheduled round-robin; a fiber runs
 * until it reaches a barrier (__syncthreads) or a wave-level exchange (__shfl*, __ballot), which are
 * implemented as barriers over the 64 fibers of the wave plus an exchange buffer.
 */
#ifndef SF_HIP_EMUL_H
#define SF_HIP_EMUL_H
#ifndef SF_EMUL
#define SF_EMUL 1
#endif

#include <ucontext.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};

namespace sfemul {
struct Fiber {
  ucontext_t ctx;
  std::vector<char> stack;
  bool done = false;
};
struct Block {
  std::vector<Fiber> fibers;
  ucontext_t sched;
  int cur = -1;
  unsigned nthreads = 0;
  // block barrier
  unsigned bar_count = 0, bar_gen = 0;
  // per-wave barrier + exchange
  std::vector<unsigned> wbar_count, wbar_gen;
  std::vector<uint64_t> exch;
  std::vector<char> smem;
  std::function<void()> body;
};
extern Block *g_blk;
extern dim3 g_blockIdx, g_gridDim, g_blockDim;
extern thread_local int dummy;

inline void yield() { swapcontext(&g_blk->fibers[g_blk->cur].ctx, &g_blk->sched); }
inline int tid() { return g_blk->cur; }

inline void block_barrier() {
  Block *b = g_blk;
  unsigned gen = b->bar_gen;
  if (++b->bar_count == b->nthreads) {
    b->bar_count = 0;
    b->bar_gen++;
  } else {
    while (b->bar_gen == gen) yield();
  }
}
inline void wave_barrier() {
  Block *b = g_blk;
  int w = tid() / 64;
  unsigned wsize = (unsigned)std::min<int>(64, (int)b->nthreads - w * 64);
  unsigned gen = b->wbar_gen[w];
  if (++b->wbar_count[w] == wsize) {
    b->wbar_count[w] = 0;
    b->wbar_gen[w]++;
  } else {
    while (b->wbar_gen[w] == gen) yield();
  }
}
void run_block(unsigned nthreads, size_t shmem, const std::function<void()> &body);
}  // namespace sfemul

struct sfemul_tidx {
  struct X { operator unsigned() const { return (unsigned)sfemul::tid(); } } x;
  unsigned y = 0, z = 0;
};
static sfemul_tidx threadIdx;
#define blockIdx sfemul::g_blockIdx
#define gridDim sfemul::g_gridDim
#define blockDim sfemul::g_blockDim

inline void __syncthreads() { sfemul::block_barrier(); }

template <typename T>
inline T __shfl(T v, int src, int width = 64) {
  (void)width;
  sfemul::Block *b = sfemul::g_blk;
  int t = sfemul::tid();
  uint64_t raw = 0;
  memcpy(&raw, &v, sizeof(T));
  b->exch[t] = raw;
  sfemul::wave_barrier();
  int base = (t / 64) * 64;
  uint64_t got = b->exch[base + (src & 63)];
  sfemul::wave_barrier();
  T out;
  memcpy(&out, &got, sizeof(T));
  return out;
}
template <typename T>
inline T __shfl_xor(T v, int mask, int width = 64) {
  return __shfl(v, (sfemul::tid() & 63) ^ mask, width);
}
template <typename T>
inline T __shfl_down(T v, unsigned delta, int width = 64) {
  int l = (sfemul::tid() & 63) + (int)delta;
  return __shfl(v, l > 63 ? (sfemul::tid() & 63) : l, width);
}
template <typename T>
inline T __shfl_up(T v, unsigned delta, int width = 64) {
  int l = (sfemul::tid() & 63) - (int)delta;
  return __shfl(v, l < 0 ? (sfemul::tid() & 63) : l, width);
}
inline unsigned long long __ballot(int pred) {
  sfemul::Block *b = sfemul::g_blk;
  int t = sfemul::tid();
  b->exch[t] = pred ? 1 : 0;
  sfemul::wave_barrier();
  int base = (t / 64) * 64;
  unsigned long long m = 0;
  for (int l = 0; l < 64 && base + l < (int)b->nthreads; l++)
    if (b->exch[base + l]) m |= 1ull << l;
  sfemul::wave_barrier();
  return m;
}
// minimum over the lanes of a wave in ONE exchange round (the kernels' DPP reduction, sf_wave_min)
inline int sfemul_wave_min(int v) {
  sfemul::Block *b = sfemul::g_blk;
  int t = sfemul::tid();
  b->exch[t] = (uint64_t)(uint32_t)v;
  sfemul::wave_barrier();
  int base = (t / 64) * 64, m = v;
  for (int l = 0; l < 64 && base + l < (int)b->nthreads; l++) {
    const int x = (int)(uint32_t)b->exch[base + l];
    if (x < m) m = x;
  }
  sfemul::wave_barrier();
  return m;
}
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
template <typename T>
inline T atomicAdd(T *p, T v) { T o = *p; *p = o + v; return o; }
template <typename T>
inline T atomicMin(T *p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <typename T>
inline T atomicOr(T *p, T v) { T o = *p; *p = o | v; return o; }

#define SF_WAVE_SYNC() sfemul::wave_barrier()
#define SF_WAVE_UNIFORM(x) (x)
#define SF_SCHED_GROUP(mask, n) ((void)0)
#define SF_SCHED_FENCE() ((void)0)
#define SF_VALU_FENCE() ((void)0)
template <class T> static inline const T *sf_const_base(const T *p) { return p; }
#define SF_PIN(x) ((void)0)
#define SF_LANE_READ(v, l) __shfl((v), (l))

/* dynamic shared memory */
#define SF_DYN_SMEM(name) char *name = sfemul::g_blk->smem.data()

/* ---- runtime subset ---- */
typedef int hipError_t;
typedef void *hipStream_t;
struct sfemul_event { std::chrono::steady_clock::time_point t; };
typedef sfemul_event *hipEvent_t;
#define hipSuccess 0
#define hipMemcpyHostToDevice 1
#define hipMemcpyDeviceToHost 2
#define hipMemcpyDeviceToDevice 3
inline const char *hipGetErrorString(hipError_t) { return "emul"; }
inline hipError_t hipGetLastError() { return 0; }
inline hipError_t hipSetDevice(int) { return 0; }
inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return 0; }
inline hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n ? n : 1); return *p ? 0 : 2; }
inline hipError_t hipFree(void *p) { free(p); return 0; }
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, int) { memcpy(d, s, n); return 0; }
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, int, hipStream_t) { memcpy(d, s, n); return 0; }
inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { memset(d, v, n); return 0; }
inline hipError_t hipMemset(void *d, int v, size_t n) { memset(d, v, n); return 0; }
inline hipError_t hipStreamCreate(hipStream_t *s) { *s = nullptr; return 0; }
inline hipError_t hipStreamDestroy(hipStream_t) { return 0; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
inline hipError_t hipDeviceSynchronize() { return 0; }
inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new sfemul_event; return 0; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return 0; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return 0; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) {
  *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
  return 0;
}
struct hipDeviceProp_t { int multiProcessorCount; char name[64]; char gcnArchName[64]; };
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) {
  p->multiProcessorCount = 2;
  strcpy(p->name, "cpu-emulation");
  strcpy(p->gcnArchName, "emul");
  return 0;
}
#define hipFuncAttributeMaxDynamicSharedMemorySize 0
template <typename F>
inline hipError_t hipFuncSetAttribute(F, int, int) { return 0; }

/* kernel launch: runs blocks sequentially */
template <typename K, typename... A>
inline void sfemul_launch(K kern, dim3 grid, dim3 block, size_t shmem, A... args) {
  sfemul::g_gridDim = grid;
  sfemul::g_blockDim = block;
  for (unsigned bx = 0; bx < grid.x; bx++) {
    sfemul::g_blockIdx = dim3(bx, 0, 0);
    sfemul::run_block(block.x, shmem, [&]() { kern(args...); });
  }
}
#define SF_LAUNCH(kern, grid, block, shmem, stream, ...) \
  do { (void)(stream); sfemul_launch(kern, dim3(grid), dim3(block), shmem, __VA_ARGS__); } while (0)

#endif
