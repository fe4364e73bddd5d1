/* TEST-ONLY: fiber scheduler behind hip_emul.h (see that header). */
#include "hip_emul.h"

namespace sfemul {
Block *g_blk = nullptr;
dim3 g_blockIdx, g_gridDim, g_blockDim;

static void trampoline() {
  Block *b = g_blk;
  b->body();
  b->fibers[b->cur].done = true;
  swapcontext(&b->fibers[b->cur].ctx, &b->sched);
}

void run_block(unsigned nthreads, size_t shmem, const std::function<void()> &body) {
  static Block blk;
  Block *b = &blk;
  g_blk = b;
  b->nthreads = nthreads;
  b->body = body;
  b->bar_count = 0;
  b->bar_gen = 0;
  unsigned nw = (nthreads + 63) / 64;
  b->wbar_count.assign(nw, 0);
  b->wbar_gen.assign(nw, 0);
  b->exch.assign(nthreads, 0);
  b->smem.assign(shmem + 64, 0);
  if (b->fibers.size() < nthreads) b->fibers.resize(nthreads);
  const size_t STK = 256 * 1024;
  for (unsigned t = 0; t < nthreads; t++) {
    Fiber &f = b->fibers[t];
    if (f.stack.size() < STK) f.stack.resize(STK);
    f.done = false;
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = f.stack.data();
    f.ctx.uc_stack.ss_size = f.stack.size();
    f.ctx.uc_link = &b->sched;
    makecontext(&f.ctx, (void (*)())trampoline, 0);
  }
  unsigned remaining = nthreads;
  while (remaining) {
    unsigned progressed = 0;
    for (unsigned t = 0; t < nthreads; t++) {
      if (b->fibers[t].done) continue;
      b->cur = (int)t;
      swapcontext(&b->sched, &b->fibers[t].ctx);
      if (b->fibers[t].done) remaining--;
      progressed++;
    }
    if (!progressed) break;
  }
  b->cur = -1;
}
}  // namespace sfemul
