// sf_shuffle.hip.h — window extraction + the r-fold shuffle background, on the device.
//
// Replaces scramble(frag, r, type) (ScanFold-Scan.py:266-282; ScanFoldFunctions.py:834-851):
//   "di"   -> dinuclShuffle: Altschul-Erikson dinucleotide-preserving shuffle in the form of
//             ScanFold-Scan.py:87-209 (weighted draw of a last edge per vertex, retry until every vertex
//             reaches the last character, remove those edges, Fisher-Yates each vertex's successor
//             list with int(random()*barrier), re-append, walk the Euler path);
//   "mono" -> randomizer: uniform permutation (ScanFold-Scan.py:248-250).
// Differences, deliberate: the random stream is Philox4x32-10 keyed by (seed, window index, shuffle
// index) instead of the process-global, never-seeded Mersenne Twister, and an N in
// the window is an ordinary fifth symbol instead of a KeyError.  One thread per output row; row 0 of a
// window is the native window.  Rows are staged in LDS and written out coalesced.
#pragma once
#include "sf_energy.h"

struct SfPhilox {
  uint32_t k0, k1;
  uint32_t c0, c1, c2, c3;
  uint32_t out[4];
  int have;
};
__device__ __forceinline__ uint32_t sf_mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
__device__ inline void sf_philox_block(SfPhilox *g) {
  uint32_t c0 = g->c0, c1 = g->c1, c2 = g->c2, c3 = g->c3, k0 = g->k0, k1 = g->k1;
  for (int r = 0; r < 10; r++) {
    const uint32_t hi0 = sf_mulhi32(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = sf_mulhi32(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  g->out[0] = c0; g->out[1] = c1; g->out[2] = c2; g->out[3] = c3;
  g->c0++;  // next block of this stream
  g->have = 4;
}
__device__ inline uint32_t sf_rand_u32(SfPhilox *g) {
  if (!g->have) sf_philox_block(g);
  return g->out[4 - g->have--];
}
// 53-bit uniform in [0,1), built like MT19937's genrand_res53 (what random.random() returns)
__device__ inline double sf_rand_double(SfPhilox *g) {
  const uint32_t a = sf_rand_u32(g) >> 5, b = sf_rand_u32(g) >> 6;
  return ((double)a * 67108864.0 + (double)b) * (1.0 / 9007199254740992.0);
}

#define SF_SHUF_BLOCK 64

// LDS per block: rows[64][W] output rows + lst[64][W] successor lists + cnt[64][25] uint16
__global__ void sf_shuffle_kernel(const uint8_t *__restrict__ transcript, int L, int W, int step, int win_begin,
                                  int n_win, int r, int kind, uint64_t seed, uint8_t *__restrict__ seqs_out) {
  SF_DYN_SMEM(smem);
  uint8_t *rows = (uint8_t *)smem;                             // [64][W]
  uint8_t *lsts = rows + (size_t)SF_SHUF_BLOCK * W;            // [64][W]
  uint16_t *cnts = (uint16_t *)(lsts + (((size_t)SF_SHUF_BLOCK * W + 3) & ~(size_t)3));  // [64][25]
  const int lane = threadIdx.x;
  const long long total = (long long)n_win * (r + 1);
  const long long row0 = (long long)blockIdx.x * SF_SHUF_BLOCK;
  const long long row = row0 + lane;
  uint8_t *out = rows + (size_t)lane * W;
  uint8_t *lst = lsts + (size_t)lane * W;
  uint16_t *cnt = cnts + lane * 25;
  (void)L;

  if (row < total) {
    const int w = (int)(row / (r + 1)), k = (int)(row % (r + 1));
    const uint8_t *src = transcript + (size_t)(win_begin + w) * step;
    if (k == 0) {
      for (int x = 0; x < W; x++) out[x] = sf_encode_nt(src[x]);
    } else {
      SfPhilox g;
      g.k0 = (uint32_t)seed; g.k1 = (uint32_t)(seed >> 32);
      g.c0 = 0; g.c1 = (uint32_t)k; g.c2 = (uint32_t)(win_begin + w); g.c3 = (uint32_t)kind;
      g.have = 0;
      if (kind == SF_SHUFFLE_MONO) {
        for (int x = 0; x < W; x++) out[x] = sf_encode_nt(src[x]);
        for (int i = W - 1; i >= 1; i--) {
          const int jj = (int)(((uint64_t)sf_rand_u32(&g) * (uint32_t)(i + 1)) >> 32);
          const uint8_t t = out[i]; out[i] = out[jj]; out[jj] = t;
        }
      } else {
        // vertex order of the reference's lists: A, C, G, U (codes 1..4), then N (0)
        const int order[5] = {1, 2, 3, 4, 0};
        for (int x = 0; x < 25; x++) cnt[x] = 0;
        uint8_t prev = sf_encode_nt(src[0]);
        const uint8_t first = prev;
        for (int x = 1; x < W; x++) {
          const uint8_t y = sf_encode_nt(src[x]);
          cnt[prev * 5 + y]++;
          prev = y;
        }
        const int lastCh = prev;
        int outdeg[5], present[5], lastedge[5];
        for (int a = 0; a < 5; a++) {
          int s = 0;
          for (int b = 0; b < 5; b++) s += cnt[a * 5 + b];
          outdeg[a] = s;
          present[a] = (s > 0) || (a == lastCh);
        }
        // draw last edges until every present vertex is connected to lastCh
        for (;;) {
          for (int oa = 0; oa < 5; oa++) {
            const int a = order[oa];
            lastedge[a] = -1;
            if (!present[a] || a == lastCh) continue;
            const double z = sf_rand_double(&g);
            const double denom = (double)outdeg[a];
            int num = 0, pick = order[4];
            for (int ob = 0; ob < 4; ob++) {
              num += cnt[a * 5 + order[ob]];
              if (z < (double)num / denom) { pick = order[ob]; break; }
            }
            lastedge[a] = pick;
          }
          int Dm[5];
          for (int a = 0; a < 5; a++) Dm[a] = (lastedge[a] == lastCh);
          for (int rnd = 0; rnd < 4; rnd++)
            for (int oa = 0; oa < 5; oa++) {
              const int a = order[oa];
              if (lastedge[a] >= 0 && Dm[lastedge[a]]) Dm[a] = 1;
            }
          int ok = 1;
          for (int a = 0; a < 5; a++)
            if (present[a] && a != lastCh && !Dm[a]) ok = 0;
          if (ok) break;
        }
        // successor lists in order of occurrence
        int start[5], fill[5];
        {
          int acc = 0;
          for (int oa = 0; oa < 5; oa++) { const int a = order[oa]; start[a] = acc; acc += outdeg[a]; fill[a] = 0; }
        }
        prev = first;
        for (int x = 1; x < W; x++) {
          const uint8_t y = sf_encode_nt(src[x]);
          lst[start[prev] + fill[prev]++] = y;
          prev = y;
        }
        // remove the first occurrence of the chosen last edge, shuffle the rest, put the edge back at the end
        for (int oa = 0; oa < 5; oa++) {
          const int a = order[oa];
          if (lastedge[a] < 0) continue;
          uint8_t *Lx = lst + start[a];
          int pos = 0;
          while (Lx[pos] != (uint8_t)lastedge[a]) pos++;
          for (int x = pos; x + 1 < outdeg[a]; x++) Lx[x] = Lx[x + 1];
        }
        for (int oa = 0; oa < 5; oa++) {
          const int a = order[oa];
          if (!present[a]) continue;
          uint8_t *Lx = lst + start[a];
          const int nlist = outdeg[a] - (lastedge[a] >= 0 ? 1 : 0);
          int barrier = nlist;
          for (int x = 0; x < nlist - 1; x++) {
            const int z = (int)(sf_rand_double(&g) * (double)barrier);
            const uint8_t t = Lx[z]; Lx[z] = Lx[barrier - 1]; Lx[barrier - 1] = t;
            barrier--;
          }
          if (lastedge[a] >= 0) Lx[nlist] = (uint8_t)lastedge[a];
        }
        // Euler walk
        int ptr[5] = {0, 0, 0, 0, 0};
        out[0] = first;
        int pc = first;
        for (int x = 1; x < W - 1; x++) {
          const uint8_t ch = lst[start[pc] + ptr[pc]++];
          out[x] = ch;
          pc = ch;
        }
        if (W > 1) out[W - 1] = (uint8_t)lastCh;
      }
    }
  }
  __syncthreads();
  // coalesced copy-out of the block's contiguous rows
  long long nrows = total - row0;
  if (nrows > SF_SHUF_BLOCK) nrows = SF_SHUF_BLOCK;
  const size_t nbytes = (size_t)nrows * W;
  uint8_t *dst = seqs_out + (size_t)row0 * W;
  for (size_t x = lane; x < nbytes; x += SF_SHUF_BLOCK) dst[x] = rows[x];
}
