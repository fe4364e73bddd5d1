/*
 * sf_shuffle_oracle.c — CPU restatement of the shuffle background of the scan.  TEST INFRASTRUCTURE ONLY
 * (same rule as sf_oracle.c: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline / verification
 * legs load it; the product never does).
 *
 * What it restates.  scramble(frag, r, type) of the reference:
 *   "di"   dinuclShuffle   ScanFold-Scan.py:187-209 with computeCountAndLists :87-115, chooseEdge :118-135,
 *                          connectedToLast :138-160, eulerian :163-175, shuffleEdgeList :178-185
 *                          (identical copy ScanFoldFunctions.py:155-277): the Altschul-Erikson shuffle as
 *                          P. Clote wrote it down in 2003;
 *   "mono" randomizer      ScanFold-Scan.py:248-250  (a uniform permutation of the window).
 * The reference draws from Python's process-global, never seeded Mersenne Twister, so its
 * shuffles cannot be reproduced even by itself.  The product therefore defines its own random stream
 * (include/scanfold_hip.h, sf_shuffle_windows): Philox4x32-10 with key = the 64-bit seed and counter =
 * (block number, shuffle index k, absolute window index, shuffle kind).  This file follows the REFERENCE's
 * algorithm step by step (per-vertex edge lists as lists, a real graph search for "connected to the last
 * character", list removal / append) on that stream, so that it can be compared bit for bit with the device
 * kernel, which is organised quite differently (flat arrays, fixed closure rounds, LDS staging).  The
 * reference's alphabet is A, C, G, U (anything else: KeyError); the product treats N as
 * a fifth vertex that comes after U, and so does this file.
 *
 * How the stream is consumed (the contract both sides implement):
 *   u32     : the next of the four 32-bit words of the current Philox block, a new block (counter word 0 + 1)
 *             when the four are used up;
 *   double  : a = u32 >> 5, b = u32 >> 6, (a * 2^26 + b) / 2^53 — the construction of random.random();
 *   mono    : Fisher-Yates from the last position down, j = floor(u32 * (i + 1) / 2^32);
 *   di      : chooseEdge draws one double per vertex that has successors and is not the last character, in
 *             the order A, C, G, U, N; the whole draw is repeated until every vertex reaches the last
 *             character; shuffleEdgeList then draws len - 1 doubles per vertex in the same vertex order.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  uint32_t key[2];
  uint32_t ctr[4];
  uint32_t word[4];
  int used; /* how many of word[] have been handed out; 4 = need a new block */
} stream_t;

static void philox_round(uint32_t c[4], const uint32_t k[2]) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
  const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0];
  const uint32_t n1 = (uint32_t)p1;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1];
  const uint32_t n3 = (uint32_t)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

static void stream_open(stream_t *s, uint64_t seed, uint32_t shuffle_index, uint32_t window, uint32_t kind) {
  s->key[0] = (uint32_t)seed;
  s->key[1] = (uint32_t)(seed >> 32);
  s->ctr[0] = 0; s->ctr[1] = shuffle_index; s->ctr[2] = window; s->ctr[3] = kind;
  s->used = 4;
}

static uint32_t next_u32(stream_t *s) {
  if (s->used == 4) {
    uint32_t c[4] = {s->ctr[0], s->ctr[1], s->ctr[2], s->ctr[3]};
    uint32_t k[2] = {s->key[0], s->key[1]};
    for (int round = 0; round < 10; round++) {
      philox_round(c, k);
      k[0] += 0x9E3779B9u; /* Weyl sequence of the key schedule */
      k[1] += 0xBB67AE85u;
    }
    memcpy(s->word, c, sizeof c);
    s->ctr[0] += 1;
    s->used = 0;
  }
  return s->word[s->used++];
}

static double next_double(stream_t *s) {
  const uint32_t a = next_u32(s) >> 5;
  const uint32_t b = next_u32(s) >> 6;
  return (a * 67108864.0 + b) / 9007199254740992.0;
}

/* codes of the product: 0 = N / anything else, 1 = A, 2 = C, 3 = G, 4 = U (T) */
static int code_of(unsigned char ch) {
  switch (ch) {
    case 'A': case 'a': case 1: return 1;
    case 'C': case 'c': case 2: return 2;
    case 'G': case 'g': case 3: return 3;
    case 'U': case 'u': case 'T': case 't': case 4: return 4;
    default: return 0;
  }
}

/* the reference iterates nuclList = ["A","C","G","U"]; N is appended as a fifth vertex */
static const int VERTEX[5] = {1, 2, 3, 4, 0};

typedef struct {
  int *item;
  int len;
} list_t;

static void list_remove_first(list_t *l, int value) { /* Python's list.remove */
  for (int i = 0; i < l->len; i++)
    if (l->item[i] == value) {
      memmove(l->item + i, l->item + i + 1, sizeof(int) * (size_t)(l->len - i - 1));
      l->len--;
      return;
    }
}

static void shuffle_mono(const int *s, int n, stream_t *rng, int *out) {
  memcpy(out, s, sizeof(int) * (size_t)n);
  for (int i = n - 1; i >= 1; i--) {
    const int j = (int)(((uint64_t)next_u32(rng) * (uint64_t)(i + 1)) >> 32);
    const int t = out[i]; out[i] = out[j]; out[j] = t;
  }
}

static void shuffle_di(const int *s, int n, stream_t *rng, int *out) {
  if (n < 2) { if (n == 1) out[0] = s[0]; return; }
  /* computeCountAndLists: dinucleotide counts and, per vertex, the successors in order of occurrence */
  int count[5][5];
  memset(count, 0, sizeof count);
  list_t succ[5];
  int *pool = (int *)malloc(sizeof(int) * 5 * (size_t)n);
  for (int v = 0; v < 5; v++) { succ[v].item = pool + (size_t)v * n; succ[v].len = 0; }
  int in_graph[5] = {0, 0, 0, 0, 0};
  for (int i = 0; i + 1 < n; i++) {
    count[s[i]][s[i + 1]]++;
    succ[s[i]].item[succ[s[i]].len++] = s[i + 1];
    in_graph[s[i]] = 1;
  }
  const int last = s[n - 1];
  in_graph[last] = 1;

  /* eulerian: draw a last edge per vertex until the edges form a tree into the last character */
  int last_edge[5];
  for (;;) {
    for (int vi = 0; vi < 5; vi++) {
      const int v = VERTEX[vi];
      last_edge[v] = -1;
      if (!in_graph[v] || v == last) continue;
      /* chooseEdge: walk the cumulative dinucleotide frequencies of v */
      const double z = next_double(rng);
      int total = 0;
      for (int w = 0; w < 5; w++) total += count[v][w];
      int cum = 0, chosen = VERTEX[4];
      for (int wi = 0; wi < 4; wi++) {
        cum += count[v][VERTEX[wi]];
        if (z < (double)cum / (double)total) { chosen = VERTEX[wi]; break; }
      }
      last_edge[v] = chosen;
    }
    /* connectedToLast, as a search from every vertex along its last edge */
    int all_connected = 1;
    for (int v = 0; v < 5 && all_connected; v++) {
      if (!in_graph[v] || v == last) continue;
      int at = v, steps = 0;
      while (at != last && steps < 6) { at = last_edge[at]; steps++; if (at < 0) break; }
      if (at != last) all_connected = 0;
    }
    if (all_connected) break;
  }

  /* dinuclShuffle: take the last edges out, shuffle what is left, put them back at the end */
  for (int vi = 0; vi < 5; vi++) {
    const int v = VERTEX[vi];
    if (last_edge[v] >= 0) list_remove_first(&succ[v], last_edge[v]);
  }
  for (int vi = 0; vi < 5; vi++) {
    const int v = VERTEX[vi];
    if (!in_graph[v]) continue;
    list_t *l = &succ[v];
    int barrier = l->len; /* shuffleEdgeList */
    for (int i = 0; i < l->len - 1; i++) {
      const int z = (int)(next_double(rng) * barrier);
      const int t = l->item[z]; l->item[z] = l->item[barrier - 1]; l->item[barrier - 1] = t;
      barrier--;
    }
  }
  for (int vi = 0; vi < 5; vi++) {
    const int v = VERTEX[vi];
    if (last_edge[v] >= 0) succ[v].item[succ[v].len++] = last_edge[v];
  }
  /* walk: pop the head of the current vertex's list, n - 2 times, then the last character */
  int head[5] = {0, 0, 0, 0, 0};
  int prev = s[0];
  out[0] = prev;
  for (int i = 1; i < n - 1; i++) {
    const int ch = succ[prev].item[head[prev]++];
    out[i] = ch;
    prev = ch;
  }
  out[n - 1] = last;
  free(pool);
}

/* Same contract as sf_shuffle_windows (include/scanfold_hip.h): n_win * (r + 1) rows of W codes, row 0 of a
 * window = the window itself, rows 1..r = its shuffles.  kind: 0 = mono, 1 = di. */
int sfo_shuffle_windows(const unsigned char *transcript, int L, int W, int step, int win_begin, int n_win, int r,
                        int kind, uint64_t seed, unsigned char *rows_out) {
  if (!transcript || !rows_out || W < 1 || step < 1 || n_win < 0 || r < 0 || (kind != 0 && kind != 1)) return -1;
  if (n_win > 0 && (long long)(win_begin + n_win - 1) * step + W > L) return -1;
  int *s = (int *)malloc(sizeof(int) * (size_t)W), *o = (int *)malloc(sizeof(int) * (size_t)W);
  for (int w = 0; w < n_win; w++) {
    const unsigned char *src = transcript + (size_t)(win_begin + w) * step;
    for (int i = 0; i < W; i++) s[i] = code_of(src[i]);
    unsigned char *dst = rows_out + (size_t)w * (r + 1) * W;
    for (int i = 0; i < W; i++) dst[i] = (unsigned char)s[i];
    for (int k = 1; k <= r; k++) {
      stream_t rng;
      stream_open(&rng, seed, (uint32_t)k, (uint32_t)(win_begin + w), (uint32_t)kind);
      if (kind == 0) shuffle_mono(s, W, &rng, o);
      else shuffle_di(s, W, &rng, o);
      unsigned char *row = dst + (size_t)k * W;
      for (int i = 0; i < W; i++) row[i] = (unsigned char)o[i];
    }
  }
  free(s);
  free(o);
  return 0;
}
