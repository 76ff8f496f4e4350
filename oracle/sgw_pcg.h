/* sgw_pcg.h -- CPU ORACLE (TEST INFRASTRUCTURE ONLY): numpy Generator(PCG64) primitives shared by the multi-agent oracles.
 * Follows numpy/random/src/pcg64/pcg64.h (pcg_setseq_128 XSL-RR 128/64, buffered next_uint32),
 * _generator.pyx (random(), shuffle) and distributions.c (random_interval). */
#ifndef SGW_PCG_H_
#define SGW_PCG_H_
#include <stdint.h>

typedef unsigned __int128 u128;
typedef struct { u128 state, inc; int has_uint32; uint32_t uinteger; } pcg_t;
static const u128 PCG_MULT = (((u128)0x2360ED051FC65DA4ULL) << 64) | 0x4385DF649FCCF645ULL;

static inline uint64_t pcg_next64(pcg_t* g) {        /* pcg64.h: pcg_setseq_128_step_r then XSL-RR 128/64 */
  g->state = g->state * PCG_MULT + g->inc;
  uint64_t hi = (uint64_t)(g->state >> 64), lo = (uint64_t)g->state;
  uint64_t x = hi ^ lo;
  unsigned rot = (unsigned)(hi >> 58);
  return (x >> rot) | (x << ((-rot) & 63));
}
static inline uint32_t pcg_next32(pcg_t* g) {        /* pcg64.h pcg64_next32: low half first, high half buffered */
  if (g->has_uint32) { g->has_uint32 = 0; return g->uinteger; }
  uint64_t n = pcg_next64(g);
  g->has_uint32 = 1; g->uinteger = (uint32_t)(n >> 32);
  return (uint32_t)n;
}
static inline double pcg_random(pcg_t* g) { return (double)(pcg_next64(g) >> 11) * (1.0 / 9007199254740992.0); }
static inline uint64_t random_interval(pcg_t* g, uint64_t max) {   /* distributions.c random_interval */
  if (max == 0) return 0;
  uint64_t mask = max, value;
  mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16; mask |= mask >> 32;
  if (max <= 0xffffffffULL) { while ((value = (pcg_next32(g) & mask)) > max) {} }
  else { while ((value = (pcg_next64(g) & mask)) > max) {} }
  return value;
}

/* distributions.c bounded_lemire_uint32 with the bitgen's buffered next_uint32 (what random_bounded_uint64(off=0, rng,
 * mask=0, use_masked=false) reaches for rng < 2^32-1): Generator.integers / choice(list) / the Floyd sampler. */
static inline uint32_t pcg_bounded_lemire32(pcg_t* g, uint32_t rng) {
  if (rng == 0) return 0;
  const uint32_t rng_excl = rng + 1;
  uint64_t m = (uint64_t)pcg_next32(g) * rng_excl;
  uint32_t leftover = (uint32_t)m;
  if (leftover < rng_excl) {
    const uint32_t threshold = (uint32_t)((0xffffffffu - rng) % rng_excl);
    while (leftover < threshold) { m = (uint64_t)pcg_next32(g) * rng_excl; leftover = (uint32_t)m; }
  }
  return (uint32_t)(m >> 32);
}
/* _generator.pyx Generator.choice(pop, size, replace=False), pop <= 10000: Floyd's algorithm (the hash set only
 * answers "was val drawn already") followed by _shuffle_int over the picks. */
static inline void pcg_choice_noreplace(pcg_t* g, int pop, int size, int* idx) {
  for (int j = pop - size, k = 0; j < pop; ++j, ++k) {
    int val = (int)pcg_bounded_lemire32(g, (uint32_t)j), seen = 0;
    for (int q = 0; q < k; ++q) seen |= (idx[q] == val);
    idx[k] = seen ? j : val;
  }
  for (int i = size - 1; i >= 1; --i) {
    int j = (int)pcg_bounded_lemire32(g, (uint32_t)i);
    int t = idx[i]; idx[i] = idx[j]; idx[j] = t;
  }
}

#endif  /* SGW_PCG_H_ */
