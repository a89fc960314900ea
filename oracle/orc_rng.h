/*
 * orc_rng.h — TEST INFRASTRUCTURE (oracle).
 *
 * The reference draws every random number from rand::thread_rng() (ChaCha12, OS-seeded,
 * one generator per rayon worker: tracing.rs:72,82,164; materials.rs:84,120;
 * geometry.rs:517).  It is unseeded and scheduled dynamically, so no two runs of the
 * reference agree; only the DISTRIBUTION of each draw is part of its behaviour.
 *
 * Spec of the replacement (DESIGN.md "RNG"): one counter-free stream per camera sample,
 * keyed by (seed, pixel index y*W+x, sample index i).  The stream serves, in program
 * order, the draws of generate_rays for that sample and then every draw of its path.
 *
 *   generator  : xoroshiro64** (Blackman & Vigna), 2 x u32 of state
 *   keying     : three rounds of the lowbias32 integer hash, see orc_rng_init
 *   f32 ranges : rand 0.8.4 UniformFloat::sample_single — 23 mantissa bits,
 *                value1_2 = bits(0x3f800000 | u32 >> 9) in [1,2); res = value1_2*scale + (low-scale)
 *   u32 ranges : rand 0.8.4 UniformInt::sample_single — widening multiply with a
 *                rejection zone (range << leading_zeros) - 1
 */
#ifndef ORC_RNG_H
#define ORC_RNG_H

#include <stdint.h>
#include <string.h>

typedef struct { uint32_t s0, s1; uint64_t draws; } orc_rng;

static inline uint32_t orc_lowbias32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
static inline uint32_t orc_rotl32(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }

static inline void orc_rng_init(orc_rng* r, uint32_t seed, uint32_t pixel, uint32_t sample) {
    uint32_t k0 = orc_lowbias32(seed ^ 0x68e31da4u);
    uint32_t p0 = orc_lowbias32(pixel + k0);
    uint32_t p1 = orc_lowbias32(p0 ^ 0xb5297a4du);
    uint32_t s0 = orc_lowbias32(p0 + sample * 0x9e3779b9u);
    uint32_t s1 = orc_lowbias32(p1 ^ (sample * 0x85ebca6bu));
    if ((s0 | s1) == 0u) s1 = 1u;
    r->s0 = s0; r->s1 = s1; r->draws = 0;
}

/* RngCore::next_u32 */
static inline uint32_t orc_next_u32(orc_rng* r) {
    uint32_t s0 = r->s0, s1 = r->s1;
    uint32_t result = orc_rotl32(s0 * 0x9e3779bbu, 5) * 5u;
    s1 ^= s0;
    r->s0 = orc_rotl32(s0, 26) ^ s1 ^ (s1 << 9);
    r->s1 = orc_rotl32(s1, 13);
    r->draws++;
    return result;
}

static inline float orc_value1_2(uint32_t bits) {
    uint32_t u = 0x3f800000u | (bits >> 9);
    float f; memcpy(&f, &u, 4);
    return f;
}
/* rng.gen_range(0.0..1.0)  (materials.rs:84,120; geometry.rs:517): scale 1, offset -1 */
static inline float orc_gen_range_01(orc_rng* r) { return orc_value1_2(orc_next_u32(r)) * 1.0f + (0.0f - 1.0f); }
/* rng.gen_range(-1.0..1.0) (tracing.rs:74,84): scale 2, offset -3 */
static inline float orc_gen_range_m11(orc_rng* r) { return orc_value1_2(orc_next_u32(r)) * 2.0f + (-1.0f - 2.0f); }
/* rng.gen_range(0..n) for u32 (tracing.rs:167-168) */
static inline uint32_t orc_gen_range_u32(orc_rng* r, uint32_t n) {
    uint32_t range = n;                                  /* (high-1) - low + 1 */
    if (range == 0u) return orc_next_u32(r);
    uint32_t zone = (range << __builtin_clz(range)) - 1u;
    for (;;) {
        uint32_t v = orc_next_u32(r);
        uint64_t m = (uint64_t)v * (uint64_t)range;
        uint32_t hi = (uint32_t)(m >> 32), lo = (uint32_t)m;
        if (lo <= zone) return hi;
    }
}

#endif /* ORC_RNG_H */
