// pt_kernels.hip — hand-written HIP kernels for gfx950 (MI355X, CDNA4) of the
// path-tracing hot path of mbk6/CS397RayTracingSP22.  No CUDA shims, no dual paths.
//
//   K1  pt_megakernel   Camera::generate_rays (tracing.rs:159-209) + Scene::shade_ray
//                       (:300-324, made iterative) + the scene hit loop (:326-346) + every
//                       intersect_ray (geometry.rs) + every scatter (materials.rs) +
//                       Texture::sample (texture.rs:26-32) + the per-pixel mean (:232-241)
//   K3  fb_unpermute    tile-major gathered buffers -> row-major W*H*3 f32
//   K4  fb_tonemap_u8   tracing.rs:244-256 (saturate toward white, gamma, quantise)
//
// The DEFAULT path is the wavefront pipeline K1w: the paths live in HBM (SoA planes, 72 / 76 bytes per
// path and pass), one kernel per phase over compacted lists, each with its own register budget:
//   wf_main       camera rays (iteration 0) / shade the pending hit, scatter, object list + mesh root tests for the new ray;
//                 compile-time forms per scene content (mesh branch, maps, rare kinds, generic volume boundaries); survivors
//                 appended per shard in two classes (A: plain Triangle / Plane hit, nothing left to walk; B: the rest)
//   wf_prefix     shard counters -> block tables of the next pass (also the walkers' work list) + the header for the host
//   wf_trav       persistent walkers over the class-B blocks, the reference's tree as a skip-link image (in LDS, or in global memory), voted steps
//   wf_trav_i     the same walk over split pools with explicit links: interior records in LDS; the leaf records in LDS too for trees whose image
//                 fits 64 KB (the default small-tree walker since round 4: 8 x 256 threads per CU), from global memory for 64 .. 150 KB (2 x 1024)
//   wf_filter_f / wf_trav_f / wf_replay   exact two-stage traversal of large meshes: root-box filter -> padded SAH tree of
//                 16-byte quantised nodes + the reference's triangle test -> replay of the reference's walk over the candidates
//   wf_reduce     per-pixel sums in sample order (tracing.rs:232-241)
// K1 (below) are the single-launch variants the pipeline grew out of; they remain selectable as structural cross-checks.
//
// Execution model of K1 (DESIGN.md "K1"):
//   * one lane = one pixel; a wave = an 8x8 pixel block; a 256-thread workgroup = a
//     32x8 strip of one 32x32 image tile.
//   * lane-persistent regeneration: a lane loops over its pixel's samples; the loop trip
//     is ONE PATH SEGMENT, and a lane whose path ended starts its next sample in the same
//     trip, so a wave never idles on short paths (64-wide waves make path-length
//     divergence the first-order loss otherwise).
//   * the object list is walked at a wave-uniform index: every lane reads the same
//     64-byte record -> scalar loads, zero divergence in the list loop.
//   * mesh BVHs are traversed STACKLESS through skip links (the reference's fixed
//     left-then-right order, geometry.rs:105-115, makes this exact), nodes and triangles
//     staged into LDS when they fit (teapot: 479 nodes + 240 triangles = 26.8 KB).
//   * VOTED variant (default, pt_megakernel_voted): each lane is a state machine and every
//     trip of the wave-wide loop votes with __ballot/__popcll which phase runs — the BVH node
//     step, not the segment, is the scheduling unit.  SIMPLE (pt_megakernel) is kept as a
//     structural cross-check.
//
// Numerics: every expression is written in the reference's evaluation order; the file
// is compiled with -ffp-contract=off (Rust never fuses a*b+c) and HIP's default
// correctly rounded f32 divide/sqrt, so each path makes bit-identical decisions to the
// CPU restatement (tests compare per-pixel path signatures bit for bit).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "pt_device.h"

#pragma clang fp contract(off)

#ifndef PT_MAIN_WAVES
#define PT_MAIN_WAVES 6     // wf_main: waves per SIMD the register allocator must allow (A/B round 2: 4 -> 90 VGPRs 99.5 ms, 6 -> 80 VGPRs 97.4 ms on cfg2)
#endif
#ifndef PT_MAIN_WAVES_NOTEX
#define PT_MAIN_WAVES_NOTEX 6 // wf_main with meshes but without maps (MESH = 1)
#endif
#ifndef PT_MAIN_WAVES_LEAN
#define PT_MAIN_WAVES_LEAN 7  // wf_main without the mesh branch (camera-ray pass, class-A parts, scenes without meshes): 60 VGPRs, no scratch, no
                              // scalar spills (the full form: 80 / 12 B / 16).  6 / 7 / 8: cfg2 82.9 / 82.8 / 83.0 ms, cfg1 41.3 / 40.8 / 41.0
#endif
#ifndef PT_TRAV_BURST
#define PT_TRAV_BURST 10    // wf_trav / wf_trav_f: interior steps per vote.  Round 1 (triangles fetched from global memory in the leaf
                            // step): 2/4/6/8 -> 52.6/46.8/45.8/46.3 ms.  Round 2 (leaves in the LDS image): 6/8/10/12/16 -> 30.7/29.6/28.6/29.4/31.0 ms
                            // on cfg2; leaving a burst early when < 24/32 lanes are still on interior nodes: +3/+7 ms
#endif
#ifndef PT_TRAVF_BURST
#define PT_TRAVF_BURST 6    // wf_trav_f: F-node steps per vote
#endif
#ifndef PT_TRAV_LEAF_W
#define PT_TRAV_LEAF_W 3    // wf_trav: a leaf step is taken when n_leaf * W > n_inner (leaves in LDS: W = 1/2/3/4 -> 33.2/30.8/30.6/30.6 ms on cfg2)
#endif
#ifndef PT_TRAV_DYN
#define PT_TRAV_DYN 0       // wf_trav: leave a burst early when fewer than this many lanes are still on interior nodes (0 = never; see above)
#endif
#ifndef PT_TRAV_LEAF2
#define PT_TRAV_LEAF2 16    // wf_trav: a leaf step tests a second triangle when at least this many lanes sit on a leaf again (the reference's
                            // tree ends in pairs of sibling leaves): 29.4 -> 28.5 ms on cfg2, 71.7 -> 69.9 on cfg4 (0 = never)
#endif
#ifndef PT_TRAV_WAVES
#define PT_TRAV_WAVES 8     // wf_trav / wf_trav_f, 256-thread blocks: waves per SIMD the register allocator must allow.  8 = 64 VGPRs and 78 SGPRs
                            // (9 scalar spills) so that eight blocks per CU really fit — 6 let the compiler take 67 VGPRs / 92 SGPRs, i.e. seven.
                            // A/B at the end of round 2, cfg2 wf_trav: 6 / 7 / 8 -> 29.2 / 29.1 / 28.1 ms (HEAD scene: unchanged)
#endif
#ifndef PT_TRAVI_BURST
#define PT_TRAVI_BURST 10   // wf_trav_i (interior nodes in LDS, leaves from global memory): interior steps per vote
#endif
#ifndef PT_TRAVI_LEAF_W
#define PT_TRAVI_LEAF_W 3   // wf_trav_i: a leaf step is taken when n_leaf * W > n_inner
#endif
#ifndef PT_TRAVI_LEAF2
#define PT_TRAVI_LEAF2 1    // wf_trav_i: a leaf step tests a second triangle when at least this many lanes sit on a leaf again (end of round 3, cfg4 walker: 24 / 16 / 8 / 4 / 1 -> 192.8 / 189.6 / 186.1 / 187.4 / 186.6 ms)
#endif
#ifndef PT_TRAVL_BURST
#define PT_TRAVL_BURST 10   // wf_trav_i with the leaves in LDS too (small trees): interior steps per vote (round 4, cfg2 walker: 6 / 8 / 10 / 14 -> 25.4 / 24.6 / 24.0 / 25.1 ms)
#endif
#ifndef PT_TRAVL_LEAF2
#define PT_TRAVL_LEAF2 1    // ... and the second triangle of a leaf step, taken when at least this many lanes sit on a leaf again (0 = never / 1 / 16 -> 24.9 / 24.0 / 24.2 ms)
#endif
#ifndef PT_PRE_ATTR
#define PT_PRE_ATTR 1       // walkers leave {attribute-record index, mesh index} of a mesh hit in the spare words of its plane-5 record: resolve_hit then
                            // fetches the mesh record and the triangle's attributes side by side instead of entry -> mesh -> attributes; 0 for A/B
#endif
#ifndef PT_SPHERE_STAGED
#define PT_SPHERE_STAGED 1  // list Spheres in two stages (sphere_stage1 / sphere_finish); 0 = one full test per sphere, for A/B
#endif
#ifndef PT_RNG_PEEP
#define PT_RNG_PEEP 1       // value1_2 as one funnel shift, genm11 as one fma (both exact; the rejection sampler's loop 44 -> 38 VALU); 0 for A/B: cfg2 75.3 -> 74.9 ms
#endif
#ifndef PT_ROT_TABLE
#define PT_ROT_TABLE 1       // sample_hemisphere's rotation for list-Triangle hits read from the table the host computed (DScene.obj_rot); 0 for A/B
#endif
#define PT_LDS_AS __attribute__((address_space(3)))
#ifndef PT_TRAVI_MED3
#define PT_TRAVI_MED3 1      // wf_trav_i (32-byte records, one mesh): the clamped box test for waves whose rays all have finite non-zero 1/d; 0 for A/B
                            // (cfg4 walker 188.5 -> 177.7 ms; in the MULTI forms it costs more than it saves: HEAD 35.1 -> 36.0 ms)
#endif
#ifndef PT_SEG_COUNT
#define PT_SEG_COUNT 1      // wf_main counts its path segments (mi_last_pipeline_counts[6]); 0 only to measure what the count costs
#endif
#ifndef PT_TRAV_PEND
#define PT_TRAV_PEND 16    // walkers over several meshes: lanes that have finished one mesh wait until this many can take the next root tests together (HEAD walker: 4 / 8 / 16 / 24 -> 40.9 / 38.1 / 36.3 / 37.0 ms; on the spot: 43.4)
#endif
#ifndef PT_MIN_WAVES
#define PT_MIN_WAVES 4      // waves per SIMD the register allocator must leave room for (<= 128 VGPRs)
#endif

namespace pt {

// ---------------------------------------------------------------- small vector math
struct f3 { float x, y, z; };
__device__ __forceinline__ f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
template <class FP> __device__ __forceinline__ f3 ld3(FP p) { return mk3(p[0], p[1], p[2]); }   // FP: any float pointer (generic or constant AS)
__device__ __forceinline__ f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
__device__ __forceinline__ f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
// InnerSpace::dot = (x + y) + z
__device__ __forceinline__ float dot(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ f3 cross(f3 a, f3 b) {
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ float mag2(f3 a) { return dot(a, a); }
// Correctly rounded 1.0f / x in 5 VALU instructions instead of the 11 of hipcc's IEEE division sequence (v_div_scale x2,
// v_rcp, 5 x fma, v_div_fmas, v_div_fixup).  v_rcp_f32 (1 ulp) followed by ONE Newton step in fused arithmetic is the
// correctly rounded reciprocal for EVERY f32 whose biased exponent lies in [2, 252] — checked exhaustively on gfx950, all
// 2^32 bit patterns, 0 mismatches against `1.0f / x` (tools/exhaustive/rcp_check.hip; the library re-runs that sweep as
// mi_selftest, tests/test_gpu_selftest.py).  Outside that range (zeros, denormals, |x| < 2^-125 or >= 2^126, inf, NaN) the
// IEEE sequence runs under an exec mask that is almost never taken.  The explicit fmaf calls are not contractions: the
// result is bit-identical to the division the reference (and the oracle) performs, for every input.
__device__ __forceinline__ float rcp_exact(float x) {
#if defined(PT_RCP_IEEE)
    return 1.0f / x;
#else
    const float ax = fabsf(x);
    float r = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    const bool out_of_range = !((ax >= 2.3509887e-38f) & (ax < 8.5070592e37f));     // NOT 2^-125 <= |x| < 2^126  <=>  biased exponent outside [2, 252]
#if defined(PT_RCP_LANE_BRANCH)
    if (out_of_range) r = 1.0f / x;
#else
    // wave-uniform fallback: one scalar branch that is almost never taken (no exec-mask bookkeeping on the hot path)
    if (__builtin_amdgcn_ballot_w64(out_of_range) != 0ull) r = out_of_range ? 1.0f / x : r;
#endif
    return r;
#endif
}
// three reciprocals (inv_d of a slab walk, geometry.rs:57) behind ONE range check and one never-taken scalar branch
__device__ __forceinline__ void rcp3_exact(float x, float y, float z, float& rx, float& ry, float& rz) {
#if defined(PT_RCP_IEEE)
    rx = 1.0f / x; ry = 1.0f / y; rz = 1.0f / z;
#else
    float a = __builtin_amdgcn_rcpf(x), b = __builtin_amdgcn_rcpf(y), c = __builtin_amdgcn_rcpf(z);
    a = __builtin_fmaf(__builtin_fmaf(-x, a, 1.0f), a, a);
    b = __builtin_fmaf(__builtin_fmaf(-y, b, 1.0f), b, b);
    c = __builtin_fmaf(__builtin_fmaf(-z, c, 1.0f), c, c);
    // all three inside [2^-125, 2^126)  <=>  min |.| >= 2^-125 and max |.| < 2^126 (a NaN operand makes one of the two compares false)
    const float lo = fminf(fminf(fabsf(x), fabsf(y)), fabsf(z)), hi = fmaxf(fmaxf(fabsf(x), fabsf(y)), fabsf(z));
    const bool nan = (x != x) | (y != y) | (z != z);
    const bool out_of_range = nan | !((lo >= 2.3509887e-38f) & (hi < 8.5070592e37f));
    if (__builtin_amdgcn_ballot_w64(out_of_range) != 0ull) {      // axis-parallel rays (a zero component) come here: IEEE for the whole wave
        a = 1.0f / x; b = 1.0f / y; c = 1.0f / z;
    }
    rx = a; ry = b; rz = c;
#endif
}
// InnerSpace::normalize = self * (1 / magnitude)
__device__ __forceinline__ f3 normalize(f3 a) { return a * rcp_exact(sqrtf(mag2(a))); }
// Matrix3 * Vector3, column-major m[9]
__device__ __forceinline__ f3 m3mul(const float* m, f3 v) {
    return mk3((m[0] * v.x + m[3] * v.y) + m[6] * v.z,
               (m[1] * v.x + m[4] * v.y) + m[7] * v.z,
               (m[2] * v.x + m[5] * v.y) + m[8] * v.z);
}
__device__ __forceinline__ float clampf(float v, float lo, float hi) {
    if (v < lo) v = lo;
    if (v > hi) v = hi;
    return v;
}

// ---------------------------------------------------------------- RNG (DESIGN.md "RNG")
struct Rng { uint32_t s0, s1; };
__device__ __forceinline__ uint32_t lowbias32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t rotl32(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }
__device__ __forceinline__ void rng_init(Rng& r, uint32_t seed_key, uint32_t pixel, uint32_t sample) {
    // seed_key = lowbias32(seed ^ 0x68e31da4), hoisted to the host
    uint32_t p0 = lowbias32(pixel + seed_key);
    uint32_t p1 = lowbias32(p0 ^ 0xb5297a4du);
    uint32_t s0 = lowbias32(p0 + sample * 0x9e3779b9u);
    uint32_t s1 = lowbias32(p1 ^ (sample * 0x85ebca6bu));
    if ((s0 | s1) == 0u) s1 = 1u;
    r.s0 = s0; r.s1 = s1;
}
// xoroshiro64**
__device__ __forceinline__ uint32_t next_u32(Rng& r) {
    uint32_t s0 = r.s0, s1 = r.s1;
    uint32_t result = rotl32(s0 * 0x9e3779bbu, 5) * 5u;
    s1 ^= s0;
    r.s0 = rotl32(s0, 26) ^ s1 ^ (s1 << 9);
    r.s1 = rotl32(s1, 13);
    return result;
}
#if PT_RNG_PEEP
// 0x3f800000 | (bits >> 9) as ONE funnel shift: the low nine bits of the high word 0x7f become the sign and exponent
__device__ __forceinline__ float value1_2(uint32_t bits) { return __uint_as_float(__builtin_amdgcn_alignbit(0x7fu, bits, 9u)); }
#else
__device__ __forceinline__ float value1_2(uint32_t bits) { return __uint_as_float(0x3f800000u | (bits >> 9)); }
#endif
// rand 0.8.4 gen_range(0.0..1.0): value1_2*1 + (0-1)      (exact)
__device__ __forceinline__ float gen01(Rng& r) { return value1_2(next_u32(r)) * 1.0f + (0.0f - 1.0f); }
// rand 0.8.4 gen_range(-1.0..1.0): value1_2*2 + (-1-2)    (exact)
#if PT_RNG_PEEP
// v in [1, 2): v * 2 is exact, so the reference's two roundings are one, and fma(v, 2, -3) is that one
__device__ __forceinline__ float genm11(Rng& r) { return __builtin_fmaf(value1_2(next_u32(r)), 2.0f, -1.0f - 2.0f); }
#else
__device__ __forceinline__ float genm11(Rng& r) { return value1_2(next_u32(r)) * 2.0f + (-1.0f - 2.0f); }
#endif
// rand 0.8.4 gen_range(0..n): widening multiply + rejection zone
__device__ __forceinline__ uint32_t gen_u32_below(Rng& r, uint32_t range, uint32_t zone) {
    for (;;) {
        uint32_t v = next_u32(r);
        uint32_t hi = __umulhi(v, range), lo = v * range;
        if (lo <= zone) return hi;
    }
}
// tracing.rs:71-79
__device__ __forceinline__ f3 rand_sphere_vec(Rng& r) {
    for (;;) {
        f3 d;
        d.x = genm11(r); d.y = genm11(r); d.z = genm11(r);
        if (mag2(d) <= 1.0f) return d;
    }
}
// tracing.rs:81-89
__device__ __forceinline__ f3 rand_disk_vec(Rng& r) {
    for (;;) {
        f3 d;
        d.x = genm11(r); d.y = genm11(r); d.z = 0.0f;
        if (mag2(d) <= 1.0f) return d;
    }
}

// f32::ln for the free-flight distance (geometry.rs:517).  One fixed sequence of f32
// operations (Cephes-style, <= 2 ulp), the same sequence the CPU checker uses.
__device__ __forceinline__ float pt_logf(float x) {
    if (!(x > 0.0f)) return (x == 0.0f) ? -__builtin_inff() : __builtin_nanf("");
    if (x == __builtin_inff()) return x;
    uint32_t ix = __float_as_uint(x);
    int e = 0;
    if (ix < 0x00800000u) { x = x * 8388608.0f; ix = __float_as_uint(x); e = -23; }
    e += (int)(ix >> 23) - 126;
    ix = (ix & 0x007fffffu) | 0x3f000000u;
    float m = __uint_as_float(ix);
    if (m < 0.70710678f) { e = e - 1; m = m + m; }
    float f = m - 1.0f;
    float z = f * f;
    float p = 7.0376836292e-2f;
    p = p * f + -1.1514610310e-1f;
    p = p * f + 1.1676998740e-1f;
    p = p * f + -1.2420140846e-1f;
    p = p * f + 1.4249322787e-1f;
    p = p * f + -1.6668057665e-1f;
    p = p * f + 2.0000714765e-1f;
    p = p * f + -2.4999993993e-1f;
    p = p * f + 3.3333331174e-1f;
    float y = (f * z) * p;
    float fe = (float)e;
    y = y + fe * -2.12194440e-4f;
    y = y - 0.5f * z;
    float r = f + y;
    r = r + fe * 0.693359375f;
    return r;
}

// ---------------------------------------------------------------- helpers tracing.rs:54-69
__device__ __forceinline__ f3 reflect(f3 v, f3 n) { return v - n * (2.0f * dot(v, n)); }
__device__ __forceinline__ float powi2(float a) { return a * a; }
__device__ __forceinline__ float powi5(float a) { float a2 = a * a; float a4 = a2 * a2; return a * a4; }
__device__ __forceinline__ float fresnel(f3 v, f3 n, float ir) {
    float r0 = powi2((ir - 1.0f) / (ir + 1.0f));
    return r0 + (1.0f - r0) * powi5(1.0f - fabsf(dot(v, n)));
}
__device__ __forceinline__ f3 refract(f3 v, f3 n, float eta) {
    float cos_theta = fminf(dot(-v, n), 1.0f);
    f3 r_out_perp = (v + n * cos_theta) * eta;
    f3 r_out_parallel = n * (-sqrtf(fabsf(1.0f - mag2(r_out_perp))));
    return r_out_perp + r_out_parallel;
}

// approx::ulps_eq!(a, b), f32 defaults (epsilon = f32::EPSILON, max_ulps = 4)
__device__ __forceinline__ bool ulps_eq(float a, float b) {
    if (fabsf(a - b) <= 1.1920929e-07f) return true;
    if ((a < 0.0f) != (b < 0.0f)) return false;
    int32_t ia = (int32_t)__float_as_uint(a), ib = (int32_t)__float_as_uint(b);
    int32_t d = ia - ib;
    if (d < 0) d = -d;
    return d <= 4;
}

// sample_hemisphere's rotation (materials.rs:176-177):
//   Basis3::between_vectors(unit_y, n).rotate_vector(dir)
// cgmath's half-way quaternion with a = (0,1,0): a.b = n.y, a x b = (n.z, 0, -n.x),
// |a|^2 = 1.  The zero y component of the quaternion is folded (x*0, x+0: exact).
__device__ __forceinline__ f3 rotate_from_unit_y(f3 n, f3 dir) {
    float k_cos_theta = n.y;
    if (ulps_eq(k_cos_theta, 1.0f)) return dir;                       // identity quaternion
    float k = sqrtf(1.0f * mag2(n));
    float qs, qx, qz;
    if (ulps_eq(k_cos_theta / k, -1.0f)) {
        // pi rotation about normalize(unit_y x unit_x) = (0,0,-1): q = (0; 0,0,-1)
        qs = 0.0f; qx = 0.0f; qz = -1.0f;
    } else {
        float s = k + k_cos_theta;
        float cx = n.z, cz = -n.x;
        float mag = sqrtf(s * s + ((cx * cx + 0.0f) + cz * cz));
        float inv = rcp_exact(mag);
        qs = s * inv; qx = cx * inv; qz = cz * inv;
    }
    float x2 = qx + qx, z2 = qz + qz;
    float xx2 = x2 * qx, xz2 = x2 * qz, zz2 = z2 * qz;
    float sz2 = z2 * qs, sx2 = x2 * qs;
    // columns of Matrix3::from(quaternion) with q.y = 0
    float c0x = 1.0f - zz2, c0y = sz2,                  c0z = xz2;
    float c1x = -sz2,       c1y = (1.0f - xx2) - zz2,   c1z = sx2;
    float c2x = xz2,        c2y = -sx2,                 c2z = 1.0f - xx2;
    return mk3((c0x * dir.x + c1x * dir.y) + c2x * dir.z,
               (c0y * dir.x + c1y * dir.y) + c2y * dir.z,
               (c0z * dir.x + c1z * dir.y) + c2z * dir.z);
}

// ---------------------------------------------------------------- primitive tests
// The primitive tests are written WITHOUT early returns: in a 64-wide wave an early return
// only saves work when all lanes take it (they almost never do), but every `if (..) return`
// costs a v_cmp + s_and_saveexec + s_cbranch + s_or pair on the scalar unit, and a wave issues
// one instruction at a time.  Each reject condition is the reference's, combined with `|`
// (NaN behaviour preserved: a comparison with NaN is false, as in the Rust source).

// Sphere::intersect_ray geometry.rs:395-413 -> parametric t, or miss
__device__ __forceinline__ bool sphere_t(f3 o, f3 d, f3 center, float r2, float t_min, float t_max, float& t_out) {
    f3 f = o - center;
    float a = mag2(d);
    float b = 2.0f * dot(f, d);
    float c = mag2(f) - r2;
    float disc = b * b - 4.0f * a * c;
    float sq = sqrtf(disc);                       // NaN when disc < 0: that lane is rejected below
    float t1 = (-b - sq) / (2.0f * a);
    float t2 = (-b + sq) / (2.0f * a);
    float t = (t1 >= t_min) ? t1 : t2;
    t_out = t;
    return !((disc < 0.0f) | (t < t_min) | (t > t_max));
}

// Triangle / IndexedTriangle Moller-Trumbore geometry.rs:331-349, 431-447
// FAST = false: the IEEE division for 1/g (the 1024-thread walker is LDS-bound, the short reciprocal's range check only costs there)
template <bool FAST = true>
__device__ __forceinline__ bool tri_t(f3 o, f3 d, f3 a, f3 e1, f3 e2, float t_min, float t_max,
                                      float& t_out, float& u_out, float& v_out) {
    f3 q = cross(d, e2);
    float g = dot(e1, q);
    float f = FAST ? rcp_exact(g) : 1.0f / g;
    f3 s = o - a;
    float u = f * dot(s, q);
    f3 r = cross(s, e1);
    float v = f * dot(d, r);
    float t = f * dot(e2, r);
    t_out = t; u_out = u; v_out = v;
    return !((fabsf(g) < 0.0001f) | (u < 0.0f) | (v < 0.0f) | (u + v > 1.0f) | (t < t_min) | (t > t_max));
}

// AABB::intersect_ray geometry.rs:52-79 with 1/d hoisted out of the node loop (the
// reference recomputes the same 1/d at every box).
// The reference tests `tmax <= tmin` after every axis.  tmin never decreases and tmax never increases
// (f32::max/min drop a NaN operand and the start values are not NaN), so a rejection on an earlier axis
// is still a rejection after the last one: ONE test at the end gives the same decision.
__device__ __forceinline__ bool slab(f3 bmin, f3 bmax, f3 o, f3 inv_d, float t_min, float t_max) {
    float tmin = t_min, tmax = t_max;
    {
        float t0 = (bmin.x - o.x) * inv_d.x, t1 = (bmax.x - o.x) * inv_d.x;
        bool sw = inv_d.x < 0.0f;
        float ta = sw ? t1 : t0, tb = sw ? t0 : t1;
        tmin = fmaxf(ta, tmin); tmax = fminf(tb, tmax);
    }
    {
        float t0 = (bmin.y - o.y) * inv_d.y, t1 = (bmax.y - o.y) * inv_d.y;
        bool sw = inv_d.y < 0.0f;
        float ta = sw ? t1 : t0, tb = sw ? t0 : t1;
        tmin = fmaxf(ta, tmin); tmax = fminf(tb, tmax);
    }
    {
        float t0 = (bmin.z - o.z) * inv_d.z, t1 = (bmax.z - o.z) * inv_d.z;
        bool sw = inv_d.z < 0.0f;
        float ta = sw ? t1 : t0, tb = sw ? t0 : t1;
        tmin = fmaxf(ta, tmin); tmax = fminf(tb, tmax);
    }
    return !(tmax <= tmin);
}

// The same test on a box delivered as {near plane, far plane} per axis — near = (1/d < 0 ? bmax : bmin), which is what the swap of
// geometry.rs:60-62 makes of t0 / t1: (near - o) * inv_d and (far - o) * inv_d are the reference's two products, already in order.
__device__ __forceinline__ bool slab_nf(float2 x, float2 y, float2 z, f3 o, f3 inv_d, float t_min, float t_max) {
    float tmin = t_min, tmax = t_max;
    { const float ta = (x.x - o.x) * inv_d.x, tb = (x.y - o.x) * inv_d.x; tmin = fmaxf(ta, tmin); tmax = fminf(tb, tmax); }
    { const float ta = (y.x - o.y) * inv_d.y, tb = (y.y - o.y) * inv_d.y; tmin = fmaxf(ta, tmin); tmax = fminf(tb, tmax); }
    { const float ta = (z.x - o.z) * inv_d.z, tb = (z.y - o.z) * inv_d.z; tmin = fmaxf(ta, tmin); tmax = fminf(tb, tmax); }
    return !(tmax <= tmin);
}

// The same decision in 4 instructions fewer, for rays whose three 1/d are finite and non-zero (`slab_plain_ray`).  Per axis the reference
// raises tmin to the near plane and lowers tmax to the far plane; v_med3_f32 CLAMPS each of them into [near, far] instead (med3(t0, t1, x) =
// clamp(x, min(t0, t1), max(t0, t1)); no sign select: for such a ray min / max of t0, t1 ARE its near / far values).  Write (a, b) for the
// clamped pair and (a_r, b_r) for the reference's.  While a <= far and b >= near the two pairs stay EQUAL (clamp(a) = max(a, near),
// clamp(b) = min(b, far)).  The first axis where that fails — a > far or b < near — is an axis where the reference pair enters the miss state
// b_r <= a_r and the clamped pair becomes equal (both far, or both near): a miss as well.  A pair in the miss state stays there: the
// reference's a_r only rises and b_r only falls, and a clamp is monotone, so b <= a implies clamp(b) <= clamp(a).  Hence `tmax <= tmin` after
// the third axis is the same boolean, flat boxes (near = far) included.  A NaN product (0 * inf, only when some 1/d is infinite) would break
// it, which is why a wave holding a ray with a non-finite or zero 1/d takes the reference form (tests/test_gpu_walkers.py axis-aligned rays).
__device__ __forceinline__ bool slab_med3(f3 bmin, f3 bmax, f3 o, f3 inv_d, float t_min, float t_max) {
    float tmin = t_min, tmax = t_max;
    { const float t0 = (bmin.x - o.x) * inv_d.x, t1 = (bmax.x - o.x) * inv_d.x; tmin = __builtin_amdgcn_fmed3f(t0, t1, tmin); tmax = __builtin_amdgcn_fmed3f(t0, t1, tmax); }
    { const float t0 = (bmin.y - o.y) * inv_d.y, t1 = (bmax.y - o.y) * inv_d.y; tmin = __builtin_amdgcn_fmed3f(t0, t1, tmin); tmax = __builtin_amdgcn_fmed3f(t0, t1, tmax); }
    { const float t0 = (bmin.z - o.z) * inv_d.z, t1 = (bmax.z - o.z) * inv_d.z; tmin = __builtin_amdgcn_fmed3f(t0, t1, tmin); tmax = __builtin_amdgcn_fmed3f(t0, t1, tmax); }
    return !(tmax <= tmin);
}
__device__ __forceinline__ bool slab_plain_ray(f3 inv_d) {
    const float ax = fabsf(inv_d.x), ay = fabsf(inv_d.y), az = fabsf(inv_d.z);
    const float big = __uint_as_float(0x7f800000u);
    return (ax > 0.0f) & (ax < big) & (ay > 0.0f) & (ay < big) & (az > 0.0f) & (az < big);
}

// (Negative result, round 1: storing nodes axis-paired so that the subtract/multiply become v_pk_add_f32 /
// v_pk_mul_f32 made wf_trav SLOWER, 43.5 -> 53.3 ms on cfg2 — 74 VGPRs instead of 64 for the aligned pairs,
// and no issue-rate gain from the packed form on this path.)

// Matrix4 (column-major) transforms, cgmath Transform3 (geometry.rs:304,307,297)
template <class FP> __device__ __forceinline__ f3 xform_point(FP m, f3 p) {
    float x = ((m[0] * p.x + m[4] * p.y) + m[8]  * p.z) + m[12] * 1.0f;
    float y = ((m[1] * p.x + m[5] * p.y) + m[9]  * p.z) + m[13] * 1.0f;
    float z = ((m[2] * p.x + m[6] * p.y) + m[10] * p.z) + m[14] * 1.0f;
    float w = ((m[3] * p.x + m[7] * p.y) + m[11] * p.z) + m[15] * 1.0f;
    float iw = 1.0f / w;
    return mk3(x * iw, y * iw, z * iw);
}
template <class FP> __device__ __forceinline__ f3 xform_vector(FP m, f3 v) {
    return mk3(((m[0] * v.x + m[4] * v.y) + m[8]  * v.z) + m[12] * 0.0f,
               ((m[1] * v.x + m[5] * v.y) + m[9]  * v.z) + m[13] * 0.0f,
               ((m[2] * v.x + m[6] * v.y) + m[10] * v.z) + m[14] * 0.0f);
}
template <class FP> __device__ __forceinline__ f3 xform_vector_transposed(FP m, f3 v) {
    return mk3(((m[0] * v.x + m[1] * v.y) + m[2]  * v.z) + m[3]  * 0.0f,
               ((m[4] * v.x + m[5] * v.y) + m[6]  * v.z) + m[7]  * 0.0f,
               ((m[8] * v.x + m[9] * v.y) + m[10] * v.z) + m[11] * 0.0f);
}

// Texture::sample texture.rs:26-32
__device__ __forceinline__ f3 tex_sample(const DScene& S, int tex, float u, float v) {
    const uint32_t t_offset = S.textures[tex].offset;
    uint32_t W = (uint32_t)S.textures[tex].width, H = (uint32_t)S.textures[tex].height;
    uint32_t x = (uint32_t)(clampf(u, 0.0f, 0.999f) * (float)W);
    if (x > W - 1u) x = W - 1u;
    uint32_t y = (uint32_t)((1.0f - clampf(v, 0.0f, 0.999f)) * (float)H);
    if (y > H - 1u) y = H - 1u;
    // texels are RGBA8 on the device (padded by the scene compiler): ONE aligned 4-byte load per fetch
    const uint32_t px = ((const PT_CONST_AS uint32_t*)(S.texels + t_offset))[(size_t)y * W + x];
    return mk3((float)(px & 0xffu) / 255.0f, (float)((px >> 8) & 0xffu) / 255.0f, (float)((px >> 16) & 0xffu) / 255.0f);
}

// ---------------------------------------------------------------- BVH storage access
typedef const PT_CONST_AS float4* cf4_ptr;
template <bool LDS> struct BvhPtr { typedef cf4_ptr type; };
template <> struct BvhPtr<true> { typedef const float4* type; };
template <bool LDS>
struct Bvh {
    typename BvhPtr<LDS>::type nodes;      // LDS, or constant-AS global
    typename BvhPtr<LDS>::type tris;
    __device__ __forceinline__ void node(int i, float4& n0, float4& n1) const { n0 = nodes[2 * i]; n1 = nodes[2 * i + 1]; }
    __device__ __forceinline__ void tri(int i, f3& a, f3& e1, f3& e2) const {
        float4 t0 = tris[3 * i], t1 = tris[3 * i + 1], t2 = tris[3 * i + 2];
        a = mk3(t0.x, t0.y, t0.z); e1 = mk3(t1.x, t1.y, t1.z); e2 = mk3(t2.x, t2.y, t2.z);
    }
};

// nodes in LDS, triangles through L1/L2 (wf_trav: halves the LDS footprint -> more resident waves)
struct BvhNodesLds {
    const float4* nodes;
    cf4_ptr tris;
    __device__ __forceinline__ void node(int i, float4& n0, float4& n1) const { n0 = nodes[2 * i]; n1 = nodes[2 * i + 1]; }
    __device__ __forceinline__ void tri(int i, f3& a, f3& e1, f3& e2) const {
        float4 t0 = tris[3 * i], t1 = tris[3 * i + 1], t2 = tris[3 * i + 2];
        a = mk3(t0.x, t0.y, t0.z); e1 = mk3(t1.x, t1.y, t1.z); e2 = mk3(t2.x, t2.y, t2.z);
    }
};

extern __shared__ float4 k1_lds[];
__device__ __forceinline__ void bvh_bind(Bvh<true>& B, const DScene&, int nn) { B.nodes = k1_lds; B.tris = k1_lds + nn; }
__device__ __forceinline__ void bvh_bind(Bvh<false>& B, const DScene& S, int) { B.nodes = (cf4_ptr)S.nodes; B.tris = (cf4_ptr)S.tris; }

// BVHNode::intersect_ray (geometry.rs:94-119) as a stackless threaded walk.
// The recursion passes t_max down unchanged to the left child and the left subtree's
// hit distance to the right child; unrolled over the whole tree that is ONE running
// bound `best_t`, lowered by every accepted leaf hit in DFS order, with `t <= best_t`
// accepted (geometry.rs:349 rejects only t > t_max), so a later equal hit replaces an
// earlier one exactly as `best_hit = hit_opt` (:114) does.
// "while-while" shape: advance every lane through interior nodes until it reaches a
// leaf (or the end), then run the triangle test for the lanes that hold a leaf.
template <class BVH>
__device__ __forceinline__ void traverse_mesh(const BVH& B, int node_begin, int node_end, int tri_begin,
                                              f3 o, f3 d, float t_min, float t_max,
                                              float& best_t, int& best_tri, float& best_u, float& best_v) {
    f3 inv_d; rcp3_exact(d.x, d.y, d.z, inv_d.x, inv_d.y, inv_d.z);      // geometry.rs:57
    best_t = t_max; best_tri = -1; best_u = 0.0f; best_v = 0.0f;
    int i = node_begin;
    while (i < node_end) {
        int leaf_tri = -1;
        while (i < node_end) {
            float4 n0, n1;
            B.node(i, n0, n1);
            int tri = __float_as_int(n1.w);
            if (tri >= 0) { leaf_tri = tri; break; }                       // geometry.rs:95
            bool hit = slab(mk3(n0.x, n0.y, n0.z), mk3(n1.x, n1.y, n1.z), o, inv_d, t_min, best_t);   // :103
            i = hit ? i + 1 : __float_as_int(n0.w);
        }
        if (leaf_tri >= 0) {
            f3 a, e1, e2;
            B.tri(tri_begin + leaf_tri, a, e1, e2);
            float t, u, v;
            if (tri_t(o, d, a, e1, e2, t_min, best_t, t, u, v)) {          // :97, :106/:113
                best_t = t; best_tri = leaf_tri; best_u = u; best_v = v;
            }
            i = i + 1;
        }
    }
}

// ---------------------------------------------------------------- path state
struct Path {
    f3 o, d;            // current ray
    f3 T, L;            // throughput, radiance of this sample
    Rng rng;
    uint32_t depth;
    uint32_t sig;
};

// closest hit so far over Scene.objects
struct Best {
    float t;            // distance of the best hit; valid when obj >= 0
    int   obj;          // index into Scene.objects, -1 = none
    int   tri;          // mesh triangle (mesh hits)
    float u, v;         // barycentrics (mesh hits)
};

// Scene::intersect_ray's replacement rule (tracing.rs:333-336): strictly closer wins;
// on equal distance the object that comes first in Scene.objects is kept.
__device__ __forceinline__ void consider(Best& b, float t, int obj, int tri, float u, float v) {
    bool take = (b.obj < 0) || (t < b.t) || (t == b.t && obj < b.obj);
    if (take) { b.t = t; b.obj = obj; b.tri = tri; b.u = u; b.v = v; }
}

// surface point handed to the scatter stage
struct Surf {
    f3 p, n;            // hitpoint (world), normal (ray-facing; zero inside a volume)
    bool frontface;
    int   kind;
    f3 albedo, emission, brdf_diffuse;   // brdf_diffuse = albedo / PI
    float roughness, metallic, ior;
    int rot;            // >= 0: entry of DScene.obj_rot that holds sample_hemisphere's rotation for this normal (list Triangles); -1: compute it
};

__device__ __forceinline__ void load_material(const DScene& S, int id, Surf& s) {
    auto m = &S.materials[id];
    s.kind = m->kind;
    s.albedo = ld3(m->albedo); s.emission = ld3(m->emission);
    s.brdf_diffuse = ld3(m->albedo_over_pi);
    s.roughness = m->roughness; s.metallic = m->metallic; s.ior = m->ior;
}

// RayHit::new (tracing.rs:121-133): face the normal against the ray
__device__ __forceinline__ void face(f3 normal, f3 dir, f3& n_out, bool& frontface) {
    frontface = dot(normal, dir) < 0.0f;
    n_out = frontface ? normal : -normal;
}

// Build the hit record of the winning object (the reference builds one per candidate
// and keeps the closest; only the winner's is observable).
// MESH = 0 compiles the mesh branch out: for launches whose pending hits cannot be mesh hits (wf_main's camera-ray pass and
// its class-A parts: a mesh hit only ever comes back from a walker, into class B).  MESH = 1: every mesh of the scene has a fixed
// material and no normal map (cfg2's teapot), so texel fetches, the material-from-maps branch and the TBN are compiled out.  2: all.
// pre_mesh / pre_attr (wavefront pipeline, PT_PRE_ATTR): the walker that found a mesh hit knows its mesh and has the mesh's first
// triangle in registers, so it leaves {mesh index, index of the triangle's DTriAttr} in the spare words of the hit's plane-5 record.
// With them the mesh record and the attribute record are fetched SIDE BY SIDE as soon as the path state has arrived, instead of
// entry -> mesh -> attributes one gather after the other (-1: not provided, the chain is followed as before).
template <int MESH = 2>
__device__ __forceinline__ void resolve_hit(const DScene& S, const Best& b, f3 o, f3 d, Surf& s, int pre_mesh = -1, int pre_attr = -1) {
    // (Round 4, measured negative: ONE 96-byte record per entry with the material inline, so that this gather and the material's are not
    // two dependent ones.  The phase stamps had shown resolve_hit as a third of a class-B wave's life; but a hop fewer did not help and
    // the wider record cost: cfg4 wf_main 76.8 -> 79.9 ms, cfg2 +1 ms.  A gather's cost here is the number of distinct cache LINES the
    // 64 lanes touch per load instruction, not the number of dependent hops: the material table is a handful of lines whichever entry a
    // lane hit, the inline copies are one line per entry.  tools/experiments/r04_shade_record.diff.)
    auto ob = &S.objects[b.obj];
    int kind = ob->kind;
    s.rot = -1;
    if (MESH && (PT_PRE_ATTR ? b.tri >= 0 : kind == OBJ_MESH)) {         // Best.tri >= 0 <=> the pending hit is a mesh triangle (list hits carry tags < 0)
        int mesh_idx = pre_mesh, attr_idx = pre_attr;
        if (!PT_PRE_ATTR || mesh_idx < 0) { mesh_idx = ob->ref; attr_idx = S.meshes[mesh_idx].tri_begin + b.tri; }
        auto M = &S.meshes[mesh_idx];
        // geometry.rs:304 — object-space ray (direction NOT renormalised)
        f3 oo = xform_point(M->inv_transform, o);
        f3 od = xform_vector(M->inv_transform, d);
        auto A = &S.triattr[attr_idx];
        float u = b.u, v = b.v, w = (1.0f - u - v);
        // geometry.rs:351 normalize(u*nb + v*nc + (1-u-v)*na)
        f3 mesh_normal = normalize((ld3(A->nb) * u + ld3(A->nc) * v) + ld3(A->na) * w);
        f3 n; bool ff;
        face(mesh_normal, od, n, ff);                                    // :352 RayHit::new in object space
        f3 hp_obj = oo + od * b.t;                                       // tracing.rs:125
        float tu = (u * A->tb[0] + v * A->tc[0]) + w * A->ta[0];         // :356
        float tv = (u * A->tb[1] + v * A->tc[1]) + w * A->ta[1];
        // all maps of this mesh in ONE 16-byte texel (DMesh.tex_comb): fetched once, unpacked where texture.rs:26-32 is applied
        uint4 comb = make_uint4(0u, 0u, 0u, 0u);
        const bool have_comb = MESH == 2 && M->tex_comb >= 0;
        if (have_comb) {
            const uint32_t t_offset = S.textures[M->tex_comb].offset;
            const uint32_t W = (uint32_t)S.textures[M->tex_comb].width, H = (uint32_t)S.textures[M->tex_comb].height;
            uint32_t x = (uint32_t)(clampf(tu, 0.0f, 0.999f) * (float)W);
            if (x > W - 1u) x = W - 1u;
            uint32_t y = (uint32_t)((1.0f - clampf(tv, 0.0f, 0.999f)) * (float)H);
            if (y > H - 1u) y = H - 1u;
            comb = ((const PT_CONST_AS uint4*)(S.texels + t_offset))[(size_t)y * W + x];
        }
        auto unpack3 = [](uint32_t px) { return mk3((float)(px & 0xffu) / 255.0f, (float)((px >> 8) & 0xffu) / 255.0f, (float)((px >> 16) & 0xffu) / 255.0f); };
        if (MESH == 2 && M->tex[4] >= 0) {                               // geometry.rs:276-283
            f3 tan_approx = ld3(A->tan);
            f3 bitangent = normalize(cross(n, tan_approx));              // :360
            f3 tangent = normalize(cross(bitangent, n));                 // :361
            f3 smp = have_comb ? unpack3(comb.z) : tex_sample(S, M->tex[4], tu, tv);
            f3 nv = smp * 2.0f - mk3(1.0f, 1.0f, 1.0f);                  // :282
            // Matrix3::from_cols(tangent, bitangent, normal) * nv            :283
            n = (tangent * nv.x + bitangent * nv.y) + n * nv.z;
        }
        s.p = xform_point(M->transform, hp_obj);                         // :307
        s.n = normalize(xform_vector_transposed(M->inv_transform, n));   // :297
        s.frontface = ff;
        if (MESH != 2 || M->material >= 0) {
            load_material(S, M->material, s);                            // :255-256
        } else {                                                         // :259-269
            s.kind = MAT_PARAMETERIZED;
            if (have_comb) {        // absent maps were filled with their defaults' bytes (0, 0, 0, 255): same values
                s.albedo = unpack3(comb.x); s.emission = unpack3(comb.y);
                s.metallic = (float)(comb.x >> 24) / 255.0f; s.roughness = (float)(comb.y >> 24) / 255.0f;
            } else {
                s.albedo   = (M->tex[0] >= 0) ? tex_sample(S, M->tex[0], tu, tv) : mk3(0.0f, 0.0f, 0.0f);
                s.emission = (M->tex[1] >= 0) ? tex_sample(S, M->tex[1], tu, tv) : mk3(0.0f, 0.0f, 0.0f);
                s.metallic  = (M->tex[2] >= 0) ? tex_sample(S, M->tex[2], tu, tv).x : 0.0f;
                s.roughness = (M->tex[3] >= 0) ? tex_sample(S, M->tex[3], tu, tv).x : 1.0f;
            }
            const float PI = 3.14159265358979323846f;
            s.brdf_diffuse = mk3(s.albedo.x / PI, s.albedo.y / PI, s.albedo.z / PI);   // materials.rs:128
            s.ior = 0.0f;
        }
        return;
    }
    load_material(S, ob->material, s);
    f3 hp = o + d * b.t;                                                 // tracing.rs:125
    s.p = hp;
    if (kind == OBJ_SPHERE) {
        f3 c = ld3(ob->f);
        face(normalize(hp - c), d, s.n, s.frontface);                    // geometry.rs:411
    } else if (kind == OBJ_TRIANGLE) {
        face(ld3(ob->f + 9), d, s.n, s.frontface);                       // geometry.rs:449
#if PT_ROT_TABLE
        s.rot = 2 * b.obj + (s.frontface ? 0 : 1);
#endif
    } else if (kind == OBJ_PLANE) {
        // geometry.rs:476-478,487: n = signum(origin_dist) * normal
        f3 normal = ld3(ob->f + 3);
        float origin_dist = dot(o - ld3(ob->f), normal);
        float sg = (origin_dist != origin_dist) ? origin_dist : (__float_as_uint(origin_dist) >> 31 ? -1.0f : 1.0f);
        face(normal * sg, d, s.n, s.frontface);
#if PT_ROT_TABLE
        // s.n is the stored normal or its negation (a product with +-1 and face()'s negation are exact): entry 0 / 1 of this object
        const bool neg = (sg < 0.0f) != (!s.frontface);
        s.rot = (sg == sg) ? 2 * b.obj + (neg ? 1 : 0) : -1;
#endif
    } else {                                                             // OBJ_VOLUME geometry.rs:520
        face(mk3(0.0f, 0.0f, 0.0f), d, s.n, s.frontface);
    }
}

// Plane::intersect_ray geometry.rs:474-489 -> parametric t, or miss
template <class OP>
__device__ __forceinline__ bool plane_t(OP ob, f3 o, f3 d, float t_min, float t_max, float& t_out) {
    f3 normal = ld3(ob->f + 3);
    float origin_dist = dot(o - ld3(ob->f), normal);
    float sg = (origin_dist != origin_dist) ? origin_dist : (__float_as_uint(origin_dist) >> 31 ? -1.0f : 1.0f);
    f3 n = normal * sg;
    float dd = dot(d, n);
    float t = fabsf(origin_dist) / fabsf(dd);
    t_out = t;
    return !(dd >= 0.0f) && !(t < t_min || t > t_max);
}

// `self.boundary.intersect_ray(ray, t_min, t_max)` of a ConvexVolume (geometry.rs:505,508) -> distance of the boundary's hit.
// The boundary is the inline sphere (ref < 0: every use in the reference), or the records S.bobjs[ref ..): one Triangle / Plane /
// StaticMesh, or the entries of a nested Scene, whose answer is its closest hit — strictly closer wins, the first entry keeps a
// tie (tracing.rs:333-336).  A StaticMesh boundary is walked in line from global memory (geometry.rs:301-314, object-space ray,
// parametric t): a rare kind, not a tuned path.  No random number is drawn in here.
// GV = false compiles the sphere only: the hot kernels exist in both forms and the host launches the GV form only for a scene
// that holds such a boundary (the generic code costs wf_main 2 VGPRs, 30 scalar spills and 5 % on cfg5 even when it never runs).
template <bool GV, class OP>
__device__ __forceinline__ bool boundary_hit(const DScene& S, OP ob, f3 o, f3 d, float t_min, float t_max, float& t_out) {
    if (!GV || ob->ref < 0) return sphere_t(o, d, ld3(ob->f), ob->f[4], t_min, t_max, t_out);
    const int first = ob->ref, n = __float_as_int(ob->f[6]);
    bool have = false;
    float best = 0.0f;
    for (int k = 0; k < n; k++) {
        auto r = &S.bobjs[first + k];
        const int kind = r->kind;
        float t = 0.0f;
        bool ok;
        if (kind == OBJ_TRIANGLE) { float u, v; ok = tri_t(o, d, ld3(r->f), ld3(r->f + 3), ld3(r->f + 6), t_min, t_max, t, u, v); }
        else if (kind == OBJ_SPHERE) ok = sphere_t(o, d, ld3(r->f), r->f[4], t_min, t_max, t);
        else if (kind == OBJ_PLANE) ok = plane_t(r, o, d, t_min, t_max, t);
        else {                                                            // OBJ_MESH
            Bvh<false> G;
            bvh_bind(G, S, 0);
            auto M = &S.meshes[r->ref];
            const f3 oo = xform_point(M->inv_transform, o), od = xform_vector(M->inv_transform, d);
            float bu, bv; int btri;
            traverse_mesh(G, M->node_begin, M->node_end, M->tri_begin, oo, od, t_min, t_max, t, btri, bu, bv);
            ok = btri >= 0;
        }
        const bool take = ok && (!have || t < best);
        best = take ? t : best;
        have = have || take;
    }
    t_out = best;
    return have;
}

// One non-mesh object of Scene.objects against the ray (wave-uniform `ob`).
template <bool GV = true, class OP>
__device__ __forceinline__ void test_object(const DScene& S, OP ob, int idx, f3 o, f3 d, float t_min, float t_max,
                                            Rng& rng, Best& best) {
    int kind = ob->kind;
    if (kind == OBJ_TRIANGLE) {
        float t, u, v;
        if (tri_t(o, d, ld3(ob->f), ld3(ob->f + 3), ld3(ob->f + 6), t_min, t_max, t, u, v))
            consider(best, t, idx, -1, 0.0f, 0.0f);
    } else if (kind == OBJ_SPHERE) {
        float t;
        if (sphere_t(o, d, ld3(ob->f), ob->f[4], t_min, t_max, t)) consider(best, t, idx, -1, 0.0f, 0.0f);
    } else if (kind == OBJ_PLANE) {                                       // geometry.rs:474-489
        float t;
        if (plane_t(ob, o, d, t_min, t_max, t)) consider(best, t, idx, -1, 0.0f, 0.0f);       // Plane: same class as Triangle
    } else if (kind == OBJ_VOLUME) {                                      // geometry.rs:502-526
        const float F32_MIN = -3.40282347e+38f, F32_MAX = 3.40282347e+38f;
        float t_entr, t_exit;
        if (boundary_hit<GV>(S, ob, o, d, F32_MIN, F32_MAX, t_entr) &&
            boundary_hit<GV>(S, ob, o, d, t_entr + 0.0001f, F32_MAX, t_exit)) {
            if (!(t_exit < t_min || t_entr > t_max)) {
                float t_start = fmaxf(t_entr, t_min);
                float t_end = fminf(t_exit, t_max);
                float dist_in_volume = t_end - t_start;
                float dist_before_scatter = ob->f[5] * pt_logf(gen01(rng));     // :517, f[5] = -1/density
                if (dist_before_scatter < dist_in_volume)
                    consider(best, t_start + dist_before_scatter, idx, -3, 0.0f, 0.0f);          // tag -3: volume hit
            }
        }
    }
}

// ---- the padded-box machinery of the exact two-stage traversal (bvh_build.hpp states the bound): used by the
//      mesh walkers wf_trav_f / wf_replay further down and by the top-level tree over the list's Triangles right below ----
// bvh_build.hpp two_stage_pad, same f32 expression
__device__ __forceinline__ bool two_stage_pad(const PT_CONST_AS DMeshF* F, f3 o, f3 d, float t_max, float& rho_out, float& dt_out) {
    const float eps = 5.9604645e-08f;
    const float dn = sqrtf((d.x * d.x + d.y * d.y) + d.z * d.z) * 1.000001f;
    const float ox = o.x - F->cx, oy = o.y - F->cy, oz = o.z - F->cz;
    const float Sr = (sqrtf((ox * ox + oy * oy) + oz * oz) + F->R) * 1.000001f;
    const float E2 = F->E2, L = F->L;
    const float B = 7.0f * eps * E2 * dn * 1.0e4f;
    const float rho = 2.0f * eps * dn * E2 * (16.0f * Sr + 14.0f * L) * 1.0e4f + 11.0f * eps * L;
    const float dt = 2.0f * (8.1f * eps * E2 * Sr * 1.0e4f + fabsf(t_max) * (B + 2.001f * eps));
    rho_out = 4.0f * rho + 16.0f * eps * Sr;
    dt_out = 2.0f * dt;
    return (B <= 0.5f) && (Sr <= 1.0e12f) && (rho_out <= 1.0e30f) && (dt_out <= 1.0e30f);
}

// AABB test of pass 1: the box grown by rho, parameter range [t_lo, t_hi].  The padding is applied to the DIFFERENCES
// (bmin - o) - rho and (bmax - o) + rho: a mesh far from its object-space origin has |o| >> rho, and o + rho would round
// the padding away.
__device__ __forceinline__ bool slab_padded(f3 bmin, f3 bmax, f3 o, float rho, f3 inv_d, float t_lo, float t_hi) {
    float tmin = t_lo, tmax = t_hi;
    {
        float t0 = ((bmin.x - o.x) - rho) * inv_d.x, t1 = ((bmax.x - o.x) + rho) * inv_d.x;
        bool sw = inv_d.x < 0.0f;
        float ta = sw ? t1 : t0, tb = sw ? t0 : t1;
        tmin = fmaxf(ta, tmin); tmax = fminf(tb, tmax);
    }
    {
        float t0 = ((bmin.y - o.y) - rho) * inv_d.y, t1 = ((bmax.y - o.y) + rho) * inv_d.y;
        bool sw = inv_d.y < 0.0f;
        float ta = sw ? t1 : t0, tb = sw ? t0 : t1;
        tmin = fmaxf(ta, tmin); tmax = fminf(tb, tmax);
    }
    {
        float t0 = ((bmin.z - o.z) - rho) * inv_d.z, t1 = ((bmax.z - o.z) + rho) * inv_d.z;
        bool sw = inv_d.z < 0.0f;
        float ta = sw ? t1 : t0, tb = sw ? t0 : t1;
        tmin = fmaxf(ta, tmin); tmax = fminf(tb, tmax);
    }
    return !(tmax <= tmin);
}

// A quantised F-node (bvh_build.hpp fq_encode): six 16-bit grid coordinates, decoded by one fmaf each — the operation the encoder
// rounded outward against — and the link word.
__device__ __forceinline__ void fq_box(float4 c, float4 g, f3& bmin, f3& bmax) {
    const uint32_t w0 = __float_as_uint(c.x), w1 = __float_as_uint(c.y), w2 = __float_as_uint(c.z);
    bmin = mk3(__builtin_fmaf((float)(w0 & 0xffffu), g.x, g.y), __builtin_fmaf((float)(w0 >> 16), g.x, g.z), __builtin_fmaf((float)(w1 & 0xffffu), g.x, g.w));
    bmax = mk3(__builtin_fmaf((float)(w1 >> 16), g.x, g.y), __builtin_fmaf((float)(w2 & 0xffffu), g.x, g.z), __builtin_fmaf((float)(w2 >> 16), g.x, g.w));
}


// Scene::intersect_ray over the non-mesh objects (tracing.rs:330-344) through the
// kind-grouped list: one tight loop per kind, the record of the next object loaded (scalar
// load, wave-uniform address) while the current one is tested.
// `tag` (< 0) records which list loop produced the hit: -1 Triangle, -2 Sphere (Best.tri >= 0 is
// reserved for mesh triangles); the wavefront pipeline uses it to class the path for free.
// `have` = "a hit is recorded" (b.obj >= 0), carried as a lane mask so that the whole decision stays in mask
// registers (testing b.obj's sign bit made the compiler rebuild it from integers: 5 more VALU per object).
__device__ __forceinline__ void consider_list(Best& b, bool& have, bool ok, float t, int obj, int tag) {
    const bool better = (t < b.t) || ((t == b.t) && (obj < b.obj));
    const bool take = ok && (!have || better);
    have = have || take;
    b.t = take ? t : b.t;
    b.obj = take ? obj : b.obj;
    b.tri = take ? tag : b.tri;
}
// Sphere::intersect_ray (geometry.rs:395-413) over a run of Sphere entries, in TWO STAGES — the cull over Scene.objects for
// incoherent rays.  The test's cheap head is the discriminant (18 VALU: f, b, c, d = b*b - 4*a*c) and its first
// exit is `if d < 0.0 { None }` (:402); the tail — a correctly rounded sqrt and two divisions, the root choice, the range test,
// ~40 VALU — is only ever needed by a lane whose ray passed that exit, and a bounce ray passes it for few of a scene's spheres.
// So stage 1 runs for every sphere (wave-uniform record in SGPRs) and a lane that passes KEEPS {b, d, index} in a one-entry stash;
// stage 2 runs for the stashed lanes when some lane is about to need its stash again, and once at the end: as often as the
// fullest lane has candidates (2-3 times for the 17 spheres of run()'s scene) instead of once per sphere.  Exact by construction,
// no error bound involved: every value is computed by the reference's operations on the reference's operands, a lane's
// candidates reach `consider_list` in list order, and a lane that is no candidate was rejected by the reference's own comparison
// (a NaN discriminant is not `< 0.0`: it stays a candidate and "hits" at NaN as it does in the reference).
struct SphereStash { float b, disc; int idx; bool full; };
__device__ __forceinline__ void sphere_finish(SphereStash& st, float two_a, float t_min, float t_max, Best& best, bool& have) {
    const float sq = sqrtf(st.disc);                                     // NaN for a NaN discriminant (never negative here)
    const float t1 = (-st.b - sq) / two_a, t2 = (-st.b + sq) / two_a;     // :405-406
    const float t = (t1 >= t_min) ? t1 : t2;                              // :407
    const bool ok = st.full & !((t < t_min) | (t > t_max));               // :409
    consider_list(best, have, ok, t, st.idx, -2);
    st.full = false;
}
template <class REC>
__device__ __forceinline__ void sphere_stage1(REC ob, f3 o, f3 d, float a, float two_a, float t_min, float t_max, SphereStash& st, Best& best, bool& have) {
    const f3 f = o - ld3(ob->f);                                          // :397
    const float b = 2.0f * dot(f, d);                                     // :399
    const float c = mag2(f) - ob->f[4];                                   // :400, f[4] = radius * radius (hoisted: same product)
    const float disc = b * b - 4.0f * a * c;                              // :401
    const bool cand = !(disc < 0.0f);                                     // :402
    if (__builtin_amdgcn_ballot_w64(cand & st.full) != 0ull) sphere_finish(st, two_a, t_min, t_max, best, have);
    st.b = cand ? b : st.b; st.disc = cand ? disc : st.disc; st.idx = cand ? ob->index : st.idx; st.full = st.full | cand;
}

// RARE = false compiles the Plane / ConvexVolume loop out (a scene without either: the Cornell configurations)
// TOP = true (long lists; the scene compiler decides): only the first n_list_lin Triangles — the large ones — are tested one by one; the others
// sit in a top-level tree (mi_rt.cpp, bvh_build.hpp FTree: SAH over the triangles' boxes, 16-byte quantised nodes) that every lane walks
// with the boxes padded by the proven bound on what the reference's f32 test can accept, running the reference's own test on the triangles of
// the leaves it reaches.  A triangle whose padded box the ray misses would have failed that test; the closest hit over the rest does not
// depend on the order of evaluation (ties: the lower Scene.objects index, which the leaf triangles carry).  A ray the bound does not cover
// (B > 1/2, anything non-finite) makes its whole wave test the rest of the list one by one.
// (Round 4, measured negative: the Triangle / Sphere records of short lists staged in LDS by wf_main and read from there — a broadcast read
// lands in VGPRs, and on gfx950 a v_mul_f32 with an SGPR operand issues at 0.6 of the rate of the same multiply on VGPRs
// (tools/microbench/valu_rates.hip; 21 of a Triangle test's 57 instructions read the record).  cfg2 wf_main 50.1 -> 51.0 ms, the walkers
// beside it 22.5 -> 23.3 ms: ten LDS reads and their waits per pair of triangles cost more than the faster multiplies give back.  Also without
// effect: settling ties inside the Triangle loop without the index comparison (ascending order, `best` empty on entry) — 73.5 vs 73.4 ms.)
template <bool GV = true, bool RARE = true, bool TOP = false>
__device__ __forceinline__ void intersect_list(const DScene& S, f3 o, f3 d, float t_min, float t_max, Rng& rng, Best& best) {
    auto L = S.list;
    int k = 0;
    bool have = best.obj >= 0;
    {   // Triangle::intersect_ray geometry.rs:431-450.  Two triangles per trip: the two
        // Moller-Trumbore chains (cross, dot, correctly rounded 1/g, ...) are independent, so the
        // in-order issue overlaps their latencies; `consider` is applied in list order.
        const int end = TOP ? S.n_list_lin : S.n_list_tri;
        // (Round 4, measured negative: the NEXT pair's records asked for before the current pair is tested, so that the scalar loads' latency
        // passes behind the tests — 20 more live SGPRs, which the allocator pays for with v_writelane / v_readlane spills around the loop: cfg2
        // 75.3 -> 76.7 ms.  Eight resident waves hide that latency already.)
        for (; k + 1 < end; k += 2) {
            auto r0 = &L[k]; auto r1 = &L[k + 1];
            // both records' constants are asked for BEFORE the first test: its reciprocal's range check is a branch, and scalar loads
            // placed behind it would wait out their latency a second time per trip
            const f3 a0 = ld3(r0->f), p0 = ld3(r0->f + 3), q0 = ld3(r0->f + 6), a1 = ld3(r1->f), p1 = ld3(r1->f + 3), q1 = ld3(r1->f + 6);
            const int i0 = r0->index, i1 = r1->index;
            float t0, u0, v0, t1, u1, v1;
            bool ok0 = tri_t(o, d, a0, p0, q0, t_min, t_max, t0, u0, v0);
            bool ok1 = tri_t(o, d, a1, p1, q1, t_min, t_max, t1, u1, v1);
            consider_list(best, have, ok0, t0, i0, -1);
            consider_list(best, have, ok1, t1, i1, -1);
        }
        for (; k < end; k++) {
            auto r0 = &L[k];
            float t0, u0, v0;
            bool ok0 = tri_t(o, d, ld3(r0->f), ld3(r0->f + 3), ld3(r0->f + 6), t_min, t_max, t0, u0, v0);
            consider_list(best, have, ok0, t0, r0->index, -1);
        }
    }
    if (TOP) {
        auto F = &S.meshf[S.top_meshf];
        float rho, dt;
        const bool covered = two_stage_pad(F, o, d, t_max, rho, dt);
        if (__builtin_amdgcn_ballot_w64(!covered) != 0ull) {
            for (k = S.n_list_lin; k < S.n_list_tri; k++) {               // the plain loop for this wave
                auto r0 = &L[k];
                float t0, u0, v0;
                bool ok0 = tri_t(o, d, ld3(r0->f), ld3(r0->f + 3), ld3(r0->f + 6), t_min, t_max, t0, u0, v0);
                consider_list(best, have, ok0, t0, r0->index, -1);
            }
        } else {
            cf4_ptr FN = (cf4_ptr)S.fnodes;
            cf4_ptr FT = (cf4_ptr)S.ftris;
            f3 inv;
            rcp3_exact(d.x, d.y, d.z, inv.x, inv.y, inv.z);
            const float t_lo = t_min - dt, t_hi = t_max + dt;
            const float4 grid = make_float4(F->qs, F->qbx, F->qby, F->qbz);
            const int fend = F->fnode_end, ftb = F->ftri_begin;
            int fi = F->fnode_begin;
            // pre-order with skip links, no stack (all hits are wanted); the wave votes between a burst of box steps and the triangle tests
            // of the leaves reached, as the walkers do (a lane that has reached a leaf waits for the leaf step)
            bool atleaf = false;
            float4 c = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (fi < fend) c = FN[fi];
            // The window's far end follows the closest hit so far: a triangle that can still win (strictly closer, or as close with a lower
            // index) passes the reference's test at a computed t <= best.t, so the exact line meets its padded box before best.t + dt (the
            // bound's dt covers computed-vs-exact); nothing beyond that can change the result.  NaN distances leave the window as it is.
            float thi = have ? fminf(t_hi, best.t + dt) : t_hi;
            while (__builtin_amdgcn_ballot_w64(fi < fend) != 0ull) {
#pragma unroll
                for (int j = 0; j < 6; j++) {
                    const bool act = (fi < fend) & !atleaf;
                    f3 bmin, bmax;
                    fq_box(c, grid, bmin, bmax);
                    const bool hit = slab_padded(bmin, bmax, o, rho, inv, t_lo, thi);
                    const int link = __float_as_int(c.w);
                    const bool leaf = link < 0;
                    const bool stop = act & hit & leaf;
                    const int nxt = (hit | leaf) ? fi + 1 : link;            // a leaf's successor is the next node either way
                    atleaf = atleaf | stop;
                    const bool move = act & !stop;
                    fi = move ? nxt : fi;
                    if (move & (fi < fend)) c = FN[fi];
                }
                const int n_leaf = __popcll(__builtin_amdgcn_ballot_w64(atleaf));
                const int n_walk = __popcll(__builtin_amdgcn_ballot_w64((fi < fend) & !atleaf));
                if (n_leaf * 3 >= n_walk) {
                    if (atleaf) {
                        const int payload = __float_as_int(c.w) & 0x7fffffff;
                        const int first = ftb + (payload >> 3), count = (payload & 7) + 1;
                        for (int j = 0; j < count; j++) {
                            const float4 t0 = FT[3 * (first + j)], t1 = FT[3 * (first + j) + 1], t2 = FT[3 * (first + j) + 2];
                            float t, u, v;
                            const bool ok = tri_t(o, d, mk3(t0.x, t0.y, t0.z), mk3(t1.x, t1.y, t1.z), mk3(t2.x, t2.y, t2.z), t_min, t_max, t, u, v);
                            consider_list(best, have, ok, t, __float_as_int(t0.w), -1);       // t0.w: the triangle's Scene.objects index
                        }
                        fi = fi + 1;
                        atleaf = false;
                        if (fi < fend) c = FN[fi];
                        thi = have ? fminf(t_hi, best.t + dt) : t_hi;
                    }
                }
            }
        }
        k = S.n_list_tri;
    }
    {   // Sphere::intersect_ray geometry.rs:395-413, staged (above)
        const int end = k + S.n_list_sphere;
#if PT_SPHERE_STAGED
        if (k < end) {
            const float a = mag2(d), two_a = 2.0f * a;                    // :398, and the divisor of :405-406
            SphereStash st = { 0.0f, 0.0f, -1, false };
            for (; k < end; k++) sphere_stage1(&L[k], o, d, a, two_a, t_min, t_max, st, best, have);
            if (__builtin_amdgcn_ballot_w64(st.full) != 0ull) sphere_finish(st, two_a, t_min, t_max, best, have);
        }
#else
        for (; k < end; k++) {
            auto ob = &L[k];
            float t;
            bool ok = sphere_t(o, d, ld3(ob->f), ob->f[4], t_min, t_max, t);
            consider_list(best, have, ok, t, ob->index, -2);
        }
#endif
    }
    if (RARE) {   // Plane (geometry.rs:474-489) and ConvexVolume (geometry.rs:502-526): rare kinds, generic code
        const int end = k + S.n_list_plane + S.n_list_volume;
        for (; k < end; k++) {
            auto ob = &L[k];
            test_object<GV>(S, ob, ob->index, o, d, t_min, t_max, rng, best);
        }
    }
}

// Same walk for CAMERA rays of one tile: entries whose bit in `mask` is clear cannot be reached by any
// ray of that tile (host-side conservative frustum test, mi_rt.cpp tile_masks), so skipping them
// skips tests that would have missed.  `mask` is wave-uniform (SGPRs); planes and volumes are never
// masked (a volume draws its random number for the whole ray LINE, geometry.rs:505).
template <bool GV = true, bool RARE = true>
__device__ __forceinline__ void intersect_list_masked(const DScene& S, unsigned long long mask, f3 o, f3 d,
                                                      float t_min, float t_max, Rng& rng, Best& best) {
    auto L = S.list;
    bool have = best.obj >= 0;
    const int n_tri = S.n_list_tri, n_ts = S.n_list_tri + S.n_list_sphere;
    unsigned long long m = mask & ((n_ts >= 64) ? ~0ull : ((1ull << n_ts) - 1ull));
    while (m != 0ull) {
        const int k = __ffsll((long long)m) - 1;
        m &= m - 1ull;
        auto r0 = &L[k];
        if (k < n_tri) {
            float t0, u0, v0;
            bool ok0 = tri_t(o, d, ld3(r0->f), ld3(r0->f + 3), ld3(r0->f + 6), t_min, t_max, t0, u0, v0);
            consider_list(best, have, ok0, t0, r0->index, -1);
        } else {
            float t;
            bool ok = sphere_t(o, d, ld3(r0->f), r0->f[4], t_min, t_max, t);
            consider_list(best, have, ok, t, r0->index, -2);
        }
    }
    const int end = RARE ? n_ts + S.n_list_plane + S.n_list_volume : n_ts;
    for (int k = n_ts; k < end; k++) {
        auto ob = &L[k];
        test_object<GV>(S, ob, ob->index, o, d, t_min, t_max, rng, best);
    }
}

// Material::scatter (materials.rs) for the resolved surface; returns the new direction
// and the throughput factor dot_term*brdf/pdf of tracing.rs:313-316.
// Material::scatter proper: new direction, attenuation (`.1` of the reference's tuple), 1/pdf.
__device__ __forceinline__ void scatter_raw(const DScene& S, const Surf& s, f3 d, Rng& rng, f3& new_d, f3& brdf, float& inv_pdf) {
    int kind = s.kind;
    bool diffuse = (kind == MAT_LAMBERTIAN);
    inv_pdf = 1.0f;
    if (kind == MAT_PARAMETERIZED) {                                     // materials.rs:116-120
        float fr = fresnel(d, s.n, 1.5f);
        float k_s = fr * (1.0f - s.roughness);
        float k_d = (1.0f - k_s) * (1.0f - s.metallic);
        diffuse = gen01(rng) < k_d;
    }
    if (kind == MAT_DIELECTRIC) {                                        // materials.rs:80-90
        float eta = s.frontface ? 1.0f / s.ior : s.ior;
        float cosv = fminf(-dot(d, s.n), 1.0f);
        bool critical_angle = eta * sqrtf(1.0f - powi2(cosv)) > 1.0f;
        float fresnel_factor = fresnel(d, s.n, s.ior);
        bool will_refract = !critical_angle && (gen01(rng) >= fresnel_factor);   // short-circuit draw
        new_d = will_refract ? refract(d, s.n, eta) : reflect(d, s.n);
        brdf = mk3(1.0f, 1.0f, 1.0f);
    } else {
        // every other material draws rand_sphere_vec next; one shared rejection loop
        f3 v = rand_sphere_vec(rng);
        if (diffuse) {                                                   // sample_hemisphere :171-178
            v.y = fabsf(v.y);
            if (s.rot >= 0) {
                // the rotation depends on the normal alone, and a list Triangle presents one of two: its matrix was computed at upload by the
                // same f32 operations (mi_rt.cpp obj_rot) — 3 loads from a table of a few cache lines instead of ~85 VALU instructions (two
                // sqrt, a division, a reciprocal, two ulps_eq) in a kernel that is bound by instruction issue
                cf4_ptr R = (cf4_ptr)S.obj_rot + 3 * s.rot;
                const float4 r0 = R[0], r1 = R[1], r2 = R[2];
                const f3 m = mk3((r0.x * v.x + r0.w * v.y) + r1.z * v.z,
                                 (r0.y * v.x + r1.x * v.y) + r1.w * v.z,
                                 (r0.z * v.x + r1.y * v.y) + r2.x * v.z);
                const bool ident = r2.y != 0.0f;
                new_d = mk3(ident ? v.x : m.x, ident ? v.y : m.y, ident ? v.z : m.z);
            } else
            new_d = rotate_from_unit_y(s.n, v);
            brdf = s.brdf_diffuse;                                       // albedo / PI
            inv_pdf = 6.28318530717958647692f;                           // pdf = 1/(2*PI)
        } else if (kind == MAT_ISOTROPIC) {                              // :161
            new_d = v;
            brdf = s.albedo;
        } else {                                                         // Metal :62 / Parameterized :137
            new_d = reflect(d, s.n) + v * s.roughness;
            if (kind == MAT_METAL) brdf = s.albedo;
            else {                                                       // lerpvec(1, albedo, metallic) tracing.rs:95-97
                float k = s.metallic;
                brdf = mk3(1.0f, 1.0f, 1.0f) * (1.0f - k) + s.albedo * k;
            }
        }
    }
}
// scatter + the estimator's weight dot_term * brdf / pdf (tracing.rs:312-316)
__device__ __forceinline__ void scatter(const DScene& S, const Surf& s, f3 d, Rng& rng, f3& new_d, f3& weight) {
    f3 brdf; float inv_pdf;
    scatter_raw(S, s, d, rng, new_d, brdf, inv_pdf);
    // tracing.rs:313
    float dot_term = (mag2(s.n) > 0.0f) ? clampf(fabsf(dot(new_d, s.n)), 0.0f, 1.0f) : 1.0f;
    float wgt = dot_term * inv_pdf;
    weight = mk3(brdf.x * wgt, brdf.y * wgt, brdf.z * wgt);
}

// Camera::generate_rays for one sample (tracing.rs:165-206)
__device__ __forceinline__ void generate_ray(const DCamera& C, uint32_t px, uint32_t py, uint32_t i, Rng& rng, f3& o, f3& d) {
    float rand_x = (float)gen_u32_below(rng, C.spp, C.zone);             // :167
    float rand_y = (float)gen_u32_below(rng, C.spp, C.zone);             // :168
    float subpixel_x = (float)(i / C.rootn_u);                           // :169
    float subpixel_y = (float)(i % C.rootn_u);                           // :170
    float off_x = (subpixel_x - C.half_rootn) * C.pixel_size / C.rootn + (rand_x - C.half_n) * C.pixel_size / C.n;
    float off_y = (subpixel_y - C.half_rootn) * C.pixel_size / C.rootn + (rand_y - C.half_n) * C.pixel_size / C.n;
    f3 center = mk3(C.pixel_size * ((float)px + C.cx_base + 0.5f) + off_x,          // :178
                    C.pixel_size * (C.cy_base - (float)py) + off_y,                 // :179
                    -C.focal_length);
    f3 focus = normalize(center) * C.focus_dist;                         // :183
    f3 lens = rand_disk_vec(rng) * C.lens_radius;                        // :184
    if (C.ortho) {                                                       // :196,200 (wave-uniform)
        o = mk3(center.x, center.y, 0.0f);
        d = ld3(C.ortho_dir);
    } else {
        o = ld3(C.eye) + m3mul(C.rot, lens);                             // :197
        d = m3mul(C.rot, normalize(focus - lens));                       // :201,204
    }
}

// path-signature steps (diagnostic; DESIGN.md "Path signature")
__device__ __forceinline__ uint32_t sig_hit(uint32_t sig, float t, int obj) {
    return lowbias32((sig ^ __float_as_uint(t)) + (uint32_t)(obj + 1) * 0x9e3779b1u);
}
__device__ __forceinline__ uint32_t sig_end_miss(uint32_t sig, const Rng& r) {
    return lowbias32(sig ^ r.s0 ^ rotl32(r.s1, 16));
}
__device__ __forceinline__ uint32_t sig_end_depth(uint32_t sig) { return lowbias32(sig ^ 0x5bd1e995u); }

// ---------------------------------------------------------------- K1
template <bool LDS, bool SIG>
__global__ __launch_bounds__(kBlock) void pt_megakernel(K1Args A) {
    const DScene& S = A.S;
    const DCamera& C = A.C;

    // ---- stage the mesh BVH into LDS (nodes then triangles) ----
    Bvh<LDS> B;
    if (LDS) {
        cf4_ptr gn = (cf4_ptr)S.nodes;
        cf4_ptr gt = (cf4_ptr)S.tris;
        int nn = (int)A.R.lds_nodes * 2, nt = (int)A.R.lds_tris * 3;
        for (int k = threadIdx.x; k < nn; k += kBlock) k1_lds[k] = gn[k];
        for (int k = threadIdx.x; k < nt; k += kBlock) k1_lds[nn + k] = gt[k];
        __syncthreads();
    }
    bvh_bind(B, S, (int)A.R.lds_nodes * 2);

    // ---- lane -> pixel ----
    const uint32_t slot = blockIdx.x / kBlocksPerTile;            // tile slot of this rank
    const uint32_t sub = blockIdx.x % kBlocksPerTile;             // 32x8 strip inside the tile
    const uint32_t tile = slot * (uint32_t)A.R.world + (uint32_t)A.R.rank;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t in_x = wave * 8 + (lane & 7), in_y = sub * 8 + (lane >> 3);
    const uint32_t out_idx = slot * kTilePixels + in_y * kTile + in_x;
    uint32_t px = 0, py = 0;
    bool in_image = false;
    if (tile < A.R.tiles_total) {
        px = (tile % A.R.tiles_x) * kTile + in_x;
        py = (tile / A.R.tiles_x) * kTile + in_y;
        in_image = (px < C.width) && (py < C.height);
    }
    const uint32_t pixel = py * C.width + px;
    const float t_min = 0.001f, t_max = C.max_trace_dist;         // tracing.rs:305

    f3 accum = mk3(0.0f, 0.0f, 0.0f);                             // tracing.rs:232
    uint32_t sigsum = 0;
    uint32_t sample = 0;
    const uint32_t spp = in_image ? C.spp : 0u;

    Path P;
    P.o = P.d = P.T = P.L = mk3(0.0f, 0.0f, 0.0f); P.depth = 0; P.sig = 0; P.rng.s0 = P.rng.s1 = 1u;
    bool fresh = true;          // lane needs a new camera ray
    bool alive = true;          // lane still has samples to do
    Best best; best.t = 0.0f; best.obj = -1; best.tri = -1; best.u = best.v = 0.0f;

    while (__any(alive)) {
        if (alive && fresh) {
            if (sample >= spp) alive = false;
            else {
                rng_init(P.rng, A.seed_key, pixel, sample);
                generate_ray(C, px, py, sample, P.rng, P.o, P.d);
                P.T = mk3(1.0f, 1.0f, 1.0f); P.L = mk3(0.0f, 0.0f, 0.0f); P.depth = 0; P.sig = 0;
                fresh = false;
                if (C.path_depth == 0u) {                                 // :301 at level 0: the background, before any intersection
                    if (SIG) sigsum += sig_end_depth(P.sig);
                    sample++;
                    fresh = true;
                }
            }
        }
        if (alive && !fresh) {
        // ---- Scene::intersect_ray (tracing.rs:330-344): objects in order, wave-uniform index ----
        best.obj = -1; best.t = 0.0f; best.tri = -1; best.u = best.v = 0.0f;
        for (int k = 0; k < S.n_objects; k++) {
            auto ob = &S.objects[k];
            if (ob->kind != OBJ_MESH) test_object(S, ob, k, P.o, P.d, t_min, t_max, P.rng, best);
        }
        for (int m = 0; m < S.n_meshes; m++) {                            // StaticMesh::intersect_ray geometry.rs:301-314
            auto M = &S.meshes[m];
            f3 oo = xform_point(M->inv_transform, P.o);
            f3 od = xform_vector(M->inv_transform, P.d);
            float bt, bu, bv; int btri;
            traverse_mesh(B, M->node_begin, M->node_end, M->tri_begin, oo, od, t_min, t_max, bt, btri, bu, bv);
            if (btri >= 0) consider(best, bt, M->object_index, btri, bu, bv);
        }
        // ---- Scene::shade_ray, one level (tracing.rs:305-321) ----
        bool end_path;
        if (best.obj < 0) {                                           // :306 background = 0
            end_path = true;
            if (SIG) P.sig = sig_end_miss(P.sig, P.rng);
        } else {
            if (SIG) P.sig = sig_hit(P.sig, best.t, best.obj);
            Surf s;
            resolve_hit(S, best, P.o, P.d, s);
            // L += T * emission                                         :321
            P.L = mk3(P.L.x + P.T.x * s.emission.x, P.L.y + P.T.y * s.emission.y, P.L.z + P.T.z * s.emission.z);
            P.depth++;
            if (P.depth >= C.path_depth) {                            // :301 at the next level
                end_path = true;
                if (SIG) P.sig = sig_end_depth(P.sig);
            } else {
                f3 nd, w;
                scatter(S, s, P.d, P.rng, nd, w);                        // :312-316
                P.o = s.p; P.d = nd;
                P.T = mk3(P.T.x * w.x, P.T.y * w.y, P.T.z * w.z);
                end_path = false;
            }
        }
        if (end_path) {
            accum = accum + P.L;                                      // :238
            if (SIG) sigsum += P.sig;
            sample++;
            fresh = true;
        }
        }   // alive
    }

    // ---- per-pixel mean (tracing.rs:241); pixels outside the image are written as 0 ----
    float n = (float)C.spp;
    float* o3 = A.out + (size_t)out_idx * 3;
    if (in_image) { o3[0] = accum.x / n; o3[1] = accum.y / n; o3[2] = accum.z / n; }
    else { o3[0] = 0.0f; o3[1] = 0.0f; o3[2] = 0.0f; }
    if (SIG && A.sig) A.sig[out_idx] = in_image ? sigsum : 0u;
}

// ---------------------------------------------------------------- ShadingMode::Phong
// Scene::phong_shade_ray (tracing.rs:277-297), the reference's debug shader: one primary hit, a point
// light with a shadow ray, ambient + diffuse*attenuation + specular.  Not a performance path: one lane
// per pixel, samples in order, BVH walked in line from global memory.
template <class BVH>
__device__ __forceinline__ void intersect_scene(const DScene& S, const BVH& B, f3 o, f3 d, float t_min, float t_max,
                                                Rng& rng, Best& best) {
    best.obj = -1; best.t = 0.0f; best.tri = -1; best.u = best.v = 0.0f;
    intersect_list(S, o, d, t_min, t_max, rng, best);
    for (int m = 0; m < S.n_meshes; m++) {                               // geometry.rs:301-314
        auto M = &S.meshes[m];
        f3 oo = xform_point(M->inv_transform, o);
        f3 od = xform_vector(M->inv_transform, d);
        float bt, bu, bv; int btri;
        traverse_mesh(B, M->node_begin, M->node_end, M->tri_begin, oo, od, t_min, t_max, bt, btri, bu, bv);
        if (btri >= 0) consider(best, bt, M->object_index, btri, bu, bv);
    }
}

// x.powf(40.0) (:287) as a fixed sequence of f32 multiplies (libm powf is not reproducible across CPU and GPU)
__device__ __forceinline__ float pow40(float x) {
    float x2 = x * x, x4 = x2 * x2, x8 = x4 * x4, x16 = x8 * x8, x32 = x16 * x16;
    return x32 * x8;
}

template <bool SIG>
__global__ __launch_bounds__(kBlock) void pt_phong(K1Args A) {
    const DScene& S = A.S;
    const DCamera& C = A.C;
    Bvh<false> B;
    bvh_bind(B, S, 0);
    const uint32_t slot = blockIdx.x / kBlocksPerTile, sub = blockIdx.x % kBlocksPerTile;
    const uint32_t tile = slot * (uint32_t)A.R.world + (uint32_t)A.R.rank;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t in_x = wave * 8 + (lane & 7), in_y = sub * 8 + (lane >> 3);
    const uint32_t out_idx = slot * kTilePixels + in_y * kTile + in_x;
    uint32_t px = 0, py = 0;
    bool in_image = false;
    if (tile < A.R.tiles_total) {
        px = (tile % A.R.tiles_x) * kTile + in_x;
        py = (tile / A.R.tiles_x) * kTile + in_y;
        in_image = (px < C.width) && (py < C.height);
    }
    const uint32_t pixel = py * C.width + px;
    const float t_max = C.max_trace_dist;
    const f3 light = ld3(C.light), ambient = ld3(C.ambient), eye = ld3(C.eye);
    f3 accum = mk3(0.0f, 0.0f, 0.0f);
    uint32_t sigsum = 0;
    const uint32_t spp = in_image ? C.spp : 0u;
    for (uint32_t sample = 0; sample < spp; sample++) {
        Rng rng; f3 o, d;
        rng_init(rng, A.seed_key, pixel, sample);
        generate_ray(C, px, py, sample, rng, o, d);
        uint32_t sig = 0;
        f3 color = mk3(0.0f, 0.0f, 0.0f);
        Best best;
        intersect_scene(S, B, o, d, 0.0f, t_max, rng, best);             // :279
        if (best.obj < 0) {
            if (SIG) sig = sig_end_miss(sig, rng);                       // :280 background = 0
        } else {
            if (SIG) sig = sig_hit(sig, best.t, best.obj);
            Surf s;
            resolve_hit(S, best, o, d, s);
            f3 to_light = normalize(light - s.p);                        // :283
            f3 to_camera = normalize(eye - s.p);                         // :284
            f3 reflected = -to_light + s.n * (2.0f * dot(to_light, s.n));   // :285
            float diffuse_weight = clampf(dot(s.n, to_light), 0.0f, 1.0f);  // :286
            float specular_weight = pow40(clampf(dot(to_camera, reflected), 0.0f, 1.0f));   // :287
            f3 so = s.p + s.n * 0.01f;                                   // :289
            float shadow_weight = 1.0f;                                  // :290-293
            Best sb;
            intersect_scene(S, B, so, to_light, 0.0f, sqrtf(mag2(light - s.p)), rng, sb);
            if (sb.obj >= 0) {
                if (SIG) sig = sig_hit(sig, sb.t, sb.obj);
                Surf ss;
                resolve_hit(S, sb, so, to_light, ss);                    // the arm's `hit` is the SHADOW hit
                shadow_weight = (sb.t * sb.t > mag2(light - ss.p)) ? 1.0f : 0.3f;
            }
            f3 nd, brdf; float inv_pdf;
            scatter_raw(S, s, d, rng, nd, brdf, inv_pdf);                // :294 `.scatter(&hit, ray).1`
            f3 sum = (ambient + brdf * diffuse_weight) + mk3(0.4f, 0.4f, 0.4f) * specular_weight;
            color = sum * shadow_weight;
        }
        accum = accum + color;                                           // :235
        if (SIG) sigsum += sig;
    }
    float n = (float)C.spp;
    float* o3 = A.out + (size_t)out_idx * 3;
    if (in_image) { o3[0] = accum.x / n; o3[1] = accum.y / n; o3[2] = accum.z / n; }
    else { o3[0] = 0.0f; o3[1] = 0.0f; o3[2] = 0.0f; }
    if (SIG && A.sig) A.sig[out_idx] = in_image ? sigsum : 0u;
}

// ---------------------------------------------------------------- path_samples > 1 (and the literal estimator)
// Scene::shade_ray as written (tracing.rs:300-324): at every hit `path_samples` scattered rays are shaded
// recursively and averaged.  No configuration uses path_samples != 1 (tracing.rs:370) and the tree grows as
// path_samples^depth, so this is a completeness path, not a performance path: one lane per pixel, the recursion
// unrolled onto an explicit per-lane stack (scratch memory), samples and random draws in the reference's
// depth-first order.  The radiance is combined exactly as :316-321 nest it — ((brdf*incoming)*dot)/pdf,
// /path_samples, emission + integral — so unlike the forward-accumulating kernels it reproduces the oracle's
// f32 image bit for bit; MI_VARIANT_RECURSIVE runs it for path_samples == 1 too (tests use that as a
// second, independent check of every device function).
constexpr int kMaxRecursion = 64;
struct BranchFrame {
    Surf s;             // the hit being integrated over
    f3 d;               // direction of the ray that produced it (scatter needs it)
    f3 integral;        // :309
    f3 brdf;            // of the sample whose incoming light is being computed
    float dot_term, pdf;
    uint32_t i;         // samples done at this level
};

template <bool SIG>
__global__ __launch_bounds__(kBlock) void pt_branch(K1Args A, uint32_t path_samples) {
    const DScene& S = A.S;
    const DCamera& C = A.C;
    Bvh<false> B;
    bvh_bind(B, S, 0);
    const uint32_t slot = blockIdx.x / kBlocksPerTile, sub = blockIdx.x % kBlocksPerTile;
    const uint32_t tile = slot * (uint32_t)A.R.world + (uint32_t)A.R.rank;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t in_x = wave * 8 + (lane & 7), in_y = sub * 8 + (lane >> 3);
    const uint32_t out_idx = slot * kTilePixels + in_y * kTile + in_x;
    uint32_t px = 0, py = 0;
    bool in_image = false;
    if (tile < A.R.tiles_total) {
        px = (tile % A.R.tiles_x) * kTile + in_x;
        py = (tile / A.R.tiles_x) * kTile + in_y;
        in_image = (px < C.width) && (py < C.height);
    }
    const uint32_t pixel = py * C.width + px;
    const float t_min = 0.001f, t_max = C.max_trace_dist;
    const float PI = 3.14159265358979323846f;
    const uint32_t depth_cap = C.path_depth < (uint32_t)kMaxRecursion ? C.path_depth : (uint32_t)kMaxRecursion;   // host checks <= 64
    BranchFrame F[kMaxRecursion];
    f3 accum = mk3(0.0f, 0.0f, 0.0f);
    uint32_t sigsum = 0;
    const uint32_t spp = in_image ? C.spp : 0u;
    for (uint32_t sample = 0; sample < spp; sample++) {
        Rng rng; f3 o, d;
        rng_init(rng, A.seed_key, pixel, sample);
        generate_ray(C, px, py, sample, rng, o, d);
        uint32_t sig = 0;
        uint32_t level = 0;
        f3 ret = mk3(0.0f, 0.0f, 0.0f);
        bool entering = true;                          // true: shade_ray(o, d, level) is being called; false: it returned `ret`
        for (;;) {
            if (entering) {
                bool have_hit = false;
                if (level >= depth_cap) {              // :301
                    if (SIG) sig = sig_end_depth(sig);
                } else {
                    Best best;
                    intersect_scene(S, B, o, d, t_min, t_max, rng, best);       // :305
                    if (best.obj < 0) { if (SIG) sig = sig_end_miss(sig, rng); }
                    else {
                        if (SIG) sig = sig_hit(sig, best.t, best.obj);
                        BranchFrame& f = F[level];
                        resolve_hit(S, best, o, d, f.s);
                        f.d = d; f.integral = mk3(0.0f, 0.0f, 0.0f); f.i = 0;
                        have_hit = true;
                    }
                }
                if (!have_hit) { ret = mk3(0.0f, 0.0f, 0.0f); entering = false; }   // background :302,306
            } else {
                if (level == 0) break;                 // shade_ray(camera ray, 0) returned
                level--;
                BranchFrame& f = F[level];             // :316  integral += (dot_term*(brdf (.) incoming)) / pdf
                f.integral = mk3(f.integral.x + ((f.brdf.x * ret.x) * f.dot_term) / f.pdf,
                                 f.integral.y + ((f.brdf.y * ret.y) * f.dot_term) / f.pdf,
                                 f.integral.z + ((f.brdf.z * ret.z) * f.dot_term) / f.pdf);
                f.i++;
                entering = true;                       // fall into the sample loop of this level
            }
            if (entering) {
                // here F[level] holds a hit; run the sample loop :310 from f.i
                BranchFrame& f = F[level];
                if (f.i < path_samples) {
                    f3 nd; float inv_pdf;
                    scatter_raw(S, f.s, f.d, rng, nd, f.brdf, inv_pdf);         // :312
                    f.pdf = (inv_pdf != 1.0f) ? 1.0f / (2.0f * PI) : 1.0f;      // sample_hemisphere's pdf (materials.rs:178)
                    f.dot_term = (mag2(f.s.n) > 0.0f) ? clampf(fabsf(dot(nd, f.s.n)), 0.0f, 1.0f) : 1.0f;   // :313
                    o = f.s.p; d = nd; level++;        // :314 shade_ray(&new_ray, depth + 1)
                } else {
                    const float n = (float)path_samples;                        // :318, :321
                    ret = mk3(f.s.emission.x + f.integral.x / n, f.s.emission.y + f.integral.y / n, f.s.emission.z + f.integral.z / n);
                    entering = false;
                }
            }
        }
        accum = accum + ret;                                                    // :238
        if (SIG) sigsum += sig;
    }
    float n = (float)C.spp;
    float* o3 = A.out + (size_t)out_idx * 3;
    if (in_image) { o3[0] = accum.x / n; o3[1] = accum.y / n; o3[2] = accum.z / n; }
    else { o3[0] = 0.0f; o3[1] = 0.0f; o3[2] = 0.0f; }
    if (SIG && A.sig) A.sig[out_idx] = in_image ? sigsum : 0u;
}

// ---------------------------------------------------------------- K1, voted state machine
// The divergent part of the path is the BVH walk: a ray that enters the teapot takes
// 20..150 node steps, its 63 neighbours take 0.  Running the walk to completion inside
// a segment (SIMPLE above) makes the wave wait for its slowest lane.  Here the
// scheduling unit is ONE NODE STEP.  Every lane is a small state machine:
//     A     not inside a mesh: shade the pending hit, start a new sample if the path
//           ended, walk the object list for the next segment, test the mesh root boxes
//     TRAV  inside a mesh BVH: holds (node index, running best) in registers
// and every trip of the wave-wide loop VOTES with __ballot/__popcll which phase to run:
//     A-trip     when lanes in A are the majority,
//     B-trip     otherwise: up to k_steps micro-steps, each again voted between the lanes
//                that sit on an interior node (slab test) and those on a leaf (triangle test).
// Lanes that are not in the voted phase keep their state and wait, so each instruction
// stream runs with most of its lanes live, and lanes re-join the A phase as soon as THEIR
// walk ends instead of when the slowest walk of the wave ends.
// Order of evaluation per lane is unchanged (list objects in order, then meshes, DFS
// left-then-right with one running bound), so results are bit-identical to SIMPLE.

// find the next mesh (index >= m) whose root the ray enters; set up the walk registers.
// returns false when no mesh is left.
template <class BVH>
__device__ __forceinline__ bool enter_next_mesh(const DScene& S, const BVH& B, int& m, f3 o, f3 d, float t_min, float t_max,
                                                f3& oo, f3& od, f3& inv_d, int& ti, int& tend, int& ttb,
                                                uint32_t mesh_mask = 0xffffffffu) {
    for (; m < S.n_meshes; m++) {
        if (m < 32 && !((mesh_mask >> m) & 1u)) continue;                // camera rays: root box out of the tile's reach
        auto M = &S.meshes[m];
        oo = xform_point(M->inv_transform, o);                           // geometry.rs:304
        od = xform_vector(M->inv_transform, d);
        rcp3_exact(od.x, od.y, od.z, inv_d.x, inv_d.y, inv_d.z);         // geometry.rs:57
        float4 n0, n1;
        B.node(M->node_begin, n0, n1);
        tend = M->node_end; ttb = M->tri_begin;
        if (__float_as_int(n1.w) >= 0) { ti = M->node_begin; return true; }          // root is a leaf (:95)
        if (slab(mk3(n0.x, n0.y, n0.z), mk3(n1.x, n1.y, n1.z), oo, inv_d, t_min, t_max)) {   // root box (:103)
            ti = M->node_begin + 1;                                      // root passed with bound t_max: go to its left child
            return true;
        }
    }
    return false;
}

template <bool LDS, bool SIG, bool DIAG, bool GV>
__global__ __launch_bounds__(kBlock, PT_MIN_WAVES) void pt_megakernel_voted(K1Args A) {
    const DScene& S = A.S;
    const DCamera& C = A.C;

    Bvh<LDS> B;
    if (LDS) {
        cf4_ptr gn = (cf4_ptr)S.nodes;
        cf4_ptr gt = (cf4_ptr)S.tris;
        int nn = (int)A.R.lds_nodes * 2, nt = (int)A.R.lds_tris * 3;
        for (int k = threadIdx.x; k < nn; k += kBlock) k1_lds[k] = gn[k];
        for (int k = threadIdx.x; k < nt; k += kBlock) k1_lds[nn + k] = gt[k];
        __syncthreads();
    }
    bvh_bind(B, S, (int)A.R.lds_nodes * 2);

    const uint32_t slot = blockIdx.x / kBlocksPerTile;
    const uint32_t sub = blockIdx.x % kBlocksPerTile;
    const uint32_t tile = slot * (uint32_t)A.R.world + (uint32_t)A.R.rank;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t in_x = wave * 8 + (lane & 7), in_y = sub * 8 + (lane >> 3);
    const uint32_t out_idx = slot * kTilePixels + in_y * kTile + in_x;
    uint32_t px = 0, py = 0;
    bool in_image = false;
    if (tile < A.R.tiles_total) {
        px = (tile % A.R.tiles_x) * kTile + in_x;
        py = (tile / A.R.tiles_x) * kTile + in_y;
        in_image = (px < C.width) && (py < C.height);
    }
    const uint32_t pixel = py * C.width + px;
    const float t_min = 0.001f, t_max = C.max_trace_dist;
    const int vote_t = (int)A.R.vote_t, vote_a = (int)A.R.vote_a, k_steps = (int)A.R.k_steps;

    f3 accum = mk3(0.0f, 0.0f, 0.0f);
    uint32_t sigsum = 0, sample = 0;
    const uint32_t spp = in_image ? C.spp : 0u;

    Path P;
    P.o = P.d = P.T = P.L = mk3(0.0f, 0.0f, 0.0f); P.depth = 0; P.sig = 0; P.rng.s0 = P.rng.s1 = 1u;
    Best best; best.t = 0.0f; best.obj = -1; best.tri = -1; best.u = best.v = 0.0f;

    enum : int { ST_A = 0, ST_TRAV = 1, ST_DEAD = 2 };
    int state = ST_A;
    bool fresh = true, pending = false;
    // walk registers (state TRAV)
    int tm = 0, ti = 0, tend = 0, ttb = 0, tbtri = -1;
    f3 too = mk3(0.0f, 0.0f, 0.0f), tod = too, tinv = too;
    float tbt = 0.0f, tbu = 0.0f, tbv = 0.0f;

    unsigned long long dg_tripsA = 0, dg_lanesA = 0, dg_tripsI = 0, dg_lanesI = 0, dg_tripsL = 0, dg_lanesL = 0, dg_tripsB = 0;
    unsigned long long dg_cycA = 0, dg_cycB = 0, dg_t0 = 0, dg_cycShade = 0, dg_cycGen = 0, dg_cycList = 0, dg_t1 = 0;
    unsigned long long dg_slabs = 0, dg_roots = 0;             // DIAG: slab tests inside trees / at mesh roots   // DIAG only: shader-clock cycles spent in A / B trips

    while (true) {
        const int nA = __popcll(__builtin_amdgcn_ballot_w64(state == ST_A));
        const int nT = __popcll(__builtin_amdgcn_ballot_w64(state == ST_TRAV));
        if (nA + nT == 0) break;
        const bool run_b = (nT > 0) && (nT * vote_t >= nA * vote_a);

        if (DIAG) dg_t0 = __builtin_amdgcn_s_memtime();
        if (!run_b) {
            if (DIAG) { dg_tripsA++; dg_lanesA += (unsigned long long)nA; }
            bool did_list = false;
            if (state == ST_A) {
                // ---- (a) Scene::shade_ray, one level, for the intersection found last trip ----
                if (DIAG) dg_t1 = __builtin_amdgcn_s_memtime();
                if (pending) {
                    pending = false;
                    bool end_path;
                    if (best.obj < 0) {
                        end_path = true;
                        if (SIG) P.sig = sig_end_miss(P.sig, P.rng);
                    } else {
                        if (SIG) P.sig = sig_hit(P.sig, best.t, best.obj);
                        Surf s;
                        resolve_hit(S, best, P.o, P.d, s);
                        P.L = mk3(P.L.x + P.T.x * s.emission.x, P.L.y + P.T.y * s.emission.y, P.L.z + P.T.z * s.emission.z);
                        P.depth++;
                        if (P.depth >= C.path_depth) {
                            end_path = true;
                            if (SIG) P.sig = sig_end_depth(P.sig);
                        } else {
                            f3 nd, w;
                            scatter(S, s, P.d, P.rng, nd, w);
                            P.o = s.p; P.d = nd;
                            P.T = mk3(P.T.x * w.x, P.T.y * w.y, P.T.z * w.z);
                            end_path = false;
                        }
                    }
                    if (end_path) {
                        accum = accum + P.L;
                        if (SIG) sigsum += P.sig;
                        sample++;
                        fresh = true;
                    }
                }
                // ---- (b) Camera::generate_rays for the next sample ----
                if (DIAG) { unsigned long long t = __builtin_amdgcn_s_memtime(); dg_cycShade += t - dg_t1; dg_t1 = t; }
                if (fresh) {
                    if (sample >= spp) state = ST_DEAD;
                    else {
                        rng_init(P.rng, A.seed_key, pixel, sample);
                        generate_ray(C, px, py, sample, P.rng, P.o, P.d);
                        P.T = mk3(1.0f, 1.0f, 1.0f); P.L = mk3(0.0f, 0.0f, 0.0f); P.depth = 0; P.sig = 0;
                        fresh = false;
                        if (C.path_depth == 0u) {                         // :301 at level 0: the background, before any intersection
                            if (SIG) sigsum += sig_end_depth(P.sig);
                            sample++;
                            fresh = true;
                        }
                    }
                }
                // ---- (c) Scene::intersect_ray: the object list, then the mesh roots ----
                if (DIAG) { unsigned long long t = __builtin_amdgcn_s_memtime(); dg_cycGen += t - dg_t1; dg_t1 = t; }
                if (state == ST_A && !fresh) {
                    best.obj = -1; best.t = 0.0f; best.tri = -1; best.u = best.v = 0.0f;
                    intersect_list<GV>(S, P.o, P.d, t_min, t_max, P.rng, best);
                    if (DIAG) did_list = true;
                    tm = 0;
                    if (enter_next_mesh(S, B, tm, P.o, P.d, t_min, t_max, too, tod, tinv, ti, tend, ttb)) {
                        state = ST_TRAV; tbt = t_max; tbtri = -1; tbu = tbv = 0.0f;
                    } else {
                        pending = true;
                    }
                }
                if (DIAG) dg_cycList += __builtin_amdgcn_s_memtime() - dg_t1;
            }
            if (DIAG) dg_roots += (unsigned long long)__popcll(__builtin_amdgcn_ballot_w64(did_list));   // wave-uniform
        } else {
            if (DIAG) dg_tripsB++;
            // ---- B-trip: BVHNode::intersect_ray (geometry.rs:94-119) ----
            // Each lane inside a BVH keeps its CURRENT NODE in registers (c0, c1).  One vote decides
            // between a BURST of up to kBurst interior steps and one leaf step.  Inside a burst there
            // is no vote and no branch: both possible successors (left child ti+1, skip link) are
            // fetched from LDS while the slab test computes, and the result selects one of them.  A
            // lane that reaches a leaf or the end of its tree just sits out the rest of the burst.
            constexpr int kBurst = 4;
            const int last_node = S.n_nodes - 1;
            float4 c0, c1;
            B.node((state == ST_TRAV) ? ti : 0, c0, c1);
            for (int k = 0; k < k_steps; k++) {
                const bool in_t = (state == ST_TRAV);
                const int tri = __float_as_int(c1.w);
                const bool at_leaf = in_t & (tri >= 0), at_inner = in_t & (tri < 0);
                const int n_leaf = __popcll(__builtin_amdgcn_ballot_w64(at_leaf));
                const int n_inner = __popcll(__builtin_amdgcn_ballot_w64(at_inner));
                if (n_leaf + n_inner == 0) break;
                if (n_inner >= n_leaf) {
                    if (DIAG) { dg_tripsI++; dg_lanesI += (unsigned long long)n_inner; }
#pragma unroll
                    for (int j = 0; j < kBurst; j++) {
                        const bool act = in_t & (ti < tend) & (__float_as_int(c1.w) < 0);
                        if (DIAG) dg_slabs += (unsigned long long)__popcll(__builtin_amdgcn_ballot_w64(act));
                        const int skip = __float_as_int(c0.w);
                        float4 l0, l1, s0, s1;
                        B.node(act ? min(ti + 1, last_node) : 0, l0, l1);
                        B.node(act ? min(skip, last_node) : 0, s0, s1);
                        const bool hit = slab(mk3(c0.x, c0.y, c0.z), mk3(c1.x, c1.y, c1.z), too, tinv, t_min, tbt);   // :103
                        const bool go = act & hit, stay = !act;
                        ti = stay ? ti : (go ? ti + 1 : skip);
                        c0.x = stay ? c0.x : (go ? l0.x : s0.x); c0.y = stay ? c0.y : (go ? l0.y : s0.y);
                        c0.z = stay ? c0.z : (go ? l0.z : s0.z); c0.w = stay ? c0.w : (go ? l0.w : s0.w);
                        c1.x = stay ? c1.x : (go ? l1.x : s1.x); c1.y = stay ? c1.y : (go ? l1.y : s1.y);
                        c1.z = stay ? c1.z : (go ? l1.z : s1.z); c1.w = stay ? c1.w : (go ? l1.w : s1.w);
                    }
                } else {
                    if (DIAG) { dg_tripsL++; dg_lanesL += (unsigned long long)n_leaf; }
                    if (at_leaf) {
                        f3 a, e1, e2;
                        B.tri(ttb + tri, a, e1, e2);
                        float t, u, v;
                        bool ok = tri_t(too, tod, a, e1, e2, t_min, tbt, t, u, v);                          // :97
                        tbt = ok ? t : tbt; tbtri = ok ? tri : tbtri; tbu = ok ? u : tbu; tbv = ok ? v : tbv;
                        ti = ti + 1;
                        B.node(min(ti, last_node), c0, c1);
                    }
                }
                if (in_t && ti >= tend) {
                    // this mesh is done: StaticMesh::intersect_ray returns (geometry.rs:305-313)
                    if (tbtri >= 0) consider(best, tbt, S.meshes[tm].object_index, tbtri, tbu, tbv);
                    tm++;
                    if (enter_next_mesh(S, B, tm, P.o, P.d, t_min, t_max, too, tod, tinv, ti, tend, ttb)) {
                        tbt = t_max; tbtri = -1; tbu = tbv = 0.0f;
                        B.node(ti, c0, c1);
                    } else {
                        state = ST_A; pending = true;
                    }
                }
            }
        }
        if (DIAG) { unsigned long long dt = __builtin_amdgcn_s_memtime() - dg_t0; if (run_b) dg_cycB += dt; else dg_cycA += dt; }
    }

    float n = (float)C.spp;
    float* o3 = A.out + (size_t)out_idx * 3;
    if (in_image) { o3[0] = accum.x / n; o3[1] = accum.y / n; o3[2] = accum.z / n; }
    else { o3[0] = 0.0f; o3[1] = 0.0f; o3[2] = 0.0f; }
    if (SIG && A.sig) A.sig[out_idx] = in_image ? sigsum : 0u;
    if (DIAG && A.diag && lane == 0) {
        atomicAdd(&A.diag[0], dg_tripsA); atomicAdd(&A.diag[1], dg_lanesA);
        atomicAdd(&A.diag[2], dg_tripsI); atomicAdd(&A.diag[3], dg_lanesI);
        atomicAdd(&A.diag[4], dg_tripsL); atomicAdd(&A.diag[5], dg_lanesL);
        atomicAdd(&A.diag[6], dg_tripsB); atomicAdd(&A.diag[7], 1ull);
        atomicAdd(&A.diag[8], dg_cycA); atomicAdd(&A.diag[9], dg_cycB);
        atomicAdd(&A.diag[10], dg_cycShade); atomicAdd(&A.diag[11], dg_cycGen); atomicAdd(&A.diag[12], dg_cycList);
        atomicAdd(&A.diag[13], dg_slabs); atomicAdd(&A.diag[14], dg_roots);
    }
}

// ---------------------------------------------------------------- K1w: wavefront pipeline
// Same per-path arithmetic as K1 (same device functions, same order of evaluation, same RNG
// stream carried in the path state), different execution model: paths live in HBM, one kernel
// per phase, every launch runs with all lanes live on a COMPACTED list of paths.
//   wf_main   iteration 0: Camera::generate_rays; later: Scene::shade_ray for the hit found
//             in the previous iteration.  Then the object list and the mesh root tests for the
//             new ray.  Paths that end write their radiance to the sample buffer; the others
//             are appended to st_out (wave-aggregated atomicAdd: one atomic per wave), and the
//             ones whose ray entered a mesh root box also to the traversal queue.
//   wf_trav   persistent waves; a lane that has finished its ray pulls the next queue entry
//             (one atomicAdd per wave and refill round), so the BVH walk runs with full waves
//             until the queue drains; small register footprint -> high occupancy for the
//             latency-bound walk.
//   wf_reduce per pixel, samples added IN ORDER (the reference's summation order), then /n.
// index of (plane p, path slot k) in the path-state buffer: plain SoA planes.  (Groups of 64 slots with their
// planes adjacent — [k / 64][p][k % 64], one wave's accesses inside 6 KB — measured the same: 105.9 vs 106.0 ms.)
__device__ __forceinline__ size_t st_idx(int p, size_t k, uint32_t cap) { return (size_t)p * (size_t)cap + k; }
// The hit record lives in the plane-4 array as three packed sub-arrays: {t, obj} (8 bytes, every path), tri (4 bytes,
// class B only: a class-A path's hit is a plain Triangle / Plane, tri = -1 by construction) and the running
// signature (4 bytes, diagnostic): 72 bytes per class-A path and direction instead of 80.
struct Hit2 { float t; int obj; };
__device__ __forceinline__ Hit2* st_hit(float4* st, size_t k, uint32_t cap) { return (Hit2*)(st + st_idx(4, 0, cap)) + k; }
__device__ __forceinline__ int* st_tri(float4* st, size_t k, uint32_t cap) { return (int*)((Hit2*)(st + st_idx(4, 0, cap)) + cap) + k; }
__device__ __forceinline__ uint32_t* st_sig(float4* st, size_t k, uint32_t cap) { return (uint32_t*)((Hit2*)(st + st_idx(4, 0, cap)) + cap) + cap + k; }
__device__ __forceinline__ uint32_t wf_append(uint32_t* counter, bool want) {
    const unsigned long long m = __builtin_amdgcn_ballot_w64(want);
    uint32_t base = 0;
    if (m != 0ull) {
        const int leader = __ffsll((long long)m) - 1;
        const int lane = (int)(threadIdx.x & 63);
        if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(m));
        base = (uint32_t)__shfl((int)base, leader);
    }
    const unsigned long long lt = (1ull << (threadIdx.x & 63)) - 1ull;
    return base + (uint32_t)__popcll(m & lt);
}

__device__ __forceinline__ void wf_pixel_of(const WfArgs& A, uint32_t pix, uint32_t& px, uint32_t& py, bool& in_image) {
    const uint32_t slot = pix / kTilePixels, in = pix % kTilePixels;
    const uint32_t tile = slot * (uint32_t)A.R.world + (uint32_t)A.R.rank;
    px = 0; py = 0; in_image = false;
    if (tile < A.R.tiles_total) {
        px = (tile % A.R.tiles_x) * kTile + (in % kTile);
        py = (tile / A.R.tiles_x) * kTile + (in / kTile);
        in_image = (px < A.C.width) && (py < A.C.height);
    }
}

// MESH = false: the lean form for launches that cannot meet a mesh hit (iteration 0 and the class-A part of a later pass)
// RARE = false: the scene holds no Plane and no ConvexVolume (their loop and the free-flight code are compiled out)
// ITER0 = true: the camera-ray pass (Camera::generate_rays instead of a state load; always the lean form)
// TOP = true: the list's Triangles sit in a top-level tree (intersect_list<.., TOP>; scenes with long lists only)
template <bool LDS, bool SIG, bool GV, int MESH, bool RARE, bool ITER0, bool TOP = false>
__global__ __launch_bounds__(kBlock, (MESH == 2 ? PT_MAIN_WAVES : (MESH == 1 ? PT_MAIN_WAVES_NOTEX : PT_MAIN_WAVES_LEAN))) void wf_main(WfArgs A) {
    const DScene& S = A.S;
    const DCamera& C = A.C;
    Bvh<LDS> B;            // only the mesh ROOT nodes are read here
    bvh_bind(B, S, 0);     // roots come from global memory (scalar load: wave-uniform address)
    const float t_min = 0.001f, t_max = C.max_trace_dist;
    const uint32_t cap = A.cap;
    // `bid` = the block's number within the whole pass.  A pass after the first may be launched in two parts (class-A blocks
    // beside the walkers of the previous pass, class-B blocks behind them) and on a grid that is only an upper bound of the part
    // (the host did not wait for the previous pass' header): the device-side block table says where the part really ends.
    uint32_t bid = blockIdx.x;
    if (!ITER0) {
        const uint32_t blocks_a = A.in_blkpfx[kWfShards], blocks_all = A.in_blkpfx[2 * kWfShards];
        if (A.part == 2u) bid += blocks_a;
        if (bid >= (A.part == 1u ? blocks_a : blocks_all)) return;
    }
    const uint32_t out_shard = bid % (uint32_t)kWfShards;

#ifdef PT_WF_STAMPS
#define WF_STAMP(k) do { __builtin_amdgcn_s_waitcnt(0); stamp[k] = __builtin_amdgcn_s_memtime(); } while (0)
    unsigned long long stamp[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    WF_STAMP(0);
#else
#define WF_STAMP(k) do { } while (0)
#endif
    Path P; Best best;
    uint32_t pix = 0, sample = 0;
    int pre_mesh = -1, pre_attr = -1;
    bool alive;
    // camera rays: what can this block's tile reach at all?  (all 256 paths of a block belong to one tile:
    // 1024 | npix, 256 | 1024, so both words are wave-uniform and stay in SGPRs)
    unsigned long long list_mask = ~0ull, mesh_word = 0xffffffffull;
    if (ITER0 && A.tile_mask) {
        const uint32_t slot0 = ((bid * kBlock) % A.npix) / kTilePixels;
        const uint32_t tile0 = slot0 * (uint32_t)A.R.world + (uint32_t)A.R.rank;
        if (tile0 < A.R.tiles_total) { list_mask = A.tile_mask[tile0]; mesh_word = A.tile_mask[A.R.tiles_total + tile0]; }
    }
    if (ITER0) {
        // ---- Camera::generate_rays (tracing.rs:159-209) ----
        const uint32_t i = bid * kBlock + threadIdx.x;
        const bool valid = i < A.n_in;
        if (!SIG && (mesh_word >> 63)) {
            // nothing is reachable from this tile: every sample is the black background (tracing.rs:306).
            // (With signatures on the rays are still generated: the signature folds in the RNG state.)
            // wf_reduce knows the same masks and does not read these slots, so nothing is written at all
            return;
        }
        alive = valid;
        pix = valid ? (i % A.npix) : 0u;
        sample = A.s_base + (valid ? (i / A.npix) : 0u);
        uint32_t px, py; bool in_image;
        wf_pixel_of(A, pix, px, py, in_image);
        P.T = mk3(1.0f, 1.0f, 1.0f); P.L = mk3(0.0f, 0.0f, 0.0f); P.depth = 0; P.sig = 0;
        P.o = P.d = mk3(0.0f, 0.0f, 0.0f); P.rng.s0 = P.rng.s1 = 1u;
        if (valid && !in_image) {
            A.samp[(size_t)(sample - A.s_base) * A.npix + pix] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            alive = false;
        }
        if (alive) {
            rng_init(P.rng, A.seed_key, py * C.width + px, sample);
            generate_ray(C, px, py, sample, P.rng, P.o, P.d);
            if (C.path_depth == 0u) {                                     // tracing.rs:301 at level 0: the background, before any intersection
                if (SIG) P.sig = sig_end_depth(P.sig);
                A.samp[(size_t)(sample - A.s_base) * A.npix + pix] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(P.sig));
                alive = false;
            }
        }
    } else {
        // which (class, shard) range of st_in does this block read?  (binary search in the block prefix:
        // entries 0..S-1 = class A of each shard, S..2S-1 = class B)
        // The shards of a class hold nearly equal counts, so interpolate and correct by a step or two (a binary
        // search is nine DEPENDENT scalar loads, ~1.4k cycles at the head of every wave, before its state
        // loads can even be issued).  Only the starting point is approximate; the walk is exact.
        uint32_t lo;
        {
            const uint32_t b = bid;
            const uint32_t blocks_a = A.in_blkpfx[kWfShards], blocks_all = A.in_blkpfx[2 * kWfShards];
            const bool in_b = b >= blocks_a;
            const uint32_t base = in_b ? (uint32_t)kWfShards : 0u, first = in_b ? blocks_a : 0u;
            const uint32_t span = in_b ? blocks_all - blocks_a : blocks_a;
            uint32_t g = (uint32_t)((float)(b - first) * ((float)kWfShards / (float)(span ? span : 1u)));
            g = base + (g < (uint32_t)kWfShards ? g : (uint32_t)kWfShards - 1u);
            g = (uint32_t)__builtin_amdgcn_readfirstlane((int)g);
            while (A.in_blkpfx[g] > b) g--;                    // in_blkpfx[0] = 0 <= b
            while (A.in_blkpfx[g + 1] <= b) g++;               // in_blkpfx[2 S] = blocks_all > b
            lo = g;
        }
        WF_STAMP(1);
        const bool cls_b = lo >= (uint32_t)kWfShards;
        const uint32_t in_shard = cls_b ? lo - (uint32_t)kWfShards : lo;
        const uint32_t local = (bid - A.in_blkpfx[lo]) * kBlock + threadIdx.x;
        const bool valid = local < A.in_count[lo];
        alive = valid;
        const uint32_t lv = valid ? local : 0u;
        const size_t k = (size_t)in_shard * A.region + (cls_b ? (A.region - 1u - lv) : lv);     // class B grows from the back
        // ---- Scene::shade_ray, one level (tracing.rs:305-321), for the hit of the previous iteration ----
        const float4 q0 = A.st_in[st_idx(0, k, cap)], q1 = A.st_in[st_idx(1, k, cap)], q2 = A.st_in[st_idx(2, k, cap)];
        const float4 q3 = A.st_in[st_idx(3, k, cap)];
        const Hit2 hr = *st_hit(A.st_in, k, cap);
        const int hr_tri = cls_b ? *st_tri(A.st_in, k, cap) : -1;       // wave-uniform choice: class-A blocks never touch the array
        P.o = mk3(q0.x, q0.y, q0.z); P.d = mk3(q0.w, q1.x, q1.y); P.T = mk3(q1.z, q1.w, q2.x); P.L = mk3(q2.y, q2.z, q2.w);
        P.rng.s0 = __float_as_uint(q3.x); P.rng.s1 = __float_as_uint(q3.y);
        pix = __float_as_uint(q3.z);
        const uint32_t sd = __float_as_uint(q3.w);
        sample = sd & 0xffffu; P.depth = sd >> 16;
        P.sig = 0;
        if (SIG) P.sig = *st_sig(A.st_in, k, cap);
        best.t = hr.t; best.obj = hr.obj; best.tri = hr_tri; best.u = 0.0f; best.v = 0.0f;
        if (MESH && valid && best.tri >= 0) {       // plane 5 (barycentrics) exists only for mesh hits
            const float4 q5 = A.st_in[st_idx(5, k, cap)];
            best.u = q5.x; best.v = q5.y;
            pre_attr = __float_as_int(q5.z); pre_mesh = __float_as_int(q5.w);       // from the walker (resolve_hit), or -1
        }
        WF_STAMP(2);
    }

    WF_STAMP(3);
    // ---- shade the pending hit, intersect the new ray — and, while enough lanes of the wave can, do it again ----
    // A path whose new ray entered no mesh root has its next hit in registers already: shading it here, in the same
    // launch, saves that path one round trip through the HBM path state (72 B out + 72 B back) and one pass' worth of
    // block look-up, state loads and stores.  The wave votes: it goes on only while at least `fuse_min` of its lanes
    // would continue (the others — ended, or waiting for a mesh walk — sit the extra trips out), at most `fuse_max`
    // times.  Per path the order of evaluation is untouched: same shading, same intersection, same RNG stream.
    bool pending = (!ITER0) && alive;            // a hit record was loaded: Scene::shade_ray, one level (tracing.rs:305-321)
    bool need = (ITER0) && alive;               // a fresh ray needs Scene::intersect_ray
    bool first = true, enters = false;
    int tm = 0;
    uint32_t fuse_left = A.fuse_max;
    if (ITER0) { best.obj = -1; best.t = 0.0f; best.tri = -1; best.u = best.v = 0.0f; }
    uint32_t n_seg = 0;                          // wave-uniform: Scene::intersect_ray evaluations (path segments) of this wave
    for (;;) {
        if (pending) {
            bool end_path;
            if (best.obj < 0) {
                end_path = true;
                if (SIG) P.sig = sig_end_miss(P.sig, P.rng);
            } else {
                if (SIG) P.sig = sig_hit(P.sig, best.t, best.obj);
                Surf s;
                resolve_hit<MESH>(S, best, P.o, P.d, s, pre_mesh, pre_attr);
                pre_mesh = -1; pre_attr = -1;                                 // a hit found inside this launch is never a mesh hit
                if (first) WF_STAMP(7);
                P.L = mk3(P.L.x + P.T.x * s.emission.x, P.L.y + P.T.y * s.emission.y, P.L.z + P.T.z * s.emission.z);
                P.depth++;
                if (P.depth >= C.path_depth) {
                    end_path = true;
                    if (SIG) P.sig = sig_end_depth(P.sig);
                } else {
                    f3 nd, w;
                    scatter(S, s, P.d, P.rng, nd, w);
                    P.o = s.p; P.d = nd;
                    P.T = mk3(P.T.x * w.x, P.T.y * w.y, P.T.z * w.z);
                    end_path = false;
                }
            }
            if (end_path) {
                A.samp[(size_t)(sample - A.s_base) * A.npix + pix] = make_float4(P.L.x, P.L.y, P.L.z, __uint_as_float(P.sig));
                alive = false;
            }
            need = alive;
            pending = false;
            if (first) WF_STAMP(8);
        }
        // ---- Scene::intersect_ray for the new ray: object list, then the mesh roots ----
#if PT_SEG_COUNT
        n_seg += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(need));
#endif
        if (need) {
            tm = 0; enters = false;
            best.obj = -1; best.t = 0.0f; best.tri = -1; best.u = best.v = 0.0f;
            if (first && ITER0 && A.tile_mask) intersect_list_masked<GV, RARE>(S, list_mask, P.o, P.d, t_min, t_max, P.rng, best);
            else intersect_list<GV, RARE, TOP>(S, P.o, P.d, t_min, t_max, P.rng, best);
            if (first) WF_STAMP(9);
            f3 oo, od, inv; int ti, tend, ttb;
            enters = enter_next_mesh(S, B, tm, P.o, P.d, t_min, t_max, oo, od, inv, ti, tend, ttb, first ? (uint32_t)mesh_word : 0xffffffffu);
            if (first) WF_STAMP(10);
            // A ray that hits no object and enters no mesh ends its path at the next shade_ray level
            // (tracing.rs:306, background = 0).  Do that level now — same operations, same RNG state —
            // instead of streaming the path through HBM once more just to terminate it.
            if (best.obj < 0 && !enters) {
                if (SIG) P.sig = sig_end_miss(P.sig, P.rng);
                A.samp[(size_t)(sample - A.s_base) * A.npix + pix] = make_float4(P.L.x, P.L.y, P.L.z, __uint_as_float(P.sig));
                alive = false;
            }
            need = false;
        }
        first = false;
        const bool cont = alive && !enters;             // its hit is known: nothing left to wait for
        // (Round 4, measured negative: the BLOCK votes instead of the wave, and the continuing paths of its four waves are packed through LDS
        // into full waves — whole waves' worth go round again 64 lanes wide, everything else is appended on the spot.  Bit-identical, and
        // slower: cfg2 wf_main 49.8 -> 54.8 ms with one extra trip, 56.5 / 59.2 / 62.3 ms with two / three / five; the barriers tie four waves
        // to the slowest one, waves left without a path keep their slots until the block ends, and the appends sit inside the loop.  Sweep of
        // the wave's own threshold on the same day: no further trip at all 84.0 ms, 48 / 40 / 32 / 24 lanes 76.1 / 75.1 / 74.6 / 75.0 ms.
        // tools/experiments/r04_block_repack.diff.)
        if (fuse_left == 0u || (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(cont)) < A.fuse_min) break;
        fuse_left--;
        pending = cont;
    }
    WF_STAMP(4);
    // ---- compact the survivors into this block's shard region, class A from the front, class B
    //      from the back (one atomic per wave and class) ----
    const bool cls_a = alive && !enters && (best.tri == -1);       // next shade: plain Triangle / Plane hit
    const bool cls_b2 = alive && !cls_a;
    // The two wave-aggregated appends (class A, class B) go out as ONE vector atomic with lanes 0 and 1 addressing the two
    // counters: one round trip per wave instead of two in a row (a returning atomic takes ~3k cycles with every CU issuing
    // them; hipcc waits after each one).  There is no traversal queue: every ray that waits for a mesh walk is a class-B path
    // (class B = enters, or a Sphere / ConvexVolume hit pending), and the walkers go through the class-B blocks of the next
    // pass' table; a class-B path that entered no root fails the walker's root tests as it failed them here (same arithmetic).
    uint32_t ia, ib;
    {
        const unsigned long long ma = __builtin_amdgcn_ballot_w64(cls_a), mb = __builtin_amdgcn_ballot_w64(cls_b2);
        const uint32_t lane = threadIdx.x & 63;
        // ... and lane 2 adds the wave's segment count to its shard's statistics counter in the same instruction (wf_prefix sums the
        // shards into the pass header: mi_last_pipeline_counts[6], what bench.py's Msegments/s is computed from)
        uint32_t* ctr = lane == 0 ? &A.out_count[out_shard] : (lane == 1 ? &A.out_count[(uint32_t)kWfShards + out_shard] : &A.trav_count[out_shard]);
        const uint32_t add = lane == 2 ? n_seg : (uint32_t)__popcll(lane == 0 ? ma : mb);
        uint32_t base = 0;
        if (lane < (PT_SEG_COUNT ? 3 : 2) && add != 0) base = atomicAdd(ctr, add);
        const unsigned long long below = (1ull << lane) - 1ull;
        ia = (uint32_t)__shfl((int)base, 0) + (uint32_t)__popcll(ma & below);
        ib = (uint32_t)__shfl((int)base, 1) + (uint32_t)__popcll(mb & below);
    }
    WF_STAMP(5);
    const size_t pos = (size_t)out_shard * A.region + (cls_a ? ia : (A.region - 1u - ib));
    if (alive) {
        A.st_out[st_idx(0, pos, cap)] = make_float4(P.o.x, P.o.y, P.o.z, P.d.x);
        A.st_out[st_idx(1, pos, cap)] = make_float4(P.d.y, P.d.z, P.T.x, P.T.y);
        A.st_out[st_idx(2, pos, cap)] = make_float4(P.T.z, P.L.x, P.L.y, P.L.z);
        A.st_out[st_idx(3, pos, cap)] = make_float4(__uint_as_float(P.rng.s0), __uint_as_float(P.rng.s1), __uint_as_float(pix),
                                                      __uint_as_float((sample & 0xffffu) | (P.depth << 16)));
        { Hit2 hw; hw.t = best.t; hw.obj = best.obj; *st_hit(A.st_out, pos, cap) = hw; }
        if (!cls_a) *st_tri(A.st_out, pos, cap) = best.tri;
        if (SIG) *st_sig(A.st_out, pos, cap) = P.sig;
    }
#ifdef PT_WF_STAMPS
    WF_STAMP(6);       // waits for the state stores too
#if PT_WF_STAMPS >= 2
    // finer form, the mesh (class-B) launches only: [0..5] the six phases, [6] waves sampled, [7] whole life, then the FIRST trip of the
    // loop split up: [8] resolve_hit, [9] emission + scatter, [10] object list, [11] mesh roots
    if (A.diag && (PT_WF_STAMPS == 3 ? MESH == 0 : MESH != 0) && !ITER0 && (bid & 63u) == 0u && threadIdx.x == 0) {     // 3: the lean (class-A) launches instead
        unsigned long long* d = A.diag;
        for (int k = 0; k < 6; k++) if (stamp[k + 1] && stamp[k]) atomicAdd(&d[k], stamp[k + 1] - stamp[k]);
        atomicAdd(&d[6], 1ull);
        atomicAdd(&d[7], stamp[6] - stamp[0]);
        if (stamp[7]) atomicAdd(&d[8], stamp[7] - stamp[3]);
        if (stamp[7] && stamp[8]) atomicAdd(&d[9], stamp[8] - stamp[7]);
        if (stamp[8] && stamp[9]) atomicAdd(&d[10], stamp[9] - stamp[8]);
        if (stamp[9] && stamp[10]) atomicAdd(&d[11], stamp[10] - stamp[9]);
    }
#else
    if (A.diag && (bid & 63u) == 0u && threadIdx.x == 0) {
        unsigned long long* d = A.diag + (ITER0 ? 8 : 0);
        for (int k = 0; k < 6; k++) if (stamp[k + 1] && stamp[k]) atomicAdd(&d[k], stamp[k + 1] - stamp[k]);
        atomicAdd(&d[6], 1ull);
        atomicAdd(&d[7], stamp[6] - stamp[0]);
    }
#endif
#endif
}

// The walkers' work list.  Every ray that waits for a mesh walk is a class-B path of the pass' output, and wf_prefix has just
// tabulated those for the next wf_main (in_count / in_blkpfx, entries kWfShards .. 2 kWfShards - 1: per shard, whole blocks of
// 256).  So the walkers take SLOTS: slot v = lane v % 256 of class-B block v / 256; the block's table entry gives the shard and
// the position in st_out by arithmetic, and a slot behind the end of its shard's list is empty (one partial block per shard).
// No queue is written, read or searched.  Class B also holds the paths whose pending hit is a Sphere or a ConvexVolume and whose
// ray entered no mesh root (cfg2: 5.6 % of class B, HEAD: 12 %): a walker's refill rejects them by the root tests wf_main ran
// on them already — a wasted slot, no wasted instruction (the test runs for the whole refill group anyway).
// table entry of class-B block `block` (counted over all blocks of the pass): wave-uniform, scalar loads
__device__ __forceinline__ uint32_t wf_entry_of(const PT_CONST_AS uint32_t* blkpfx, uint32_t block) {
    uint32_t lo = (uint32_t)kWfShards, hi = 2u * (uint32_t)kWfShards;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (blkpfx[mid] <= block) lo = mid; else hi = mid; }
    return lo;
}
// The wave's cursor into the table: the entry of the block its chunk STARTS in, with that entry's first block and path count —
// wave-uniform, fetched by scalar loads once per chunk.  A chunk is at most 256 slots, so nearly every slot a wave hands out lies
// in that very block and its position is arithmetic on registers; only a chunk that straddles two blocks (an odd chunk length in a
// short work list) sends lanes through the table.
struct WfCur { uint32_t lo, block, first, count; };
__device__ __forceinline__ WfCur wf_cur_at(const WfArgs& A, uint32_t blocks_a, uint32_t slot) {     // slot: wave-uniform
    WfCur c;
    c.block = blocks_a + (slot >> 8);
    c.lo = wf_entry_of(A.in_blkpfx, c.block);
    c.first = A.in_blkpfx[c.lo]; c.count = A.in_count[c.lo];
    return c;
}
// per lane: slot v -> position in st_out; false = empty slot
__device__ __forceinline__ bool wf_slot(const WfArgs& A, uint32_t blocks_a, uint32_t v, const WfCur& cur, uint32_t& pos) {
    const uint32_t b = blocks_a + (v >> 8);
    uint32_t lo = cur.lo, first = cur.first, count = cur.count;
    if (b != cur.block) {                                    // rare: the chunk straddles a block boundary
        while (A.in_blkpfx[lo + 1] <= b) lo++;               // in_blkpfx[2 kWfShards] = all blocks > b
        first = A.in_blkpfx[lo]; count = A.in_count[lo];
    }
    const uint32_t local = (b - first) * (uint32_t)kBlock + (v & 255u);
    pos = (lo - (uint32_t)kWfShards) * A.region + (A.region - 1u - local);      // class B grows from the back of its shard's region
    return local < count;
}

// persistent BVH walker with per-lane dynamic refill from the sharded queues
// LDS: 0 = BVH in global memory, 2 = nodes (leaves carry a and e1 of their triangle) + the e2 vectors in LDS
template <int LDS> struct TravBvh { typedef Bvh<false> type; };
template <> struct TravBvh<2> { typedef BvhNodesLds type; };
__device__ __forceinline__ void bvh_bind(BvhNodesLds& B, const DScene& S, int) { B.nodes = k1_lds; B.tris = (cf4_ptr)S.tris; }

// BS: threads per block.  256 for the small-LDS modes; 1024 (one block per CU) when the node array needs most of a
// CU's 160 KB of LDS.  (Round 3: trees of that size — the drone's 3471 nodes = 140 KB with their leaves — now take wf_trav_i below,
// which keeps only the interior nodes in LDS and runs TWO 1024-thread blocks per CU; this form remains for trees whose interior
// nodes alone exceed 78 KB while the whole image still fits 156 KB, and as a cross-check: tests/test_gpu_walkers.py.)
// (Negative result, round 2: TWO RAYS PER LANE in the 1024-thread form — it runs 4 waves per SIMD whatever its register count,
// PMC shows 43 % of a wave's life parked at s_waitcnt, so two interleaved walks per lane looked free.  Built as a template
// parameter with the per-ray state slimmed to fit 128 VGPRs without spills (o, d re-read for a further mesh, the hit record
// merged in memory at the end of a walk, u and v recomputed by one more triangle test): bit-exact, but cfg4 wf_trav
// 270 -> 321 ms, HEAD 54 -> 67 ms; the slimmed state alone, one ray per lane: 310 / 63 ms and cfg2 28.4 -> 34 ms.  The
// walker is bound by VALU issue and LDS bank conflicts of its random 32-byte node reads, not by uncovered latency.)
// MULTI = false: this launch walks ONE mesh (bit 0 of trav_mask alone): the step to a further mesh is compiled out and the world-space
// ray is dead once it has been taken to object space (6 VGPRs of a 64-register budget)
template <int LDS, int BS, bool MULTI>
__global__ __launch_bounds__(BS, (BS == 256 ? PT_TRAV_WAVES : 4)) void wf_trav(WfArgs A) {
    const DScene& S = A.S;
    // Work distribution over the slots of the class-B blocks (wf_slot above):
    // wave w owns chunk w outright, later chunks come from one shared cursor.  The static first chunk
    // matters: 8192 waves all opening with an atomicAdd on ONE word cost ~240 us per launch (same-address
    // atomics retire at ~88 / us), 2.6 ms per frame and most of the multi-rank overhead; a queue that
    // the static chunks cover never touches the cursor at all.
    const uint32_t blocks_a = A.in_blkpfx[kWfShards];
    const uint32_t n_q = (A.in_blkpfx[2 * kWfShards] - blocks_a) * (uint32_t)kBlock;      // slots
    const uint32_t n_waves = gridDim.x * ((uint32_t)BS / 64u);
    // 256 rays per grab (128: the cursor's round trip shows, trav x3; 1024: long tails, +5 %).  A queue too short to give every
    // wave 256 rays is dealt out evenly instead, in whole waves' worth of 64: 200 k rays are then one walk's time on 3 k waves,
    // not four walks in a row on 800 (the late passes of a frame, and every pass of a small tile share)
    uint32_t chunk = 256u;
    if (n_q < n_waves * 256u) chunk = max(64u, ((n_q + n_waves - 1u) / n_waves + 63u) & ~63u);
    const bool shared_part = n_waves * chunk < n_q;                  // anything beyond the static chunks?
    if (blockIdx.x * ((uint32_t)BS / 64u) * chunk >= n_q) return;       // nothing for this block (then nothing is left over either)
    typename TravBvh<LDS>::type B;
    const int lds_nn = (int)A.R.lds_nodes * 2;                       // float4 slots of the staged nodes; the e2 vectors follow
    if (LDS != 0) {
        cf4_ptr gn = (cf4_ptr)S.nodes;
        cf4_ptr ge = (cf4_ptr)S.e2s;
        const int ne = (int)A.R.lds_tris;
        for (int k = threadIdx.x; k < lds_nn; k += BS) k1_lds[k] = gn[k];
        for (int k = threadIdx.x; k < ne; k += BS) k1_lds[lds_nn + k] = ge[k];
        __syncthreads();
    }
    bvh_bind(B, S, lds_nn);
    const float t_min = 0.001f, t_max = A.C.max_trace_dist;
    const uint32_t cap = A.cap;
    // clamp of the prefetch that follows the last node of a tree: inside the LDS image when there is one (it holds the head of
    // the node pool only: a clamp to the pool's last node would read behind the staged nodes), else inside the pool
    const int last_node = (LDS != 0 ? (int)A.R.lds_nodes : S.n_nodes) - 1;
    const uint32_t lane = threadIdx.x & 63;

    // wave-uniform work cursor: a chunk [wnext, wend) of the queue
    const uint32_t wave_id = blockIdx.x * ((uint32_t)BS / 64u) + (threadIdx.x >> 6);
    uint32_t wnext = wave_id * chunk, wend = min(wnext + chunk, n_q);
    bool drained = false;
    if (wnext >= n_q) { wnext = wend = 0; drained = true; }
    // shard of the chunk's first entry, kept wave-uniform (SGPRs): a lane then finds its own shard with a step or two
    // instead of an eight-deep chain of dependent vector loads at every refill
    WfCur wcur = { (uint32_t)kWfShards, 0u, 0u, 0u };
    if (!drained) wcur = wf_cur_at(A, blocks_a, (uint32_t)__builtin_amdgcn_readfirstlane((int)wnext));
    bool have = false;
    uint32_t pos = 0;
    f3 o = mk3(0.0f, 0.0f, 0.0f), d = o, too = o, tod = o, tinv = o;
    Best best; best.t = 0.0f; best.obj = -1; best.tri = -1; best.u = best.v = 0.0f;
    int tm = 0, ti = 0, tend = 0, ttb = 0, te2 = 0, tbtri = -1;
    float tbt = 0.0f, tbu = 0.0f, tbv = 0.0f;
    float4 c0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), c1 = c0;
    // Single-mesh form: mesh 0's record is read ONCE, here, instead of by dependent scalar loads in every refill (as wf_trav_i)
    float m0[16]; float4 r0 = c0, r1 = c0; int nb0 = 0, ne0 = 0, tb0 = 0, e20 = 0, obj0 = 0;
    if (!MULTI) {
        auto M = &S.meshes[0];
#pragma unroll
        for (int k = 0; k < 16; k++) m0[k] = M->inv_transform[k];
        nb0 = M->node_begin; ne0 = M->node_end; tb0 = M->tri_begin; e20 = M->e2_begin; obj0 = M->object_index;
        B.node(nb0, r0, r1);
    }
#ifdef PT_TRAV_DIAG
    // developer build: where do the lanes go?  (wave-uniform counters, summed into A.diag at exit)
    unsigned long long dg_trips = 0, dg_bsteps = 0, dg_blanes = 0, dg_lsteps = 0, dg_llanes = 0, dg_refills = 0, dg_rlanes = 0, dg_have = 0;
#endif

    while (true) {
        // ---- refill idle lanes ----
        unsigned long long need = __builtin_amdgcn_ballot_w64(!have);
        const uint32_t n_idle = (uint32_t)__popcll(need);
        if ((n_idle >= A.refill_min || n_idle == 64u) && !drained) {
            if (wnext == wend) {
                // shared cursor, counted from the end of the static chunks
                uint32_t base = n_q;
                if (shared_part) {
                    if (lane == 0) base = n_waves * chunk + atomicAdd(&A.trav_head[0], chunk);
                    base = (uint32_t)__shfl((int)base, 0);
                }
                if (base >= n_q) drained = true;
                else { wnext = base; wend = min(base + chunk, n_q); wcur = wf_cur_at(A, blocks_a, (uint32_t)__builtin_amdgcn_readfirstlane((int)base)); }
            }
            const uint32_t avail = wend - wnext;
            const uint32_t rank = (uint32_t)__popcll(need & ((1ull << lane) - 1ull));
            const bool take = !have && rank < avail && wf_slot(A, blocks_a, wnext + rank, wcur, pos);
            if (take) {
                const float4 q0 = A.st_out[st_idx(0, pos, cap)], q1 = A.st_out[st_idx(1, pos, cap)];
                const Hit2 hr = *st_hit(A.st_out, pos, cap);
                o = mk3(q0.x, q0.y, q0.z); d = mk3(q0.w, q1.x, q1.y);
                best.t = hr.t; best.obj = hr.obj; best.tri = -1; best.u = 0.0f; best.v = 0.0f;
                // the ray entered SOME mesh's root box in wf_main; find the first one of THIS launch's meshes it enters
                // (same root tests, same arithmetic; a ray that enters none of them has nothing to do here)
                tm = 0;
                bool in;
                if (!MULTI) {
                    too = xform_point(m0, o); tod = xform_vector(m0, d);                         // as enter_next_mesh: geometry.rs:304, :57, :95, :103
                    rcp3_exact(tod.x, tod.y, tod.z, tinv.x, tinv.y, tinv.z);
                    const bool leaf_root = __float_as_int(r1.w) >= 0;
                    in = leaf_root || slab(mk3(r0.x, r0.y, r0.z), mk3(r1.x, r1.y, r1.z), too, tinv, t_min, t_max);
                    ti = leaf_root ? nb0 : nb0 + 1; tend = ne0; ttb = tb0; te2 = e20;
                } else {
                    in = enter_next_mesh(S, B, tm, o, d, t_min, t_max, too, tod, tinv, ti, tend, ttb, A.trav_mask);
                    if (in) te2 = S.meshes[tm].e2_begin;
                }
                if (in) {
                    tbt = t_max; tbtri = -1; tbu = tbv = 0.0f;
                    B.node(ti, c0, c1);
                    have = true;
                }
            }
            wnext += min(avail, n_idle);
#ifdef PT_TRAV_DIAG
            dg_refills++; dg_rlanes += (unsigned long long)__popcll(__builtin_amdgcn_ballot_w64(take));
#endif
        }
        if (__builtin_amdgcn_ballot_w64(have) == 0ull) {
            if (drained) break;
            continue;
        }
#ifdef PT_TRAV_DIAG
        dg_trips++; dg_have += (unsigned long long)__popcll(__builtin_amdgcn_ballot_w64(have));
#endif

        // ---- one voted step: a burst of interior nodes or one leaf (same code as the voted K1) ----
        // (Negative result, round 2: DEFERRING the leaves — a lane notes up to three leaves and walks on as if their
        // triangle tests had failed, the noted tests run together later, and a test that passes throws the speculation
        // away and resumes at that leaf's successor with the new bound — is exact (all 153 GPU tests green) but slower:
        // wf_trav 35.6 -> 43.5 ms on cfg2, 81 -> 96 ms on cfg4: 74 instead of 64 VGPRs (6 instead of 8 waves per SIMD),
        // 20 % more instructions per node step, and every hit replays part of the walk.)
        const int tri = __float_as_int(c1.w);
        const bool walking = have & (!MULTI | (ti < tend));             // MULTI: a lane that has finished a mesh may be waiting for company (below)
        const bool at_leaf = walking & (tri >= 0), at_inner = walking & (tri < 0);
        const int n_leaf = __popcll(__builtin_amdgcn_ballot_w64(at_leaf));
        const int n_inner = __popcll(__builtin_amdgcn_ballot_w64(at_inner));
        if (n_inner >= n_leaf * PT_TRAV_LEAF_W) {
            // burst of interior steps, no vote in between.  At 6 waves/SIMD the LDS latency of the
            // dependent node fetch is covered by the other waves, and this form is 20 instructions
            // per step shorter than prefetching both successors and selecting (used in K1).
#pragma unroll
            for (int j = 0; j < PT_TRAV_BURST; j++) {
                const bool act = have & (ti < tend) & (__float_as_int(c1.w) < 0);
#ifdef PT_TRAV_DIAG
                dg_bsteps++; dg_blanes += (unsigned long long)__popcll(__builtin_amdgcn_ballot_w64(act));
#endif
                const bool hit = slab(mk3(c0.x, c0.y, c0.z), mk3(c1.x, c1.y, c1.z), too, tinv, t_min, tbt);
                const int nxt = hit ? ti + 1 : __float_as_int(c0.w);
                ti = act ? nxt : ti;
                if (act) B.node(min(ti, last_node), c0, c1);
                if (PT_TRAV_DYN > 0 && j + 1 < PT_TRAV_BURST &&
                    __popcll(__builtin_amdgcn_ballot_w64(have & (ti < tend) & (__float_as_int(c1.w) < 0))) < PT_TRAV_DYN) break;
            }
        } else {
            // the leaf node just fetched IS the triangle's a and e1 (pt_device.h DScene.e2s); e2 is one more 16-byte read
            // from the LDS image (or from the e2 pool when the tree is walked from global memory)
            for (int k = 0; k < (PT_TRAV_LEAF2 > 0 ? 2 : 1); k++) {
                const int ltri = __float_as_int(c1.w);
                const bool lf = have & (ti < tend) & (ltri >= 0);
                if (k > 0 && __popcll(__builtin_amdgcn_ballot_w64(lf)) < PT_TRAV_LEAF2) break;
                if (lf) {
                    float4 ev;
                    if (LDS != 0) ev = k1_lds[lds_nn + te2 + ltri]; else ev = ((cf4_ptr)S.e2s)[te2 + ltri];
                    float t, u, v;
                    bool ok = tri_t<BS == 256>(too, tod, mk3(c0.x, c0.y, c0.z), mk3(c1.x, c1.y, c1.z), mk3(ev.x, ev.y, ev.z), t_min, tbt, t, u, v);
                    tbt = ok ? t : tbt; tbtri = ok ? ltri : tbtri; tbu = ok ? u : tbu; tbv = ok ? v : tbv;
                    ti = ti + 1;
                    B.node(min(ti, last_node), c0, c1);
                }
            }
        }
#ifdef PT_TRAV_DIAG
        if (n_inner < n_leaf) { dg_lsteps++; dg_llanes += (unsigned long long)n_leaf; }
#endif
        // MULTI: the root tests of a further mesh wait until PT_TRAV_PEND lanes need them, or nobody is walking (see wf_trav_i)
        const bool ended = have && ti >= tend;
        if (MULTI ? (__popcll(__builtin_amdgcn_ballot_w64(ended)) >= PT_TRAV_PEND || __builtin_amdgcn_ballot_w64(have & (ti < tend)) == 0ull) : true) {
            if (ended) {
                if (tbtri >= 0) consider(best, tbt, MULTI ? S.meshes[tm].object_index : obj0, tbtri, tbu, tbv);
                tm++;
                if (MULTI && enter_next_mesh(S, B, tm, o, d, t_min, t_max, too, tod, tinv, ti, tend, ttb, A.trav_mask)) {
                    tbt = t_max; tbtri = -1; tbu = tbv = 0.0f;
                    te2 = S.meshes[tm].e2_begin;
                    B.node(ti, c0, c1);
                } else {
                    // StaticMesh results merged: hand the closest hit back to the path.  70 % of the rays that enter
                    // a root box hit no triangle closer than the list's hit: their record is already right.
                    if (best.tri >= 0) {
                        Hit2 hw; hw.t = best.t; hw.obj = best.obj;
                        *st_hit(A.st_out, pos, cap) = hw;                 // the signature has its own words: no read-modify-write
                        *st_tri(A.st_out, pos, cap) = best.tri;
                        // the spare words: {index of the triangle's attribute record, mesh index} for resolve_hit — known here without a
                        // look-up only in the single-mesh form (mesh 0, its first triangle in a register); -1 = wf_main follows the chain
                        A.st_out[st_idx(5, pos, cap)] = make_float4(best.u, best.v, __int_as_float((!MULTI && PT_PRE_ATTR) ? tb0 + best.tri : -1),
                                                                     __int_as_float((!MULTI && PT_PRE_ATTR) ? 0 : -1));
                    }
                    have = false;
                }
            }
        }
    }
#ifdef PT_TRAV_DIAG
    if (A.diag && lane == 0) {
        atomicAdd(&A.diag[0], dg_trips); atomicAdd(&A.diag[1], dg_have); atomicAdd(&A.diag[2], dg_bsteps); atomicAdd(&A.diag[3], dg_blanes);
        atomicAdd(&A.diag[4], dg_lsteps); atomicAdd(&A.diag[5], dg_llanes); atomicAdd(&A.diag[6], dg_refills); atomicAdd(&A.diag[7], dg_rlanes);
    }
#endif
}

// ---------------------------------------------------------------- wf_trav_i: the walker for trees of 64 .. 150 KB
// Same walk, same arithmetic, same results as wf_trav; different storage.  A tree whose whole image (nodes with their leaves'
// triangles) exceeds 64 KB forces wf_trav into ONE 1024-thread block per CU: 4 waves per SIMD, which cannot cover the latency of
// the conflicted LDS node reads (16 waves queue behind one LDS pipe) AND the dependent VALU chain of the slab test — PMC showed the
// LDS-busy time and the VALU-issue time of that kernel ADDING up to its duration instead of overlapping.  Here only the INTERIOR
// nodes live in LDS (pt_device.h DScene.inodes: half the bytes, every link explicit), so two such blocks fit a CU — 8 waves per
// SIMD — and a leaf is three 16-byte reads from the leaf pool in global memory (L2-resident: 11 of the ~96 steps of a drone ray).
// LEAF_LDS = true (round 4): the leaf records staged in LDS as well, behind the interior ones — for trees whose WHOLE split image fits 64 KB
// (the teapot: 7.6 + 11.5 KB, eight 256-thread blocks per CU as wf_trav<2, 256>).  Same storage cost as wf_trav's image, but the explicit
// links make an interior step four VALU instructions shorter (no `ti < tend`, no `ti + 1`, no clamp of the prefetch behind the last node).
// PAIR = true (round 4; needs LEAF_LDS): the interior records are re-laid while they are staged, 56 bytes each —
//   {bmin.x, bmax.x, bmax.x, bmin.x}{.y ...}{.z ...}{link on a miss, link on a hit}, links to interior records as LDS byte offsets —
// so that ONE 8-byte read at `record + (1/d < 0 ? 8 : 0)` per axis delivers {near plane, far plane} already in the order geometry.rs:60-62's
// swap produces: the six v_cndmask per box test (sw ? t1 : t0) disappear.  Same subtractions and products on the same operands; what changes
// is which register they arrive in.  On gfx950 a v_cndmask / v_min / v_max / shift issues in 4 cycles, an f32 add / mul or an integer add in 2
// (tools/microbench/valu_rates.hip), so the step's issue time goes from 12 x 2 + 15 x 4 to 15 x 2 + 7 x 4 cycles.  The 56-byte stride
// (14 words: gcd with 64 banks = 2) spreads the 8-byte reads of 32 lanes over all bank pairs.
template <int BS, bool MULTI, bool LEAF_LDS = false, bool PAIR = false>
__global__ __launch_bounds__(BS, 8) void wf_trav_i(WfArgs A) {
    static_assert(!PAIR || LEAF_LDS, "the paired layout keeps the leaf records in LDS");
    const DScene& S = A.S;
    const uint32_t blocks_a = A.in_blkpfx[kWfShards];
    const uint32_t n_q = (A.in_blkpfx[2 * kWfShards] - blocks_a) * (uint32_t)kBlock;      // slots of the class-B blocks (wf_slot)
    const uint32_t n_waves = gridDim.x * ((uint32_t)BS / 64u);
    uint32_t chunk = 256u;                                           // as wf_trav
    if (n_q < n_waves * 256u) chunk = max(64u, ((n_q + n_waves - 1u) / n_waves + 63u) & ~63u);
    const bool shared_part = n_waves * chunk < n_q;
    if (blockIdx.x * ((uint32_t)BS / 64u) * chunk >= n_q) return;
    // PAIR: byte offset of the leaf records behind the 56-byte interior records (16-byte aligned)
    const int pair_leaf_base = (((int)A.R.lds_nodes * kPairStride + 15) & ~15);
    {
        cf4_ptr gi = (cf4_ptr)S.inodes;
        const int nn = (int)A.R.lds_nodes * 2;
        if (!PAIR) {
            // links to interior records as ABSOLUTE LDS addresses (as in the paired layout below): a step reads `link` and `link + 16` directly
            const int lds0 = (int)(uint32_t)(uintptr_t)(PT_LDS_AS char*)k1_lds;
            for (int k = threadIdx.x; k < nn; k += BS) {
                float4 v = gi[k];
                const int l = __float_as_int(v.w);
                v.w = __int_as_float(l >= 0 ? lds0 + l * 32 : l);
                k1_lds[k] = v;
            }
        } else {
            float* w = (float*)k1_lds;
            // links to interior records become ABSOLUTE LDS addresses (the image's own LDS offset added here, once): a step then reads at
            // `link + axis offset` through an address-space-3 pointer made from the integer, with no base to add (one VALU instruction per step)
            const int lds0 = (int)(uint32_t)(uintptr_t)(PT_LDS_AS char*)k1_lds;
            auto link = [lds0](float f) { const int l = __float_as_int(f); return __int_as_float(l >= 0 ? lds0 + l * kPairStride : l); };
            for (int k = threadIdx.x; k < (int)A.R.lds_nodes; k += BS) {
                const float4 a = gi[2 * k], b = gi[2 * k + 1];
                float* r = w + k * (kPairStride / 4);
                r[0] = a.x; r[1] = b.x; r[2] = b.x; r[3] = a.x;
                r[4] = a.y; r[5] = b.y; r[6] = b.y; r[7] = a.y;
                r[8] = a.z; r[9] = b.z; r[10] = b.z; r[11] = a.z;
                r[12] = link(a.w); r[13] = link(b.w);
            }
        }
        if (LEAF_LDS) {
            cf4_ptr gl = (cf4_ptr)S.lnodes;
            const int nl = (int)A.R.lds_tris * 3;                    // float4 slots of the staged leaf records
            if (!PAIR) {
                const int lds0 = (int)(uint32_t)(uintptr_t)(PT_LDS_AS char*)k1_lds;
                for (int k = threadIdx.x; k < nl; k += BS) {
                    float4 v = gl[k];
                    if (k % 3 == 0) { const int l = __float_as_int(v.w); v.w = __int_as_float(l >= 0 ? lds0 + l * 32 : l); }   // the link to the next node
                    k1_lds[nn + k] = v;
                }
            } else {
                float4* ll = (float4*)((char*)k1_lds + pair_leaf_base);
                const int lds0 = (int)(uint32_t)(uintptr_t)(PT_LDS_AS char*)k1_lds;
                for (int k = threadIdx.x; k < nl; k += BS) {
                    float4 v = gl[k];
                    if (k % 3 == 0) { const int l = __float_as_int(v.w); v.w = __int_as_float(l >= 0 ? lds0 + l * kPairStride : l); }   // the link to the next node
                    ll[k] = v;
                }
            }
        }
        // (Tried and dropped: links turned into LDS byte offsets while staging, so that a node fetch needs no address arithmetic — the shift
        // was folded into the address add already: walker 23.8 -> 23.8 ms on cfg2, 190.0 -> 190.9 ms on cfg4 with its leaves' links scaled on the fly.)
        __syncthreads();
    }
    const float4* IN = k1_lds;
    const int lds0_abs = (int)(uint32_t)(uintptr_t)(PT_LDS_AS char*)k1_lds;
    const float4* LL = PAIR ? (const float4*)((const char*)k1_lds + pair_leaf_base) : k1_lds + (int)A.R.lds_nodes * 2;   // LEAF_LDS: the leaf records
    const char* PB = (const char*)k1_lds;                            // PAIR: interior records by byte offset
    // (Round 4, measured negative: the LDS image in two planes — all first halves, then all second halves, so that a 16-byte read of
    // node i starts at bank 4 i mod 64 and reaches all sixteen bank quads where the 32-byte stride reaches eight.  PMC, same box:
    // SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.50 -> 0.39 here and 0.52 -> 0.41 in wf_trav, LDS-busy cycles -19 %, but the second plane's
    // address is one more VALU per node fetch (+3.7 % instructions) and the kernel got SLOWER by what the instructions cost: cfg4 184.6 ->
    // 190.1 ms, cfg2 25.05 -> 25.90 ms, HEAD 34.8 -> 35.8 ms.  The walkers are bound by instruction issue; the conflicts are hidden
    // behind it.  tools/experiments/r04_lds_planar.diff)
    typedef float vf4 __attribute__((ext_vector_type(4)));
    auto lds4 = [](int addr) { const vf4 v = *(const PT_LDS_AS vf4*)(uintptr_t)(uint32_t)addr; return make_float4(v.x, v.y, v.z, v.w); };
#define IN0(a) lds4(a)               // `a`: the record's absolute LDS address (what the staged links hold)
#define IN1(a) lds4((a) + 16)
#define IN1_AT(i) IN[2 * (i) + 1]    // by record index: the root, whose id comes from the mesh record
    cf4_ptr LN = (cf4_ptr)S.lnodes;
    Bvh<false> B;                                                    // mesh ROOT boxes come from the ordinary node pool (enter_next_mesh)
    bvh_bind(B, S, 0);
    const float t_min = 0.001f, t_max = A.C.max_trace_dist;
    const uint32_t cap = A.cap;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave_id = blockIdx.x * ((uint32_t)BS / 64u) + (threadIdx.x >> 6);
    uint32_t wnext = wave_id * chunk, wend = min(wnext + chunk, n_q);
    bool drained = false;
    if (wnext >= n_q) { wnext = wend = 0; drained = true; }
    WfCur wcur = { (uint32_t)kWfShards, 0u, 0u, 0u };
    if (!drained) wcur = wf_cur_at(A, blocks_a, (uint32_t)__builtin_amdgcn_readfirstlane((int)wnext));
    bool have = false;
    uint32_t pos = 0;
    f3 o = mk3(0.0f, 0.0f, 0.0f), d = o, too = o, tod = o, tinv = o;
    Best best; best.t = 0.0f; best.obj = -1; best.tri = -1; best.u = best.v = 0.0f;
    int tm = 0, id = kIdEnd, tbtri = -1;                             // id: the node the lane stands on (>= 0 interior: c0, c1 hold it)
    float tbt = 0.0f, tbu = 0.0f, tbv = 0.0f;
    float4 c0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), c1 = c0;
    // PAIR: the record the lane stands on as {near, far} per axis and its two links; sx / sy / sz = the axis' byte offset inside a record, + 8 where 1/d < 0
    float2 px = make_float2(0.0f, 0.0f), py = px, pz = px; int2 plk = make_int2(0, 0);
    int sx = 0, sy = 16, sz = 32;
    auto pair_load = [&]() {
        typedef float vf2 __attribute__((ext_vector_type(2)));
        typedef int vi2 __attribute__((ext_vector_type(2)));
        const vf2 ax = *(const PT_LDS_AS vf2*)(uintptr_t)(uint32_t)(id + sx), ay = *(const PT_LDS_AS vf2*)(uintptr_t)(uint32_t)(id + sy),
                  az = *(const PT_LDS_AS vf2*)(uintptr_t)(uint32_t)(id + sz);
        const vi2 al = *(const PT_LDS_AS vi2*)(uintptr_t)(uint32_t)(id + 48);
        px = make_float2(ax.x, ax.y); py = make_float2(ay.x, ay.y); pz = make_float2(az.x, az.y); plk = make_int2(al.x, al.y);
    };
    auto pair_signs = [&]() { sx = tinv.x < 0.0f ? 8 : 0; sy = tinv.y < 0.0f ? 24 : 16; sz = tinv.z < 0.0f ? 40 : 32; };   // geometry.rs:60 `if inv_d < 0.0`

    // Single-mesh form: mesh 0's record — inverse transform, root box, first node, object index — is read ONCE, here, instead of by
    // four dependent scalar loads in every refill (same values, same operations on them afterwards)
    float m0[16]; float4 r0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), r1 = r0; int root0 = 0, obj0 = 0, tb0 = 0;
    if (!MULTI) {
        auto M = &S.meshes[0];
#pragma unroll
        for (int k = 0; k < 16; k++) m0[k] = M->inv_transform[k];
        B.node(M->node_begin, r0, r1);
        root0 = M->i_root; obj0 = M->object_index; tb0 = M->tri_begin;
        if (root0 >= 0) root0 = PAIR ? *(const int*)(PB + (root0 * kPairStride + 52)) : __float_as_int(IN1_AT(root0).w);   // the node a ray stands on after a passed root test
    }
    // the ray has passed (or skipped, for a root that is a leaf) mesh tm's root test: stand on the first node to visit
    bool texo = false;               // this lane's ray has a zero or non-finite 1/d: its wave takes the reference form of the box test
    auto start_mesh = [&]() {
        if (PAIR) pair_signs();
        if (!PAIR && !MULTI && PT_TRAVI_MED3) texo = !slab_plain_ray(tinv);
        if (!MULTI) { id = root0; if (id >= 0) { if (PAIR) pair_load(); else { c0 = IN0(id); c1 = IN1(id); } } tbt = t_max; tbtri = -1; tbu = tbv = 0.0f; return; }
        const int root = S.meshes[tm].i_root;
        id = root;
        if (root >= 0) { id = PAIR ? *(const int*)(PB + (root * kPairStride + 52)) : __float_as_int(IN1_AT(root).w); }   // root box passed with bound t_max: its left child
        if (id >= 0) { if (PAIR) pair_load(); else { c0 = IN0(id); c1 = IN1(id); } }
        tbt = t_max; tbtri = -1; tbu = tbv = 0.0f;
    };

#ifdef PT_TRAVI_STAMPS      // developer build (MI_RT_WF_STAMPS=1 tools/probe.py prints A.diag): [8] wave life, [9] cycles inside refills (s_memtime), [10] refills,
                            // [11] rays they brought, [12] trips, [13] lanes holding a ray summed over the trips, [14] waves
    unsigned long long ts_life = __builtin_amdgcn_s_memtime(), ts_refill = 0, n_refill = 0, n_taken = 0, n_trips = 0, n_have = 0;
#endif
    while (true) {
        // ---- refill idle lanes (as wf_trav) ----
        unsigned long long need = __builtin_amdgcn_ballot_w64(!have);
        const uint32_t n_idle = (uint32_t)__popcll(need);
#ifdef PT_TRAVI_STAMPS
        __builtin_amdgcn_s_waitcnt(0); const unsigned long long ts_r0 = __builtin_amdgcn_s_memtime();
        const bool did_refill = (n_idle >= A.refill_min || n_idle == 64u) && !drained;
#endif
        if ((n_idle >= A.refill_min || n_idle == 64u) && !drained) {
            if (wnext == wend) {
                uint32_t base = n_q;
                if (shared_part) {
                    if (lane == 0) base = n_waves * chunk + atomicAdd(&A.trav_head[0], chunk);
                    base = (uint32_t)__shfl((int)base, 0);
                }
                if (base >= n_q) drained = true;
                else { wnext = base; wend = min(base + chunk, n_q); wcur = wf_cur_at(A, blocks_a, (uint32_t)__builtin_amdgcn_readfirstlane((int)base)); }
            }
            const uint32_t avail = wend - wnext;
            const uint32_t rank = (uint32_t)__popcll(need & ((1ull << lane) - 1ull));
            const bool take = !have && rank < avail && wf_slot(A, blocks_a, wnext + rank, wcur, pos);
            if (take) {
                const float4 q0 = A.st_out[st_idx(0, pos, cap)], q1 = A.st_out[st_idx(1, pos, cap)];
                const Hit2 hr = *st_hit(A.st_out, pos, cap);
                o = mk3(q0.x, q0.y, q0.z); d = mk3(q0.w, q1.x, q1.y);
                best.t = hr.t; best.obj = hr.obj; best.tri = -1; best.u = 0.0f; best.v = 0.0f;
                tm = 0;
                if (!MULTI) {
                    too = xform_point(m0, o); tod = xform_vector(m0, d);                         // as enter_next_mesh: geometry.rs:304, :57, :95, :103
                    rcp3_exact(tod.x, tod.y, tod.z, tinv.x, tinv.y, tinv.z);
                    if (__float_as_int(r1.w) >= 0 || slab(mk3(r0.x, r0.y, r0.z), mk3(r1.x, r1.y, r1.z), too, tinv, t_min, t_max)) { start_mesh(); have = true; }
                } else {
                    int ti, tend, ttb;
                    if (enter_next_mesh(S, B, tm, o, d, t_min, t_max, too, tod, tinv, ti, tend, ttb, A.trav_mask)) { start_mesh(); have = true; }
                }
            }
            wnext += min(avail, n_idle);
        }
#ifdef PT_TRAVI_STAMPS
        if (did_refill) { __builtin_amdgcn_s_waitcnt(0); ts_refill += __builtin_amdgcn_s_memtime() - ts_r0; n_refill++; n_taken += (unsigned long long)__popcll(__builtin_amdgcn_ballot_w64(have)) - (64ull - n_idle); }
#endif
        if (__builtin_amdgcn_ballot_w64(have) == 0ull) {
            if (drained) break;
            continue;
        }
#ifdef PT_TRAVI_STAMPS
        n_trips++; n_have += (unsigned long long)__popcll(__builtin_amdgcn_ballot_w64(have));
#endif
        // ---- one voted step: a burst of interior nodes, or the leaves the lanes stand on ----
        const bool at_leaf = have & (id < 0) & (id != kIdEnd), at_inner = have & (id >= 0);
        const int n_leaf = __popcll(__builtin_amdgcn_ballot_w64(at_leaf));
        const int n_inner = __popcll(__builtin_amdgcn_ballot_w64(at_inner));
        if (n_inner >= n_leaf * PT_TRAVI_LEAF_W) {
            if (PAIR) {
                // lanes leave the burst when they reach a leaf or the end and stay out: the steps nest, no per-step select of `id`
                bool on = have & (id >= 0);
#pragma unroll
                for (int j = 0; j < PT_TRAVL_BURST; j++) {
                    if (on) {
                        const bool hit = slab_nf(px, py, pz, too, tinv, t_min, tbt);                               // geometry.rs:103
                        id = hit ? plk.y : plk.x;
                        on = id >= 0;
                        if (on) pair_load();
                    }
                }
            } else if (PT_TRAVI_MED3 && !MULTI && __builtin_amdgcn_ballot_w64(have & texo) == 0ull) {
                // every ray of the wave has finite non-zero 1/d (practically always): the clamped form of the box test, 6 v_med3 for 6 v_cndmask + 4 min / max
#pragma unroll
                for (int j = 0; j < (LEAF_LDS ? PT_TRAVL_BURST : PT_TRAVI_BURST); j++) {
                    const bool act = have & (id >= 0);
                    const bool hit = slab_med3(mk3(c0.x, c0.y, c0.z), mk3(c1.x, c1.y, c1.z), too, tinv, t_min, tbt);
                    const int nxt = hit ? __float_as_int(c1.w) : __float_as_int(c0.w);
                    id = act ? nxt : id;
                    if (act & (id >= 0)) { c0 = IN0(id); c1 = IN1(id); }
                }
            } else if (PT_TRAVI_MED3 && !MULTI) {
                // a ray with a zero or infinite 1/d is in the wave: the reference form, as a ROLLED loop (rare; keeps the kernel small)
#pragma nounroll
                for (int j = 0; j < (LEAF_LDS ? PT_TRAVL_BURST : PT_TRAVI_BURST); j++) {
                    const bool act = have & (id >= 0);
                    const bool hit = slab(mk3(c0.x, c0.y, c0.z), mk3(c1.x, c1.y, c1.z), too, tinv, t_min, tbt);      // geometry.rs:103
                    const int nxt = hit ? __float_as_int(c1.w) : __float_as_int(c0.w);
                    id = act ? nxt : id;
                    if (act & (id >= 0)) { c0 = IN0(id); c1 = IN1(id); }
                }
            } else {
                // (the nested form of the paired layout's burst, tried on the 32-byte records too: cfg4 walker 186.1 -> 186.7 ms, HEAD unchanged)
#pragma unroll
                for (int j = 0; j < (LEAF_LDS ? PT_TRAVL_BURST : PT_TRAVI_BURST); j++) {
                    const bool act = have & (id >= 0);
                    const bool hit = slab(mk3(c0.x, c0.y, c0.z), mk3(c1.x, c1.y, c1.z), too, tinv, t_min, tbt);      // geometry.rs:103
                    const int nxt = hit ? __float_as_int(c1.w) : __float_as_int(c0.w);
                    id = act ? nxt : id;
                    if (act & (id >= 0)) { c0 = IN0(id); c1 = IN1(id); }
                }
            }
        } else {
            for (int k = 0; k < ((LEAF_LDS ? PT_TRAVL_LEAF2 : PT_TRAVI_LEAF2) > 0 ? 2 : 1); k++) {
                const bool lf = have & (id < 0) & (id != kIdEnd);
                if (k > 0 && __popcll(__builtin_amdgcn_ballot_w64(lf)) < (LEAF_LDS ? PT_TRAVL_LEAF2 : PT_TRAVI_LEAF2)) break;
                if (lf) {
                    const int li = ~id;
                    float4 l0, l1, l2;
                    if (LEAF_LDS) { l0 = LL[3 * li]; l1 = LL[3 * li + 1]; l2 = LL[3 * li + 2]; }
                    else { l0 = LN[3 * li]; l1 = LN[3 * li + 1]; l2 = LN[3 * li + 2]; }
                    float t, u, v;
                    const bool ok = tri_t(too, tod, mk3(l0.x, l0.y, l0.z), mk3(l1.x, l1.y, l1.z), mk3(l2.x, l2.y, l2.z), t_min, tbt, t, u, v);   // :97
                    const int ltri = __float_as_int(l1.w);
                    tbt = ok ? t : tbt; tbtri = ok ? ltri : tbtri; tbu = ok ? u : tbu; tbv = ok ? v : tbv;
                    id = __float_as_int(l0.w);
                    if (!PAIR && !LEAF_LDS && id >= 0) id = lds0_abs + id * 32;            // a leaf record read from global memory carries a node id
                    if (id >= 0) { if (PAIR) pair_load(); else { c0 = IN0(id); c1 = IN1(id); } }
                }
            }
        }
        // this mesh is done: StaticMesh::intersect_ray returns (geometry.rs:305-313).  MULTI: the root tests of a further mesh are ~120
        // instructions; run on the spot they serve the one or two lanes that finish in this very step, every other step.  So a lane
        // that finishes a mesh only notes it (id stays kIdEnd, the lane sits out the steps), and the entry tests run once
        // PT_TRAV_PEND lanes wait for one — or nobody is walking any more.
        const bool ended = have && id == kIdEnd;
        if (MULTI ? (__popcll(__builtin_amdgcn_ballot_w64(ended)) >= PT_TRAV_PEND || __builtin_amdgcn_ballot_w64(have & (id != kIdEnd)) == 0ull) : true) {
            if (ended) {
                if (tbtri >= 0) consider(best, tbt, MULTI ? S.meshes[tm].object_index : obj0, tbtri, tbu, tbv);
                tm++;
                int ti, tend, ttb;
                if (MULTI && enter_next_mesh(S, B, tm, o, d, t_min, t_max, too, tod, tinv, ti, tend, ttb, A.trav_mask)) start_mesh();
                else {
                    if (best.tri >= 0) {
                        Hit2 hw; hw.t = best.t; hw.obj = best.obj;
                        *st_hit(A.st_out, pos, cap) = hw;
                        *st_tri(A.st_out, pos, cap) = best.tri;
                        A.st_out[st_idx(5, pos, cap)] = make_float4(best.u, best.v, __int_as_float((!MULTI && PT_PRE_ATTR) ? tb0 + best.tri : -1),
                                                                     __int_as_float((!MULTI && PT_PRE_ATTR) ? 0 : -1));      // as wf_trav
                    }
                    have = false;
                }
            }
        }
    }
#ifdef PT_TRAVI_STAMPS
    if (A.diag && lane == 0) {
        atomicAdd(&A.diag[8], __builtin_amdgcn_s_memtime() - ts_life); atomicAdd(&A.diag[9], ts_refill); atomicAdd(&A.diag[10], n_refill);
        atomicAdd(&A.diag[11], n_taken); atomicAdd(&A.diag[12], n_trips); atomicAdd(&A.diag[13], n_have); atomicAdd(&A.diag[14], 1ull);
    }
#endif
}

#undef IN0
#undef IN1
#undef IN1_AT

// ---------------------------------------------------------------- exact two-stage mesh traversal
// For meshes whose file order makes the reference's index-range tree useless (obj/sphere.obj: 3800 box tests and 960
// triangle tests per entering ray).  bvh_build.hpp states the argument; in short:
//   pass 1 (wf_trav_f)  walks a spatial SAH tree over the triangles with boxes padded, per ray, by a proven bound on what the
//                       reference's f32 Moller-Trumbore test can accept, and runs THAT test (tri_t, same operations) on the
//                       leaves it reaches: every triangle the reference's test would pass becomes a candidate {t, index};
//   pass 2 (wf_replay)  replays BVHNode::intersect_ray (geometry.rs:94-119) over the root-to-candidate paths only, in index
//                       order with the reference's running bound: box rejections (flat boxes included), `t <= best`
//                       acceptance and the later-equal-hit-wins rule are evaluated by the reference's own arithmetic on
//                       the reference's own boxes.  Triangles that are no candidates fail the reference's test wherever the
//                       reference reaches them and cannot change its result.
// A ray the bound does not cover (B > 1/2, non-finite) or with more than kCandMax candidates in one mesh takes the plain
// reference walk for that mesh inside wf_replay.

// next two-stage mesh (index >= m, bit set in `mask`) whose REFERENCE root box the ray enters (geometry.rs:103 at the root:
// nothing below can be reached otherwise); sets up the object-space ray and the padded walk.  `fallback` receives the bits
// of entered meshes the bound does not cover for this ray.
__device__ __forceinline__ bool enter_next_mesh_f(const DScene& S, int& m, uint32_t mask, f3 o, f3 d, float t_min, float t_max,
                                                  f3& oo, f3& od, f3& inv_d, float& rho_out, float& t_lo, float& t_hi,
                                                  int& fi, int& fend, int& ftb, uint32_t& fallback, float4& grid) {
    cf4_ptr gn = (cf4_ptr)S.nodes;
    for (; m < S.n_meshes && m < kTwoStageMaxMeshes; m++) {
        if (!((mask >> m) & 1u)) continue;
        auto M = &S.meshes[m];
        oo = xform_point(M->inv_transform, o);                           // geometry.rs:304
        od = xform_vector(M->inv_transform, d);
        rcp3_exact(od.x, od.y, od.z, inv_d.x, inv_d.y, inv_d.z);         // geometry.rs:57
        const float4 n0 = gn[2 * M->node_begin], n1 = gn[2 * M->node_begin + 1];
        if (__float_as_int(n1.w) < 0 &&                                   // a root that is a leaf is never box-tested (:95)
            !slab(mk3(n0.x, n0.y, n0.z), mk3(n1.x, n1.y, n1.z), oo, inv_d, t_min, t_max)) continue;
        auto F = &S.meshf[m];
        float rho, dt;
        if (!two_stage_pad(F, oo, od, t_max, rho, dt)) { fallback |= 1u << m; continue; }
        rho_out = rho;
        t_lo = t_min - dt; t_hi = t_max + dt;
        fi = F->fnode_begin; fend = F->fnode_end; ftb = F->ftri_begin;
        grid = make_float4(F->qs, F->qbx, F->qby, F->qbz);
        return true;
    }
    return false;
}

// Does the ray enter the reference root box of ANY two-stage mesh in `mask`?  (The test enter_next_mesh_f starts with.)
__device__ __forceinline__ bool enters_two_stage_root(const DScene& S, uint32_t mask, f3 o, f3 d, float t_min, float t_max) {
    cf4_ptr gn = (cf4_ptr)S.nodes;
    bool any = false;
    for (int m = 0; m < S.n_meshes && m < kTwoStageMaxMeshes; m++) {
        if (!((mask >> m) & 1u)) continue;
        auto M = &S.meshes[m];
        const f3 oo = xform_point(M->inv_transform, o), od = xform_vector(M->inv_transform, d);
        f3 inv_d;
        rcp3_exact(od.x, od.y, od.z, inv_d.x, inv_d.y, inv_d.z);
        const float4 n0 = gn[2 * M->node_begin], n1 = gn[2 * M->node_begin + 1];
        any = any | (int)(__float_as_int(n1.w) >= 0) | (int)slab(mk3(n0.x, n0.y, n0.z), mk3(n1.x, n1.y, n1.z), oo, inv_d, t_min, t_max);
    }
    return any;
}

// pass 0 of the two-stage traversal: the queue holds every ray that entered SOME mesh root; most of them are none of the
// two-stage meshes' business (HEAD scene: 77 % miss the sphere's root box).  Inside the persistent walkers such an entry costs a
// whole refill round — a chain of dependent loads (cursor, shard table, queue, path state) of ~16 us per wave for 64 rejected
// rays: 10.5 of wf_trav_f's 23.5 ms.  This plain streaming pass over the queue — every load independent, one thread per entry —
// applies the root test and keeps the rest: cand_hdr[j].x = the state position of the j-th kept ray, trav_head[2] = how many.
// wf_trav_f and wf_replay then work on that list.  Blocks take the shards round-robin, so nothing is searched.
__global__ __launch_bounds__(256) void wf_filter_f(WfArgs A) {
    const DScene& S = A.S;
    const float t_min = 0.001f, t_max = A.C.max_trace_dist;
    const uint32_t cap = A.cap;
    const uint32_t shard = blockIdx.x % (uint32_t)kWfShards, sub = blockIdx.x / (uint32_t)kWfShards, n_sub = gridDim.x / (uint32_t)kWfShards;
    const uint32_t count = A.in_count[(uint32_t)kWfShards + shard];            // this shard's class-B paths: its rays that wait for a mesh walk are among them
    const uint32_t lane = threadIdx.x & 63u;
    // The kept entries are staged in LDS and leave in runs of >= 1024 behind ONE atomic on the list's counter (a single address
    // retires ~90 atomics per microsecond: one per wave and 64 entries took 29 ms here)
    __shared__ uint32_t buf[1024 + 256];
    __shared__ uint32_t n_buf, g_base;
    if (threadIdx.x == 0) n_buf = 0;
    __syncthreads();
    auto flush = [&]() {                                  // block-uniform call
        const uint32_t n = n_buf;
        if (threadIdx.x == 0 && n) g_base = atomicAdd(&A.trav_head[2], n);
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n; i += 256u) A.cand_hdr[g_base + i] = make_uint2(buf[i], 0u);
        __syncthreads();
        if (threadIdx.x == 0) n_buf = 0;
        __syncthreads();
    };
    for (uint32_t k0 = sub * 256u; k0 < count; k0 += n_sub * 256u) {
        const uint32_t k = k0 + threadIdx.x;
        uint32_t pos = 0;
        bool keep = false;
        if (k < count) {
            pos = shard * A.region + (A.region - 1u - k);
            const float4 q0 = A.st_out[st_idx(0, pos, cap)], q1 = A.st_out[st_idx(1, pos, cap)];
            keep = enters_two_stage_root(S, A.trav_mask, mk3(q0.x, q0.y, q0.z), mk3(q0.w, q1.x, q1.y), t_min, t_max);
        }
        const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
        uint32_t base = 0;
        if (lane == 0 && m) base = atomicAdd(&n_buf, (uint32_t)__popcll(m));
        base = (uint32_t)__shfl((int)base, 0);
        if (keep) buf[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = pos;
        __syncthreads();
        const uint32_t n_now = n_buf;       // read between two barriers: no wave may add to n_buf (next trip) before every wave has
        __syncthreads();                    // read it, or the waves could disagree about entering flush() and its barriers
        if (n_now >= 1024u) flush();
    }
    flush();
}

// pass 1: persistent walkers over the traversal queue (work distribution and refill exactly as wf_trav)
// (Negative result, round 3: the top levels of the F-trees staged in LDS — pool re-ordered to [top levels][deeper subtrees], every link
// explicit, 2048 nodes = 64 KB per 1024-thread block — is bit-exact and changes nothing: wf_trav_f 25.1 -> 25.0 ms on the HEAD scene.  A
// wave's step waits for its slowest lane, and some lane is always below the top levels; more resident waves do not help either
// (4 / 6 / 8 blocks per CU: 30.0 / 24.6 / 25.2 ms) although VALU issue is only 0.34: the limit is the line traffic of the gathers (every
// 32-byte node read moves a whole cache line from L2); a node that fills its line would cut it.  DESIGN.md section 4, tools/experiments/.)
// (Negative result, round 2: flag bits in the queue word — "enters a reference-walk mesh" / "enters a two-stage mesh", so that each
// walker skips the entries that are not for it without touching the path state — need wf_main to test EVERY mesh root instead
// of stopping at the first one entered: wf_main +2.5 ms on cfg2, +1.2 ms on the HEAD scene, wf_trav_f only -0.3 ms.)
__global__ __launch_bounds__(256, PT_TRAV_WAVES) void wf_trav_f(WfArgs A) {
    const DScene& S = A.S;
    const uint32_t n_q = A.trav_head[2];                 // the rays wf_filter_f kept
    const uint32_t n_waves = gridDim.x * 4u;
    uint32_t chunk = 256u;                               // as wf_trav: short queues are dealt out evenly
    if (n_q < n_waves * 256u) chunk = max(64u, ((n_q + n_waves - 1u) / n_waves + 63u) & ~63u);
    const bool shared_part = n_waves * chunk < n_q;
    if (blockIdx.x * 4u * chunk >= n_q) return;
    cf4_ptr FN = (cf4_ptr)S.fnodes;
    cf4_ptr FT = (cf4_ptr)S.ftris;
    const float t_min = 0.001f, t_max = A.C.max_trace_dist;
    const uint32_t cap = A.cap;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave_id = blockIdx.x * 4u + (threadIdx.x >> 6);
    uint32_t wnext = wave_id * chunk, wend = min(wnext + chunk, n_q);
    bool drained = false;
    if (wnext >= n_q) { wnext = wend = 0; drained = true; }

    bool have = false, atleaf = false;
    uint32_t vi = 0, pos = 0, nc = 0, fb = 0;
    f3 too = mk3(0.0f, 0.0f, 0.0f), tod = too, tinv = too;
    float trho = 0.0f, t_lo = 0.0f, t_hi = 0.0f;
    int tm = 0, fi = 0, fend = 0, ftb = 0;
    float4 c0 = make_float4(0.0f, 0.0f, 0.0f, 0.0f), tgrid = c0;      // c0: the node the lane stands on (16 bytes: ONE load per step)
    const int last_fnode = S.n_fnodes - 1;          // clamp for the load that follows the last node of a tree
#ifdef PT_TRAV_DIAG
    unsigned long long dg_trips = 0, dg_bsteps = 0, dg_blanes = 0, dg_lsteps = 0, dg_llanes = 0, dg_refills = 0, dg_rlanes = 0, dg_walks = 0;
#endif

    while (true) {
        // ---- refill idle lanes ----
        // (Negative result, round 2: only 23 % of the HEAD scene's queue entries enter the sphere's root box, so one refill round
        // leaves most idle lanes idle — 24.6 of 64 lanes live in an F-node step — but looping the refill up to 8 times until the
        // wave is busy made wf_trav_f slower, 7.5 -> 7.85 ms: each round is a dependent queue -> state gather.)
        unsigned long long need = __builtin_amdgcn_ballot_w64(!have);
        const uint32_t n_idle = (uint32_t)__popcll(need);
        if ((n_idle >= A.refill_min || n_idle == 64u) && !drained) {
            if (wnext == wend) {
                uint32_t base = n_q;
                if (shared_part) {
                    if (lane == 0) base = n_waves * chunk + atomicAdd(&A.trav_head[1], chunk);
                    base = (uint32_t)__shfl((int)base, 0);
                }
                if (base >= n_q) drained = true;
                else { wnext = base; wend = min(base + chunk, n_q); }
            }
            const uint32_t avail = wend - wnext;
            const uint32_t rank = (uint32_t)__popcll(need & ((1ull << lane) - 1ull));
            const bool take = !have && rank < avail;
            if (take) {
                vi = wnext + rank;
                pos = A.cand_hdr[vi].x;
                const float4 q0 = A.st_out[st_idx(0, pos, cap)], q1 = A.st_out[st_idx(1, pos, cap)];
                const f3 o = mk3(q0.x, q0.y, q0.z), d = mk3(q0.w, q1.x, q1.y);
                nc = 0; fb = 0; tm = 0; atleaf = false;
                if (enter_next_mesh_f(S, tm, A.trav_mask, o, d, t_min, t_max, too, tod, tinv, trho, t_lo, t_hi, fi, fend, ftb, fb, tgrid)) {
                    c0 = FN[fi];
                    have = true;
                } else {
                    A.cand_hdr[vi] = make_uint2(pos, fb << 8);          // entered none of these meshes (or only uncovered ones)
                }
            }
            wnext += min(avail, n_idle);
#ifdef PT_TRAV_DIAG
            dg_refills++; dg_rlanes += (unsigned long long)__popcll(__builtin_amdgcn_ballot_w64(take)); dg_walks += (unsigned long long)__popcll(__builtin_amdgcn_ballot_w64(take & have));
#endif
        }
        if (__builtin_amdgcn_ballot_w64(have) == 0ull) {
            if (drained) break;
            continue;
        }
#ifdef PT_TRAV_DIAG
        dg_trips++;
#endif

        // ---- one voted step: a burst of F-node box tests, or the triangle tests of the leaves reached ----
        const bool walking = have & !atleaf;
        const int n_leaf = __popcll(__builtin_amdgcn_ballot_w64(have & atleaf));
        const int n_walk = __popcll(__builtin_amdgcn_ballot_w64(walking));
        if (n_walk >= n_leaf) {
#pragma unroll
            for (int j = 0; j < PT_TRAVF_BURST; j++) {
                const bool act = have & !atleaf & (fi < fend);
#ifdef PT_TRAV_DIAG
                dg_bsteps++; dg_blanes += (unsigned long long)__popcll(__builtin_amdgcn_ballot_w64(act));
#endif
                f3 bmin, bmax;
                fq_box(c0, tgrid, bmin, bmax);
                const bool hit = slab_padded(bmin, bmax, too, trho, tinv, t_lo, t_hi);
                const int link = __float_as_int(c0.w);
                const bool leaf = link < 0;
                const bool stop = act & hit & leaf;                      // reached a leaf: its triangles are tested in a leaf step
                const int nxt = (hit | leaf) ? fi + 1 : link;            // a leaf's successor is the next node either way
                atleaf = atleaf | stop;
                const bool move = act & !stop;
                fi = move ? nxt : fi;
                if (move) c0 = FN[min(fi, last_fnode)];
            }
        } else if (have & atleaf) {
#ifdef PT_TRAV_DIAG
            dg_lsteps++; dg_llanes += (unsigned long long)n_leaf;
#endif
            const int payload = __float_as_int(c0.w) & 0x7fffffff;
            const int first = ftb + (payload >> 3), count = (payload & 7) + 1;
            for (int k = 0; k < count; k++) {
                const float4 t0 = FT[3 * (first + k)], t1 = FT[3 * (first + k) + 1], t2 = FT[3 * (first + k) + 2];
                float t, u, v;
                const bool ok = tri_t(too, tod, mk3(t0.x, t0.y, t0.z), mk3(t1.x, t1.y, t1.z), mk3(t2.x, t2.y, t2.z), t_min, t_max, t, u, v);
                if (ok) {
                    if (nc < (uint32_t)kCandMax) { A.cand[(size_t)vi * kCandMax + nc] = make_uint2(__float_as_uint(t), ((uint32_t)tm << 24) | (uint32_t)__float_as_int(t0.w)); nc++; }
                    else fb |= 1u << tm;                                  // too many passing triangles: reference walk for this mesh
                }
            }
            atleaf = false;
            fi = fi + 1;                                                  // a leaf's successor in pre-order
            c0 = FN[min(fi, last_fnode)];
        }
        if (have && !atleaf && fi >= fend) {
            tm++;
            const float4 q0 = A.st_out[st_idx(0, pos, cap)], q1 = A.st_out[st_idx(1, pos, cap)];
            const f3 o = mk3(q0.x, q0.y, q0.z), d = mk3(q0.w, q1.x, q1.y);
            if (enter_next_mesh_f(S, tm, A.trav_mask, o, d, t_min, t_max, too, tod, tinv, trho, t_lo, t_hi, fi, fend, ftb, fb, tgrid)) {
                c0 = FN[fi];
            } else {
                A.cand_hdr[vi] = make_uint2(pos, nc | (fb << 8));
                have = false;
            }
        }
    }
#ifdef PT_TRAV_DIAG
    if (A.diag && lane == 0) {
        atomicAdd(&A.diag[8], dg_trips); atomicAdd(&A.diag[9], dg_walks); atomicAdd(&A.diag[10], dg_bsteps); atomicAdd(&A.diag[11], dg_blanes);
        atomicAdd(&A.diag[12], dg_lsteps); atomicAdd(&A.diag[13], dg_llanes); atomicAdd(&A.diag[14], dg_refills); atomicAdd(&A.diag[15], dg_rlanes);
    }
#endif
}

// pass 2: one thread per queued ray
__global__ __launch_bounds__(256) void wf_replay(WfArgs A) {
    const DScene& S = A.S;
    const uint32_t n_q = A.trav_head[2];                 // the rays wf_filter_f kept
    const float t_min = 0.001f, t_max = A.C.max_trace_dist;
    const uint32_t cap = A.cap;
    cf4_ptr RN = (cf4_ptr)S.nodes;
    Bvh<false> B;
    bvh_bind(B, S, 0);
    // Most queued rays have no candidate at all (they entered a root box and passed no triangle): with one thread per
    // queue entry 9.5 of 64 lanes did any work (PMC, HEAD scene).  So every wave COMPACTS first: it scans the headers 64 at
    // a time, collects the entries that need a replay in a small LDS queue of its own (ballot + prefix count, no atomics),
    // and runs the replay body only on full batches of 64.
    __shared__ volatile uint32_t rq[4][128];
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t n_waves = gridDim.x * 4u;
    uint32_t scan = (blockIdx.x * 4u + wv) * 64u;        // wave-uniform cursor over the queue, strided by the waves
    uint32_t qn = 0;                                       // entries waiting in this wave's LDS queue (wave-uniform)
    for (;;) {
        while (qn < 64u && scan < n_q) {
            const uint32_t e = scan + lane;
            const bool want = e < n_q && A.cand_hdr[e].y != 0u;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(want);
            if (want) rq[wv][qn + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = e;
            qn += (uint32_t)__popcll(m);
            scan += n_waves * 64u;
        }
        if (qn == 0u) break;
        const uint32_t take = qn < 64u ? qn : 64u;
        const bool active = lane < take;
        const uint32_t vi = active ? rq[wv][lane] : 0u;
        const uint32_t moved = (lane < qn - take) ? rq[wv][64u + lane] : 0u;      // the rest moves to the front of the queue
        __builtin_amdgcn_wave_barrier();
        if (lane < qn - take) rq[wv][lane] = moved;
        qn -= take;
        if (active) {
        const uint2 hdr = A.cand_hdr[vi];
        const uint32_t nc = hdr.y & 0xffu, fb = hdr.y >> 8;
        const size_t pos = hdr.x;
        const float4 q0 = A.st_out[st_idx(0, pos, cap)], q1 = A.st_out[st_idx(1, pos, cap)];
        const f3 o = mk3(q0.x, q0.y, q0.z), d = mk3(q0.w, q1.x, q1.y);
        const Hit2 hr = *st_hit(A.st_out, pos, cap);
        Best best; best.t = hr.t; best.obj = hr.obj; best.tri = -1; best.u = best.v = 0.0f;
        int best_mesh = -1;
        // candidates into registers (unused slots get the largest key)
        uint32_t key[kCandMax]; float ct[kCandMax];
#pragma unroll
        for (int k = 0; k < kCandMax; k++) {
            uint2 c = make_uint2(0u, 0xffffffffu);
            if ((uint32_t)k < nc) c = A.cand[(size_t)vi * kCandMax + k];
            ct[k] = __uint_as_float(c.x); key[k] = c.y;
        }
        long long last = -1;
        int cur = -1, n_tris = 0, node_begin = 0;
        f3 oo = o, od = d, inv = d;
        float wt = t_max; int win = -1;
        int prev = -1, fail_depth = 0x7fffffff;
        for (uint32_t round = 0; round < nc; round++) {
            // next candidate in (mesh, triangle index) order
            uint32_t kmin = 0xffffffffu; float tsel = 0.0f;
#pragma unroll
            for (int k = 0; k < kCandMax; k++) {
                const bool better = ((long long)key[k] > last) && (key[k] < kmin);
                kmin = better ? key[k] : kmin; tsel = better ? ct[k] : tsel;
            }
            if (kmin == 0xffffffffu) break;
            last = (long long)kmin;
            const int m = (int)(kmin >> 24), tri = (int)(kmin & 0xffffffu);
            if ((fb >> m) & 1u) continue;                                 // this mesh takes the reference walk below
            if (m != cur) {
                if (cur >= 0 && win >= 0) {
                    const Best before = best;
                    consider(best, wt, S.meshes[cur].object_index, win, 0.0f, 0.0f);
                    if (best.tri != before.tri || best.obj != before.obj || best.t != before.t) best_mesh = cur;
                }
                cur = m;
                auto M = &S.meshes[m];
                oo = xform_point(M->inv_transform, o); od = xform_vector(M->inv_transform, d);
                rcp3_exact(od.x, od.y, od.z, inv.x, inv.y, inv.z);
                n_tris = M->n_tris; node_begin = M->node_begin;
                wt = t_max; win = -1; prev = -1; fail_depth = 0x7fffffff;
            }
            // BVHNode::intersect_ray restricted to the path root -> leaf `tri`; ancestors shared with the previous candidate
            // were decided when it was walked (the reference tests a node once, with the bound it held then)
            int s = 0, e = n_tris, me = 0, depth = 0, new_fail = 0x7fffffff;
            bool dead = false;
            while (e - s > 1) {
                const bool shared = (prev >= s) & (prev < e);
                if (shared) {
                    if (depth == fail_depth) { dead = true; new_fail = fail_depth; break; }
                } else {
                    const float4 n0 = RN[2 * (node_begin + me)], n1 = RN[2 * (node_begin + me) + 1];
                    if (!slab(mk3(n0.x, n0.y, n0.z), mk3(n1.x, n1.y, n1.z), oo, inv, t_min, wt)) { dead = true; new_fail = depth; break; }   // :103
                }
                const int mid = s + (e - s) / 2;                          // geometry.rs:209
                if (tri < mid) { me = me + 1; e = mid; } else { me = me + 2 * (mid - s); s = mid; }
                depth++;
            }
            fail_depth = dead ? new_fail : 0x7fffffff;
            prev = tri;
            if (!dead && tsel <= wt) { wt = tsel; win = tri; }            // :349 rejects only t > t_max: a later equal hit replaces
        }
        if (cur >= 0 && win >= 0) {
            const Best before = best;
            consider(best, wt, S.meshes[cur].object_index, win, 0.0f, 0.0f);
            if (best.tri != before.tri || best.obj != before.obj || best.t != before.t) best_mesh = cur;
        }
        // meshes the bound did not cover for this ray, or with more passing triangles than candidate slots: the plain walk
        for (int m = 0; m < S.n_meshes && m < kTwoStageMaxMeshes; m++) {
            if (!((fb >> m) & 1u)) continue;
            auto M = &S.meshes[m];
            const f3 po = xform_point(M->inv_transform, o), pd = xform_vector(M->inv_transform, d);
            float bt, bu, bv; int btri;
            traverse_mesh(B, M->node_begin, M->node_end, M->tri_begin, po, pd, t_min, t_max, bt, btri, bu, bv);
            if (btri >= 0) {
                const Best before = best;
                consider(best, bt, M->object_index, btri, bu, bv);
                if (best.tri != before.tri || best.obj != before.obj || best.t != before.t) best_mesh = m;
            }
        }
        if (best.tri >= 0) {
            // the winner's barycentrics: the reference's test once more on that triangle (same operations, same bits)
            auto M = &S.meshes[best_mesh];
            const f3 po = xform_point(M->inv_transform, o), pd = xform_vector(M->inv_transform, d);
            f3 a, e1, e2;
            B.tri(M->tri_begin + best.tri, a, e1, e2);
            float t, u, v;
            (void)tri_t(po, pd, a, e1, e2, t_min, t_max, t, u, v);
            Hit2 hw; hw.t = best.t; hw.obj = best.obj;
            *st_hit(A.st_out, pos, cap) = hw;
            *st_tri(A.st_out, pos, cap) = best.tri;
            A.st_out[st_idx(5, pos, cap)] = make_float4(u, v, __int_as_float(PT_PRE_ATTR ? M->tri_begin + best.tri : -1), __int_as_float(PT_PRE_ATTR ? best_mesh : -1));
        }
        }   // active
    }
}

// After every wf_main: turn the per-shard append counters into the tables the next kernels index with —
// in_count / in_blkpfx (class A shards, then class B shards) for the next wf_main, trav_pfx for wf_trav —
// and a 3-word header (blocks of the next pass, live paths, queue length).  One block; keeps the host out
// of the critical path: the header lands in pinned host memory while wf_trav runs.
__global__ __launch_bounds__(256) void wf_prefix(uint32_t* __restrict__ out_count, uint32_t* __restrict__ trav_count,
                                                 uint32_t* __restrict__ in_count, uint32_t* __restrict__ in_blkpfx,
                                                 uint32_t* __restrict__ trav_pfx, uint32_t* __restrict__ hdr,
                                                 volatile uint32_t* host_hdr, uint32_t seq) {
    __shared__ uint32_t sc[4][256];
    __shared__ uint32_t tot_b, tot_seg;
    const uint32_t t = threadIdx.x;
    if (t == 0) { tot_b = 0; tot_seg = 0; }
    __syncthreads();
    const uint32_t a = out_count[t], b = out_count[kWfShards + t], q = b;   // the walkers' work list = the class-B paths
    atomicAdd(&tot_b, b);
    atomicAdd(&tot_seg, trav_count[t]);                                     // path segments of the pass (statistics), per shard
    trav_count[t] = 0;
    out_count[t] = 0; out_count[kWfShards + t] = 0;                         // ready for the next wf_main
    if (t < 4) trav_count[kWfShards + t] = 0;                               // trav_head[0] / [1]: the shared cursors of wf_trav / wf_trav_f; [2]: wf_filter_f's count
    in_count[t] = a; in_count[kWfShards + t] = b;
    const uint32_t va = (a + kBlock - 1) / kBlock, vb = (b + kBlock - 1) / kBlock;
    uint32_t x0 = va, x1 = vb, x2 = q, x3 = a + b;
    sc[0][t] = x0; sc[1][t] = x1; sc[2][t] = x2; sc[3][t] = x3;
    __syncthreads();
    for (uint32_t d = 1; d < 256; d <<= 1) {           // inclusive Hillis-Steele scans, four at once
        uint32_t y0 = 0, y1 = 0, y2 = 0, y3 = 0;
        if (t >= d) { y0 = sc[0][t - d]; y1 = sc[1][t - d]; y2 = sc[2][t - d]; y3 = sc[3][t - d]; }
        __syncthreads();
        x0 += y0; x1 += y1; x2 += y2; x3 += y3;
        sc[0][t] = x0; sc[1][t] = x1; sc[2][t] = x2; sc[3][t] = x3;
        __syncthreads();
    }
    const uint32_t blocks_a = sc[0][255];
    in_blkpfx[t] = x0 - va;
    in_blkpfx[kWfShards + t] = blocks_a + (x1 - vb);
    trav_pfx[t] = x2 - q;
    if (t == 255) {
        in_blkpfx[2 * kWfShards] = blocks_a + x1;
        trav_pfx[kWfShards] = x2;
        hdr[0] = blocks_a + x1; hdr[1] = x3; hdr[2] = x2; hdr[3] = 0;
        // the host's copy goes straight into pinned host memory (no copy kernel that would queue behind the
        // persistent walkers): data, system-scope fence, then the sequence number the host polls
        host_hdr[0] = blocks_a + x1; host_hdr[1] = x3; host_hdr[2] = x2; host_hdr[4] = tot_b;     // [4]: class-B paths (statistics only)
        host_hdr[6] = tot_seg;                                                                     // [6]: Scene::intersect_ray evaluations of this pass (statistics only)
        host_hdr[5] = blocks_a;                                                                    // [5]: the class-A blocks come first
        __threadfence_system();
        host_hdr[3] = seq;
    }
}

// per-pixel in-order accumulation of one batch of samples (tracing.rs:232-241)
__global__ __launch_bounds__(256) void wf_reduce(WfArgs A, uint32_t first_batch, uint32_t last_batch) {
    const uint32_t pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= A.npix) return;
    float4 acc = first_batch ? make_float4(0.0f, 0.0f, 0.0f, 0.0f) : A.accum[pix];
    uint32_t sigsum = __float_as_uint(acc.w);
    // dead tile (nothing reachable, wf_main generated no ray and wrote no sample): every sample is +0.0
    bool dead = false;
    if (A.tile_mask && !A.sig) {
        const uint32_t tile = (pix / kTilePixels) * (uint32_t)A.R.world + (uint32_t)A.R.rank;
        dead = tile < A.R.tiles_total && (A.tile_mask[A.R.tiles_total + tile] >> 63) != 0ull;
    }
    for (uint32_t s = 0; s < (dead ? 0u : A.s_count); s++) {
        const float4 v = A.samp[(size_t)s * A.npix + pix];
        acc.x = acc.x + v.x; acc.y = acc.y + v.y; acc.z = acc.z + v.z;      // final_color += shade_ray(..)  :238
        sigsum += __float_as_uint(v.w);
    }
    acc.w = __uint_as_float(sigsum);
    A.accum[pix] = acc;
    if (last_batch) {
        uint32_t px, py; bool in_image;
        wf_pixel_of(A, pix, px, py, in_image);
        const float n = (float)A.C.spp;
        float* o3 = A.out + (size_t)pix * 3;
        if (in_image) { o3[0] = acc.x / n; o3[1] = acc.y / n; o3[2] = acc.z / n; }       // :241
        else { o3[0] = 0.0f; o3[1] = 0.0f; o3[2] = 0.0f; }
        if (A.sig) A.sig[pix] = in_image ? sigsum : 0u;
    }
}

// ---------------------------------------------------------------- K3: un-permute
// gathered[world][tiles_padded][1024][3] -> image[H][W][3].  One thread per pixel.
__global__ __launch_bounds__(256) void fb_unpermute(const float* __restrict__ gathered, float* __restrict__ image,
                                                    uint32_t width, uint32_t height, uint32_t tiles_x,
                                                    uint32_t world, uint32_t tiles_padded) {
    uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= width * height) return;
    uint32_t x = idx % width, y = idx / width;
    uint32_t tile = (y / kTile) * tiles_x + (x / kTile);
    uint32_t rank = tile % world, slot = tile / world;
    size_t src = (((size_t)rank * tiles_padded + slot) * kTilePixels + (y % kTile) * kTile + (x % kTile)) * 3;
    size_t dst = (size_t)idx * 3;
    image[dst] = gathered[src]; image[dst + 1] = gathered[src + 1]; image[dst + 2] = gathered[src + 2];
}

// same mapping for the u32 signature plane
__global__ __launch_bounds__(256) void sig_unpermute(const uint32_t* __restrict__ gathered, uint32_t* __restrict__ image,
                                                     uint32_t width, uint32_t height, uint32_t tiles_x,
                                                     uint32_t world, uint32_t tiles_padded) {
    uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= width * height) return;
    uint32_t x = idx % width, y = idx / width;
    uint32_t tile = (y / kTile) * tiles_x + (x / kTile);
    uint32_t rank = tile % world, slot = tile / world;
    image[idx] = gathered[((size_t)rank * tiles_padded + slot) * kTilePixels + (y % kTile) * kTile + (x % kTile)];
}

// ---------------------------------------------------------------- K4: tone-map
// tracing.rs:244-256: saturate toward white, clamp, pow(1/gamma), *255.9999, `as u8`
__global__ __launch_bounds__(256) void fb_tonemap_u8(const float* __restrict__ image, uint8_t* __restrict__ out,
                                                     uint32_t n_pixels, float inv_gamma) {
    uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_pixels) return;
    float tmp[3] = { image[3 * idx], image[3 * idx + 1], image[3 * idx + 2] };
    float fc[3] = { tmp[0], tmp[1], tmp[2] };
#pragma unroll
    for (int i = 0; i < 3; i++) {
        float d = tmp[i] - 1.0f;
        if (d > 0.0f) { fc[(i + 1) % 3] += d; fc[(i + 2) % 3] += d; }
    }
#pragma unroll
    for (int i = 0; i < 3; i++) {
        float q = powf(clampf(fc[i], 0.0f, 1.0f), inv_gamma) * 255.9999f;
        uint8_t b = (q != q) ? 0 : (q <= 0.0f ? 0 : (q >= 255.0f ? 255 : (uint8_t)q));
        out[3 * idx + i] = b;
    }
}

// ---------------------------------------------------------------- self-test of the arithmetic shortcuts
// rcp_exact against the IEEE division, over ALL 2^32 f32 bit patterns (NaN results compare equal as NaNs).
__global__ __launch_bounds__(256) void selftest_rcp(unsigned long long* mismatches) {
    unsigned long long bad = 0;
    const unsigned long long n = 1ull << 32, stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long i = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; i < n; i += stride) {
        const float x = __uint_as_float((uint32_t)i);
        const float a = rcp_exact(x), b = 1.0f / x;
        if (!(__float_as_uint(a) == __float_as_uint(b) || (a != a && b != b))) bad++;
    }
    if (bad) atomicAdd(mismatches, bad);
}

// ---------------------------------------------------------------- launch wrappers
hipError_t launch_selftest_rcp(unsigned long long* d_mismatches, hipStream_t stream) {
    hipLaunchKernelGGL(selftest_rcp, dim3(4096), dim3(256), 0, stream, d_mismatches);
    return hipGetLastError();
}
hipError_t launch_megakernel(const K1Args& args, uint32_t n_blocks, bool lds, bool sig, size_t lds_bytes, hipStream_t stream) {
    dim3 grid(n_blocks), block(kBlock);
#define PT_LAUNCH(L, G) hipLaunchKernelGGL((pt_megakernel<L, G>), grid, block, (L) ? lds_bytes : 0, stream, args)
    if (lds) { if (sig) PT_LAUNCH(true, true); else PT_LAUNCH(true, false); }
    else     { if (sig) PT_LAUNCH(false, true); else PT_LAUNCH(false, false); }
#undef PT_LAUNCH
    return hipGetLastError();
}

hipError_t launch_megakernel_voted(const K1Args& args, uint32_t n_blocks, bool lds, bool sig, bool diag, bool gv,
                                   size_t lds_bytes, hipStream_t stream) {
    dim3 grid(n_blocks), block(kBlock);
    // gv: the scene holds a ConvexVolume whose boundary is not the inline sphere (boundary_hit); the diagnostic form always carries it
#define PT_LAUNCH_V(L, G, D, V) hipLaunchKernelGGL((pt_megakernel_voted<L, G, D, V>), grid, block, (L) ? lds_bytes : 0, stream, args)
    if (diag) { if (lds) PT_LAUNCH_V(true, true, true, true); else PT_LAUNCH_V(false, true, true, true); }
    else if (gv) {
        if (lds) { if (sig) PT_LAUNCH_V(true, true, false, true); else PT_LAUNCH_V(true, false, false, true); }
        else     { if (sig) PT_LAUNCH_V(false, true, false, true); else PT_LAUNCH_V(false, false, false, true); }
    }
    else if (lds) { if (sig) PT_LAUNCH_V(true, true, false, false); else PT_LAUNCH_V(true, false, false, false); }
    else          { if (sig) PT_LAUNCH_V(false, true, false, false); else PT_LAUNCH_V(false, false, false, false); }
#undef PT_LAUNCH_V
    return hipGetLastError();
}

hipError_t launch_phong(const K1Args& a, uint32_t n_blocks, bool sig, hipStream_t stream) {
    dim3 grid(n_blocks), block(kBlock);
    if (sig) hipLaunchKernelGGL((pt_phong<true>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((pt_phong<false>), grid, block, 0, stream, a);
    return hipGetLastError();
}
hipError_t launch_branch(const K1Args& a, uint32_t n_blocks, uint32_t path_samples, bool sig, hipStream_t stream) {
    dim3 grid(n_blocks), block(kBlock);
    if (sig) hipLaunchKernelGGL((pt_branch<true>), grid, block, 0, stream, a, path_samples);
    else hipLaunchKernelGGL((pt_branch<false>), grid, block, 0, stream, a, path_samples);
    return hipGetLastError();
}
hipError_t launch_wf_main(const WfArgs& a, uint32_t n_blocks, bool sig, bool gv, bool tex, hipStream_t stream) {
    dim3 grid(n_blocks), block(kBlock);
    // gv: the scene holds a ConvexVolume whose boundary is not the inline sphere.  tex: some mesh takes its material from maps or has a
    // normal map.  The lean form (no mesh branch) serves the launches that cannot meet a mesh hit: the camera-ray pass, the class-A
    // part of a later pass, every pass of a scene without meshes.
    const bool lean = a.iter0 != 0u || a.part == 1u || a.S.n_meshes == 0;
    const bool rare = a.S.n_list_plane + a.S.n_list_volume > 0;          // (gv implies rare: it is a kind of ConvexVolume)
    const bool top = a.S.top_meshf >= 0;                                  // the list's Triangles in a top-level tree (long lists)
#define PT_WF_MAIN2(G, V, M, R, I) do { if (top) hipLaunchKernelGGL((wf_main<false, G, V, M, R, I, true>), grid, block, 0, stream, a); \
                                        else hipLaunchKernelGGL((wf_main<false, G, V, M, R, I, false>), grid, block, 0, stream, a); } while (0)
#define PT_WF_SIG(V, M, R, I) do { if (sig) PT_WF_MAIN2(true, V, M, R, I); else PT_WF_MAIN2(false, V, M, R, I); } while (0)
#define PT_WF_MESH(V, R) do { if (a.iter0) PT_WF_SIG(V, 0, R, true); else if (lean) PT_WF_SIG(V, 0, R, false); \
                              else if (tex) PT_WF_SIG(V, 2, R, false); else PT_WF_SIG(V, 1, R, false); } while (0)
    if (!rare) PT_WF_MESH(false, false);
    else if (gv) PT_WF_MESH(true, true);
    else PT_WF_MESH(false, true);
#undef PT_WF_MESH
#undef PT_WF_SIG
#undef PT_WF_MAIN2
    return hipGetLastError();
}
// big_lds_enabled: the calling context's record of the > 64 KB dynamic-LDS opt-in.  The attribute belongs to the
// function ON THE CURRENT DEVICE, so it is kept per context (one context = one device), not per process.
hipError_t launch_wf_trav(const WfArgs& a, uint32_t n_blocks, int lds_mode, size_t lds_bytes, bool* big_lds_enabled, hipStream_t stream) {
    dim3 grid(n_blocks), block(kBlock);
    const bool multi = a.trav_mask != 1u || a.S.n_meshes > 32;      // more than mesh 0 to walk in this launch
    if (lds_mode == 3) {           // nodes in LDS, one 1024-thread block per CU
        if (!*big_lds_enabled) {
            hipError_t e = hipFuncSetAttribute((const void*)wf_trav<2, 1024, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e == hipSuccess) e = hipFuncSetAttribute((const void*)wf_trav<2, 1024, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            *big_lds_enabled = true;
        }
        if (multi) hipLaunchKernelGGL((wf_trav<2, 1024, true>), grid, dim3(1024), lds_bytes, stream, a);
        else hipLaunchKernelGGL((wf_trav<2, 1024, false>), grid, dim3(1024), lds_bytes, stream, a);
    }
    else if (lds_mode == 2) { if (multi) hipLaunchKernelGGL((wf_trav<2, 256, true>), grid, block, lds_bytes, stream, a); else hipLaunchKernelGGL((wf_trav<2, 256, false>), grid, block, lds_bytes, stream, a); }
    else { if (multi) hipLaunchKernelGGL((wf_trav<0, 256, true>), grid, block, 0, stream, a); else hipLaunchKernelGGL((wf_trav<0, 256, false>), grid, block, 0, stream, a); }
    return hipGetLastError();
}
hipError_t launch_wf_trav_p(const WfArgs& a, uint32_t n_blocks, size_t lds_bytes, hipStream_t stream) {      // the paired layout: 512-thread blocks
    if (a.trav_mask != 1u || a.S.n_meshes > 32) hipLaunchKernelGGL((wf_trav_i<512, true, true, true>), dim3(n_blocks), dim3(512), lds_bytes, stream, a);
    else hipLaunchKernelGGL((wf_trav_i<512, false, true, true>), dim3(n_blocks), dim3(512), lds_bytes, stream, a);
    return hipGetLastError();
}
hipError_t launch_wf_trav_i(const WfArgs& a, uint32_t n_blocks, size_t lds_bytes, bool leaf_lds, bool* big_lds_enabled, hipStream_t stream) {
    if (leaf_lds) {                // the whole split image (interior + leaf records) fits 64 KB: 256-thread blocks, several per CU
        if (a.trav_mask != 1u || a.S.n_meshes > 32) hipLaunchKernelGGL((wf_trav_i<256, true, true>), dim3(n_blocks), dim3(256), lds_bytes, stream, a);
        else hipLaunchKernelGGL((wf_trav_i<256, false, true>), dim3(n_blocks), dim3(256), lds_bytes, stream, a);
        return hipGetLastError();
    }
    if (!*big_lds_enabled) {       // the attribute belongs to the function on the current device: kept per context
        hipError_t e = hipFuncSetAttribute((const void*)wf_trav_i<1024, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e == hipSuccess) e = hipFuncSetAttribute((const void*)wf_trav_i<1024, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        *big_lds_enabled = true;
    }
    const bool multi = a.trav_mask != 1u || a.S.n_meshes > 32;
    if (multi) hipLaunchKernelGGL((wf_trav_i<1024, true>), dim3(n_blocks), dim3(1024), lds_bytes, stream, a);
    else hipLaunchKernelGGL((wf_trav_i<1024, false>), dim3(n_blocks), dim3(1024), lds_bytes, stream, a);
    return hipGetLastError();
}
hipError_t launch_wf_filter_f(const WfArgs& a, uint32_t blocks_per_shard, hipStream_t stream) {
    hipLaunchKernelGGL(wf_filter_f, dim3((uint32_t)kWfShards * blocks_per_shard), dim3(256), 0, stream, a);
    return hipGetLastError();
}
hipError_t launch_wf_trav_f(const WfArgs& a, uint32_t n_blocks, hipStream_t stream) {
    hipLaunchKernelGGL(wf_trav_f, dim3(n_blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}
hipError_t launch_wf_replay(const WfArgs& a, uint32_t n_blocks, hipStream_t stream) {
    hipLaunchKernelGGL(wf_replay, dim3(n_blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}
hipError_t launch_wf_prefix(uint32_t* out_count, uint32_t* trav_count, uint32_t* in_count, uint32_t* in_blkpfx,
                            uint32_t* trav_pfx, uint32_t* hdr, uint32_t* host_hdr, uint32_t seq, hipStream_t stream) {
    hipLaunchKernelGGL(wf_prefix, dim3(1), dim3(256), 0, stream, out_count, trav_count, in_count, in_blkpfx, trav_pfx, hdr,
                       (volatile uint32_t*)host_hdr, seq);
    return hipGetLastError();
}
hipError_t launch_wf_reduce(const WfArgs& a, bool first_batch, bool last_batch, hipStream_t stream) {
    hipLaunchKernelGGL(wf_reduce, dim3((a.npix + 255) / 256), dim3(256), 0, stream, a, first_batch ? 1u : 0u, last_batch ? 1u : 0u);
    return hipGetLastError();
}

hipError_t launch_unpermute(const float* gathered, float* image, uint32_t width, uint32_t height, uint32_t tiles_x,
                            uint32_t world, uint32_t tiles_padded, hipStream_t stream) {
    uint32_t n = width * height;
    hipLaunchKernelGGL(fb_unpermute, dim3((n + 255) / 256), dim3(256), 0, stream, gathered, image, width, height,
                       tiles_x, world, tiles_padded);
    return hipGetLastError();
}
hipError_t launch_sig_unpermute(const uint32_t* gathered, uint32_t* image, uint32_t width, uint32_t height,
                                uint32_t tiles_x, uint32_t world, uint32_t tiles_padded, hipStream_t stream) {
    uint32_t n = width * height;
    hipLaunchKernelGGL(sig_unpermute, dim3((n + 255) / 256), dim3(256), 0, stream, gathered, image, width, height,
                       tiles_x, world, tiles_padded);
    return hipGetLastError();
}
hipError_t launch_tonemap(const float* image, uint8_t* out, uint32_t n_pixels, float inv_gamma, hipStream_t stream) {
    hipLaunchKernelGGL(fb_tonemap_u8, dim3((n_pixels + 255) / 256), dim3(256), 0, stream, image, out, n_pixels, inv_gamma);
    return hipGetLastError();
}

}  // namespace pt
