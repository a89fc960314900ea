// bvh_build.hpp — host-side tree builders of the scene compiler (mi_rt.cpp); header-only so that the CPU test of the
// two-stage traversal (tests/cpp/two_stage_check.cpp) builds the very same trees.
//
//   RefTree   the REFERENCE's BVH of a StaticMesh (geometry.rs:175-217): median split of the triangle INDEX range,
//             leaf i = triangle i, exact union boxes — emitted in DFS pre-order with skip links (node = 8 floats:
//             {bmin.xyz, skip}{bmax.xyz, tri or -1}).  Pre-order makes the topology arithmetic: the node of range
//             [s, e) at index me has its left child [s, mid) at me + 1 and its right child [mid, e) at me + 2 (mid - s).
//   FTree     the "fast" tree of the exact two-stage traversal ("two-stage"): a spatial SAH hierarchy
//             over the mesh's TRIANGLES (leaves of <= 2 triangles, any indices), boxes = the triangles' own bounds.
//
// What the F-tree is for.  The reference's tree splits the triangle INDEX range, so its boxes are only as tight as the
// file's face order is coherent (obj/sphere.obj: consecutive quads lie far apart, every box above the quad level is the
// whole sphere, 3800 box tests and 960 triangle tests per entering ray).  Pass 1 walks the F-tree instead and runs the
// reference's own Moller-Trumbore test (geometry.rs:331-349, same f32 operations) on the triangles of the leaves it
// reaches; every triangle whose test PASSES becomes a candidate.  Pass 2 (pt_kernels.hip wf_replay) replays the reference's
// depth-first walk over the root-to-candidate paths only, in index order, with the reference's running bound, so box
// rejections (flat boxes included), the `t <= best` acceptance and its tie rule are the reference's own.  A triangle that
// is not a candidate fails the reference's triangle test whenever the reference reaches it, changes nothing there, and
// may be skipped: the result is the reference's, provided pass 1 finds EVERY passing triangle.
//
// Why pass 1 finds every passing triangle: a forward error bound on the test, written out in DESIGN.md section 4.
// With eps = 2^-24, per triangle e1 = b - a, e2 = c - a as stored, ray (o, d), s = o - a and the test's computed
// g, u, v, t:   |g - g^| <= 7 eps |e1||d||e2|,   |s.q - s^.q^| <= 7.5 eps |s||d||e2|,   |d.r - d.r^| <= 8.1 eps |d||s||e1|,
// |e2.r - e2.r^| <= 8.1 eps |e2||s||e1|  (^ = exact).  A pass needs the COMPUTED |g| >= 1e-4 (:338), so with
// B = 7 eps E2 |d| / 1e-4 <= 1/2 (E2 = max |e1||e2| over the mesh) the exact barycentrics and distance differ from the
// computed ones by at most  du|e1| + dv|e2| <= rho = 2 eps |d| E2 (16 Sr + 14 L) / 1e-4 + 11 eps L  in space and
// dt = 2 (8.1 eps E2 Sr / 1e-4 + t_max (B + 2.001 eps))  along the ray  (Sr >= |o - vertex|, L = longest stored edge).
// Hence: the test passes => the exact line meets the triangle's box grown by rho at a parameter in [t_min - dt, t_max + dt].
// The device pads by 4 rho + 16 eps Sr, applied to the differences (bmin - o) - pad and (bmax - o) + pad — never to o itself:
// a mesh far from its object-space origin has |o| >> pad — (its own f32 slab arithmetic moves a box face by <= 3 eps Sr) and hands any ray with
// B > 1/2 or non-finite inputs to the plain reference walk.  Meshes whose scale makes the 1e-4 test void (obj/drone.obj in
// object space: |e1||e2||d| ~ 1e5) never qualify: there the reference can accept arbitrarily grazing triangles from
// arbitrarily far away, and no finite padding is conservative.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace pt {
namespace build {

struct V3 { float x, y, z; };
struct Box { V3 mn, mx; };

inline Box tri_box(V3 a, V3 b, V3 c) {                 // IndexedTriangle::bounding_box geometry.rs:367-381
    Box r;
    r.mn = V3{ fminf(a.x, fminf(b.x, c.x)), fminf(a.y, fminf(b.y, c.y)), fminf(a.z, fminf(b.z, c.z)) };
    r.mx = V3{ fmaxf(a.x, fmaxf(b.x, c.x)), fmaxf(a.y, fmaxf(b.y, c.y)), fmaxf(a.z, fmaxf(b.z, c.z)) };
    return r;
}
inline Box surround(const Box& a, const Box& b) {      // AABB::aabb_surrounding geometry.rs:28-41
    Box r;
    r.mn = V3{ fminf(a.mn.x, b.mn.x), fminf(a.mn.y, b.mn.y), fminf(a.mn.z, b.mn.z) };
    r.mx = V3{ fmaxf(a.mx.x, b.mx.x), fmaxf(a.mx.y, b.mx.y), fmaxf(a.mx.z, b.mx.z) };
    return r;
}
inline void put_node(float* n, const Box& b, int w0, int w1) {
    n[0] = b.mn.x; n[1] = b.mn.y; n[2] = b.mn.z; memcpy(&n[3], &w0, 4);
    n[4] = b.mx.x; n[5] = b.mx.y; n[6] = b.mx.z; memcpy(&n[7], &w1, 4);
}
inline Box get_box(const float* n) { return Box{ V3{ n[0], n[1], n[2] }, V3{ n[4], n[5], n[6] } }; }

// StaticMesh::build_bvh_helper geometry.rs:190-217, emitted in DFS pre-order into `nodes` (8 floats per node; skip links
// are indices into `nodes`).  The reference's random-axis sort (:200-207) only permutes a scratch vector whose entries are
// never read back (the leaf is built from `idx: start`, :194), so the tree is this index-range median split whatever the
// RNG does.
struct RefTree {
    const float* positions; const uint32_t* indices;
    std::vector<float>* nodes;
    V3 vpos(int tri, int corner) const { const float* p = &positions[3 * (size_t)indices[3 * (size_t)tri + corner]]; return V3{ p[0], p[1], p[2] }; }
    Box build(int start, int end) {
        const size_t me = nodes->size() / 8;
        nodes->resize(nodes->size() + 8);
        Box box;
        int tri = -1;
        if (end - start == 1) {                                         // :192
            box = tri_box(vpos(start, 0), vpos(start, 1), vpos(start, 2));   // :195
            tri = start;                                                // :194
        } else {
            const int mid = start + (end - start) / 2;                  // :209
            const Box l = build(start, mid);                            // :210
            const Box r = build(mid, end);                              // :211
            box = surround(l, r);                                       // :212
        }
        const int skip = (int)(nodes->size() / 8);                      // first node after this subtree
        put_node(&(*nodes)[me * 8], box, skip, tri);
        return box;
    }
};

// Per-mesh constants of the padding bound (computed in double, rounded up).
struct FConst { float E2, L, cx, cy, cz, R; };

// Binned-SAH hierarchy over the triangles' boxes, DFS pre-order with skip links, appended to `fnodes` (8 floats per node:
// {bmin.xyz, skip}{bmax.xyz, leaf}); leaf = (first << 3) | (count - 1) >= 0 indexes `ftris`, interior = -1.  Skip links are
// indices into `fnodes`.  (The builder's form; the device gets the 16-byte quantised nodes of fq_encode below.)  `ftris` receives the leaves' triangles in leaf order, 12 floats each: {a.xyz, bits of the
// triangle's index}{e1.xyz, 0}{e2.xyz, 0} — the reference's operands exactly as the reference walk reads them.
struct FTree {
    const float* tris;                   // this mesh's triangles, 12 floats each: {a, 0}{e1, 0}{e2, 0}
    int n_tris;
    std::vector<float>* fnodes;
    std::vector<float>* ftris;
    int ftri_base;                       // index (in triangles) of this mesh's first entry in ftris
    int leaf_max;
    std::vector<int> order;
    std::vector<V3> cen;
    std::vector<Box> tb;

    static float area(const Box& b) {
        const float dx = b.mx.x - b.mn.x, dy = b.mx.y - b.mn.y, dz = b.mx.z - b.mn.z;
        return 2.0f * (dx * dy + dy * dz + dz * dx);
    }
    static float up(double v) { float f = (float)v; if ((double)f < v) f = nextafterf(f, INFINITY); return f; }
    static float dn(double v) { float f = (float)v; if ((double)f > v) f = nextafterf(f, -INFINITY); return f; }
    // bounds of the triangle (a, a + e1, a + e2) with e1, e2 as STORED (that is the triangle the test works on), outward rounded
    static Box stored_box(const float* T) {
        double lo[3], hi[3];
        for (int k = 0; k < 3; k++) {
            const double a = T[k], b = (double)T[k] + (double)T[4 + k], c = (double)T[k] + (double)T[8 + k];
            lo[k] = std::min(a, std::min(b, c)); hi[k] = std::max(a, std::max(b, c));
        }
        return Box{ V3{ dn(lo[0]), dn(lo[1]), dn(lo[2]) }, V3{ up(hi[0]), up(hi[1]), up(hi[2]) } };
    }
    FConst run() {
        order.resize((size_t)n_tris); cen.resize((size_t)n_tris); tb.resize((size_t)n_tris);
        double E2 = 0.0, L = 0.0, lo[3] = { 1e300, 1e300, 1e300 }, hi[3] = { -1e300, -1e300, -1e300 };
        for (int i = 0; i < n_tris; i++) {
            const float* T = tris + (size_t)i * 12;
            order[(size_t)i] = i;
            tb[(size_t)i] = stored_box(T);
            const Box& b = tb[(size_t)i];
            cen[(size_t)i] = V3{ 0.5f * (b.mn.x + b.mx.x), 0.5f * (b.mn.y + b.mx.y), 0.5f * (b.mn.z + b.mx.z) };
            const double l1 = std::sqrt((double)T[4] * T[4] + (double)T[5] * T[5] + (double)T[6] * T[6]);
            const double l2 = std::sqrt((double)T[8] * T[8] + (double)T[9] * T[9] + (double)T[10] * T[10]);
            E2 = std::max(E2, l1 * l2); L = std::max(L, std::max(l1, l2));
            const float bl[3] = { b.mn.x, b.mn.y, b.mn.z }, bh[3] = { b.mx.x, b.mx.y, b.mx.z };
            for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], (double)bl[k]); hi[k] = std::max(hi[k], (double)bh[k]); }
        }
        build(0, n_tris);
        FConst c;
        const double cx = 0.5 * (lo[0] + hi[0]), cy = 0.5 * (lo[1] + hi[1]), cz = 0.5 * (lo[2] + hi[2]);
        const double R = 0.5 * std::sqrt((hi[0] - lo[0]) * (hi[0] - lo[0]) + (hi[1] - lo[1]) * (hi[1] - lo[1]) + (hi[2] - lo[2]) * (hi[2] - lo[2]));
        c.E2 = up(E2 * (1.0 + 1e-6)); c.L = up(L * (1.0 + 1e-6));
        c.cx = (float)cx; c.cy = (float)cy; c.cz = (float)cz;
        c.R = up(R * (1.0 + 1e-5) + 1e-6 * (std::fabs(cx) + std::fabs(cy) + std::fabs(cz)));      // covers the rounding of the centre
        return c;
    }
    Box build(int lo, int hi) {
        const size_t me = fnodes->size() / 8;
        fnodes->resize(fnodes->size() + 8);
        Box box;
        int leaf = -1;
        if (hi - lo <= leaf_max) {
            const int first = (int)(ftris->size() / 12) - ftri_base;
            for (int i = lo; i < hi; i++) {
                const int t = order[(size_t)i];
                const float* T = tris + (size_t)t * 12;
                float rec[12];
                memcpy(rec, T, sizeof rec);
                memcpy(&rec[3], &t, 4);
                ftris->insert(ftris->end(), rec, rec + 12);
                box = (i == lo) ? tb[(size_t)t] : surround(box, tb[(size_t)t]);
            }
            leaf = (first << 3) | (hi - lo - 1);
        } else {
            const int mid = split(lo, hi);
            const Box l = build(lo, mid);
            const Box r = build(mid, hi);
            box = surround(l, r);
        }
        put_node(&(*fnodes)[me * 8], box, (int)(fnodes->size() / 8), leaf);
        return box;
    }
    // partitions order[lo, hi) and returns the split point (lo < mid < hi)
    int split(int lo, int hi) {
        constexpr int kBins = 16;
        V3 cmn = cen[(size_t)order[(size_t)lo]], cmx = cmn;
        for (int i = lo + 1; i < hi; i++) {
            const V3 c = cen[(size_t)order[(size_t)i]];
            cmn = V3{ fminf(cmn.x, c.x), fminf(cmn.y, c.y), fminf(cmn.z, c.z) };
            cmx = V3{ fmaxf(cmx.x, c.x), fmaxf(cmx.y, c.y), fmaxf(cmx.z, c.z) };
        }
        float best_cost = INFINITY; int best_axis = -1, best_bin = -1;
        const float lo3[3] = { cmn.x, cmn.y, cmn.z }, hi3[3] = { cmx.x, cmx.y, cmx.z };
        for (int axis = 0; axis < 3; axis++) {
            const float ext = hi3[axis] - lo3[axis];
            if (!(ext > 0.0f) || !std::isfinite(ext)) continue;
            Box bb[kBins]; int cnt[kBins]; bool used[kBins];
            for (int b = 0; b < kBins; b++) { cnt[b] = 0; used[b] = false; }
            const float scale = (float)kBins / ext;
            for (int i = lo; i < hi; i++) {
                const int a = order[(size_t)i];
                const float c = axis == 0 ? cen[(size_t)a].x : axis == 1 ? cen[(size_t)a].y : cen[(size_t)a].z;
                int b = (int)((c - lo3[axis]) * scale);
                b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
                bb[b] = used[b] ? surround(bb[b], tb[(size_t)a]) : tb[(size_t)a];
                used[b] = true; cnt[b]++;
            }
            float la[kBins]; int ln[kBins];
            Box acc{}; bool have = false; int n = 0;
            for (int b = 0; b < kBins; b++) {
                if (used[b]) { acc = have ? surround(acc, bb[b]) : bb[b]; have = true; n += cnt[b]; }
                la[b] = have ? area(acc) : 0.0f; ln[b] = n;
            }
            have = false; n = 0;
            for (int b = kBins - 1; b >= 1; b--) {
                if (used[b]) { acc = have ? surround(acc, bb[b]) : bb[b]; have = true; n += cnt[b]; }
                if (n == 0 || ln[b - 1] == 0) continue;
                const float cost = la[b - 1] * (float)ln[b - 1] + area(acc) * (float)n;
                if (cost < best_cost) { best_cost = cost; best_axis = axis; best_bin = b; }
            }
        }
        int mid;
        if (best_axis < 0) mid = lo + (hi - lo) / 2;        // all centroids equal: any split is as good
        else {
            const float ext = hi3[best_axis] - lo3[best_axis], scale = (float)kBins / ext;
            auto bin_of = [&](int a) {
                const float c = best_axis == 0 ? cen[(size_t)a].x : best_axis == 1 ? cen[(size_t)a].y : cen[(size_t)a].z;
                int b = (int)((c - lo3[best_axis]) * scale);
                return b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
            };
            mid = (int)(std::partition(order.begin() + lo, order.begin() + hi, [&](int a) { return bin_of(a) < best_bin; }) - order.begin());
            if (mid <= lo || mid >= hi) mid = lo + (hi - lo) / 2;
        }
        return mid;
    }
};

// What the device walks: the F-nodes QUANTISED to 16 bytes — {qmin.x | qmin.y << 16, qmin.z | qmax.x << 16, qmax.y | qmax.z << 16, link},
// link = the skip index of an interior node, 0x80000000 | (first << 3 | count - 1) for a leaf (whose successor is the next node
// either way).  A coordinate decodes as fmaf((float)q, s, b): ONE correctly rounded operation, the same on host and device, with
// s a power of two per mesh and b the F-root's lower corner.  The encoder rounds OUTWARD against that very decode, so the decoded
// box CONTAINS the node's box, and the padded slab test — evaluated on the decoded box with the arithmetic it always had — is
// monotone in the box (IEEE rounding is monotone: a lower bmin or higher bmax can only move an axis' entry earlier and its exit
// later): every ray that passes the test on the exact box passes it on the decoded one.  The bound above therefore holds
// unchanged; the grid costs a few false-positive node visits (1/65534 of the mesh's extent per face).  Why quantise at all: the
// walk is bound by L1 tag look-ups — one per lane and load INSTRUCTION — and a 32-byte node takes two loads.
struct FQuant { float s, bx, by, bz; };
inline float fq_decode(uint32_t q, float s, float b) { return fmaf((float)q, s, b); }
// `fnodes`: n float nodes as FTree emits them (node 0 = the root, whose box bounds every other); `out` receives 4 words per node.
// false: the mesh cannot be put on the grid (non-finite bounds, or a face that does not fit 16 bits) — it does not qualify then.
inline bool fq_encode(const float* fnodes, size_t n, FQuant* g, std::vector<uint32_t>* out) {
    if (n == 0) { *g = FQuant{ 1.0f, 0.0f, 0.0f, 0.0f }; return true; }
    const Box root = get_box(fnodes);
    const double ext = std::max((double)root.mx.x - (double)root.mn.x, std::max((double)root.mx.y - (double)root.mn.y, (double)root.mx.z - (double)root.mn.z));
    if (!std::isfinite(ext) || !std::isfinite(root.mn.x) || !std::isfinite(root.mn.y) || !std::isfinite(root.mn.z)) return false;
    int k = -126;
    if (ext > 0.0) { int e; (void)std::frexp(ext / 65534.0, &e); k = std::max(e, -126); }      // ext / 65534 <= 2^e
    if (k > 100) return false;
    g->s = std::ldexp(1.0f, k); g->bx = root.mn.x; g->by = root.mn.y; g->bz = root.mn.z;
    const float b3[3] = { g->bx, g->by, g->bz };
    out->reserve(out->size() + n * 4);
    for (size_t i = 0; i < n; i++) {
        const float* nd = fnodes + i * 8;
        uint32_t q[6];
        for (int a = 0; a < 3; a++) {
            const float lo = nd[a], hi = nd[4 + a];
            if (!std::isfinite(lo) || !std::isfinite(hi)) return false;
            double ql = std::floor(((double)lo - (double)b3[a]) / (double)g->s), qh = std::ceil(((double)hi - (double)b3[a]) / (double)g->s);
            ql = std::min(std::max(ql, 0.0), 65535.0); qh = std::min(std::max(qh, 0.0), 70000.0);
            uint32_t l = (uint32_t)ql, h = (uint32_t)qh;
            while (l > 0 && fq_decode(l, g->s, b3[a]) > lo) l--;
            if (fq_decode(l, g->s, b3[a]) > lo) return false;
            while (h <= 65535u && fq_decode(h, g->s, b3[a]) < hi) h++;
            if (h > 65535u) return false;
            q[a] = l; q[3 + a] = h;
        }
        int skip, leaf; memcpy(&skip, &nd[3], 4); memcpy(&leaf, &nd[7], 4);
        out->push_back(q[0] | (q[1] << 16)); out->push_back(q[2] | (q[3] << 16)); out->push_back(q[4] | (q[5] << 16));
        out->push_back(leaf >= 0 ? (0x80000000u | (uint32_t)leaf) : (uint32_t)skip);
    }
    return true;
}
inline Box fq_box(const uint32_t* w, const FQuant& g) {
    return Box{ V3{ fq_decode(w[0] & 0xffffu, g.s, g.bx), fq_decode(w[0] >> 16, g.s, g.by), fq_decode(w[1] & 0xffffu, g.s, g.bz) },
                V3{ fq_decode(w[1] >> 16, g.s, g.bx), fq_decode(w[2] & 0xffffu, g.s, g.by), fq_decode(w[2] >> 16, g.s, g.bz) } };
}

// The per-ray padding of pass 1 (the same f32 expression runs on the device: pt_kernels.hip two_stage_pad).
// Returns false when the bound does not apply to this ray (B > 1/2, or anything non-finite): reference walk.
struct FPad { float rho, dt; };
inline bool two_stage_pad(const FConst& c, const float o[3], const float d[3], float t_max, FPad* out) {
    const float eps = 5.9604645e-08f;                                  // 2^-24
    const float dn = sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]) * 1.000001f;
    const float ox = o[0] - c.cx, oy = o[1] - c.cy, oz = o[2] - c.cz;
    const float Sr = (sqrtf((ox * ox + oy * oy) + oz * oz) + c.R) * 1.000001f;
    const float B = 7.0f * eps * c.E2 * dn * 1.0e4f;
    const float rho = 2.0f * eps * dn * c.E2 * (16.0f * Sr + 14.0f * c.L) * 1.0e4f + 11.0f * eps * c.L;
    const float dt = 2.0f * (8.1f * eps * c.E2 * Sr * 1.0e4f + fabsf(t_max) * (B + 2.001f * eps));
    out->rho = 4.0f * rho + 16.0f * eps * Sr;
    out->dt = 2.0f * dt;
    return (B <= 0.5f) && (Sr <= 1.0e12f) && (out->rho <= 1.0e30f) && (out->dt <= 1.0e30f);      // false for NaN
}

}  // namespace build
}  // namespace pt
