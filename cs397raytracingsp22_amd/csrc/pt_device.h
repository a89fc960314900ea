// pt_device.h — device-resident scene layout shared by the host scene compiler
// (mi_rt.cpp) and the HIP kernels (pt_kernels.hip).  gfx950 only.
//
// Layout rules (DESIGN.md "Data layout in HBM"):
//  * The linear object list of Scene.objects (tracing.rs:215,330) is an array of
//    fixed 64-byte records read at a wave-uniform index, so every lane of a wave
//    reads the same record and the compiler serves it with scalar (SMEM) loads.
//  * Each mesh BVH (reference topology, geometry.rs:190-217) is stored THREADED in
//    DFS pre-order: node i's left child is i+1 and `skip` is the node that follows i's
//    subtree.  The reference always descends left-then-right (geometry.rs:105-115),
//    so a skip link replaces the traversal stack exactly.
//  * Nodes are 2 x float4 (32 B), triangles 3 x float4 (a, e1, e2 — 48 B) so a lane
//    fetches a node / triangle with 16-byte loads from LDS or L1/L2.  Leaf nodes hold {a, skip}{e1, tri} (see DScene.e2s).
//  * Per-triangle shading attributes (normals, uvs, tangent) are a separate array
//    touched once per mesh HIT, not per candidate.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>

// Read-only scene data is typed as CONSTANT address space on the device: a read at a
// wave-uniform address then compiles to a scalar (SMEM) load into SGPRs even when the kernel
// also stores to global memory (with plain global pointers the compiler must assume the
// stores may alias and falls back to per-lane vector loads).  Same 8-byte pointers on the host.
#if defined(__HIP_DEVICE_COMPILE__)
#define PT_CONST_AS __attribute__((address_space(4)))
#else
#define PT_CONST_AS
#endif

namespace pt {

constexpr int kTile = 32;            // MI_TILE
constexpr int kTilePixels = kTile * kTile;
constexpr int kBlock = 256;          // threads per workgroup = 4 waves of 64
constexpr int kBlocksPerTile = kTilePixels / kBlock;

enum : int { OBJ_SPHERE = 0, OBJ_TRIANGLE = 1, OBJ_PLANE = 2, OBJ_VOLUME = 3, OBJ_MESH = 4 };
enum : int { MAT_LAMBERTIAN = 0, MAT_METAL = 1, MAT_DIELECTRIC = 2, MAT_PARAMETERIZED = 3, MAT_ISOTROPIC = 4 };

// One entry of Scene.objects, 16 words.  Word use by kind:
//   SPHERE   f[0..2] center, f[3] radius, f[4] radius*radius
//   TRIANGLE f[0..2] a, f[3..5] e1=b-a, f[6..8] e2=c-a, f[9..11] normalize(e1 x e2)
//   PLANE    f[0..2] point, f[3..5] normal
//   VOLUME   f[0..2] boundary center, f[3] radius, f[4] radius*radius, f[5] -1/density; ref = -1: that inline sphere is the boundary,
//            else ref = first of int(f[6]) boundary records in DScene.bobjs (a Triangle / Plane / StaticMesh, or a nested Scene's entries)
//   MESH     ref = mesh index
// Every derived value is computed on the host with the same f32 operation the
// reference performs per ray (geometry.rs:400,434-435,449,517), so hoisting it is exact.
struct alignas(16) DObject {
    int32_t kind;
    int32_t material;    // material index (phase function for VOLUME); unused for MESH
    int32_t ref;         // mesh index for MESH
    int32_t index;       // position in Scene.objects (used by the kind-grouped `list` copy)
    float   f[12];
};
static_assert(sizeof(DObject) == 64, "DObject must be 64 bytes");

// Material, 16 words: kind, albedo, emission, roughness, metallic, ior, albedo/PI
struct alignas(16) DMaterial {
    int32_t kind;
    float   albedo[3];
    float   emission[3];
    float   roughness;
    float   metallic;
    float   ior;
    float   albedo_over_pi[3];   // materials.rs:41,128 `self.albedo / PI`, hoisted (exact: same f32 division)
    float   pad[3];
};
static_assert(sizeof(DMaterial) == 64, "DMaterial must be 64 bytes");

struct DTexture {
    uint32_t offset;     // byte offset of the RGB8 texels in the texel pool
    int32_t  width;
    int32_t  height;
    int32_t  pad;
};

// StaticMesh (geometry.rs:127-134)
struct alignas(16) DMesh {
    float    transform[16];      // column-major
    float    inv_transform[16];  // column-major
    int32_t  node_begin;         // index of the root node in the node pool
    int32_t  node_end;           // one past the last node of this mesh
    int32_t  tri_begin;          // first triangle in the triangle pools
    int32_t  n_tris;
    int32_t  material;           // -1 = ParameterizedMaterial from textures
    int32_t  tex[5];             // -1 = None
    int32_t  object_index;       // position in Scene.objects (tie-breaking)
    int32_t  e2_begin;           // first entry of this mesh in the e2 pool (leaf-embedded triangles, see DScene.e2s)
    // When all the maps this mesh binds have the same size, their texels are ALSO stored interleaved, 16 bytes per texel:
    // {albedo.rgb, metallic.x}{emission.rgb, roughness.x}{normal.rgb, 0}{0}: one 16-byte fetch per hit instead of up to five
    // 4-byte fetches from five different cache lines (incoherent rays: every fetch is a line of its own).  Same bytes, same
    // (float)b / 255 conversions, same texel index -> same values.  -1 = not available (sizes differ / fixed material).
    int32_t  tex_comb;           // index into DScene.textures of the interleaved array
    int32_t  i_root;             // id of this mesh's root in the split pools (DScene.inodes / lnodes): >= 0 interior record, < 0 = ~leaf record
    int32_t  pad2[2];
};
static_assert(sizeof(DMesh) % 16 == 0, "DMesh must be 16-byte sized");

// Two-stage traversal data of a StaticMesh (bvh_build.hpp FTree): where its F-tree lives and the constants of the
// per-ray padding bound.  qualifies = 0: the bound never applies to this mesh (object-space scale makes the reference's
// absolute 1e-4 determinant test void) and it is always walked through the reference's own tree.
struct alignas(16) DMeshF {
    int32_t fnode_begin, fnode_end;   // this mesh's F-nodes in the F-node pool
    int32_t ftri_begin;               // first entry of this mesh in the F-ordered triangle pool
    int32_t qualifies;
    float   E2, L;                    // max |e1||e2|, longest stored edge
    float   cx, cy, cz, R;            // bounding sphere of the vertices (object space)
    float   pad[2];
    float   qs, qbx, qby, qbz;        // grid of the quantised F-nodes: coordinate = fmaf(q, qs, qb) (bvh_build.hpp fq_encode)
};
static_assert(sizeof(DMeshF) == 64, "DMeshF must be 64 bytes");

// per-triangle shading attributes, 20 floats = 5 float4
struct alignas(16) DTriAttr {
    float na[3], nb[3], nc[3];   // corner normals          geometry.rs:350
    float ta[2], tb[2], tc[2];   // corner uvs              geometry.rs:355
    float tan[3];                // StaticMesh::get_tangent geometry.rs:245-250, hoisted (exact)
    float pad[2];
};
static_assert(sizeof(DTriAttr) == 80, "DTriAttr must be 80 bytes");

struct DScene {
    const PT_CONST_AS DObject*   objects;
    const PT_CONST_AS DMaterial* materials;
    const PT_CONST_AS DMesh*     meshes;
    const PT_CONST_AS float*     nodes;      // float4 pairs: {bmin.xyz, skip(int)} {bmax.xyz, tri(int, -1 = interior)}
    const PT_CONST_AS float*     tris;       // float4 triples: {a.xyz, 0} {e1.xyz, 0} {e2.xyz, 0}
    // LEAF nodes carry their triangle instead of a box (the reference never box-tests a leaf, geometry.rs:95, so no
    // kernel reads a leaf's box): {a.xyz, skip}{e1.xyz, tri}; the third operand comes from this pool, float4 {e2.xyz, 0} per
    // triangle, ordered like the node pool.  A walker that has just fetched a leaf node from LDS then needs ONE more
    // 16-byte LDS read for the triangle test — no trip to global memory inside the walk.
    const PT_CONST_AS float*     e2s;
    // The same trees once more with interior nodes and leaves in SEPARATE pools, every link explicit (wf_trav_i: only the
    // interior records are staged in LDS — half the bytes, so a tree that needs one 1024-thread block per CU as a whole image
    // runs two of them).  A link is an id: >= 0 = interior record, < 0 = ~(leaf record), kIdEnd = the walk of this mesh is over.
    //   inodes: float4 pairs {bmin.xyz, id on a miss (skip link)}{bmax.xyz, id on a hit (left child)}
    //   lnodes: float4 triples {a.xyz, id of the next node}{e1.xyz, triangle index}{e2.xyz, 0}
    const PT_CONST_AS float*     inodes;
    const PT_CONST_AS float*     lnodes;
    const PT_CONST_AS DTriAttr*  triattr;
    const PT_CONST_AS DTexture*  textures;
    const PT_CONST_AS uint8_t*   texels;
    // the non-mesh objects again, GROUPED BY KIND (triangles, spheres, planes, volumes; stable
    // within a kind), each record carrying its Scene.objects index: the hit loop runs one
    // tight loop per kind with no per-object dispatch.  Exact: the closest hit is order
    // independent, ties go to the lower Scene.objects index (tracing.rs:335), and only
    // volumes draw random numbers, in their original relative order (geometry.rs:517).
    const PT_CONST_AS DObject*   list;
    // boundary records of ConvexVolumes whose boundary is not the inline sphere (same record format; MESH: ref = entry of `meshes`
    // at or behind n_meshes — the table holds the Scene.objects meshes first, then the boundary-only ones)
    const PT_CONST_AS DObject*   bobjs;
    // sample_hemisphere's rotation for the two normals of every list Triangle, hoisted to the host (mi_rt.cpp): 3 float4 per entry,
    // entry 2 i + (frontface ? 0 : 1) of Scene.objects entry i
    const PT_CONST_AS float*     obj_rot;
    // two-stage traversal (meshes that qualify): F-tree nodes (same 2 x float4 node format; leaf word = (first << 3) | (count - 1)
    // into ftris), the F-ordered triangles {a.xyz, tri index}{e1.xyz, 0}{e2.xyz, 0}, and the per-mesh constants
    const PT_CONST_AS float*     fnodes;
    const PT_CONST_AS float*     ftris;
    const PT_CONST_AS DMeshF*    meshf;
    int32_t n_list_tri, n_list_sphere, n_list_plane, n_list_volume;
    // Top-level tree over the list's Triangles (long lists only): the first n_list_lin Triangles of `list` (the large
    // ones) are tested one by one, the others sit in a padded SAH tree whose record is meshf[top_meshf] (the F-tree machinery of the
    // two-stage mesh traversal, in world space); top_meshf = -1: no tree, n_list_lin = n_list_tri
    int32_t n_list_lin, top_meshf;
    int32_t n_objects;
    int32_t n_meshes;
    int32_t n_nodes;
    int32_t n_tris;
    int32_t n_fnodes;
};

// Camera::generate_rays constants (tracing.rs:160-163,187-191), computed once on the
// host with the reference's f32 operations.
struct DCamera {
    float eye[3];
    float rot[9];          // Matrix3::from_cols(normalize(view x up), up, -view), column-major
    float pixel_size;      // 1 / H
    float n;               // aa_sample_count as f32
    float rootn;           // sqrt(n)
    float half_rootn;      // 0.5 * rootn
    float half_n;          // 0.5 * n
    float cx_base;         // -0.5*W   (x as f32 + cx_base + 0.5)
    float cy_base;         // 0.5 + 0.5*H
    float focal_length;
    float focus_dist;
    float lens_radius;
    float max_trace_dist;
    uint32_t rootn_u;      // rootn as u32
    uint32_t spp;          // aa_sample_count
    uint32_t zone;         // rand UniformInt rejection zone for range = spp
    uint32_t path_depth;
    uint32_t width, height;
    // CameraProjectionMode::Orthographic (tracing.rs:196,200,204): origin = (centre.x, centre.y, 0) in
    // camera space as it is, direction = rot * view_dir (the constant, rotated once on the host)
    uint32_t ortho;
    float ortho_dir[3];
    // ShadingMode::Phong (tracing.rs:277-297): Scene.point_light_pos / ambient
    float light[3];
    float ambient[3];
};

struct DRender {
    uint32_t seed;
    int32_t  rank, world;
    uint32_t tiles_x, tiles_y, tiles_total;
    uint32_t my_tiles;       // tiles owned by this rank
    uint32_t lds_nodes;      // number of BVH nodes staged into LDS (0 = read from global)
    uint32_t lds_tris;       // number of triangles staged into LDS
    uint32_t vote_t, vote_a; // VOTED kernel: run the BVH phase when n_trav*vote_t >= n_a*vote_a
    uint32_t k_steps;        // VOTED kernel: node micro-steps per BVH trip
};

// kernel argument block of K1 (passed by value: lives in the kernarg segment / SGPRs)
struct K1Args {
    DScene  S;
    DCamera C;
    DRender R;
    uint32_t seed_key;   // lowbias32(seed ^ 0x68e31da4)
    float*    out;       // [tiles_padded][1024][3]
    uint32_t* sig;       // [tiles_padded][1024] or nullptr
    unsigned long long* diag;   // 16 counters (diagnostic build only) or nullptr
};

// ---- wavefront pipeline (pt_kernels.hip "K1w") ----
// Path state streamed through HBM as 6 float4 planes [6][cap] (coalesced 16-byte accesses):
//   q0 o.xyz d.x | q1 d.yz T.xy | q2 T.z L.xyz | q3 rng.s0 rng.s1 pix sample|depth<<16
//   plane-4 array: packed sub-arrays {t, obj} 8 B x cap | tri 4 B x cap (class B only) | signature 4 B x cap; q5 best.u best.v mesh
constexpr int kWfPlanes = 6;
// Appends are SHARDED: block b of wf_main appends its survivors to region (b % kWfShards) of
// st_out with that shard's own counter, so no single address sees more than ~1/256 of the
// atomics (one word saturates at ~88 atomics/us on gfx950).  A shard's region can never
// overflow: it receives at most the paths its own input blocks hold.
constexpr int kWfShards = 256;
constexpr int kCandMax = 8;          // two-stage: candidate slots per queued ray (more passing triangles -> reference walk for that mesh)
constexpr int kTwoStageMaxMeshes = 24;   // mesh index bits in cand_hdr
constexpr int32_t kIdEnd = (int32_t)0x80000000;   // split pools: "no further node"
constexpr int kPairStride = 56;                    // wf_trav_i<.., PAIR>: bytes of an interior record in the paired LDS layout (pt_kernels.hip)
struct WfArgs {
    DScene  S;
    DCamera C;
    DRender R;
    uint32_t seed_key;
    float4* st_in;        // state of the live paths of the previous iteration (nullptr at iteration 0)
    float4* st_out;       // state of the paths that continue, compacted per shard region
    uint32_t cap;         // plane stride = kWfShards * region
    uint32_t region;      // slots per shard region (multiple of 256)
    uint32_t n_in;        // iteration 0: number of new paths (npix * s_count)
    uint32_t n_blocks_in; // iteration > 0: blocks to run = sum over shards of ceil(count/256)
    uint32_t part;        // a pass after the first may be launched in two parts: 0 = all blocks, 1 = the class-A blocks only, 2 = the class-B blocks
                          // only.  The grid may be larger than the part (a host-side upper bound): surplus blocks return at once
    uint32_t iter0;       // 1 = generate camera rays (first iteration of a batch)
    uint32_t s_base;      // first sample index of this batch
    uint32_t s_count;     // samples per pixel in this batch
    uint32_t npix;        // tile-major pixels of this rank = tiles_padded * 1024
    uint32_t refill_min;  // wf_trav: refill idle lanes only when at least this many are idle (amortises the gather latency)
    uint32_t fuse_max;    // wf_main: at most this many further shade + intersect rounds inside one launch ...
    uint32_t fuse_min;    // ... each taken only while at least this many lanes of the wave can continue
    // Every shard region is filled from BOTH ENDS: class A (the next shade is a plain Triangle /
    // Plane hit: the common, cheap case) grows from the front, class B (sphere hits, rays waiting
    // for a mesh walk, volumes) from the back; A + B can never exceed the region.  The next pass
    // runs all A blocks, then all B blocks, so most waves execute one shading branch only.
    const PT_CONST_AS uint32_t* in_count;    // [2*kWfShards] live paths per (class, shard) of st_in: A shards, then B shards
    const PT_CONST_AS uint32_t* in_blkpfx;   // [2*kWfShards + 1] exclusive prefix of ceil(count/256)
    uint32_t* out_count;  // [2*kWfShards] appended to st_out per (class, shard)
    uint32_t* trav_count; // [kWfShards] statistics: path segments (Scene::intersect_ray evaluations) wf_main ran, per output shard; summed and zeroed by wf_prefix.  trav_head sits behind it
    uint32_t* trav_head;  // [0]: consumption head over the CONCATENATED per-shard queues (wf_trav grabs 256 entries per atomic); [1]: the same for wf_trav_f
    const PT_CONST_AS uint32_t* trav_pfx;   // [kWfShards + 1] exclusive prefix of the class-B counts (statistics)
    const PT_CONST_AS uint32_t* hdr;   // [4] written by wf_prefix after every wf_main: blocks of the next pass, live paths, class-B paths (the walkers' work list)
    float4* samp;         // [s_count][npix] finished samples: L.xyz, signature bits
    float4* accum;        // [npix] running per-pixel sum (xyz) and signature sum (w bits)
    float*    out;        // compact framebuffer [tiles_padded][1024][3]
    uint32_t* sig;        // or nullptr
    // Primary rays only: bit k of tile_mask[global tile] = entry k of the kind-grouped list can be hit by a
    // camera ray of that 32x32 tile (conservative host-side frustum test; nullptr = test everything).
    // tile_mask[tiles_total + tile]: low 32 bits = the same for the meshes' root boxes, bit 63 = nothing
    // at all is reachable from this tile.
    const PT_CONST_AS unsigned long long* tile_mask;
    // mesh walks: bit m of trav_mask = mesh m is walked by THIS launch (wf_trav: the meshes walked through the reference's
    // tree; wf_trav_f / wf_replay: the two-stage meshes).  Results of successive launches merge through the hit record.
    uint32_t trav_mask;
    // two-stage: candidates of queue entry vi = cand[vi * kCandMax .. ) as {t, (mesh << 24) | triangle}; cand_hdr[vi] =
    // {position of the path in st_out, count | bit (8 + m): mesh m must take the reference walk (overflow, bound not applicable)}
    uint2* cand;
    uint2* cand_hdr;
    unsigned long long* diag;   // developer builds with -DPT_WF_STAMPS: [16] summed s_memtime deltas of sampled wf_main waves
};

}  // namespace pt
