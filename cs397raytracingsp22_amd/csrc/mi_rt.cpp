// mi_rt.cpp — host runtime behind the C ABI of include/mi_rt.h (libmi_rt.so).
//
//   * scene compiler: flattens the POD description of the reference's Scene.objects
//     into the device layout of pt_device.h, including the BVH of every StaticMesh with
//     the REFERENCE's topology (geometry.rs:190-217: median split of the triangle index
//     range, leaf i = triangle i, exact union boxes), emitted in DFS pre-order with skip
//     links for the stackless traversal of pt_kernels.hip.
//   * render entry points: whole image on one GPU (mi_render) and the device-pointer
//     building blocks used with one process per GPU (tiles, un-permute, tone-map).
//
// There is no CPU implementation of the render path in this library: every entry
// point that produces pixels launches HIP kernels and reports an error when it cannot.
// Compiled with hipcc, -ffp-contract=off: the values hoisted out of the per-ray code
// (e1, e2, r*r, normalize(e1 x e2), -1/density, albedo/PI, the triangle tangent, the
// camera basis) are computed here with the same f32 operations, in the same order,
// the reference performs per ray, so hoisting them does not change a single bit.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <thread>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/mi_rt.h"
#include "pt_device.h"
#include "bvh_build.hpp"

#pragma clang fp contract(off)

namespace pt {
hipError_t launch_megakernel(const K1Args& args, uint32_t n_blocks, bool lds, bool sig, size_t lds_bytes, hipStream_t stream);
hipError_t launch_megakernel_voted(const K1Args& args, uint32_t n_blocks, bool lds, bool sig, bool diag, bool gv,
                                   size_t lds_bytes, hipStream_t stream);
hipError_t launch_wf_main(const WfArgs& a, uint32_t n_blocks, bool sig, bool gv, bool tex, hipStream_t stream);
hipError_t launch_phong(const K1Args& a, uint32_t n_blocks, bool sig, hipStream_t stream);
hipError_t launch_branch(const K1Args& a, uint32_t n_blocks, uint32_t path_samples, bool sig, hipStream_t stream);
hipError_t launch_wf_trav(const WfArgs& a, uint32_t n_blocks, int lds_mode, size_t lds_bytes, bool* big_lds_enabled, hipStream_t stream);
hipError_t launch_wf_trav_i(const WfArgs& a, uint32_t n_blocks, size_t lds_bytes, bool leaf_lds, bool* big_lds_enabled, hipStream_t stream);
hipError_t launch_wf_trav_p(const WfArgs& a, uint32_t n_blocks, size_t lds_bytes, hipStream_t stream);
hipError_t launch_wf_filter_f(const WfArgs& a, uint32_t blocks_per_shard, hipStream_t stream);
hipError_t launch_wf_trav_f(const WfArgs& a, uint32_t n_blocks, hipStream_t stream);
hipError_t launch_wf_replay(const WfArgs& a, uint32_t n_blocks, hipStream_t stream);
hipError_t launch_wf_prefix(uint32_t* out_count, uint32_t* trav_count, uint32_t* in_count, uint32_t* in_blkpfx,
                            uint32_t* trav_pfx, uint32_t* hdr, uint32_t* host_hdr, uint32_t seq, hipStream_t stream);
hipError_t launch_wf_reduce(const WfArgs& a, bool first_batch, bool last_batch, hipStream_t stream);
hipError_t launch_unpermute(const float* gathered, float* image, uint32_t width, uint32_t height, uint32_t tiles_x,
                            uint32_t world, uint32_t tiles_padded, hipStream_t stream);
hipError_t launch_sig_unpermute(const uint32_t* gathered, uint32_t* image, uint32_t width, uint32_t height,
                                uint32_t tiles_x, uint32_t world, uint32_t tiles_padded, hipStream_t stream);
hipError_t launch_tonemap(const float* image, uint8_t* out, uint32_t n_pixels, float inv_gamma, hipStream_t stream);
hipError_t launch_selftest_rcp(unsigned long long* d_mismatches, hipStream_t stream);
}  // namespace pt

using namespace pt;

// ------------------------------------------------------------------ errors
static thread_local std::string g_err;
static int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    g_err = buf;
    return code;
}
#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) return fail(MI_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// ------------------------------------------------------------------ host f32 math (reference order)
namespace {
struct h3 { float x, y, z; };
inline h3 H3(float x, float y, float z) { h3 r = { x, y, z }; return r; }
inline h3 H3p(const float* p) { return H3(p[0], p[1], p[2]); }
inline h3 sub(h3 a, h3 b) { return H3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline h3 add(h3 a, h3 b) { return H3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline h3 scale(h3 a, float s) { return H3(a.x * s, a.y * s, a.z * s); }
inline float dot(h3 a, h3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline h3 cross(h3 a, h3 b) { return H3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
inline h3 normalize(h3 a) { return scale(a, 1.0f / sqrtf(dot(a, a))); }
inline uint32_t lowbias32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x;
}
}  // namespace

// ------------------------------------------------------------------ context
struct mi_ctx {
    int device = 0;
    hipStream_t stream = nullptr;            // own stream for mi_render
    hipStream_t aux_stream = nullptr;        // wavefront pipeline: the class-A part of a pass runs here, beside the walkers of the previous pass
    hipStream_t aux2_stream = nullptr;       // ... and wf_trav_f here, beside wf_trav (scenes with meshes of both kinds)
    hipEvent_t ev_pfx = nullptr, ev_part = nullptr, ev_travf = nullptr;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    bool ev_recorded = false;

    // device scene
    void* blob = nullptr; size_t blob_bytes = 0;
    DScene S{};
    bool have_scene = false;
    bool mesh_maps = false;                  // some mesh takes its material from maps or has a normal map: wf_main's MESH = 2 form
    bool gen_volumes = false;                // a ConvexVolume whose boundary is not the inline sphere: the kernels' GV forms
    bool list_tree = false;                  // the list's Triangles sit in a top-level tree (long lists): wf_main's TOP forms
    uint32_t lds_bytes = 0;                  // bytes needed to stage nodes + tris, 0 = no meshes
    // per live mesh (Scene.objects order): end of its nodes in the node pool, does the two-stage bound apply to it at all,
    // is it walked two-stage by default (qualifies and large enough for the F-tree to pay)
    std::vector<int> mesh_node_end, mesh_e2_end, mesh_inode_end; std::vector<uint8_t> mesh_qualifies, mesh_default_ts;
    void* d_cand = nullptr; size_t cand_bytes = 0;           // two-stage candidates [cap][kCandMax] {t, key}
    void* d_cand_hdr = nullptr; size_t cand_hdr_bytes = 0;   // [cap] {pos, count | flags}

    // scratch for mi_render
    float* d_compact = nullptr; size_t compact_bytes = 0;
    float* d_image = nullptr;   size_t image_bytes = 0;
    uint8_t* d_u8 = nullptr;    size_t u8_bytes = 0;
    uint32_t* d_sigc = nullptr; size_t sigc_bytes = 0;
    uint32_t* d_sigi = nullptr; size_t sigi_bytes = 0;
    unsigned long long* d_diag = nullptr;    // 16 counters of the diagnostic variant
    float point_light_pos[3] = {0.0f, 1.0f, 5.0f}, ambient[3] = {0.1f, 0.1f, 0.1f};   // Scene fields read by Phong
    hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;     // mi_render's whole-call timer
    bool big_lds_enabled = false;                    // wf_trav<2,1024>'s > 64 KB dynamic-LDS opt-in, set on THIS context's device
    bool big_lds_enabled_i = false;                  // the same for wf_trav_i<1024>
    // wavefront pipeline buffers
    void* d_wf_a = nullptr; size_t wf_a_bytes = 0;   // path state ping
    void* d_wf_b = nullptr; size_t wf_b_bytes = 0;   // path state pong
    void* d_wf_samp = nullptr; size_t wf_samp_bytes = 0;
    void* d_wf_acc = nullptr; size_t wf_acc_bytes = 0;
    uint32_t* d_wf_cnt = nullptr;
    uint32_t* h_hdr = nullptr;                       // pinned + device-mapped ring of 8-word slots: wf_prefix writes {blocks, live, queue, seq, live class B, class-A blocks}
    uint64_t wf_counts[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };   // last frame: passes, class-A paths streamed, class-B paths, queue entries, samples, pixels
    uint32_t* h_hdr_dev = nullptr;                   // its device-side address
    uint32_t hdr_seq = 0;
    // per-tile primary-ray masks over the kind-grouped list (see tile_masks)
    std::vector<DObject> h_list; int h_n_tri = 0, h_n_sphere = 0, h_n_unmasked = 0;   // planes + volumes: never masked
    struct MeshBox { bool cullable; double corner[8][3]; };
    std::vector<MeshBox> h_mesh_box;                 // world-space corners of every live mesh's root box
    std::vector<unsigned long long> h_tile_mask; void* d_tile_mask = nullptr; size_t tile_mask_bytes = 0;
    mi_camera_desc mask_cam{}; uint32_t mask_stride = 0; bool mask_valid = false;
    std::vector<hipEvent_t> wf_ev;                   // event pool for per-kernel timing of the pipeline
    float wf_ms[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };     // last frame: wf_main, wf_trav, wf_reduce totals (ms), launches, wf_trav_f, wf_replay
    int n_cus = 256;
    // Developer knobs (MI_RT_* environment variables), read ONCE in mi_ctx_create; none is needed for
    // normal operation and none changes a result — what a caller may want to control is in mi_render_opts.
    struct Tuning {
        uint32_t vote_t = 2, vote_a = 1, k_steps = 8;   // voted megakernel
        uint32_t lds_pad = 0;                           // occupancy experiments
        uint32_t refill_min = 16;                       // wf_trav: refill idle lanes when at least this many are idle (A/B round 2: 32 / 16 / 8 -> 36.9 / 35.8 / 38.6 ms on cfg2)
        uint32_t fuse_max = 0, fuse_min = 32;           // wf_main: in-launch continuation (rounds: 0 = automatic; lanes needed)
        int trav_lds = -1;                              // wf_trav LDS mode override (-1 = automatic)
        int trav_bpc = 0;                               // wf_trav blocks per CU override (0 = automatic)
        int travf_bpc = 0;                              // wf_trav_f blocks per CU override (0 = default)
        int kernel_timing = -1;                         // per-launch HIP events: -1 = single-rank renders only
        bool global_bvh = false;                        // never stage a BVH in LDS
        bool wf_stamps = false;                         // -DPT_WF_STAMPS builds: collect wf_main phase stamps
        bool debug_mask = false;                        // print tile-mask statistics
        bool dump_launches = false;                     // print every pipeline launch's duration (needs per-launch events)
        int split = 1;                                  // wf_main in two parts, class A beside the previous pass' walkers (0 = one launch per pass)
        int conc = 1, conc_trav_bpc = 0, conc_travf_bpc = 0;   // wf_trav and wf_trav_f on two streams (0 = one after the other); their blocks per CU then (0 = the usual)
        uint32_t tail_paths = 0xffffffffu;              // a pass that starts with at most this many live paths runs every path as far as it can inside the launch (0 = never; default: automatic)
        uint32_t nowait_blocks = 16384;                 // passes whose grid bound is at most this many blocks are launched without waiting for the previous header (0 = always wait)
        uint32_t spin_timeout_ms = 120000;              // header wait: give up after this long without progress
    } tune;
};

static int ensure(void** p, size_t* have, size_t want) {
    if (*have >= want && *p) return MI_OK;
    if (*p) { (void)hipFree(*p); *p = nullptr; *have = 0; }
    hipError_t e = hipMalloc(p, want);
    if (e != hipSuccess) return fail(MI_ERR_OOM, "hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    *have = want;
    return MI_OK;
}

extern "C" int mi_abi_version(void) { return MI_RT_ABI_VERSION; }
extern "C" const char* mi_last_error(void) { return g_err.c_str(); }

extern "C" void mi_ctx_destroy(mi_ctx* c);

static int ctx_init(mi_ctx* c, const hipDeviceProp_t& prop) {
    HIP_TRY(hipStreamCreate(&c->stream));
    HIP_TRY(hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&c->ev_pfx, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&c->ev_part, hipEventDisableTiming));
    HIP_TRY(hipStreamCreateWithFlags(&c->aux2_stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&c->ev_travf, hipEventDisableTiming));
    HIP_TRY(hipEventCreate(&c->ev_start));
    HIP_TRY(hipEventCreate(&c->ev_stop));
    HIP_TRY(hipEventCreate(&c->ev_t0));
    HIP_TRY(hipEventCreate(&c->ev_t1));
    HIP_TRY(hipMalloc((void**)&c->d_diag, 16 * sizeof(unsigned long long)));
    c->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    HIP_TRY(hipMalloc((void**)&c->d_wf_cnt, (9 * 256 + 64) * sizeof(uint32_t)));
    HIP_TRY(hipHostMalloc((void**)&c->h_hdr, 16 * 8 * sizeof(uint32_t), hipHostMallocMapped | hipHostMallocCoherent));     // kHdrRing slots of 8 words
    memset(c->h_hdr, 0, 16 * 8 * sizeof(uint32_t));
    HIP_TRY(hipHostGetDevicePointer((void**)&c->h_hdr_dev, c->h_hdr, 0));
    // developer knobs: read here, once (never on the render path)
    mi_ctx::Tuning& t = c->tune;
    auto env_u = [](const char* name, uint32_t& v) { if (const char* e = getenv(name)) v = (uint32_t)atoi(e); };
    auto env_i = [](const char* name, int& v) { if (const char* e = getenv(name)) v = atoi(e); };
    env_u("MI_RT_VOTE_T", t.vote_t); env_u("MI_RT_VOTE_A", t.vote_a); env_u("MI_RT_KSTEPS", t.k_steps);
    if (const char* e = getenv("MI_RT_LDS_PAD_KB")) t.lds_pad = (uint32_t)atoi(e) * 1024u;
    env_u("MI_RT_WF_REFILL", t.refill_min);
    env_u("MI_RT_WF_FUSE_MAX", t.fuse_max); env_u("MI_RT_WF_FUSE_MIN", t.fuse_min);
    env_i("MI_RT_WF_SPLIT", t.split); env_u("MI_RT_WF_NOWAIT_BLOCKS", t.nowait_blocks); env_u("MI_RT_WF_TAIL_PATHS", t.tail_paths);
    env_i("MI_RT_WF_CONC", t.conc); env_i("MI_RT_WF_CONC_TRAV_BPC", t.conc_trav_bpc); env_i("MI_RT_WF_CONC_TRAVF_BPC", t.conc_travf_bpc);
    env_i("MI_RT_WF_TRAV_LDS", t.trav_lds); env_i("MI_RT_WF_TRAV_BPC", t.trav_bpc); env_i("MI_RT_WF_TRAVF_BPC", t.travf_bpc); env_i("MI_RT_WF_KERNEL_TIMING", t.kernel_timing);
    t.global_bvh = getenv("MI_RT_GLOBAL_BVH") != nullptr;
    t.wf_stamps = getenv("MI_RT_WF_STAMPS") != nullptr;
    t.debug_mask = getenv("MI_RT_DEBUG_MASK") != nullptr;
    t.dump_launches = getenv("MI_RT_WF_DUMP_LAUNCHES") != nullptr;
    env_u("MI_RT_SPIN_TIMEOUT_MS", t.spin_timeout_ms);
    if (t.vote_t < 1) t.vote_t = 1;
    if (t.k_steps < 1) t.k_steps = 1;
    if (t.refill_min < 1) t.refill_min = 1;
    if (t.refill_min > 64) t.refill_min = 64;
    return MI_OK;
}

extern "C" int mi_ctx_create(int device, mi_ctx** out) {
    if (!out) return fail(MI_ERR_INVALID, "mi_ctx_create: out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(MI_ERR_NO_DEVICE, "no HIP device available (%s); the render path has no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(MI_ERR_INVALID, "device %d out of range (have %d)", device, n);
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(MI_ERR_NO_DEVICE, "device %d is %s; this library carries gfx950 (MI355X) code only", device, prop.gcnArchName);
    mi_ctx* c = new mi_ctx();
    c->device = device;
    const int rc = ctx_init(c, prop);
    if (rc != MI_OK) {                 // release whatever was created (the message of the failure is kept)
        const std::string msg = g_err;
        mi_ctx_destroy(c);
        g_err = msg;
        return rc;
    }
    *out = c;
    return MI_OK;
}

extern "C" void mi_ctx_destroy(mi_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    if (c->blob) (void)hipFree(c->blob);
    if (c->d_compact) (void)hipFree(c->d_compact);
    if (c->d_image) (void)hipFree(c->d_image);
    if (c->d_u8) (void)hipFree(c->d_u8);
    if (c->d_sigc) (void)hipFree(c->d_sigc);
    if (c->d_sigi) (void)hipFree(c->d_sigi);
    if (c->d_diag) (void)hipFree(c->d_diag);
    if (c->d_wf_a) (void)hipFree(c->d_wf_a);
    if (c->d_wf_b) (void)hipFree(c->d_wf_b);
    if (c->d_wf_samp) (void)hipFree(c->d_wf_samp);
    if (c->d_wf_acc) (void)hipFree(c->d_wf_acc);
    if (c->d_cand) (void)hipFree(c->d_cand);
    if (c->d_cand_hdr) (void)hipFree(c->d_cand_hdr);
    if (c->d_tile_mask) (void)hipFree(c->d_tile_mask);
    if (c->d_wf_cnt) (void)hipFree(c->d_wf_cnt);
    if (c->h_hdr) (void)hipHostFree(c->h_hdr);
    for (hipEvent_t e : c->wf_ev) (void)hipEventDestroy(e);
    if (c->ev_start) (void)hipEventDestroy(c->ev_start);
    if (c->ev_stop) (void)hipEventDestroy(c->ev_stop);
    if (c->ev_t0) (void)hipEventDestroy(c->ev_t0);
    if (c->ev_t1) (void)hipEventDestroy(c->ev_t1);
    if (c->ev_pfx) (void)hipEventDestroy(c->ev_pfx);
    if (c->ev_part) (void)hipEventDestroy(c->ev_part);
    if (c->aux_stream) (void)hipStreamDestroy(c->aux_stream);
    if (c->ev_travf) (void)hipEventDestroy(c->ev_travf);
    if (c->aux2_stream) (void)hipStreamDestroy(c->aux2_stream);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

// ------------------------------------------------------------------ scene compiler
namespace {

bool finite16(const float* m) { for (int i = 0; i < 16; i++) if (!std::isfinite(m[i])) return false; return true; }

}  // namespace

extern "C" int mi_scene_upload(mi_ctx* c, const mi_scene_desc* d) {
    if (!c || !d) return fail(MI_ERR_INVALID, "mi_scene_upload: NULL argument");
    if (d->n_objects < 0 || d->n_materials < 0 || d->n_meshes < 0 || d->n_textures < 0)
        return fail(MI_ERR_INVALID, "negative count");
    if (d->n_objects > 0 && !d->objects) return fail(MI_ERR_INVALID, "objects is NULL");
    HIP_TRY(hipSetDevice(c->device));

    std::vector<DObject> objs((size_t)d->n_objects);
    std::vector<DMaterial> mats((size_t)d->n_materials);
    std::vector<DMesh> meshes((size_t)d->n_meshes);
    std::vector<DMeshF> meshf((size_t)d->n_meshes);
    struct MeshBuild { std::vector<float> nodes, fnodes, ftris; std::vector<uint32_t> fq; bool qualifies = false, default_ts = false; int inode_end = 0; };
    std::vector<MeshBuild> mb((size_t)d->n_meshes);
    std::vector<float> nodes, tris, ftris, e2s, inodes, lnodes;
    std::vector<uint32_t> fnodes;              // the F-trees as the device walks them: 4 words per node (bvh_build.hpp fq_encode)
    std::vector<DTriAttr> attrs;
    std::vector<DTexture> texs((size_t)d->n_textures);
    std::vector<uint8_t> texels;

    const float PI = 3.14159265358979323846f;
    for (int i = 0; i < d->n_materials; i++) {
        const mi_material& s = d->materials[i];
        if (s.kind < MI_MAT_LAMBERTIAN || s.kind > MI_MAT_ISOTROPIC) return fail(MI_ERR_INVALID, "material %d: bad kind %d", i, s.kind);
        DMaterial& m = mats[(size_t)i];
        memset(&m, 0, sizeof m);
        m.kind = s.kind;
        for (int k = 0; k < 3; k++) { m.albedo[k] = s.albedo[k]; m.emission[k] = s.emission[k]; m.albedo_over_pi[k] = s.albedo[k] / PI; }
        if (s.kind == MI_MAT_DIELECTRIC) for (int k = 0; k < 3; k++) m.emission[k] = 0.0f;      // materials.rs:102
        m.roughness = s.roughness; m.metallic = s.metallic; m.ior = s.idx_of_refraction;
    }
    auto mat_ok = [&](int id) { return id >= 0 && id < d->n_materials; };

    for (int i = 0; i < d->n_textures; i++) {
        const mi_texture& t = d->textures[i];
        if (t.width <= 0 || t.height <= 0 || !t.rgb) return fail(MI_ERR_INVALID, "texture %d: bad size or NULL texels", i);
        while (texels.size() % 16) texels.push_back(0);
        texs[(size_t)i].offset = (uint32_t)texels.size();
        texs[(size_t)i].width = t.width; texs[(size_t)i].height = t.height; texs[(size_t)i].pad = 0;
        // texels are padded to RGBA8 on the device: one aligned 4-byte load per fetch instead of three byte loads
        const size_t np = (size_t)t.width * t.height, at = texels.size();
        texels.resize(at + np * 4);
        for (size_t k = 0; k < np; k++) {
            texels[at + 4 * k] = t.rgb[3 * k]; texels[at + 4 * k + 1] = t.rgb[3 * k + 1]; texels[at + 4 * k + 2] = t.rgb[3 * k + 2];
            texels[at + 4 * k + 3] = 255;
        }
    }

    // meshes: BVH + de-indexed triangle pools
    for (int mi = 0; mi < d->n_meshes; mi++) {
        const mi_mesh& s = d->meshes[mi];
        if (!s.positions || !s.normals || !s.texcoords || !s.indices || s.n_triangles < 1 || s.n_vertices < 1)
            return fail(MI_ERR_INVALID, "mesh %d: positions, normals, texcoords and indices are all required (geometry.rs:350,355)", mi);
        for (size_t k = 0; k < 3 * (size_t)s.n_triangles; k++)
            if (s.indices[k] >= (uint32_t)s.n_vertices) return fail(MI_ERR_INVALID, "mesh %d: index %u out of range", mi, s.indices[k]);
        if (!finite16(s.transform) || !finite16(s.inv_transform)) return fail(MI_ERR_INVALID, "mesh %d: non-finite transform", mi);
        if (s.material >= d->n_materials) return fail(MI_ERR_INVALID, "mesh %d: bad material", mi);
        DMesh& M = meshes[(size_t)mi];
        memset(&M, 0, sizeof M);
        memcpy(M.transform, s.transform, sizeof M.transform);
        memcpy(M.inv_transform, s.inv_transform, sizeof M.inv_transform);
        M.material = s.material < 0 ? -1 : s.material;
        for (int k = 0; k < 5; k++) {
            if (s.textures[k] >= d->n_textures) return fail(MI_ERR_INVALID, "mesh %d: bad texture index", mi);
            M.tex[k] = s.textures[k] < 0 ? -1 : s.textures[k];
        }
        M.object_index = -1;
        M.tex_comb = -1;
        if (s.material < 0) {
            // interleaved copy of this mesh's maps (pt_device.h DMesh.tex_comb) when every bound map has the same size
            int w = 0, h = 0, bound = 0; bool same = true;
            for (int k = 0; k < 5; k++) if (s.textures[k] >= 0) {
                const mi_texture& t = d->textures[s.textures[k]];
                if (bound == 0) { w = t.width; h = t.height; } else if (t.width != w || t.height != h) same = false;
                bound++;
            }
            if (bound >= 2 && same && (uint64_t)w * (uint64_t)h * 16u < (1ull << 30)) {
                while (texels.size() % 16) texels.push_back(0);
                DTexture T; T.offset = (uint32_t)texels.size(); T.width = w; T.height = h; T.pad = 0;
                const size_t np = (size_t)w * h, at = texels.size();
                texels.resize(at + np * 16, 0);
                const uint8_t* src[5];
                for (int k = 0; k < 5; k++) src[k] = s.textures[k] >= 0 ? d->textures[s.textures[k]].rgb : nullptr;
                for (size_t px = 0; px < np; px++) {
                    uint8_t* o = &texels[at + px * 16];
                    // absent maps: albedo 0, emission 0, metallic 0, roughness 1.0 = 255 / 255 (geometry.rs:260-263)
                    for (int ch = 0; ch < 3; ch++) { o[ch] = src[0] ? src[0][px * 3 + ch] : 0; o[4 + ch] = src[1] ? src[1][px * 3 + ch] : 0; o[8 + ch] = src[4] ? src[4][px * 3 + ch] : 0; }
                    o[3] = src[2] ? src[2][px * 3] : 0;
                    o[7] = src[3] ? src[3][px * 3] : 255;
                }
                M.tex_comb = (int)texs.size();
                texs.push_back(T);
            }
        }
        M.tri_begin = (int)(tris.size() / 12);
        M.n_tris = s.n_triangles;
        MeshBuild& B = mb[(size_t)mi];
        build::RefTree rt{ s.positions, s.indices, &B.nodes };           // mesh-local node indices; relocated into the pool below
        rt.build(0, s.n_triangles);                                      // geometry.rs:185
        for (int t = 0; t < s.n_triangles; t++) {
            uint32_t ia = s.indices[3 * (size_t)t], ib = s.indices[3 * (size_t)t + 1], ic = s.indices[3 * (size_t)t + 2];
            h3 a = H3p(&s.positions[3 * (size_t)ia]), b = H3p(&s.positions[3 * (size_t)ib]), cc = H3p(&s.positions[3 * (size_t)ic]);
            h3 e1 = sub(b, a), e2 = sub(cc, a);                         // geometry.rs:336-337
            float rec[12] = { a.x, a.y, a.z, 0.0f, e1.x, e1.y, e1.z, 0.0f, e2.x, e2.y, e2.z, 0.0f };
            tris.insert(tris.end(), rec, rec + 12);
            DTriAttr A; memset(&A, 0, sizeof A);
            memcpy(A.na, &s.normals[3 * (size_t)ia], 12); memcpy(A.nb, &s.normals[3 * (size_t)ib], 12); memcpy(A.nc, &s.normals[3 * (size_t)ic], 12);
            memcpy(A.ta, &s.texcoords[2 * (size_t)ia], 8); memcpy(A.tb, &s.texcoords[2 * (size_t)ib], 8); memcpy(A.tc, &s.texcoords[2 * (size_t)ic], 8);
            // StaticMesh::get_tangent geometry.rs:245-250
            float u1 = A.ta[0], u2 = A.tb[0], u3 = A.tc[0], v1 = A.ta[1], v2 = A.tb[1], v3 = A.tc[1];
            h3 num = sub(scale(sub(b, a), (v3 - v1)), scale(sub(cc, a), (v2 - v1)));
            float den = (u2 - u1) * (v3 - v1) - (v2 - v1) * (u3 - u1);
            A.tan[0] = num.x / den; A.tan[1] = num.y / den; A.tan[2] = num.z / den;
            attrs.push_back(A);
        }
        // Two-stage traversal (bvh_build.hpp): does the padding bound apply to this mesh?  B = 7 eps E2 |d_obj| / 1e-4 must
        // stay <= 1/2 for every ray; |d_obj| <= |inv_transform's 3x3|_F |d_world|, and world directions of up to 8 units are
        // covered with a factor 10 to spare (a longer one takes the reference walk for that ray, decided on the device).
        DMeshF& F = meshf[(size_t)mi];
        memset(&F, 0, sizeof F);
        {
            build::FTree ft{ tris.data() + (size_t)M.tri_begin * 12, s.n_triangles, &B.fnodes, &B.ftris, 0, 2, {}, {}, {} };
            const build::FConst fc = ft.run();
            double fro = 0.0;
            for (int cc = 0; cc < 3; cc++) for (int r = 0; r < 3; r++) fro += (double)s.inv_transform[cc * 4 + r] * (double)s.inv_transform[cc * 4 + r];
            const double b_ref = 7.0 * 5.9604645e-08 * (double)fc.E2 * (std::sqrt(fro) * 8.0) * 1.0e4;
            const bool affine = s.inv_transform[3] == 0.0f && s.inv_transform[7] == 0.0f && s.inv_transform[11] == 0.0f && s.inv_transform[15] == 1.0f;
            B.qualifies = affine && std::isfinite(b_ref) && b_ref <= 0.05 && s.n_triangles < (1 << 24) && std::isfinite(fc.R) && std::isfinite(fc.L);
            build::FQuant fq{ 1.0f, 0.0f, 0.0f, 0.0f };
            if (B.qualifies) B.qualifies = build::fq_encode(B.fnodes.data(), B.fnodes.size() / 8, &fq, &B.fq);
            F.qs = fq.s; F.qbx = fq.bx; F.qby = fq.by; F.qbz = fq.bz;
            B.fnodes.clear(); B.fnodes.shrink_to_fit();
            B.default_ts = B.qualifies && s.n_triangles >= 1024;        // below that the reference's tree sits in LDS and the F-tree does not pay
            F.qualifies = B.qualifies ? 1 : 0;
            F.E2 = fc.E2; F.L = fc.L; F.cx = fc.cx; F.cy = fc.cy; F.cz = fc.cz; F.R = fc.R;
            if (!B.qualifies) { B.fq.clear(); B.ftris.clear(); }
        }
    }
    // Pool placement: meshes walked through the reference's tree first, so that one LDS window over the head of the node
    // pool covers exactly the trees wf_trav needs.  (The order of Scene.objects — ties, RNG draws — is not touched.)
    // Who references which mesh: Scene.objects entries, and ConvexVolume boundaries (a StaticMesh, or a nested Scene's entries).
    // Trees nobody references are not placed at all; boundary-only trees go last (they are walked from global memory).
    std::vector<uint8_t> obj_ref((size_t)d->n_meshes, 0), bnd_ref((size_t)d->n_meshes, 0);
    if (d->n_boundary_objects < 0 || (d->n_boundary_objects > 0 && !d->boundary_objects)) return fail(MI_ERR_INVALID, "bad boundary_objects");
    for (int i = 0; i < d->n_objects; i++)
        if (d->objects[i].kind == MI_OBJ_MESH) {
            if (d->objects[i].index < 0 || d->objects[i].index >= d->n_meshes) return fail(MI_ERR_INVALID, "object %d: bad mesh index", i);
            obj_ref[(size_t)d->objects[i].index] = 1;
        }
    for (int v = 0; v < d->n_volumes && d->volumes; v++) {
        const mi_volume& vo = d->volumes[v];
        auto mark = [&](int kind, int index) -> int {
            if (kind == MI_OBJ_MESH) {
                if (index < 0 || index >= d->n_meshes) return fail(MI_ERR_INVALID, "volume %d: bad boundary mesh index", v);
                bnd_ref[(size_t)index] = 1;
            }
            return MI_OK;
        };
        if (vo.boundary_kind == MI_OBJ_SCENE) {
            if (vo.boundary_index < 0 || vo.boundary_count < 0 || (int64_t)vo.boundary_index + vo.boundary_count > d->n_boundary_objects)
                return fail(MI_ERR_INVALID, "volume %d: boundary entries out of range", v);
            for (int k = 0; k < vo.boundary_count; k++) {
                const mi_object& e = d->boundary_objects[vo.boundary_index + k];
                const int rcm = mark(e.kind, e.index);
                if (rcm != MI_OK) return rcm;
            }
        } else {
            const int rcm = mark(vo.boundary_kind, vo.boundary_index);
            if (rcm != MI_OK) return rcm;
        }
    }
    {
        std::vector<int> order;
        for (int pass = 0; pass < 3; pass++)
            for (int mi = 0; mi < d->n_meshes; mi++) {
                const int cls = obj_ref[(size_t)mi] ? (mb[(size_t)mi].default_ts ? 1 : 0) : (bnd_ref[(size_t)mi] ? 2 : 3);
                if (cls == pass) order.push_back(mi);
            }
        for (int mi : order) {
            MeshBuild& B = mb[(size_t)mi];
            DMesh& M = meshes[(size_t)mi];
            DMeshF& F = meshf[(size_t)mi];
            const int nbase = (int)(nodes.size() / 8), fbase = (int)(fnodes.size() / 4);
            M.node_begin = nbase;
            for (size_t k = 0; k < B.nodes.size(); k += 8) { int sk; memcpy(&sk, &B.nodes[k + 3], 4); sk += nbase; memcpy(&B.nodes[k + 3], &sk, 4); }
            // leaf nodes carry {a, skip}{e1, tri} instead of their (never tested) box; e2 goes to its own small pool
            M.e2_begin = (int)(e2s.size() / 4);
            for (size_t k = 0; k < B.nodes.size(); k += 8) {
                int tri; memcpy(&tri, &B.nodes[k + 7], 4);
                if (tri < 0) continue;
                const float* T = &tris[((size_t)M.tri_begin + (size_t)tri) * 12];
                B.nodes[k + 0] = T[0]; B.nodes[k + 1] = T[1]; B.nodes[k + 2] = T[2];
                B.nodes[k + 4] = T[4]; B.nodes[k + 5] = T[5]; B.nodes[k + 6] = T[6];
            }
            for (int t = 0; t < M.n_tris; t++) {
                const float* T = &tris[((size_t)M.tri_begin + (size_t)t) * 12];
                const float rec[4] = { T[8], T[9], T[10], 0.0f };
                e2s.insert(e2s.end(), rec, rec + 4);
            }
            // the same tree with interior nodes and leaves in separate pools and explicit links (pt_device.h DScene.inodes)
            {
                const int n_local = (int)(B.nodes.size() / 8);
                const int ibase = (int)(inodes.size() / 8), lbase = (int)(lnodes.size() / 12);
                std::vector<int32_t> id((size_t)n_local);
                int ni = 0, nl = 0;
                for (int j = 0; j < n_local; j++) {
                    int tri; memcpy(&tri, &B.nodes[(size_t)j * 8 + 7], 4);
                    id[(size_t)j] = tri < 0 ? ibase + ni++ : ~(lbase + nl++);
                }
                auto id_of = [&](int j) -> int32_t { return j >= n_local ? kIdEnd : id[(size_t)j]; };
                for (int j = 0; j < n_local; j++) {
                    const float* N = &B.nodes[(size_t)j * 8];
                    int sk, tri; memcpy(&sk, &N[3], 4); memcpy(&tri, &N[7], 4);
                    if (tri < 0) {
                        float rec[8] = { N[0], N[1], N[2], 0.0f, N[4], N[5], N[6], 0.0f };
                        const int32_t miss = id_of(sk - nbase), hit = id_of(j + 1);
                        memcpy(&rec[3], &miss, 4); memcpy(&rec[7], &hit, 4);
                        inodes.insert(inodes.end(), rec, rec + 8);
                    } else {
                        const float* T = &tris[((size_t)M.tri_begin + (size_t)tri) * 12];
                        float rec[12] = { T[0], T[1], T[2], 0.0f, T[4], T[5], T[6], 0.0f, T[8], T[9], T[10], 0.0f };
                        const int32_t next = id_of(j + 1);
                        memcpy(&rec[3], &next, 4); memcpy(&rec[7], &tri, 4);
                        lnodes.insert(lnodes.end(), rec, rec + 12);
                    }
                }
                M.i_root = id_of(0);
                B.inode_end = (int)(inodes.size() / 8);
            }
            nodes.insert(nodes.end(), B.nodes.begin(), B.nodes.end());
            M.node_end = (int)(nodes.size() / 8);
            for (size_t k = 0; k < B.fq.size(); k += 4) if (!(B.fq[k + 3] & 0x80000000u)) B.fq[k + 3] += (uint32_t)fbase;      // interior nodes: skip links into the pool
            F.fnode_begin = fbase; F.ftri_begin = (int)(ftris.size() / 12);
            fnodes.insert(fnodes.end(), B.fq.begin(), B.fq.end());
            ftris.insert(ftris.end(), B.ftris.begin(), B.ftris.end());
            F.fnode_end = (int)(fnodes.size() / 4);
            B.nodes.clear(); B.nodes.shrink_to_fit(); B.fq.clear(); B.ftris.clear();
        }
    }

    // one Sphere / Triangle / Plane record (Scene.objects entry or boundary entry), derived constants hoisted
    auto fill_primitive = [&](int kind, int index, const char* what, int i, DObject& D) -> int {
        switch (kind) {
        case MI_OBJ_SPHERE: {
            if (index < 0 || index >= d->n_spheres || !d->spheres) return fail(MI_ERR_INVALID, "%s %d: bad sphere index", what, i);
            const mi_sphere& s = d->spheres[index];
            if (!mat_ok(s.material)) return fail(MI_ERR_INVALID, "%s %d: bad material", what, i);
            D.material = s.material;
            D.f[0] = s.center[0]; D.f[1] = s.center[1]; D.f[2] = s.center[2]; D.f[3] = s.radius;
            D.f[4] = s.radius * s.radius;                               // geometry.rs:400
            return MI_OK;
        }
        case MI_OBJ_TRIANGLE: {
            if (index < 0 || index >= d->n_triangles || !d->triangles) return fail(MI_ERR_INVALID, "%s %d: bad triangle index", what, i);
            const mi_triangle& t = d->triangles[index];
            if (!mat_ok(t.material)) return fail(MI_ERR_INVALID, "%s %d: bad material", what, i);
            D.material = t.material;
            h3 a = H3p(t.a), e1 = sub(H3p(t.b), a), e2 = sub(H3p(t.c), a);      // geometry.rs:434-435
            h3 n = normalize(cross(e1, e2));                                    // geometry.rs:449
            D.f[0] = a.x; D.f[1] = a.y; D.f[2] = a.z;
            D.f[3] = e1.x; D.f[4] = e1.y; D.f[5] = e1.z;
            D.f[6] = e2.x; D.f[7] = e2.y; D.f[8] = e2.z;
            D.f[9] = n.x; D.f[10] = n.y; D.f[11] = n.z;
            return MI_OK;
        }
        case MI_OBJ_PLANE: {
            if (index < 0 || index >= d->n_planes || !d->planes) return fail(MI_ERR_INVALID, "%s %d: bad plane index", what, i);
            const mi_plane& p = d->planes[index];
            if (!mat_ok(p.material)) return fail(MI_ERR_INVALID, "%s %d: bad material", what, i);
            D.material = p.material;
            for (int k = 0; k < 3; k++) { D.f[k] = p.point[k]; D.f[3 + k] = p.normal[k]; }
            return MI_OK;
        }
        default: return fail(MI_ERR_INVALID, "%s %d: unknown kind %d", what, i, kind);
        }
    };
    // boundary records of the ConvexVolumes that are not plain spheres; boundary meshes get entries of the device's mesh table
    // BEHIND the Scene.objects meshes (bmesh_of: mesh index -> entry, filled below once the live table exists)
    std::vector<DObject> bobjs;
    std::vector<std::pair<size_t, int>> bmesh_fix;        // (record in bobjs, mi_mesh index) to be pointed at its table entry

    // Scene.objects in order
    for (int i = 0; i < d->n_objects; i++) {
        const mi_object& o = d->objects[i];
        DObject& D = objs[(size_t)i];
        memset(&D, 0, sizeof D);
        D.kind = o.kind;
        D.index = i;
        D.ref = -1;
        switch (o.kind) {
        case MI_OBJ_SPHERE: case MI_OBJ_TRIANGLE: case MI_OBJ_PLANE: {
            const int rcp = fill_primitive(o.kind, o.index, "object", i, D);
            if (rcp != MI_OK) return rcp;
            break;
        }
        case MI_OBJ_VOLUME: {
            if (o.index < 0 || o.index >= d->n_volumes || !d->volumes) return fail(MI_ERR_INVALID, "object %d: bad volume index", i);
            const mi_volume& v = d->volumes[o.index];
            if (!mat_ok(v.phase_material)) return fail(MI_ERR_INVALID, "object %d: bad phase material", i);
            D.material = v.phase_material;
            D.f[5] = -1.0f / v.density;                                 // geometry.rs:517
            if (v.boundary_kind == MI_OBJ_SPHERE) {                     // the inline sphere: what every use in the reference is
                for (int k = 0; k < 3; k++) D.f[k] = v.boundary_center[k];
                D.f[3] = v.boundary_radius;
                D.f[4] = v.boundary_radius * v.boundary_radius;         // geometry.rs:400 via :505
                break;
            }
            // any other `Arc<dyn Intersectable>` (geometry.rs:496): its records, tested twice per ray by the kernels (:505,508)
            std::vector<mi_object> entries;
            if (v.boundary_kind == MI_OBJ_SCENE) for (int k = 0; k < v.boundary_count; k++) entries.push_back(d->boundary_objects[v.boundary_index + k]);
            else { mi_object e; e.kind = v.boundary_kind; e.index = v.boundary_index; entries.push_back(e); }
            D.ref = (int)bobjs.size();
            { const int n = (int)entries.size(); memcpy(&D.f[6], &n, 4); }
            for (size_t k = 0; k < entries.size(); k++) {
                DObject R; memset(&R, 0, sizeof R);
                R.kind = entries[k].kind; R.index = (int)k; R.ref = -1;
                if (entries[k].kind == MI_OBJ_MESH) { R.material = -1; bmesh_fix.emplace_back(bobjs.size(), entries[k].index); }
                else if (entries[k].kind == MI_OBJ_VOLUME || entries[k].kind == MI_OBJ_SCENE)
                    return fail(MI_ERR_UNSUPPORTED, "object %d: a ConvexVolume or a Scene inside a ConvexVolume boundary", i);
                else { const int rcp = fill_primitive(entries[k].kind, entries[k].index, "boundary entry of object", i, R); if (rcp != MI_OK) return rcp; }
                bobjs.push_back(R);
            }
            break;
        }
        case MI_OBJ_MESH: {
            // the same StaticMesh may appear several times (Arc sharing, tracing.rs:215): every appearance is an entry of its own
            // in the device's mesh table — sharing the nodes, triangles and attributes in the pools — with its own object index
            D.ref = o.index; D.material = -1;
            break;
        }
        default: return fail(MI_ERR_INVALID, "object %d: unknown kind %d", i, o.kind);
        }
    }
    // The device's mesh table: one entry per MESH entry of Scene.objects, in that order (S.n_meshes of them: what the hit loop
    // walks), then one per boundary mesh (reached only through a ConvexVolume's boundary record).
    std::vector<DMesh> live;
    std::vector<DMeshF> livef;
    c->mesh_node_end.clear(); c->mesh_e2_end.clear(); c->mesh_inode_end.clear(); c->mesh_qualifies.clear(); c->mesh_default_ts.clear();
    for (int i = 0; i < d->n_objects; i++)
        if (objs[(size_t)i].kind == OBJ_MESH) {
            int r = objs[(size_t)i].ref; objs[(size_t)i].ref = (int)live.size();
            live.push_back(meshes[(size_t)r]); livef.push_back(meshf[(size_t)r]);
            live.back().object_index = i;
            c->mesh_node_end.push_back(meshes[(size_t)r].node_end);
            c->mesh_e2_end.push_back(meshes[(size_t)r].e2_begin + meshes[(size_t)r].n_tris);
            c->mesh_inode_end.push_back(mb[(size_t)r].inode_end);
            c->mesh_qualifies.push_back(mb[(size_t)r].qualifies ? 1 : 0); c->mesh_default_ts.push_back(mb[(size_t)r].default_ts ? 1 : 0);
        }
    const size_t n_scene_meshes = live.size();
    for (auto& fx : bmesh_fix) {
        bobjs[fx.first].ref = (int)live.size();
        live.push_back(meshes[(size_t)fx.second]); livef.push_back(meshf[(size_t)fx.second]);
        live.back().object_index = -1;
    }

    // kind-grouped copy of the non-mesh objects (stable within a kind)
    std::vector<DObject> list;
    int n_list[4] = { 0, 0, 0, 0 };
    {
        const int order[4] = { OBJ_TRIANGLE, OBJ_SPHERE, OBJ_PLANE, OBJ_VOLUME };
        for (int g = 0; g < 4; g++)
            for (size_t i = 0; i < objs.size(); i++)
                if (objs[i].kind == order[g]) { list.push_back(objs[i]); n_list[g]++; }
    }
    // ---- top-level tree over the list's Triangles ("top-level BVH over Scene.objects"; long lists only) ----
    // A list of hundreds of Triangles is hundreds of Moller-Trumbore tests per path segment.  The exact two-stage machinery of the
    // meshes applies to them unchanged, in world space: a SAH tree over the triangles' boxes, walked per ray with the boxes padded by the
    // proven bound on what the reference's f32 test can accept (bvh_build.hpp), and the reference's own test on the triangles of the leaves
    // reached — a triangle whose padded box the ray misses would have failed that test, and the closest hit over the rest is
    // order-independent (ties: the lower Scene.objects index).  The bound scales with E2 = max |e1||e2| over the tree, so the LARGE
    // triangles (walls: 32 x the median product and more) stay in front of the list and are tested one by one; the tree needs >= 96 of
    // the others.  A ray the bound does not cover (B > 1/2, non-finite) makes its wave test the whole list one by one.
    int n_list_lin = n_list[0], top_meshf = -1;
    {
        constexpr int kTopMinTris = 96;       // measured (tools/probe_list_tree.py): 40 small triangles 0.85 x, 105: 1.1 x, 400: 1.6 x, 2000: 1.9 x of the plain loop
        const int nt = n_list[0];
        if (nt >= kTopMinTris) {
            std::vector<double> prod((size_t)nt);
            for (int k = 0; k < nt; k++) {
                const float* f = list[(size_t)k].f;
                prod[(size_t)k] = std::sqrt((double)f[3] * f[3] + (double)f[4] * f[4] + (double)f[5] * f[5]) * std::sqrt((double)f[6] * f[6] + (double)f[7] * f[7] + (double)f[8] * f[8]);
            }
            std::vector<double> sorted = prod;
            std::nth_element(sorted.begin(), sorted.begin() + nt / 2, sorted.end());
            const double big = 32.0 * sorted[(size_t)nt / 2];
            // large triangles (and anything non-finite) to the front, order kept within each part (stable: ties between equal hits are decided by index anyway)
            std::vector<DObject> front, rest;
            for (int k = 0; k < nt; k++) ((!(prod[(size_t)k] <= big) || !std::isfinite(prod[(size_t)k])) ? front : rest).push_back(list[(size_t)k]);
            if ((int)rest.size() >= kTopMinTris) {
                std::vector<float> lt(rest.size() * 12, 0.0f), tn, tt;
                for (size_t k = 0; k < rest.size(); k++) {
                    const float* f = rest[k].f; float* T = &lt[k * 12];
                    for (int q = 0; q < 3; q++) { T[q] = f[q]; T[4 + q] = f[3 + q]; T[8 + q] = f[6 + q]; }
                }
                build::FTree ft{ lt.data(), (int)rest.size(), &tn, &tt, 0, 2, {}, {}, {} };
                const build::FConst fc = ft.run();
                build::FQuant fq{ 1.0f, 0.0f, 0.0f, 0.0f };
                std::vector<uint32_t> tq;
                // (the bound's B = 7 eps E2 |d| / 1e-4 is checked per ray on the device; here only: is it finite, and below 1/2 for a unit direction at all)
                const double b_unit = 7.0 * 5.9604645e-08 * (double)fc.E2 * 1.0e4;
                if (std::isfinite(b_unit) && b_unit <= 0.25 && std::isfinite(fc.R) && std::isfinite(fc.L) && rest.size() < (1u << 24) &&
                    build::fq_encode(tn.data(), tn.size() / 8, &fq, &tq)) {
                    const int fbase = (int)(fnodes.size() / 4);
                    for (size_t k = 0; k < tq.size(); k += 4) if (!(tq[k + 3] & 0x80000000u)) tq[k + 3] += (uint32_t)fbase;      // interior nodes: skip links into the pool
                    for (size_t e = 0; e < tt.size() / 12; e++) {          // a leaf triangle carries its Scene.objects index where a mesh triangle carries its number
                        int t; memcpy(&t, &tt[e * 12 + 3], 4);
                        const int32_t idx = rest[(size_t)t].index;
                        memcpy(&tt[e * 12 + 3], &idx, 4);
                    }
                    DMeshF F; memset(&F, 0, sizeof F);
                    F.fnode_begin = fbase; F.ftri_begin = (int)(ftris.size() / 12);
                    fnodes.insert(fnodes.end(), tq.begin(), tq.end());
                    ftris.insert(ftris.end(), tt.begin(), tt.end());
                    F.fnode_end = (int)(fnodes.size() / 4);
                    F.qualifies = 1; F.E2 = fc.E2; F.L = fc.L; F.cx = fc.cx; F.cy = fc.cy; F.cz = fc.cz; F.R = fc.R;
                    F.qs = fq.s; F.qbx = fq.bx; F.qby = fq.by; F.qbz = fq.bz;
                    top_meshf = (int)livef.size();
                    livef.push_back(F);
                    n_list_lin = (int)front.size();
                    for (size_t k = 0; k < front.size(); k++) list[k] = front[k];
                    for (size_t k = 0; k < rest.size(); k++) list[front.size() + k] = rest[k];
                }
            }
        }
    }
    // world-space corners of the root boxes (tile masks).  The rays reach object space through inv_transform
    // (geometry.rs:304), so the corners come from ITS inverse (f64), not from `transform`; a projective
    // inv_transform or a single-triangle mesh (no root box) is never culled.
    c->h_mesh_box.assign(n_scene_meshes, mi_ctx::MeshBox{ false, {} });
    for (size_t m = 0; m < n_scene_meshes; m++) {
        const DMesh& M = live[m];
        const float* it = M.inv_transform;
        if (!(it[3] == 0.0f && it[7] == 0.0f && it[11] == 0.0f && it[15] == 1.0f)) continue;
        const float* n0 = &nodes[(size_t)M.node_begin * 8];
        int32_t tri_id; memcpy(&tri_id, &n0[7], 4);
        if (tri_id >= 0) continue;
        double a[4][8];                                   // [inv | I], Gauss-Jordan with partial pivoting
        for (int r = 0; r < 4; r++) for (int q = 0; q < 4; q++) { a[r][q] = (double)it[q * 4 + r]; a[r][4 + q] = r == q ? 1.0 : 0.0; }
        bool ok = true;
        for (int col = 0; col < 4 && ok; col++) {
            int piv = col;
            for (int r = col + 1; r < 4; r++) if (fabs(a[r][col]) > fabs(a[piv][col])) piv = r;
            if (!(fabs(a[piv][col]) > 1e-12)) { ok = false; break; }
            if (piv != col) for (int q = 0; q < 8; q++) std::swap(a[piv][q], a[col][q]);
            const double inv = 1.0 / a[col][col];
            for (int q = 0; q < 8; q++) a[col][q] *= inv;
            for (int r = 0; r < 4; r++) if (r != col) { const double f = a[r][col]; for (int q = 0; q < 8; q++) a[r][q] -= f * a[col][q]; }
        }
        if (!ok) continue;
        mi_ctx::MeshBox& B = c->h_mesh_box[m];
        B.cullable = true;
        for (int k = 0; k < 8; k++) {
            const double q[3] = { (double)((k & 1) ? n0[4] : n0[0]), (double)((k & 2) ? n0[5] : n0[1]), (double)((k & 4) ? n0[6] : n0[2]) };
            for (int r = 0; r < 3; r++) {
                B.corner[k][r] = a[r][4] * q[0] + a[r][5] * q[1] + a[r][6] * q[2] + a[r][7];
                if (!std::isfinite(B.corner[k][r])) B.cullable = false;
            }
        }
    }
    // sample_hemisphere's rotation Basis3::between_vectors(unit_y, n) (materials.rs:176) as a matrix, for the two normals a list Triangle or Plane can
    // present to a ray (its stored normal and the negation: RayHit::new, tracing.rs:118-123; Plane: geometry.rs:476-478): the same f32 operations in the same order as
    // pt_kernels.hip rotate_from_unit_y performs per scatter (this file is compiled with -ffp-contract=off; sqrtf and the divisions are
    // correctly rounded on both sides, rcp_exact IS 1.0f / x), so reading the table is exact.  12 floats per entry:
    // {c0.xyz, c1.x}{c1.yz, c2.xy}{c2.z, 1 = identity (the function returns `dir` untouched), 0, 0}; entry 2 i + (frontface ? 0 : 1) of object i.
    std::vector<float> obj_rot(objs.size() * 24, 0.0f);
    {
        auto ulps_eq = [](float a, float b) {                       // approx::ulps_eq!, f32 defaults (pt_kernels.hip ulps_eq)
            if (fabsf(a - b) <= 1.1920929e-07f) return true;
            if ((a < 0.0f) != (b < 0.0f)) return false;
            int32_t ia, ib; memcpy(&ia, &a, 4); memcpy(&ib, &b, 4);
            int32_t d = (int32_t)((uint32_t)ia - (uint32_t)ib);
            if (d < 0) d = (int32_t)(0u - (uint32_t)d);
            return d <= 4;
        };
        auto rot = [&](float nx, float ny, float nz, float* out) {
            const float k_cos_theta = ny;
            if (ulps_eq(k_cos_theta, 1.0f)) { out[9] = 1.0f; return; }
            const float k = sqrtf(1.0f * ((nx * nx + ny * ny) + nz * nz));
            float qs, qx, qz;
            if (ulps_eq(k_cos_theta / k, -1.0f)) { qs = 0.0f; qx = 0.0f; qz = -1.0f; }
            else {
                const float sq = k + k_cos_theta;
                const float cx = nz, cz = -nx;
                const float mag = sqrtf(sq * sq + ((cx * cx + 0.0f) + cz * cz));
                const float inv = 1.0f / mag;
                qs = sq * inv; qx = cx * inv; qz = cz * inv;
            }
            const float x2 = qx + qx, z2 = qz + qz;
            const float xx2 = x2 * qx, xz2 = x2 * qz, zz2 = z2 * qz;
            const float sz2 = z2 * qs, sx2 = x2 * qs;
            out[0] = 1.0f - zz2; out[1] = sz2; out[2] = xz2;                       // c0
            out[3] = -sz2; out[4] = (1.0f - xx2) - zz2; out[5] = sx2;            // c1
            out[6] = xz2; out[7] = -sx2; out[8] = 1.0f - xx2;                     // c2
        };
        for (size_t i = 0; i < objs.size(); i++) if (objs[i].kind == OBJ_TRIANGLE || objs[i].kind == OBJ_PLANE) {
            const float* n = objs[i].f + (objs[i].kind == OBJ_TRIANGLE ? 9 : 3);       // the Triangle's stored normal / the Plane's
            rot(n[0], n[1], n[2], &obj_rot[i * 24]);
            rot(-n[0], -n[1], -n[2], &obj_rot[i * 24 + 12]);
        }
    }
    // one blob: objects | list | materials | meshes | nodes | tris | attrs | textures | texels
    auto align = [](size_t x) { return (x + 255) & ~(size_t)255; };
    size_t off_obj = 0;
    size_t off_list = align(off_obj + objs.size() * sizeof(DObject));
    size_t off_bobj = align(off_list + (list.size() + 1) * sizeof(DObject));     // +1: the loop prefetches one record ahead
    size_t off_rot = align(off_bobj + (bobjs.size() + 1) * sizeof(DObject));
    size_t off_mat = align(off_rot + obj_rot.size() * 4 + 16);
    size_t off_mesh = align(off_mat + mats.size() * sizeof(DMaterial));
    size_t off_meshf = align(off_mesh + live.size() * sizeof(DMesh));
    size_t off_fnodes = align(off_meshf + livef.size() * sizeof(DMeshF));
    size_t off_ftris = align(off_fnodes + fnodes.size() * 4 + 32);
    size_t off_nodes = align(off_ftris + ftris.size() * 4 + 48);
    size_t off_e2 = align(off_nodes + nodes.size() * 4);
    size_t off_inodes = align(off_e2 + e2s.size() * 4 + 16);
    size_t off_lnodes = align(off_inodes + inodes.size() * 4 + 32);
    size_t off_tris = align(off_lnodes + lnodes.size() * 4 + 48);
    size_t off_attr = align(off_tris + tris.size() * 4);
    size_t off_tex = align(off_attr + attrs.size() * sizeof(DTriAttr));
    size_t off_texel = align(off_tex + texs.size() * sizeof(DTexture));
    size_t total = align(off_texel + texels.size() + 16);
    if (total > 0xffffffffull) return fail(MI_ERR_UNSUPPORTED, "scene larger than 4 GiB");
    std::vector<uint8_t> host(total, 0);
    auto put = [&](size_t off, const void* p, size_t n) { if (n) memcpy(host.data() + off, p, n); };
    put(off_obj, objs.data(), objs.size() * sizeof(DObject));
    put(off_list, list.data(), list.size() * sizeof(DObject));
    put(off_bobj, bobjs.data(), bobjs.size() * sizeof(DObject));
    put(off_rot, obj_rot.data(), obj_rot.size() * 4);
    put(off_mat, mats.data(), mats.size() * sizeof(DMaterial));
    put(off_mesh, live.data(), live.size() * sizeof(DMesh));
    put(off_meshf, livef.data(), livef.size() * sizeof(DMeshF));
    put(off_fnodes, fnodes.data(), fnodes.size() * 4);
    put(off_ftris, ftris.data(), ftris.size() * 4);
    put(off_nodes, nodes.data(), nodes.size() * 4);
    put(off_e2, e2s.data(), e2s.size() * 4);
    put(off_inodes, inodes.data(), inodes.size() * 4);
    put(off_lnodes, lnodes.data(), lnodes.size() * 4);
    put(off_tris, tris.data(), tris.size() * 4);
    put(off_attr, attrs.data(), attrs.size() * sizeof(DTriAttr));
    put(off_tex, texs.data(), texs.size() * sizeof(DTexture));
    put(off_texel, texels.data(), texels.size());

    if (c->blob) { (void)hipFree(c->blob); c->blob = nullptr; c->blob_bytes = 0; c->have_scene = false; }
    hipError_t e = hipMalloc(&c->blob, total);
    if (e != hipSuccess) return fail(MI_ERR_OOM, "hipMalloc(%zu) for the scene failed: %s", total, hipGetErrorString(e));
    c->blob_bytes = total;
    HIP_TRY(hipMemcpy(c->blob, host.data(), total, hipMemcpyHostToDevice));
    uint8_t* b = (uint8_t*)c->blob;
    c->S.objects = (const DObject*)(b + off_obj);
    c->S.list = (const DObject*)(b + off_list);
    c->S.bobjs = (const DObject*)(b + off_bobj);
    c->S.obj_rot = (const float*)(b + off_rot);
    c->gen_volumes = !bobjs.empty();
    c->mesh_maps = false;
    for (const DMesh& M : live) if (M.material < 0 || M.tex[4] >= 0) c->mesh_maps = true;
    c->h_list = list; c->h_n_tri = n_list[0]; c->h_n_sphere = n_list[1]; c->h_n_unmasked = n_list[2] + n_list[3]; c->mask_valid = false;
    c->S.n_list_tri = n_list[0]; c->S.n_list_sphere = n_list[1]; c->S.n_list_plane = n_list[2]; c->S.n_list_volume = n_list[3];
    c->S.n_list_lin = n_list_lin; c->S.top_meshf = top_meshf; c->list_tree = top_meshf >= 0;
    c->S.materials = (const DMaterial*)(b + off_mat);
    c->S.meshes = (const DMesh*)(b + off_mesh);
    c->S.meshf = (const DMeshF*)(b + off_meshf);
    c->S.fnodes = (const float*)(b + off_fnodes);
    c->S.ftris = (const float*)(b + off_ftris);
    c->S.n_fnodes = (int)(fnodes.size() / 4);
    c->S.nodes = (const float*)(b + off_nodes);
    c->S.e2s = (const float*)(b + off_e2);
    c->S.inodes = (const float*)(b + off_inodes);
    c->S.lnodes = (const float*)(b + off_lnodes);
    c->S.tris = (const float*)(b + off_tris);
    c->S.triattr = (const DTriAttr*)(b + off_attr);
    c->S.textures = (const DTexture*)(b + off_tex);
    c->S.texels = (const uint8_t*)(b + off_texel);
    c->S.n_objects = (int)objs.size();
    c->S.n_meshes = (int)n_scene_meshes;
    c->S.n_nodes = (int)(nodes.size() / 8);
    c->S.n_tris = (int)(tris.size() / 12);
    c->lds_bytes = (uint32_t)((nodes.size() + tris.size()) * 4);
    for (int k = 0; k < 3; k++) { c->point_light_pos[k] = d->point_light_pos[k]; c->ambient[k] = d->ambient[k]; }
    c->have_scene = true;
    return MI_OK;
}

// ------------------------------------------------------------------ render
static int check_camera(const mi_camera_desc* cam) {
    if (!cam) return fail(MI_ERR_INVALID, "camera is NULL");
    if (cam->projection_mode != MI_PROJ_PERSPECTIVE && cam->projection_mode != MI_PROJ_ORTHOGRAPHIC)
        return fail(MI_ERR_INVALID, "unknown projection_mode %d", cam->projection_mode);
    if (cam->shading_mode != MI_SHADE_PATHTRACE && cam->shading_mode != MI_SHADE_PHONG)
        return fail(MI_ERR_INVALID, "unknown shading_mode %d", cam->shading_mode);
    if (cam->path_samples == 0) return fail(MI_ERR_INVALID, "path_samples must be >= 1 (tracing.rs:318 divides by it)");
    if (cam->screen_width == 0 || cam->screen_height == 0 || cam->screen_width > 32768 || cam->screen_height > 32768)
        return fail(MI_ERR_INVALID, "bad image size %ux%u", cam->screen_width, cam->screen_height);
    if (cam->aa_sample_count == 0) return fail(MI_ERR_INVALID, "aa_sample_count must be >= 1");
    if ((uint32_t)sqrtf((float)cam->aa_sample_count) == 0) return fail(MI_ERR_INVALID, "aa_sample_count too small");
    if (!(cam->gamma > 0.0f) || !std::isfinite(cam->gamma)) return fail(MI_ERR_INVALID, "gamma must be finite and > 0 (tracing.rs:254 raises to 1/gamma)");
    // A camera that makes every ray non-finite is refused, not rendered.  The reference would render it: its tests then "hit"
    // with a NaN distance wherever every reject comparison is false (geometry.rs:338-349, 401-410) and Scene keeps the FIRST
    // such hit in Scene.objects order (tracing.rs:335, NaN < x is false) — a result that depends on the evaluation order
    // of unordered comparisons, which the kind-grouped object list of the kernels does not keep.
    for (int k = 0; k < 3; k++)
        if (!std::isfinite(cam->eyepoint[k]) || !std::isfinite(cam->view_dir[k]) || !std::isfinite(cam->up[k]))
            return fail(MI_ERR_INVALID, "camera eyepoint / view_dir / up must be finite");
    if (!std::isfinite(cam->focal_length) || !std::isfinite(cam->focus_dist) || !std::isfinite(cam->lens_radius) || std::isnan(cam->max_trace_dist))
        return fail(MI_ERR_INVALID, "camera focal_length / focus_dist / lens_radius must be finite, max_trace_dist not NaN");
    {
        // tracing.rs:188: rotation.x = view_dir.cross(up).normalize(), in f32 as the kernels evaluate it
        const float* v = cam->view_dir; const float* u = cam->up;
        const float cx = v[1] * u[2] - v[2] * u[1], cy = v[2] * u[0] - v[0] * u[2], cz = v[0] * u[1] - v[1] * u[0];
        const float m2 = (cx * cx + cy * cy) + cz * cz;
        if (!(m2 > 0.0f) || !std::isfinite(1.0f / sqrtf(m2)))
            return fail(MI_ERR_INVALID, "view_dir x up is zero or not finite: the camera basis is singular and every ray would be NaN (tracing.rs:188)");
    }
    return MI_OK;
}

// The tile grid of the partition.  Tiles are numbered row-major over a grid whose ROW LENGTH `tx` is the image's tile columns
// rounded up to the next integer coprime with `world` (tile t -> rank t % world, slot t / world): a row length that shares a factor
// with the rank count repeats the same few column classes for a rank in every row (60 columns over 8 ranks: two classes, and the
// ranks whose classes cross the expensive middle of the frame took 7 % longer than the others); a coprime one walks every rank
// through all classes.  The extra columns hold no pixel: their tiles are rendered as "outside the image" (zeros) and never
// copied anywhere.  world = 1 (mi_render) keeps the plain grid.
static void tile_counts(const mi_camera_desc* cam, int world, uint32_t* tx, uint32_t* ty, uint32_t* total, uint32_t* padded) {
    uint32_t stride = (cam->screen_width + MI_TILE - 1) / MI_TILE;
    auto gcd = [](uint32_t a, uint32_t b) { while (b) { const uint32_t r = a % b; a = b; b = r; } return a; };
    while (gcd(stride, (uint32_t)world) != 1u) stride++;
    *tx = stride;
    *ty = (cam->screen_height + MI_TILE - 1) / MI_TILE;
    *total = *tx * *ty;
    *padded = (*total + (uint32_t)world - 1) / (uint32_t)world;
}

extern "C" int mi_compact_size(const mi_camera_desc* cam, int32_t world, uint32_t* tiles_total, uint32_t* tiles_padded) {
    if (!cam || world < 1 || !tiles_total || !tiles_padded) return fail(MI_ERR_INVALID, "mi_compact_size: bad argument");
    uint32_t tx, ty;
    tile_counts(cam, world, &tx, &ty, tiles_total, tiles_padded);
    return MI_OK;
}

static void make_camera(const mi_camera_desc* cam, DCamera* C) {
    memset(C, 0, sizeof *C);
    h3 view = H3p(cam->view_dir), up = H3p(cam->up);
    h3 c0 = normalize(cross(view, up));                                 // tracing.rs:188
    C->eye[0] = cam->eyepoint[0]; C->eye[1] = cam->eyepoint[1]; C->eye[2] = cam->eyepoint[2];
    C->rot[0] = c0.x; C->rot[1] = c0.y; C->rot[2] = c0.z;
    C->rot[3] = up.x; C->rot[4] = up.y; C->rot[5] = up.z;               // :189
    C->rot[6] = -view.x; C->rot[7] = -view.y; C->rot[8] = -view.z;      // :190
    C->pixel_size = 1.0f / (float)cam->screen_height;                   // :160
    C->n = (float)cam->aa_sample_count;                                 // :162
    C->rootn = sqrtf(C->n);                                             // :163
    C->half_rootn = 0.5f * C->rootn;
    C->half_n = 0.5f * C->n;
    C->cx_base = -(0.5f * (float)cam->screen_width);                    // :178
    C->cy_base = 0.5f + 0.5f * (float)cam->screen_height;               // :179
    C->focal_length = cam->focal_length; C->focus_dist = cam->focus_dist; C->lens_radius = cam->lens_radius;
    C->max_trace_dist = cam->max_trace_dist;
    C->rootn_u = (uint32_t)C->rootn;                                    // :169 `rootn as u32`
    C->spp = cam->aa_sample_count;
    C->zone = (C->spp << __builtin_clz(C->spp)) - 1u;                   // rand 0.8.4 UniformInt::sample_single
    C->path_depth = cam->path_depth;
    C->width = cam->screen_width; C->height = cam->screen_height;
    C->ortho = cam->projection_mode == MI_PROJ_ORTHOGRAPHIC ? 1u : 0u;
    // :200,204  rotation * view_dir with cgmath's Matrix3 * Vector3 order: (c0*v.x + c1*v.y) + c2*v.z
    h3 c2 = H3(-view.x, -view.y, -view.z);
    h3 od = add(add(scale(c0, view.x), scale(up, view.y)), scale(c2, view.z));
    C->ortho_dir[0] = od.x; C->ortho_dir[1] = od.y; C->ortho_dir[2] = od.z;
}

// K1w: the wavefront pipeline (pt_kernels.hip).  The host enqueues one iteration per path segment:
// wf_main, wf_prefix (device-side bookkeeping of the shard counters), wf_trav.  It needs one number back
// per iteration — the grid of the next wf_main — which arrives on a second stream while wf_trav runs, so
// the compute stream never waits for the host; the call still returns only when the frame is done.
//
// Memory: path state is streamed through HBM — 2 x 96 B (ping/pong) + 16 B sample slot per path
// (kWfBytesPerPath; the walkers read the class-B lists, there is no traversal queue).  The batch is sized to the free HBM (288 GB on MI355X: the whole 1080p/256 spp
// frame, 531 M paths = 112 GB, is ONE batch), halved on allocation failure.
static const int kHdrRing = 16;        // pinned header slots (wf_prefix -> host), one per pass in flight
static const int kRunAhead = 3;        // passes the host may launch before it has read the header of an earlier one
static const size_t kWfBytesPerPath = 2 * (size_t)kWfPlanes * sizeof(float4) + sizeof(float4);      // ping / pong state, sample slot
static const size_t kWfBytesPerPathTwoStage = (size_t)kCandMax * sizeof(uint2) + sizeof(uint2);     // candidates + header per queue slot

// which meshes this render walks two-stage (bit m = live mesh m)
static uint32_t two_stage_mask(const mi_ctx* c, uint32_t flags) {
    if (flags & MI_OPT_REFERENCE_WALK) return 0u;
    uint32_t m = 0;
    for (size_t i = 0; i < c->mesh_qualifies.size() && i < (size_t)kTwoStageMaxMeshes; i++)
        if (c->mesh_qualifies[i] && ((flags & MI_OPT_TWO_STAGE) || c->mesh_default_ts[i])) m |= 1u << i;
    return m;
}

static int wf_alloc(mi_ctx* c, WfArgs& a, uint32_t spp, uint64_t max_state_bytes, bool two_stage, uint32_t& s_batch) {
    uint64_t max_paths;
    const size_t per_path = kWfBytesPerPath + (two_stage ? kWfBytesPerPathTwoStage : 0);
    if (max_state_bytes != 0) max_paths = max_state_bytes / per_path;        // the caller's budget (mi_render_opts)
    else {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = (size_t)8 << 30;
        // what this context already holds for the pipeline can be reused
        free_b += c->wf_a_bytes + c->wf_b_bytes + c->wf_samp_bytes + c->cand_bytes + c->cand_hdr_bytes;
        max_paths = (uint64_t)((double)free_b * 0.6 / (double)per_path);
    }
    if (max_paths > (1ull << 31)) max_paths = 1ull << 31;           // 32-bit path indices
    uint64_t sb = max_paths / a.npix;
    // the smallest batch is one sample of every (padded) pixel of this rank: a caller's budget below that cannot be honoured
    if (sb < 1 && max_state_bytes != 0)
        return fail(MI_ERR_INVALID, "max_state_bytes = %llu is below the pipeline's minimum for this image: one sample per pixel = %llu bytes",
                    (unsigned long long)max_state_bytes, (unsigned long long)((uint64_t)a.npix * per_path));
    if (sb < 1) sb = 1;
    if (sb > spp) sb = spp;
    for (;;) {
        s_batch = (uint32_t)sb;
        const uint32_t paths = a.npix * s_batch;
        const uint32_t max_blocks = (paths + kBlock - 1) / kBlock;
        // a shard receives at most the paths of its own input blocks: ceil(n_blocks / shards) blocks, where
        // n_blocks <= max_blocks + 2 * shards (one partial block per (class, shard) range)
        a.region = ((max_blocks + kWfShards - 1) / kWfShards + 3) * kBlock;
        a.cap = a.region * (uint32_t)kWfShards;
        const size_t st_bytes = (size_t)kWfPlanes * sizeof(float4) * a.cap;
        int rc = ensure(&c->d_wf_a, &c->wf_a_bytes, st_bytes);
        if (rc == MI_OK) rc = ensure(&c->d_wf_b, &c->wf_b_bytes, st_bytes);
        if (rc == MI_OK) rc = ensure(&c->d_wf_samp, &c->wf_samp_bytes, (size_t)paths * sizeof(float4));
        if (rc == MI_OK) rc = ensure(&c->d_wf_acc, &c->wf_acc_bytes, (size_t)a.npix * sizeof(float4));
        if (rc == MI_OK && two_stage) rc = ensure(&c->d_cand, &c->cand_bytes, (size_t)a.cap * kCandMax * sizeof(uint2));
        if (rc == MI_OK && two_stage) rc = ensure(&c->d_cand_hdr, &c->cand_hdr_bytes, (size_t)a.cap * sizeof(uint2));
        if (rc == MI_OK) return MI_OK;
        if (rc != MI_ERR_OOM || sb == 1) return rc;
        (void)hipGetLastError();
        sb = (sb + 1) / 2;                                           // back off and retry with half the batch
    }
}

// sizes and allocates (host-side work: must happen BEFORE the timing start event is recorded)
static int wf_prepare(mi_ctx* c, const mi_camera_desc* cam, uint32_t padded, uint64_t max_state_bytes, bool two_stage, WfArgs& a, uint32_t& s_batch) {
    memset(&a, 0, sizeof a);
    a.npix = padded * (uint32_t)kTilePixels;
    const uint32_t spp = cam->aa_sample_count;
    if (spp > 0xffffu) return fail(MI_ERR_UNSUPPORTED, "wavefront variant: aa_sample_count must be <= 65535");
    if (cam->path_depth > 0xffffu) return fail(MI_ERR_UNSUPPORTED, "wavefront variant: path_depth must be <= 65535");
    return wf_alloc(c, a, spp, max_state_bytes, two_stage, s_batch);
}

// Primary-ray culling for the wavefront pipeline.  For every 32x32 tile: which Triangle / Sphere entries of
// the kind-grouped list can a camera ray of that tile reach?  A perspective camera with lens_radius 0
// sends every ray of a tile from the eye through the tile's pixel footprints — pixel centre +- 1 px
// (tracing.rs:171-181: both jitter terms span +-0.5 px) — so the rays lie inside the pyramid spanned by
// the four corner directions of the footprint, here widened by one more pixel (>= 5e-4 rad at any
// resolution up to 2k rows, against f32 rounding of ~1e-7 in the generated directions).  An object that is
// entirely on the outer side of one of the pyramid's four planes through the eye cannot be hit; it is
// dropped from the tile's mask and its test — which would have missed — is not run.  f64 on the host,
// a further 1e-4 scene-unit slack; any non-finite value or a singular camera basis keeps everything.
// Planes and ConvexVolumes are never masked.  Returns false when masking does not apply.
static bool tile_masks(mi_ctx* c, const mi_camera_desc* cam, uint32_t flags, uint32_t stride) {
    const int n_ts = c->h_n_tri + c->h_n_sphere;
    const int n_mesh = (int)c->h_mesh_box.size();
    if ((n_ts == 0 && n_mesh == 0) || n_ts > 64 || n_mesh > 32) return false;
    // rays must leave the eye itself (no lens) towards the image plane (focus_dist > 0 keeps the direction's sign)
    if (cam->projection_mode != MI_PROJ_PERSPECTIVE || cam->lens_radius != 0.0f || !(cam->focus_dist > 0.0f) || (flags & MI_OPT_NO_TILE_MASKS)) return false;
    if (c->mask_valid && c->mask_stride == stride && memcmp(&c->mask_cam, cam, sizeof *cam) == 0) return true;
    const double W = cam->screen_width, H = cam->screen_height, p = 1.0 / H;
    const double view[3] = { cam->view_dir[0], cam->view_dir[1], cam->view_dir[2] };
    const double up[3] = { cam->up[0], cam->up[1], cam->up[2] };
    double c0[3] = { view[1] * up[2] - view[2] * up[1], view[2] * up[0] - view[0] * up[2], view[0] * up[1] - view[1] * up[0] };
    const double l0 = sqrt(c0[0] * c0[0] + c0[1] * c0[1] + c0[2] * c0[2]);
    if (!(l0 > 1e-12) || !std::isfinite(l0)) return false;
    for (double& v : c0) v /= l0;
    // det of R = [c0 up -view]: a (near-)singular basis flattens the pyramid
    const double det = c0[0] * (up[1] * -view[2] - up[2] * -view[1]) - up[0] * (c0[1] * -view[2] - c0[2] * -view[1])
                     + -view[0] * (c0[1] * up[2] - c0[2] * up[1]);
    if (!(fabs(det) > 1e-6) || !std::isfinite(det)) return false;
    auto dir = [&](double px, double py, double* o) {           // R * (camera-space point on the image plane)
        const double x = p * (px - 0.5 * W + 0.5), y = p * (0.5 + 0.5 * H - py), z = -(double)cam->focal_length;
        for (int k = 0; k < 3; k++) o[k] = c0[k] * x + up[k] * y + -view[k] * z;
    };
    // `stride` >= the image's tile columns: the row length of the tile numbering (tile_counts); the surplus columns hold no pixel
    const uint32_t tx = stride, tx_image = (cam->screen_width + MI_TILE - 1) / MI_TILE, ty = (cam->screen_height + MI_TILE - 1) / MI_TILE;
    // [0, tiles): list masks; [tiles, 2*tiles): low 32 bits = mesh mask, bit 63 = DEAD tile (nothing reachable:
    // every camera ray of the tile leaves the scene at once)
    const size_t n_tiles = (size_t)tx * ty;
    c->h_tile_mask.assign(2 * n_tiles, ~0ull);
    for (size_t t = 0; t < n_tiles; t++) c->h_tile_mask[n_tiles + t] = 0xffffffffull;
    const double margin_px = 2.0, slack = 1e-4;
    const double eye[3] = { cam->eyepoint[0], cam->eyepoint[1], cam->eyepoint[2] };
    // Test shapes.  The geometric argument needs the f32 intersection tests to be WELL CONDITIONED for every
    // camera ray, or a test could "hit" a triangle it passes far from.  Moller-Trumbore (geometry.rs:434-446)
    // computes u = (s.h)/a with a = -d.n: the rounding error of u is ~2^-22 |s||h| / |a|, and a test that passes
    // t <= t_max has |a| >= |n| h_E / t_max (t's numerator s.n = h_E |n| does not depend on d; h_E = distance
    // of the eye from the triangle's plane).  So with  G = 2^-22 S t_reach / (h_E alt_min)  (S = eye to
    // farthest vertex, alt_min = smallest altitude, t_reach = max_trace_dist x the column norms of the camera
    // basis) small enough (below) every accepted ray passes inside the triangle scaled by 1.01 about its
    // centroid — which is the shape tested against the pyramid.  Triangles that fail the
    // guard (eye almost in their plane, slivers, huge max_trace_dist) are simply never masked.
    // Spheres (geometry.rs:395-408): the discriminant's rounding moves the silhouette by ~1e-6 relative;
    // the radius is padded by 0.1 % plus 1e-5 of the centre distance.
    struct Shape { bool cullable; double v[3][3]; double r; };
    std::vector<Shape> shapes((size_t)n_ts);
    {
        double colmax = 0.0;
        { const double lu = sqrt(up[0] * up[0] + up[1] * up[1] + up[2] * up[2]), lv = sqrt(view[0] * view[0] + view[1] * view[1] + view[2] * view[2]);
          colmax = 1.0 + lu + lv; }                            // |R d| <= (|c0| + |up| + |view|) |d|, |c0| = 1
        const double t_reach = (double)cam->max_trace_dist * colmax;
        for (int e = 0; e < n_ts; e++) {
            const DObject& ob = c->h_list[(size_t)e];
            Shape& sh = shapes[(size_t)e];
            sh.cullable = false; sh.r = 0.0;
            if (e < c->h_n_tri) {
                double P[3][3], cen[3] = { 0, 0, 0 };
                for (int vtx = 0; vtx < 3; vtx++) for (int q = 0; q < 3; q++) {
                    P[vtx][q] = (double)ob.f[q] + (vtx == 1 ? (double)ob.f[3 + q] : vtx == 2 ? (double)ob.f[6 + q] : 0.0);
                    cen[q] += P[vtx][q] / 3.0;
                }
                const double e1[3] = { P[1][0] - P[0][0], P[1][1] - P[0][1], P[1][2] - P[0][2] };
                const double e2[3] = { P[2][0] - P[0][0], P[2][1] - P[0][1], P[2][2] - P[0][2] };
                const double e3[3] = { P[2][0] - P[1][0], P[2][1] - P[1][1], P[2][2] - P[1][2] };
                const double nn[3] = { e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0] };
                const double area2 = sqrt(nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2]);
                auto len = [](const double* w) { return sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]); };
                const double emax = std::max(len(e1), std::max(len(e2), len(e3)));
                double S = 0.0, hE = 0.0;
                for (int vtx = 0; vtx < 3; vtx++) { const double w[3] = { P[vtx][0] - eye[0], P[vtx][1] - eye[1], P[vtx][2] - eye[2] }; S = std::max(S, len(w)); }
                for (int q = 0; q < 3; q++) hE += (eye[q] - P[0][q]) * nn[q];
                hE = fabs(hE) / area2;
                const double alt_min = area2 / emax;
                const double G = ldexp(1.0, -22) * S * t_reach / (hE * alt_min);
                // (u, v) off by G moves the point by <= 2 G emax in the plane; scaling by 1.01 about the centroid moves
                // every edge out by >= 0.0033 alt_min: G <= 1e-3 alt_min / emax keeps the ray inside the scaled triangle
                sh.cullable = std::isfinite(G) && area2 > 0.0 && G <= 1e-3 * alt_min / emax;
                for (int vtx = 0; vtx < 3; vtx++) for (int q = 0; q < 3; q++) sh.v[vtx][q] = cen[q] + (P[vtx][q] - cen[q]) * 1.01;
            } else {
                double dist = 0.0;
                for (int q = 0; q < 3; q++) { sh.v[0][q] = (double)ob.f[q]; dist += (sh.v[0][q] - eye[q]) * (sh.v[0][q] - eye[q]); }
                sh.r = fabs((double)ob.f[3]) * 1.001 + 1e-5 * sqrt(dist);
                sh.cullable = std::isfinite(sh.r) && std::isfinite(dist);
            }
        }
    }
    for (uint32_t j = 0; j < ty; j++) for (uint32_t i = 0; i < tx; i++) {
        if (i >= tx_image) {                     // a column beyond the image: nothing to render, whatever the scene holds
            c->h_tile_mask[(size_t)j * tx + i] = 0ull; c->h_tile_mask[n_tiles + (size_t)j * tx + i] = 1ull << 63;
            continue;
        }
        const double x0 = (double)i * MI_TILE - margin_px, x1 = std::min<double>(W, (i + 1.0) * MI_TILE) - 1.0 + margin_px;
        const double y0 = (double)j * MI_TILE - margin_px, y1 = std::min<double>(H, (j + 1.0) * MI_TILE) - 1.0 + margin_px;
        double cs[4][3], ctr[3], n[4][3];
        dir(x0, y0, cs[0]); dir(x1, y0, cs[1]); dir(x1, y1, cs[2]); dir(x0, y1, cs[3]);
        dir(0.5 * (x0 + x1), 0.5 * (y0 + y1), ctr);
        bool ok = true;
        for (int k = 0; k < 4 && ok; k++) {
            const double* a = cs[k]; const double* b = cs[(k + 1) & 3];
            double v[3] = { a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0] };
            const double l = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
            if (!(l > 0.0) || !std::isfinite(l)) { ok = false; break; }
            const double sgn = (v[0] * ctr[0] + v[1] * ctr[1] + v[2] * ctr[2]) < 0.0 ? -1.0 : 1.0;      // inward
            for (int q = 0; q < 3; q++) n[k][q] = sgn * v[q] / l;
        }
        if (!ok) continue;
        unsigned long long mask = ~0ull;
        for (int e = 0; e < n_ts; e++) {
            const Shape& sh = shapes[(size_t)e];
            if (!sh.cullable) continue;
            bool cull = false;
            for (int k = 0; k < 4 && !cull; k++) {
                if (e < c->h_n_tri) {                       // the three (scaled) vertices all outside plane k
                    bool all_out = true;
                    for (int vtx = 0; vtx < 3 && all_out; vtx++) {
                        double d = 0.0;
                        for (int q = 0; q < 3; q++) d += (sh.v[vtx][q] - eye[q]) * n[k][q];
                        all_out = d < -slack;              // false for NaN
                    }
                    cull = all_out;
                } else {
                    double d = 0.0;
                    for (int q = 0; q < 3; q++) d += (sh.v[0][q] - eye[q]) * n[k][q];
                    cull = d < -(sh.r + slack);
                }
            }
            if (cull) mask &= ~(1ull << e);
        }
        c->h_tile_mask[(size_t)j * tx + i] = mask;
        unsigned long long mm = 0xffffffffull;
        for (int m = 0; m < n_mesh && m < 32; m++) {          // the mesh word has 32 bits: meshes 32, 33, ... are never culled
            const mi_ctx::MeshBox& B = c->h_mesh_box[(size_t)m];
            if (!B.cullable) continue;
            bool cull = false;
            for (int k = 0; k < 4 && !cull; k++) {
                bool all_out = true;
                for (int v = 0; v < 8 && all_out; v++) {
                    double d = 0.0, len = 0.0;
                    for (int q = 0; q < 3; q++) { const double w = B.corner[v][q] - eye[q]; d += w * n[k][q]; len += fabs(B.corner[v][q]) + fabs(eye[q]); }
                    all_out = d < -(slack + 1e-5 * len);       // f32 rounding of the object-space ray and slabs
                }
                cull = all_out;
            }
            if (cull) mm &= ~(1ull << m);
        }
        const unsigned long long ts_bits = (n_ts >= 64) ? ~0ull : ((1ull << n_ts) - 1ull);
        const unsigned long long mesh_bits = (n_mesh >= 32) ? 0xffffffffull : ((1ull << n_mesh) - 1ull);
        if ((mask & ts_bits) == 0ull && (mm & mesh_bits) == 0ull && c->h_n_unmasked == 0 && n_mesh <= 32) mm |= 1ull << 63;
        c->h_tile_mask[n_tiles + (size_t)j * tx + i] = mm;
    }
    c->mask_cam = *cam; c->mask_stride = stride; c->mask_valid = false;              // valid once uploaded
    if (c->tune.debug_mask) {
        size_t bits = 0, mbits = 0, dead = 0;
        for (size_t t = 0; t < n_tiles; t++) {
            bits += (size_t)__builtin_popcountll(c->h_tile_mask[t] & ((n_ts >= 64) ? ~0ull : ((1ull << n_ts) - 1ull)));
            mbits += (size_t)__builtin_popcountll(c->h_tile_mask[n_tiles + t] & ((n_mesh >= 32) ? 0xffffffffull : ((1ull << n_mesh) - 1ull)));
            dead += (size_t)(c->h_tile_mask[n_tiles + t] >> 63);
        }
        fprintf(stderr, "[mi_rt] tile masks: %zu tiles, %.2f of %d list entries and %.2f of %d meshes kept per tile, %zu dead tiles\n",
                n_tiles, (double)bits / (double)n_tiles, n_ts, (double)mbits / (double)n_tiles, n_mesh, dead);
    }
    return true;
}

// Samples [begin, end) of every pixel, added in order to `accum` (nullptr = the context's own buffer).
// begin == 0 starts the sums from zero; end == aa_sample_count also writes the per-pixel means.
struct SampleRange { uint32_t begin, end; float4* accum; };

static int render_tiles_wavefront(mi_ctx* c, const K1Args& k, const mi_camera_desc* cam, WfArgs a, uint32_t s_batch, bool lds, uint32_t flags,
                                  float* d_compact, uint32_t* d_sig, SampleRange range, hipStream_t stream) {
    a.S = k.S; a.C = k.C; a.R = k.R; a.seed_key = k.seed_key;
    const uint32_t spp = cam->aa_sample_count;
    int rc = MI_OK;
    (void)rc;
    // counters: [0..2S) out_count (class A shards, class B shards), [2S..3S) trav_count, [3S] trav_head,
    // then in_count [2S], in_blkpfx [2S+1], trav_pfx [S+1]
    const size_t S_ = (size_t)kWfShards;
    uint32_t* cnt = c->d_wf_cnt;
    a.out_count = cnt; a.trav_count = cnt + 2 * S_; a.trav_head = cnt + 3 * S_;
    uint32_t* d_in_count = cnt + 3 * S_ + 8;
    uint32_t* d_in_pfx = d_in_count + 2 * S_;
    a.in_count = d_in_count; a.in_blkpfx = d_in_pfx;
    uint32_t* d_trav_pfx = d_in_pfx + 2 * S_ + 8;
    a.trav_pfx = d_trav_pfx;
    uint32_t* d_hdr = d_trav_pfx + S_ + 8;
    a.hdr = d_hdr;
    a.samp = (float4*)c->d_wf_samp; a.accum = range.accum ? range.accum : (float4*)c->d_wf_acc;
    a.out = d_compact; a.sig = d_sig;
    a.tile_mask = nullptr;
    if (tile_masks(c, cam, flags, a.R.tiles_x)) {
        if (!c->mask_valid) {
            int rcm = ensure(&c->d_tile_mask, &c->tile_mask_bytes, c->h_tile_mask.size() * sizeof(unsigned long long));
            if (rcm != MI_OK) return rcm;
            HIP_TRY(hipMemcpyAsync(c->d_tile_mask, c->h_tile_mask.data(), c->h_tile_mask.size() * sizeof(unsigned long long), hipMemcpyHostToDevice, stream));
            c->mask_valid = true;
        }
        a.tile_mask = (const unsigned long long*)c->d_tile_mask;
    }
    a.diag = nullptr;           // developer builds (-DPT_WF_STAMPS): phase stamps of wf_main
    if (c->tune.wf_stamps) { a.diag = c->d_diag; HIP_TRY(hipMemsetAsync(c->d_diag, 0, 16 * sizeof(unsigned long long), stream)); }
    a.refill_min = c->tune.refill_min;
    // further shade + intersect rounds inside one wf_main launch: one when meshes park part of every wave's rays for the walker
    // (cfg2: 1 / 2 / 3 rounds -> 99 / 101 / 103 ms), two in a scene without meshes, where every live lane can go on
    // (cfg5 at 512 spp: 1 / 2 / 3 / 5 rounds -> 281.6 / 271.8 / 278.4 / 287.7 ms; the cfg1 scene at 1080p: 46.6 / 42.9 / 47.4 / 48.1 ms)
    a.fuse_max = c->tune.fuse_max ? c->tune.fuse_max : (c->S.n_meshes == 0 ? 2u : 1u);
    a.fuse_min = c->tune.fuse_min < 1 ? 1 : c->tune.fuse_min;
    const uint32_t fuse_max_normal = a.fuse_max, fuse_min_normal = a.fuse_min;
    // Which meshes are walked how: the two-stage meshes (wf_trav_f + wf_replay), the rest through the reference's tree (wf_trav).
    const uint32_t all_meshes = c->S.n_meshes >= 32 ? 0xffffffffu : ((1u << c->S.n_meshes) - 1u);
    const uint32_t ts_mask = two_stage_mask(c, flags) & all_meshes;
    const uint32_t ref_mask = all_meshes & ~ts_mask;
    // wf_trav LDS mode: 2 = BVH nodes in LDS, triangles through L1 (default when the nodes fit 64 KB:
    // teapot 15 KB -> 8 blocks per CU; 122.6 ms vs 126.0 ms for mode 1 on cfg2 1080p/256), 1 = nodes +
    // triangles (what the megakernels stage), 0 = everything from global memory.  The LDS window is the head of the node
    // pool up to the last tree wf_trav walks (the scene compiler places those trees first).
    int ref_nodes = 0, ref_e2 = 0, ref_inodes = 0;
    for (size_t m = 0; m < c->mesh_node_end.size(); m++) if (m >= 32 || ((ref_mask >> m) & 1u)) {     // meshes 32, 33, ... have no mask bit: always walked here
        ref_nodes = std::max(ref_nodes, c->mesh_node_end[m]); ref_e2 = std::max(ref_e2, c->mesh_e2_end[m]);
        ref_inodes = std::max(ref_inodes, c->mesh_inode_end[m]);
    }
    // LDS image of a walker block: the nodes (leaves carry a and e1 of their triangle) + the triangles' e2 vectors
    const size_t node_bytes = (size_t)ref_nodes * 32 + (size_t)ref_e2 * 16;
    const size_t inode_bytes = (size_t)ref_inodes * 32;               // wf_trav_i: the interior nodes only
    const int ref_lnodes = ref_nodes - ref_inodes;                    // the split pools hold every tree in the node pool's order: leaves = nodes - interior nodes
    const size_t split_bytes = inode_bytes + (size_t)ref_lnodes * 48; // wf_trav_i<.., LEAF_LDS>: interior and leaf records
    const size_t pair_bytes = (((size_t)ref_inodes * (size_t)kPairStride + 15) & ~(size_t)15) + (size_t)ref_lnodes * 48;   // wf_trav_i<.., PAIR>: 56-byte interior records
    // 3 = the image needs most of a CU's 160 KB: ONE 1024-thread block per CU (16 waves).  4 = wf_trav_i: interior nodes in LDS, leaves
    // from global memory — chosen over 3 whenever two of its blocks fit a CU (8 waves per SIMD instead of 4), and the only LDS mode
    // left for images beyond 156 KB whose interior nodes still fit
    int trav_lds_mode = 0;
    if (ref_nodes > 0 && !c->tune.global_bvh) {
        // one mesh to walk: the single-mesh forms over 32-byte records test boxes with the clamped form (slab_med3), 22 VALU per step like the
        // paired layout's, on half the LDS and eight 256-thread blocks per CU (cfg2 walker 22.4 against 22.8 ms); several small meshes: the
        // paired layout (the MULTI forms have no register left for the clamped form's class test)
        const bool one_mesh = ref_mask == 1u && c->S.n_meshes <= 32;
        if (one_mesh && split_bytes <= 64u * 1024u) trav_lds_mode = 5;
        else if (pair_bytes <= 40u * 1024u) trav_lds_mode = 6;      // four 512-thread blocks per CU = 8 waves per SIMD
        else if (split_bytes <= 64u * 1024u) trav_lds_mode = 5;     // (= node_bytes: both images hold every node once and every triangle once)
        else if (node_bytes <= 64u * 1024u) trav_lds_mode = 2;
        else if (inode_bytes <= 78u * 1024u) trav_lds_mode = 4;
        else if (node_bytes <= 156u * 1024u) trav_lds_mode = 3;
        else if (inode_bytes <= 156u * 1024u) trav_lds_mode = 4;
    }
    // 5 = wf_trav_i with the leaf records in LDS too (the whole split image within 64 KB): the explicit links make its interior step shorter than wf_trav's
    // 6 = the same with the interior records in the paired layout (near / far plane per axis behind one 8-byte read: no selects in the box test)
    if (c->tune.trav_lds >= 0) {
        const int m = c->tune.trav_lds;
        if (m == 0 || (m == 2 && node_bytes <= 64u * 1024u) || (m == 3 && node_bytes <= 156u * 1024u) || (m == 4 && inode_bytes <= 156u * 1024u && ref_nodes > 0) ||
            (m == 5 && split_bytes <= 64u * 1024u && ref_nodes > 0) || (m == 6 && pair_bytes <= 40u * 1024u && ref_nodes > 0)) trav_lds_mode = m;
    }
    const size_t trav_lds_bytes = trav_lds_mode == 6 ? pair_bytes : trav_lds_mode == 5 ? split_bytes : (trav_lds_mode == 4 ? inode_bytes : (trav_lds_mode >= 2 ? node_bytes : 0));
    a.R.lds_nodes = trav_lds_mode >= 4 ? (uint32_t)ref_inodes : (trav_lds_mode ? (uint32_t)ref_nodes : 0);
    a.R.lds_tris = (trav_lds_mode == 2 || trav_lds_mode == 3) ? (uint32_t)ref_e2 : (trav_lds_mode >= 5 ? (uint32_t)ref_lnodes : 0);   // e2 entries / leaf records staged behind the nodes
    a.cand = (uint2*)c->d_cand; a.cand_hdr = (uint2*)c->d_cand_hdr;
    float4* bufs[2] = { (float4*)c->d_wf_a, (float4*)c->d_wf_b };
    uint32_t trav_bpc = 6;                  // resident blocks per CU: bounded by LDS (160 KB) and by 8 waves/SIMD
    if (trav_lds_mode == 2) { trav_bpc = (uint32_t)((160u * 1024u) / (node_bytes ? node_bytes : 1)); if (trav_bpc > 8) trav_bpc = 8; if (trav_bpc < 2) trav_bpc = 2; }
    if (trav_lds_mode == 3) trav_bpc = 1;
    if (trav_lds_mode == 4) trav_bpc = inode_bytes <= 78u * 1024u ? 2 : 1;
    if (trav_lds_mode == 5) { trav_bpc = (uint32_t)((160u * 1024u) / (split_bytes ? split_bytes : 1)); if (trav_bpc > 8) trav_bpc = 8; if (trav_bpc < 2) trav_bpc = 2; }
    if (trav_lds_mode == 6) { trav_bpc = (uint32_t)((160u * 1024u) / (pair_bytes ? pair_bytes : 1)); if (trav_bpc > 4) trav_bpc = 4; if (trav_bpc < 1) trav_bpc = 1; }
    if (c->tune.trav_bpc > 0) trav_bpc = (uint32_t)c->tune.trav_bpc;
    const uint32_t trav_blocks = (uint32_t)c->n_cus * trav_bpc;
    const uint32_t travf_blocks = (uint32_t)c->n_cus * (c->tune.travf_bpc > 0 ? (uint32_t)c->tune.travf_bpc : 6u), replay_blocks = (uint32_t)c->n_cus * 8u;
    const uint32_t conc_trav_bpc = c->tune.conc_trav_bpc > 0 ? (uint32_t)c->tune.conc_trav_bpc : trav_bpc;
    const uint32_t conc_travf_bpc = c->tune.conc_travf_bpc > 0 ? (uint32_t)c->tune.conc_travf_bpc : travf_blocks / (uint32_t)c->n_cus;

    // per-kernel timing: one event pair per launch, summed after the frame
    size_t ev_used = 0;
    std::vector<int> ev_kind;
    // one event pair per launch is ~0.4 ms per frame of extra barriers: nothing on a whole frame (109 ms), 3 %
    // of a 1/8 share, so multi-rank renders skip it unless asked (MI_RT_WF_KERNEL_TIMING=0/1 overrides)
    bool per_kernel_timing = a.R.world == 1;
    if (c->tune.kernel_timing >= 0) per_kernel_timing = c->tune.kernel_timing != 0;
    auto stamp = [&](int kind, hipStream_t on) -> int {      // kind: 0 wf_main, 1 wf_trav, 2 wf_reduce, 3 wf_trav_f, 4 wf_replay, 5 wf_main's class-A part on the second stream; call before AND after the launch
        if (!per_kernel_timing) return MI_OK;
        if (ev_used == c->wf_ev.size()) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return MI_ERR_HIP; c->wf_ev.push_back(e); }
        if (hipEventRecord(c->wf_ev[ev_used++], on) != hipSuccess) return MI_ERR_HIP;
        ev_kind.push_back(kind);
        return MI_OK;
    };
#define WF_TIMED_ON(kind, on, call) do { if (stamp(kind, on) != MI_OK) return fail(MI_ERR_HIP, "event"); HIP_TRY(call); if (stamp(kind, on) != MI_OK) return fail(MI_ERR_HIP, "event"); } while (0)
#define WF_TIMED(kind, call) WF_TIMED_ON(kind, stream, call)

    uint64_t counts[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    counts[4] = (uint64_t)a.npix * (range.end - range.begin); counts[5] = a.npix;
    // samples of dead tiles (nothing reachable from the tile: no ray is generated, no slot written or read) — only without signatures
    if (a.tile_mask && !d_sig) {
        uint64_t dead_px = 0;
        for (uint32_t t = (uint32_t)a.R.rank; t < a.R.tiles_total; t += (uint32_t)a.R.world) {
            if (!(c->h_tile_mask[(size_t)a.R.tiles_total + t] >> 63)) continue;
            const uint32_t x0 = (t % a.R.tiles_x) * MI_TILE, y0 = (t / a.R.tiles_x) * MI_TILE;
            if (x0 >= cam->screen_width || y0 >= cam->screen_height) continue;
            dead_px += (uint64_t)std::min<uint32_t>(MI_TILE, cam->screen_width - x0) * std::min<uint32_t>(MI_TILE, cam->screen_height - y0);
        }
        counts[7] = dead_px * (range.end - range.begin);
    }
    const bool have_walkers = ref_mask || ts_mask || c->S.n_meshes > 32;
    // Headers: wf_prefix stores {blocks, live paths, queue length, seq, class-B paths, class-A blocks} of every pass into a RING of
    // pinned host slots (slot = seq % kHdrRing), so the host may run a few passes ahead of the device and still read every header.
    struct PassHdr { uint32_t blocks, live, queue, live_b, blocks_a, segments; };
    auto header_ready = [&](uint32_t seq) { return ((volatile uint32_t*)c->h_hdr)[(size_t)(seq % kHdrRing) * 8 + 3] == seq; };
    auto read_header = [&](uint32_t seq) {
        __atomic_thread_fence(__ATOMIC_ACQUIRE);
        const volatile uint32_t* h = (volatile uint32_t*)c->h_hdr + (size_t)(seq % kHdrRing) * 8;
        PassHdr r = { h[0], h[1], h[2], h[4], h[5], h[6] };
        return r;
    };
    auto wait_header = [&](uint32_t seq) -> int {
        const auto t_wait = std::chrono::steady_clock::now();
        for (uint32_t spin = 1; !header_ready(seq); spin++) {
            if ((spin & 63u) == 0) {
                hipError_t q = hipStreamQuery(stream);
                if (q == hipSuccess) { if (header_ready(seq)) break; return fail(MI_ERR_HIP, "wavefront pipeline: stream drained without a header"); }
                if (q != hipErrorNotReady) return fail(MI_ERR_HIP, "wavefront pipeline: %s", hipGetErrorString(q));
                // a wedged stream neither drains nor errors: bound the wait by wall clock (one pass is milliseconds)
                const auto waited = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t_wait).count();
                if (waited > (long long)c->tune.spin_timeout_ms)
                    return fail(MI_ERR_HIP, "wavefront pipeline: no header from the device after %lld ms (stream wedged?)", (long long)waited);
            }
            // the header of a small pass is there within tens of microseconds: poll without sleeping at first (a sleep
            // costs ~60 us whatever it asks for), then back off
            if (spin < 4096u && std::chrono::steady_clock::now() - t_wait < std::chrono::microseconds(150)) {
#if defined(__x86_64__)
                __builtin_ia32_pause();
#else
                std::this_thread::yield();
#endif
            } else std::this_thread::sleep_for(std::chrono::microseconds(20));
        }
        return MI_OK;
    };
    for (uint32_t s0 = range.begin; s0 < range.end; s0 += s_batch) {
        a.s_base = s0; a.s_count = (s0 + s_batch <= range.end) ? s_batch : (range.end - s0);
        int cur = 0;
        a.iter0 = 1;
        a.n_in = a.npix * a.s_count;
        HIP_TRY(hipMemsetAsync(cnt, 0, (3 * S_ + 8) * sizeof(uint32_t), stream));   // wf_prefix re-zeroes them after every pass
        const uint32_t seq0 = c->hdr_seq + 1u;          // seq of this batch's pass 0
        uint32_t seen = 0;                              // headers of passes [0, seen) have been read
        PassHdr last = { (a.n_in + kBlock - 1) / kBlock, a.n_in, 0u, 0u, 0u, 0u };      // "header of pass -1": the camera rays
        bool all_dead = false;
        auto consume = [&](bool count_it) {             // read header `seen` (it has arrived)
            last = read_header(seq0 + seen);
            seen++;
            if (count_it) { counts[0] += 1; counts[1] += last.live - last.live_b; counts[2] += last.live_b; counts[3] += last.queue; }
            counts[6] += last.segments;                 // every launched pass counts: a pass behind the one that ended every path ran no segment
            if (last.live == 0) all_dead = true;
        };
        uint32_t it = 0;
        for (; it <= cam->path_depth + 1u; it++) {
            while (seen < it && !all_dead && header_ready(seq0 + seen)) consume(true);
            if (all_dead) break;
            // The grid of pass `it` comes from the header of pass it - 1.  While that pass is still running the host would
            // have to wait for it (the header is on its way while the walkers run, so for a big pass the wait is hidden); a SMALL
            // pass is launched at once instead, on a grid that is an upper bound — live paths only decrease, and every (class,
            // shard) list may end in a partial block — whose surplus blocks leave at their first instruction (wf_main compares
            // its block number with the device-side table).  The host runs at most kRunAhead passes ahead of the headers.
            bool exact = seen == it;
            if (!exact) {
                const uint64_t bound = (uint64_t)last.live / kBlock + 2u * (uint64_t)kWfShards;
                if (c->tune.nowait_blocks == 0 || bound > (uint64_t)c->tune.nowait_blocks || it - seen > (uint32_t)kRunAhead) {
                    while (seen < it && !all_dead) { int rcw = wait_header(seq0 + seen); if (rcw != MI_OK) return rcw; consume(true); }
                    if (all_dead) break;
                    exact = true;
                }
            }
            uint32_t grid_all, grid_a;
            if (exact) { grid_all = last.blocks; grid_a = last.blocks_a; }
            else { grid_all = grid_a = (uint32_t)((uint64_t)last.live / kBlock) + 2u * (uint32_t)kWfShards; }
            if (it == 0) { grid_all = last.blocks; grid_a = 0; }
            if (grid_all == 0) break;
            // THE TAIL.  Once the live paths no longer fill the chip (last.live bounds this pass' input: paths only end), thin waves
            // cost nothing — there is nobody to give their lanes to — while every further pass costs two launches and a header.
            // So each wave keeps shading as long as ANY of its lanes can go on (a path that enters a mesh root still parks for
            // the walker).  Per path the operations and their order are those of the pass-by-pass schedule.  Without meshes
            // nothing ever parks: this launch ends every path and is the last one.
            // Threshold (MI_RT_WF_TAIL_PATHS): without meshes 4 Mi paths, the size below which passes are launched without waiting
            // (cfg1 as BASELINE states it, 400x400 / 16 spp: 0.64 ms at 0, 0.55 at 64 Ki ... 2 Mi, 0.49 from 3 Mi on; cfg1 / cfg5 at
            // 1080p unchanged up to 8 Mi); with meshes 1 Mi (a 1/8 share of cfg2: 11.0 ms up to 2 Mi, 11.1 at 4 Mi, 11.3 at 8 Mi).
            const uint32_t tail_paths = c->tune.tail_paths != 0xffffffffu ? c->tune.tail_paths : (have_walkers ? (1u << 20) : (4u << 20));
            const bool tail = it > 0 && tail_paths != 0 && last.live <= tail_paths;      // (the camera pass in this form too: 0.49 -> 0.58 ms on cfg1 as stated)
            a.fuse_max = tail ? cam->path_depth + 2u : fuse_max_normal;
            a.fuse_min = tail ? 1u : fuse_min_normal;
            a.st_in = a.iter0 ? nullptr : bufs[cur];
            a.st_out = bufs[cur ^ 1];
            a.n_blocks_in = grid_all;
            // A pass after the first is launched in TWO PARTS.  Its class-A blocks (paths whose pending hit is a plain Triangle /
            // Plane: nothing a walker could still change) go to a second stream, ordered only behind the previous pass' wf_prefix:
            // they fill the CUs the persistent walkers of that pass leave idle as their queue runs out (a walker launch ends
            // with ~0.1 ms of tail whatever its queue size, eleven times per frame and per rank).  The class-B blocks follow the
            // walkers on the main stream; wf_prefix waits for both parts.  Same blocks, same work, another schedule.
            const bool split = !a.iter0 && have_walkers && c->tune.split != 0 && (exact ? (grid_a >= 64u && grid_a < grid_all) : true);
            if (split) {
                HIP_TRY(hipStreamWaitEvent(c->aux_stream, c->ev_pfx, 0));
                a.part = 1;
                WF_TIMED_ON(5, c->aux_stream, launch_wf_main(a, grid_a, d_sig != nullptr, c->gen_volumes, c->mesh_maps, c->aux_stream));     // kind 5: its span includes waiting for CUs
                HIP_TRY(hipEventRecord(c->ev_part, c->aux_stream));
                a.part = 2;
                WF_TIMED(0, launch_wf_main(a, exact ? grid_all - grid_a : grid_all, d_sig != nullptr, c->gen_volumes, c->mesh_maps, stream));
                HIP_TRY(hipStreamWaitEvent(stream, c->ev_part, 0));
            } else {
                a.part = 0;
                WF_TIMED(0, launch_wf_main(a, grid_all, d_sig != nullptr, c->gen_volumes, c->mesh_maps, stream));
            }
            // device-side bookkeeping: tables for the next pass and for wf_trav, and the header the host needs (grid of the
            // next pass, anything alive?), which wf_prefix stores straight into pinned host memory: the compute stream never
            // waits for the host
            const uint32_t seq = ++c->hdr_seq;
            HIP_TRY(launch_wf_prefix(cnt, cnt + 2 * S_, d_in_count, d_in_pfx, d_trav_pfx, d_hdr, c->h_hdr_dev + (size_t)(seq % kHdrRing) * 8, seq, stream));
            if (have_walkers && c->tune.split != 0) HIP_TRY(hipEventRecord(c->ev_pfx, stream));
            // persistent walkers; they leave at once when the queue is empty.  Successive launches merge their meshes' hits
            // into the hit record (strictly closer wins, ties go to the lower Scene.objects index: order-independent)
            const bool ref_walk = ref_mask || c->S.n_meshes > 32;      // meshes 32, 33, ... have no mask bit: they always take the reference walk
            // Meshes of both kinds: wf_trav and wf_trav_f read the same work list and write different things (the hit record / the
            // candidate lists; each has its own cursor), so they go to two streams and wf_replay, which merges into the hit
            // record, follows both.  With full grids the F-tree walkers move in as the reference walkers run out of queue and leave
            // (HEAD 98.5 -> 96.5 ms).  Sharing every CU from the start — half the wave slots each — gains nothing: 68 ms for the
            // pair, exactly the 43 + 25 ms they take one after the other (VALU issue 0.71 + 0.34: together they saturate it).
            const bool side_by_side = ref_walk && ts_mask && c->tune.conc != 0;
            if (side_by_side) {
                HIP_TRY(hipEventRecord(c->ev_pfx, stream));
                HIP_TRY(hipStreamWaitEvent(c->aux2_stream, c->ev_pfx, 0));
            }
            if (ref_walk) {
                a.trav_mask = ref_mask;
                const uint32_t blocks = side_by_side ? (uint32_t)c->n_cus * conc_trav_bpc : trav_blocks;
                if (trav_lds_mode == 6) WF_TIMED(1, launch_wf_trav_p(a, blocks, trav_lds_bytes, stream));
                else if (trav_lds_mode >= 4) WF_TIMED(1, launch_wf_trav_i(a, blocks, trav_lds_bytes, trav_lds_mode == 5, &c->big_lds_enabled_i, stream));
                else WF_TIMED(1, launch_wf_trav(a, blocks, trav_lds_mode, trav_lds_bytes, &c->big_lds_enabled, stream));
            }
            if (ts_mask) {
                a.trav_mask = ts_mask;
                // wf_filter_f keeps the class-B paths whose ray enters a two-stage mesh's root box; wf_trav_f and wf_replay work on that list
                if (side_by_side) {
                    WF_TIMED_ON(3, c->aux2_stream, launch_wf_filter_f(a, 8u, c->aux2_stream));
                    WF_TIMED_ON(3, c->aux2_stream, launch_wf_trav_f(a, (uint32_t)c->n_cus * conc_travf_bpc, c->aux2_stream));
                    HIP_TRY(hipEventRecord(c->ev_travf, c->aux2_stream));
                    HIP_TRY(hipStreamWaitEvent(stream, c->ev_travf, 0));
                } else {
                    WF_TIMED(3, launch_wf_filter_f(a, 8u, stream));
                    WF_TIMED(3, launch_wf_trav_f(a, travf_blocks, stream));
                }
                WF_TIMED(4, launch_wf_replay(a, replay_blocks, stream));
            }
            cur ^= 1;
            a.iter0 = 0;
            if (tail && !have_walkers) { it++; break; }
        }
        WF_TIMED(2, launch_wf_reduce(a, s0 == 0, s0 + a.s_count >= spp, stream));
        // the headers not read yet (statistics; passes launched behind the one that ended every path are not counted)
        const uint32_t launched = c->hdr_seq + 1u - seq0;
        while (seen < launched) { int rcw = wait_header(seq0 + seen); if (rcw != MI_OK) return rcw; consume(!all_dead); }
    }
#undef WF_TIMED
#undef WF_TIMED_ON
    HIP_TRY(hipStreamSynchronize(stream));
    for (int k = 0; k < 8; k++) c->wf_counts[k] = counts[k];
    for (int k = 0; k < 8; k++) c->wf_ms[k] = 0.0f;
    c->wf_ms[3] = (float)(ev_used / 2);
    for (size_t e = 0; e + 1 < ev_used; e += 2) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, c->wf_ev[e], c->wf_ev[e + 1]) == hipSuccess) c->wf_ms[ev_kind[e] < 3 ? ev_kind[e] : ev_kind[e] + 1] += ms;
        if (c->tune.dump_launches) {
            static const char* const names[6] = { "wf_main", "wf_trav", "wf_reduce", "wf_trav_f", "wf_replay", "wf_main (class A, beside the walkers)" };
            fprintf(stderr, "[mi_rt] launch %zu %s %.4f ms\n", e / 2, names[ev_kind[e]], ms);
        }
    }
    return MI_OK;
}

static int render_tiles(mi_ctx* c, const mi_camera_desc* cam, const mi_render_opts* o, float* d_compact,
                        uint32_t* d_sig, hipStream_t stream, mi_stats* st, const SampleRange* partial = nullptr) {
    int rc = check_camera(cam);
    if (rc != MI_OK) return rc;
    if (!c->have_scene) return fail(MI_ERR_NO_SCENE, "no scene uploaded");
    if (!o || o->world < 1 || o->rank < 0 || o->rank >= o->world) return fail(MI_ERR_INVALID, "bad rank/world");
    SampleRange range = { 0u, cam->aa_sample_count, nullptr };
    if (partial) {
        range = *partial;
        if (range.begin >= range.end || range.end > cam->aa_sample_count)
            return fail(MI_ERR_INVALID, "sample range [%u, %u) is not inside [0, %u)", range.begin, range.end, cam->aa_sample_count);
        if (!range.accum) return fail(MI_ERR_INVALID, "accumulator buffer is NULL");
        if (cam->shading_mode != MI_SHADE_PATHTRACE || cam->path_samples != 1 || (o->variant != MI_VARIANT_DEFAULT && o->variant != MI_VARIANT_WAVEFRONT))
            return fail(MI_ERR_UNSUPPORTED, "progressive rendering runs on the default (wavefront) path-tracing variant only");
    }
    const bool writes_image = range.end == cam->aa_sample_count;
    if (!d_compact && writes_image) return fail(MI_ERR_INVALID, "output buffer is NULL");
    K1Args a;
    a.S = c->S;
    if (o->flags & MI_OPT_NO_LIST_TREE) { a.S.top_meshf = -1; a.S.n_list_lin = a.S.n_list_tri; }      // every list Triangle one by one (cross-check)
    make_camera(cam, &a.C);
    for (int k = 0; k < 3; k++) { a.C.light[k] = c->point_light_pos[k]; a.C.ambient[k] = c->ambient[k]; }
    const bool phong = cam->shading_mode == MI_SHADE_PHONG;      // debug shader: own kernel, `variant` is ignored
    // path_samples != 1 (tracing.rs:310) branches at every hit: the literal, recursive estimator (pt_branch)
    const bool recursive = !phong && (cam->path_samples != 1 || o->variant == MI_VARIANT_RECURSIVE);
    if (recursive && cam->path_depth > 64) return fail(MI_ERR_UNSUPPORTED, "recursive estimator: path_depth must be <= 64");
    uint32_t tx, ty, total, padded;
    tile_counts(cam, o->world, &tx, &ty, &total, &padded);
    a.R.seed = o->seed; a.R.rank = o->rank; a.R.world = o->world;
    a.R.tiles_x = tx; a.R.tiles_y = ty; a.R.tiles_total = total;
    a.R.my_tiles = (total > (uint32_t)o->rank) ? (total - (uint32_t)o->rank + (uint32_t)o->world - 1) / (uint32_t)o->world : 0;
    bool lds = c->S.n_meshes > 0 && c->lds_bytes <= 64u * 1024u && !c->tune.global_bvh;
    a.R.lds_nodes = lds ? (uint32_t)c->S.n_nodes : 0;
    a.R.lds_tris = lds ? (uint32_t)c->S.n_tris : 0;
    a.seed_key = lowbias32(o->seed ^ 0x68e31da4u);
    a.out = d_compact;
    a.sig = (o->want_signature && d_sig) ? d_sig : nullptr;
    int variant = o->variant == MI_VARIANT_DEFAULT ? MI_VARIANT_WAVEFRONT : o->variant;
    if (variant != MI_VARIANT_SIMPLE && variant != MI_VARIANT_VOTED && variant != MI_VARIANT_VOTED_DIAG &&
        variant != MI_VARIANT_WAVEFRONT && variant != MI_VARIANT_RECURSIVE)
        return fail(MI_ERR_INVALID, "unknown variant %d (2, 5 and 6 were removed in ABI 3)", o->variant);
    const bool diag = variant == MI_VARIANT_VOTED_DIAG;
    a.R.vote_t = c->tune.vote_t; a.R.vote_a = c->tune.vote_a; a.R.k_steps = c->tune.k_steps;
    a.diag = nullptr;
    uint32_t n_blocks = padded * (uint32_t)kBlocksPerTile;
    if (diag) {
        a.diag = c->d_diag;
        HIP_TRY(hipMemsetAsync(c->d_diag, 0, 16 * sizeof(unsigned long long), stream));
    }
    WfArgs wa; uint32_t wf_batch = 1;
    if (variant == MI_VARIANT_WAVEFRONT && !phong && !recursive) {
        int rcp = wf_prepare(c, cam, padded, o->max_state_bytes, two_stage_mask(c, o->flags) != 0u, wa, wf_batch);
        if (rcp != MI_OK) return rcp;
    }
    HIP_TRY(hipEventRecord(c->ev_start, stream));
    if (phong)
        HIP_TRY(launch_phong(a, n_blocks, a.sig != nullptr, stream));
    else if (recursive)
        HIP_TRY(launch_branch(a, n_blocks, cam->path_samples, a.sig != nullptr, stream));
    else if (variant == MI_VARIANT_WAVEFRONT) {
        int rcw = render_tiles_wavefront(c, a, cam, wa, wf_batch, lds, o->flags, d_compact, a.sig, range, stream);
        if (rcw != MI_OK) return rcw;
    } else if (variant == MI_VARIANT_VOTED || variant == MI_VARIANT_VOTED_DIAG)
        HIP_TRY(launch_megakernel_voted(a, n_blocks, lds, a.sig != nullptr, diag, c->gen_volumes, c->lds_bytes + c->tune.lds_pad, stream));
    else
        HIP_TRY(launch_megakernel(a, n_blocks, lds, a.sig != nullptr, c->lds_bytes, stream));
    HIP_TRY(hipEventRecord(c->ev_stop, stream));
    c->ev_recorded = true;
    if (st) {
        memset(st, 0, sizeof *st);
        uint64_t pixels = 0;
        for (uint32_t t = (uint32_t)o->rank; t < total; t += (uint32_t)o->world) {
            uint32_t x0 = (t % tx) * MI_TILE, y0 = (t / tx) * MI_TILE;
            if (x0 >= cam->screen_width) continue;               // a column of the numbering beyond the image (tile_counts)
            uint32_t w = cam->screen_width - x0 < MI_TILE ? cam->screen_width - x0 : MI_TILE;
            uint32_t h = cam->screen_height - y0 < MI_TILE ? cam->screen_height - y0 : MI_TILE;
            pixels += (uint64_t)w * h;
        }
        st->pixels = pixels;
        st->samples = pixels * (range.end - range.begin);
        st->tiles = a.R.my_tiles; st->tiles_padded = padded;
        st->scene_bytes = (uint32_t)c->blob_bytes; st->scene_in_lds = lds ? 1u : 0u;
    }
    return MI_OK;
}

extern "C" int mi_render_tiles_device(mi_ctx* c, const mi_camera_desc* cam, const mi_render_opts* opts,
                                      void* d_compact_f32, void* d_sig_u32, void* stream, mi_stats* stats) {
    if (!c) return fail(MI_ERR_INVALID, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    return render_tiles(c, cam, opts, (float*)d_compact_f32, (uint32_t*)d_sig_u32, (hipStream_t)stream, stats);
}

extern "C" int mi_render_samples_device(mi_ctx* c, const mi_camera_desc* cam, const mi_render_opts* opts,
                                        uint32_t sample_begin, uint32_t sample_end, void* d_accum_f32x4,
                                        void* d_compact_f32, void* d_sig_u32, void* stream, mi_stats* stats) {
    if (!c) return fail(MI_ERR_INVALID, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    SampleRange r = { sample_begin, sample_end, (float4*)d_accum_f32x4 };
    return render_tiles(c, cam, opts, (float*)d_compact_f32, (uint32_t*)d_sig_u32, (hipStream_t)stream, stats, &r);
}

extern "C" int mi_unpermute_device(mi_ctx* c, const mi_camera_desc* cam, int32_t world, const void* d_gathered_f32,
                                   void* d_image_f32, void* stream) {
    if (!c || !cam || world < 1 || !d_gathered_f32 || !d_image_f32) return fail(MI_ERR_INVALID, "mi_unpermute_device: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    uint32_t tx, ty, total, padded;
    tile_counts(cam, world, &tx, &ty, &total, &padded);
    HIP_TRY(launch_unpermute((const float*)d_gathered_f32, (float*)d_image_f32, cam->screen_width, cam->screen_height, tx,
                             (uint32_t)world, padded, (hipStream_t)stream));
    return MI_OK;
}

extern "C" int mi_tonemap_device(mi_ctx* c, const mi_camera_desc* cam, const void* d_image_f32, void* d_image_u8, void* stream) {
    if (!c || !cam || !d_image_f32 || !d_image_u8) return fail(MI_ERR_INVALID, "mi_tonemap_device: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(launch_tonemap((const float*)d_image_f32, (uint8_t*)d_image_u8, cam->screen_width * cam->screen_height,
                           1.0f / cam->gamma, (hipStream_t)stream));                            // tracing.rs:254
    return MI_OK;
}

extern "C" int mi_last_kernel_ms(mi_ctx* c, float* ms) {
    if (!c || !ms) return fail(MI_ERR_INVALID, "mi_last_kernel_ms: bad argument");
    if (!c->ev_recorded) return fail(MI_ERR_INVALID, "no kernel has been launched on this context");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventSynchronize(c->ev_stop));
    HIP_TRY(hipEventElapsedTime(ms, c->ev_start, c->ev_stop));
    return MI_OK;
}

extern "C" int mi_reserve(mi_ctx* c, const mi_camera_desc* cam, int32_t world, uint64_t max_state_bytes) {
    if (!c) return fail(MI_ERR_INVALID, "ctx is NULL");
    int rc = check_camera(cam);
    if (rc != MI_OK) return rc;
    if (world < 1) return fail(MI_ERR_INVALID, "bad world");
    HIP_TRY(hipSetDevice(c->device));
    uint32_t tx, ty, total, padded;
    tile_counts(cam, world, &tx, &ty, &total, &padded);
    WfArgs a;
    memset(&a, 0, sizeof a);
    a.npix = padded * (uint32_t)kTilePixels;
    uint32_t s_batch = 1;
    return wf_alloc(c, a, cam->aa_sample_count, max_state_bytes, two_stage_mask(c, 0u) != 0u, s_batch);
}

extern "C" int mi_last_pipeline_ms(mi_ctx* c, float* out8) {
    if (!c || !out8) return fail(MI_ERR_INVALID, "mi_last_pipeline_ms: bad argument");
    for (int i = 0; i < 8; i++) out8[i] = c->wf_ms[i];
    return MI_OK;
}

extern "C" int mi_last_pipeline_counts(mi_ctx* c, uint64_t* out8) {
    if (!c || !out8) return fail(MI_ERR_INVALID, "mi_last_pipeline_counts: bad argument");
    for (int i = 0; i < 8; i++) out8[i] = c->wf_counts[i];
    return MI_OK;
}

extern "C" int mi_last_diag(mi_ctx* c, uint64_t* out16) {
    if (!c || !out16) return fail(MI_ERR_INVALID, "mi_last_diag: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out16, c->d_diag, 16 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return MI_OK;
}

extern "C" int mi_selftest(mi_ctx* c, uint64_t* out4) {
    if (!c || !out4) return fail(MI_ERR_INVALID, "mi_selftest: bad argument");
    HIP_TRY(hipSetDevice(c->device));
    for (int k = 0; k < 4; k++) out4[k] = 0;
    HIP_TRY(hipMemsetAsync(c->d_diag, 0, 16 * sizeof(unsigned long long), c->stream));
    HIP_TRY(launch_selftest_rcp(c->d_diag, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    unsigned long long bad = 0;
    HIP_TRY(hipMemcpy(&bad, c->d_diag, sizeof bad, hipMemcpyDeviceToHost));
    out4[0] = bad; out4[1] = 1ull << 32;
    return MI_OK;
}

extern "C" int mi_render(mi_ctx* c, const mi_camera_desc* cam, const mi_render_opts* opts, float* out_rgb_f32,
                         uint8_t* out_rgb_u8, uint32_t* out_sig, mi_stats* stats) {
    if (!c) return fail(MI_ERR_INVALID, "ctx is NULL");
    if (!opts) return fail(MI_ERR_INVALID, "opts is NULL");
    if (opts->rank != 0 || opts->world != 1) return fail(MI_ERR_INVALID, "mi_render renders a whole image: rank/world must be 0/1");
    int rc = check_camera(cam);
    if (rc != MI_OK) return rc;
    HIP_TRY(hipSetDevice(c->device));
    const hipEvent_t t0 = c->ev_t0, t1 = c->ev_t1;      // owned by the context: no early return can leak them
    uint32_t tx, ty, total, padded;
    tile_counts(cam, 1, &tx, &ty, &total, &padded);
    size_t npix = (size_t)cam->screen_width * cam->screen_height;
    size_t cbytes = (size_t)padded * kTilePixels * 3 * sizeof(float);
    if ((rc = ensure((void**)&c->d_compact, &c->compact_bytes, cbytes)) != MI_OK) return rc;
    if ((rc = ensure((void**)&c->d_image, &c->image_bytes, npix * 3 * sizeof(float))) != MI_OK) return rc;
    bool want_sig = opts->want_signature && out_sig;
    if (want_sig) {
        if ((rc = ensure((void**)&c->d_sigc, &c->sigc_bytes, (size_t)padded * kTilePixels * 4)) != MI_OK) return rc;
        if ((rc = ensure((void**)&c->d_sigi, &c->sigi_bytes, npix * 4)) != MI_OK) return rc;
    }
    mi_render_opts o = *opts;
    o.want_signature = want_sig ? 1 : 0;
    HIP_TRY(hipEventRecord(t0, c->stream));
    rc = render_tiles(c, cam, &o, c->d_compact, want_sig ? c->d_sigc : nullptr, c->stream, stats);
    if (rc != MI_OK) return rc;
    HIP_TRY(launch_unpermute(c->d_compact, c->d_image, cam->screen_width, cam->screen_height, tx, 1, padded, c->stream));
    if (out_rgb_f32) HIP_TRY(hipMemcpyAsync(out_rgb_f32, c->d_image, npix * 3 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    if (out_rgb_u8) {
        if ((rc = ensure((void**)&c->d_u8, &c->u8_bytes, npix * 3)) != MI_OK) return rc;
        HIP_TRY(launch_tonemap(c->d_image, c->d_u8, (uint32_t)npix, 1.0f / cam->gamma, c->stream));
        HIP_TRY(hipMemcpyAsync(out_rgb_u8, c->d_u8, npix * 3, hipMemcpyDeviceToHost, c->stream));
    }
    if (want_sig) {
        HIP_TRY(launch_sig_unpermute(c->d_sigc, c->d_sigi, cam->screen_width, cam->screen_height, tx, 1, padded, c->stream));
        HIP_TRY(hipMemcpyAsync(out_sig, c->d_sigi, npix * 4, hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(hipEventRecord(t1, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (stats) {
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, c->ev_start, c->ev_stop)); stats->kernel_ms = ms;
        HIP_TRY(hipEventElapsedTime(&ms, t0, t1)); stats->total_ms = ms;
    }
    return MI_OK;
}

// ------------------------------------------------------------------ multi-GPU behind the ABI
// One process drives N devices: one mi_ctx, one stream and (per frame) one host thread per device, the image cut into
// 32x32 tiles with tile t on device t mod N (the partition of dist.py), and exactly ONE exchange per frame: every peer
// SENDS its compact tile buffer to device 0 over its own xGMI link (RCCL ncclSend / ncclRecv inside one group: a fan-in,
// not a ring — xGMI is point-to-point, 7 links per GPU, so the 7 transfers of 3.1 MB run side by side), then K3 + K4 on
// device 0.  RCCL is resolved at run time (dlopen), so single-GPU users of this library do not need it installed.
#include <dlfcn.h>
#include <rccl/rccl.h>

namespace {
struct Rccl {
    void* so = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool load(std::string* why) {
        if (so) return true;
        const char* names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
        for (const char* n : names) { so = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (so) break; }
        if (!so) { *why = std::string("cannot load RCCL: ") + (dlerror() ? dlerror() : "?"); return false; }
#define RCCL_SYM(field, name) field = (decltype(field))dlsym(so, name); if (!field) { *why = std::string("RCCL lacks ") + name; return false; }
        RCCL_SYM(CommInitAll, "ncclCommInitAll"); RCCL_SYM(CommDestroy, "ncclCommDestroy"); RCCL_SYM(GroupStart, "ncclGroupStart");
        RCCL_SYM(GroupEnd, "ncclGroupEnd"); RCCL_SYM(Send, "ncclSend"); RCCL_SYM(Recv, "ncclRecv"); RCCL_SYM(GetErrorString, "ncclGetErrorString");
#undef RCCL_SYM
        return true;
    }
};
Rccl g_rccl;
}  // namespace

struct mi_multi {
    std::vector<int> devices;
    std::vector<mi_ctx*> ctx;
    std::vector<ncclComm_t> comm;
    // device 0: gathered[world][tiles_padded][1024][3] (its own share is rendered straight into slice 0);
    // every other device: its compact buffer.  Signatures likewise.
    std::vector<void*> d_compact; std::vector<size_t> compact_bytes;
    std::vector<void*> d_sig; std::vector<size_t> sig_bytes;
    void* d_sig_image = nullptr; size_t sig_image_bytes = 0;
    // test transport (mi_multi_create_loopback): every rank on ONE device, no RCCL; the exchange is a D2D copy per peer on the
    // peer's stream, ordered into device 0's stream by ev_sent[r]
    bool loopback = false;
    std::vector<hipEvent_t> ev_sent;
};

#define RCCL_TRY(expr)                                                                              \
    do {                                                                                            \
        ncclResult_t r_ = (expr);                                                                   \
        if (r_ != ncclSuccess) return fail(MI_ERR_HIP, "%s failed: %s", #expr, g_rccl.GetErrorString(r_)); \
    } while (0)

extern "C" void mi_multi_destroy(mi_multi* m) {
    if (!m) return;
    for (size_t r = 0; r < m->ctx.size(); r++) {
        if (!m->ctx[r]) continue;
        (void)hipSetDevice(m->devices[r]);
        (void)hipDeviceSynchronize();
        if (r < m->comm.size() && m->comm[r]) (void)g_rccl.CommDestroy(m->comm[r]);
        if (r < m->ev_sent.size() && m->ev_sent[r]) (void)hipEventDestroy(m->ev_sent[r]);
        if (r < m->d_compact.size() && m->d_compact[r]) (void)hipFree(m->d_compact[r]);
        if (r < m->d_sig.size() && m->d_sig[r]) (void)hipFree(m->d_sig[r]);
        if (r == 0 && m->d_sig_image) (void)hipFree(m->d_sig_image);
        mi_ctx_destroy(m->ctx[r]);
    }
    delete m;
}

extern "C" int mi_multi_create(int n_devices, const int* devices, mi_multi** out) {
    if (!out) return fail(MI_ERR_INVALID, "mi_multi_create: out is NULL");
    *out = nullptr;
    if (n_devices < 1 || n_devices > 64) return fail(MI_ERR_INVALID, "mi_multi_create: n_devices %d out of range", n_devices);
    mi_multi* m = new mi_multi();
    for (int r = 0; r < n_devices; r++) {
        m->devices.push_back(devices ? devices[r] : r);
        for (int q = 0; q < r; q++) if (m->devices[(size_t)q] == m->devices[(size_t)r]) {
            const int dup = m->devices[(size_t)r];
            delete m;
            return fail(MI_ERR_INVALID, "mi_multi_create: device %d listed twice", dup);
        }
    }
    m->ctx.assign((size_t)n_devices, nullptr); m->comm.assign((size_t)n_devices, nullptr);
    m->d_compact.assign((size_t)n_devices, nullptr); m->compact_bytes.assign((size_t)n_devices, 0);
    m->d_sig.assign((size_t)n_devices, nullptr); m->sig_bytes.assign((size_t)n_devices, 0);
    for (int r = 0; r < n_devices; r++) {
        const int rc = mi_ctx_create(m->devices[(size_t)r], &m->ctx[(size_t)r]);
        if (rc != MI_OK) { const std::string msg = g_err; mi_multi_destroy(m); g_err = msg; return rc; }
    }
    std::string why;
    if (!g_rccl.load(&why)) { mi_multi_destroy(m); return fail(MI_ERR_UNSUPPORTED, "%s", why.c_str()); }
    const ncclResult_t r = g_rccl.CommInitAll(m->comm.data(), n_devices, m->devices.data());
    if (r != ncclSuccess) {
        for (auto& cm : m->comm) cm = nullptr;
        mi_multi_destroy(m);
        return fail(MI_ERR_HIP, "ncclCommInitAll(%d devices) failed: %s", n_devices, g_rccl.GetErrorString(r));
    }
    *out = m;
    return MI_OK;
}

extern "C" int mi_multi_create_loopback(int n_contexts, int device, mi_multi** out) {
    if (!out) return fail(MI_ERR_INVALID, "mi_multi_create_loopback: out is NULL");
    *out = nullptr;
    if (n_contexts < 1 || n_contexts > 64) return fail(MI_ERR_INVALID, "mi_multi_create_loopback: n_contexts %d out of range", n_contexts);
    mi_multi* m = new mi_multi();
    m->loopback = true;
    m->devices.assign((size_t)n_contexts, device);
    m->ctx.assign((size_t)n_contexts, nullptr); m->ev_sent.assign((size_t)n_contexts, nullptr);
    m->d_compact.assign((size_t)n_contexts, nullptr); m->compact_bytes.assign((size_t)n_contexts, 0);
    m->d_sig.assign((size_t)n_contexts, nullptr); m->sig_bytes.assign((size_t)n_contexts, 0);
    for (int r = 0; r < n_contexts; r++) {
        int rc = mi_ctx_create(device, &m->ctx[(size_t)r]);
        if (rc == MI_OK && hipEventCreateWithFlags(&m->ev_sent[(size_t)r], hipEventDisableTiming) != hipSuccess)
            rc = fail(MI_ERR_HIP, "hipEventCreateWithFlags failed");
        if (rc != MI_OK) { const std::string msg = g_err; mi_multi_destroy(m); g_err = msg; return rc; }
    }
    *out = m;
    return MI_OK;
}

extern "C" int mi_multi_device_count(const mi_multi* m) { return m ? (int)m->ctx.size() : 0; }
extern "C" mi_ctx* mi_multi_context(const mi_multi* m, int rank) {
    if (!m || rank < 0 || (size_t)rank >= m->ctx.size()) { (void)fail(MI_ERR_INVALID, "mi_multi_context: rank %d out of range", rank); return nullptr; }
    return m->ctx[(size_t)rank];
}

// Run fn(rank) on one host thread per device; the first failure (code + message) is reported in the caller's thread.
template <class F> static int on_every_device(mi_multi* m, F fn) {
    const size_t n = m->ctx.size();
    std::vector<int> rc(n, MI_OK); std::vector<std::string> msg(n);
    auto body = [&](size_t r) { rc[r] = fn((int)r); if (rc[r] != MI_OK) msg[r] = g_err; };
    std::vector<std::thread> th;
    for (size_t r = 1; r < n; r++) th.emplace_back(body, r);
    body(0);
    for (auto& t : th) t.join();
    for (size_t r = 0; r < n; r++) if (rc[r] != MI_OK) { g_err = "device " + std::to_string(m->devices[r]) + ": " + msg[r]; return rc[r]; }
    return MI_OK;
}

// Loopback ranks share one card: "60 % of the free HBM" per rank would be claimed `world` times over by threads that all look at
// the same free figure.  With no budget given each rank gets an even share of that default (what the ranks already hold counts as
// free).  ~0 = failure (message recorded).  The RCCL route (one rank per device) passes the caller's value through.
static uint64_t loopback_budget(const mi_multi* m, uint64_t max_state_bytes) {
    if (!m->loopback || max_state_bytes != 0 || m->ctx.size() < 2) return max_state_bytes;
    size_t free_b = 0, total_b = 0;
    if (hipSetDevice(m->devices[0]) != hipSuccess || hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)fail(MI_ERR_HIP, "hipMemGetInfo failed"); return ~0ull; }
    for (const mi_ctx* c : m->ctx) free_b += c->wf_a_bytes + c->wf_b_bytes + c->wf_samp_bytes + c->cand_bytes + c->cand_hdr_bytes;
    return (uint64_t)((double)free_b * 0.6 / (double)m->ctx.size());
}

extern "C" int mi_multi_scene_upload(mi_multi* m, const mi_scene_desc* scene) {
    if (!m || !scene) return fail(MI_ERR_INVALID, "mi_multi_scene_upload: NULL argument");
    // the scene is small (KBs .. a few hundred MB of textures) and 288 GB per GPU make replication free
    return on_every_device(m, [&](int r) { return mi_scene_upload(m->ctx[(size_t)r], scene); });
}

extern "C" int mi_multi_reserve(mi_multi* m, const mi_camera_desc* cam, uint64_t max_state_bytes) {
    if (!m) return fail(MI_ERR_INVALID, "mi_multi_reserve: NULL argument");
    const int world = (int)m->ctx.size();
    if ((max_state_bytes = loopback_budget(m, max_state_bytes)) == ~0ull) return MI_ERR_HIP;
    return on_every_device(m, [&](int r) { return mi_reserve(m->ctx[(size_t)r], cam, world, max_state_bytes); });
}

extern "C" int mi_multi_render(mi_multi* m, const mi_camera_desc* cam, const mi_render_opts* opts, float* out_rgb_f32,
                               uint8_t* out_rgb_u8, uint32_t* out_sig, mi_stats* stats) {
    if (!m || !opts) return fail(MI_ERR_INVALID, "mi_multi_render: NULL argument");
    int rc = check_camera(cam);
    if (rc != MI_OK) return rc;
    const int world = (int)m->ctx.size();
    uint32_t tx, ty, total, padded;
    tile_counts(cam, world, &tx, &ty, &total, &padded);
    const size_t slice_f = (size_t)padded * kTilePixels * 3, slice_s = (size_t)padded * kTilePixels;     // elements per rank
    const size_t npix = (size_t)cam->screen_width * cam->screen_height;
    const bool want_sig = opts->want_signature && out_sig;
    const auto t_begin = std::chrono::steady_clock::now();
    // buffers: device 0 holds the gathered array, the peers their own slice
    for (int r = 0; r < world; r++) {
        HIP_TRY(hipSetDevice(m->devices[(size_t)r]));
        const size_t nf = (r == 0 ? (size_t)world : 1) * slice_f * sizeof(float), ns = (r == 0 ? (size_t)world : 1) * slice_s * 4;
        if ((rc = ensure(&m->d_compact[(size_t)r], &m->compact_bytes[(size_t)r], nf)) != MI_OK) return rc;
        if (want_sig && (rc = ensure(&m->d_sig[(size_t)r], &m->sig_bytes[(size_t)r], ns)) != MI_OK) return rc;
    }
    mi_ctx* c0 = m->ctx[0];
    HIP_TRY(hipSetDevice(m->devices[0]));
    if ((rc = ensure((void**)&c0->d_image, &c0->image_bytes, npix * 3 * sizeof(float))) != MI_OK) return rc;
    if ((rc = ensure((void**)&c0->d_u8, &c0->u8_bytes, npix * 3)) != MI_OK) return rc;     // K4 always runs: the u8 image stays resident on device 0
    if (want_sig && (rc = ensure(&m->d_sig_image, &m->sig_image_bytes, npix * 4)) != MI_OK) return rc;

    // ---- every device renders its tiles (one host thread each; the wavefront pipeline drives its passes from the host) ----
    std::vector<mi_stats> st((size_t)world);
    const uint64_t budget = loopback_budget(m, opts->max_state_bytes);
    if (budget == ~0ull) return MI_ERR_HIP;
    rc = on_every_device(m, [&](int r) {
        mi_ctx* c = m->ctx[(size_t)r];
        if (hipSetDevice(c->device) != hipSuccess) return fail(MI_ERR_HIP, "hipSetDevice failed");
        mi_render_opts o = *opts;
        o.max_state_bytes = budget;
        o.rank = r; o.world = world; o.want_signature = want_sig ? 1 : 0;
        const int rr = render_tiles(c, cam, &o, (float*)m->d_compact[(size_t)r], want_sig ? (uint32_t*)m->d_sig[(size_t)r] : nullptr, c->stream, &st[(size_t)r]);
        if (rr != MI_OK) return rr;
        if (hipStreamSynchronize(c->stream) != hipSuccess) return fail(MI_ERR_HIP, "hipStreamSynchronize failed");
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, c->ev_start, c->ev_stop) == hipSuccess) st[(size_t)r].kernel_ms = ms;
        return (int)MI_OK;
    });
    if (rc != MI_OK) return rc;

    // ---- the frame's single exchange: fan-in of the compact buffers to device 0 (one group, issued from this thread) ----
    // A failure between GroupStart and GroupEnd must not leave the group open (every later RCCL call of this thread would
    // silently join it): the first error is kept, the remaining calls of the group are skipped, GroupEnd always runs.
    if (world > 1 && m->loopback) {
        // test transport: the same slices to the same offsets, each Send / Recv pair replaced by one copy on the SENDER's stream
        // (behind its render) that device 0's stream waits for — the ordering the RCCL pair gives
        for (int r = 1; r < world; r++) {
            hipStream_t sr = m->ctx[(size_t)r]->stream;
            HIP_TRY(hipMemcpyAsync((float*)m->d_compact[0] + (size_t)r * slice_f, m->d_compact[(size_t)r], slice_f * sizeof(float), hipMemcpyDeviceToDevice, sr));
            if (want_sig)
                HIP_TRY(hipMemcpyAsync((uint32_t*)m->d_sig[0] + (size_t)r * slice_s, m->d_sig[(size_t)r], slice_s * 4, hipMemcpyDeviceToDevice, sr));
            HIP_TRY(hipEventRecord(m->ev_sent[(size_t)r], sr));
            HIP_TRY(hipStreamWaitEvent(c0->stream, m->ev_sent[(size_t)r], 0));
        }
    } else if (world > 1) {
        ncclResult_t first = ncclSuccess; const char* what = "";
        auto step = [&](ncclResult_t r, const char* name) { if (first == ncclSuccess && r != ncclSuccess) { first = r; what = name; } return first == ncclSuccess; };
        RCCL_TRY(g_rccl.GroupStart());
        for (int r = 1; r < world; r++) {
            if (!step(g_rccl.Recv((float*)m->d_compact[0] + (size_t)r * slice_f, slice_f, ncclFloat, r, m->comm[0], c0->stream), "ncclRecv")) break;
            if (!step(g_rccl.Send(m->d_compact[(size_t)r], slice_f, ncclFloat, 0, m->comm[(size_t)r], m->ctx[(size_t)r]->stream), "ncclSend")) break;
            if (want_sig) {
                if (!step(g_rccl.Recv((uint32_t*)m->d_sig[0] + (size_t)r * slice_s, slice_s, ncclUint32, r, m->comm[0], c0->stream), "ncclRecv (signatures)")) break;
                if (!step(g_rccl.Send(m->d_sig[(size_t)r], slice_s, ncclUint32, 0, m->comm[(size_t)r], m->ctx[(size_t)r]->stream), "ncclSend (signatures)")) break;
            }
        }
        const ncclResult_t e_end = g_rccl.GroupEnd();
        if (first != ncclSuccess) return fail(MI_ERR_HIP, "%s failed inside the frame's exchange: %s", what, g_rccl.GetErrorString(first));
        if (e_end != ncclSuccess) return fail(MI_ERR_HIP, "ncclGroupEnd failed: %s", g_rccl.GetErrorString(e_end));
    }
    // ---- K3 + K4 on device 0, in stream order behind the receives ----
    HIP_TRY(hipSetDevice(m->devices[0]));
    HIP_TRY(launch_unpermute((const float*)m->d_compact[0], c0->d_image, cam->screen_width, cam->screen_height, tx, (uint32_t)world, padded, c0->stream));
    if (out_rgb_f32) HIP_TRY(hipMemcpyAsync(out_rgb_f32, c0->d_image, npix * 3 * sizeof(float), hipMemcpyDeviceToHost, c0->stream));
    HIP_TRY(launch_tonemap(c0->d_image, c0->d_u8, (uint32_t)npix, 1.0f / cam->gamma, c0->stream));
    if (out_rgb_u8) HIP_TRY(hipMemcpyAsync(out_rgb_u8, c0->d_u8, npix * 3, hipMemcpyDeviceToHost, c0->stream));
    if (want_sig) {
        HIP_TRY(launch_sig_unpermute((const uint32_t*)m->d_sig[0], (uint32_t*)m->d_sig_image, cam->screen_width, cam->screen_height, tx, (uint32_t)world, padded, c0->stream));
        HIP_TRY(hipMemcpyAsync(out_sig, m->d_sig_image, npix * 4, hipMemcpyDeviceToHost, c0->stream));
    }
    for (int r = world - 1; r >= 0; r--) {          // the peers' sends, then device 0
        HIP_TRY(hipSetDevice(m->devices[(size_t)r]));
        HIP_TRY(hipStreamSynchronize(m->ctx[(size_t)r]->stream));
    }
    if (stats) {
        memset(stats, 0, sizeof *stats);
        for (int r = 0; r < world; r++) {
            stats->samples += st[(size_t)r].samples; stats->pixels += st[(size_t)r].pixels; stats->tiles += st[(size_t)r].tiles;
            stats->kernel_ms = std::max(stats->kernel_ms, st[(size_t)r].kernel_ms);             // the slowest device's pipeline pass
        }
        stats->tiles_padded = padded; stats->scene_bytes = st[0].scene_bytes; stats->scene_in_lds = st[0].scene_in_lds;
        stats->total_ms = (float)std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t_begin).count() * 1e-3f;
    }
    return MI_OK;
}
