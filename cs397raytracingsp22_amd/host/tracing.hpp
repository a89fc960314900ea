// tracing.hpp — C++ host mirror of src/util/tracing.rs: Camera and Scene with the reference's
// field names.  Scene::render_to_image() keeps its meaning (tracing.rs:221-263) and becomes
// flatten -> mi_scene_upload -> mi_render through the C ABI of include/mi_rt.h.  Errors that the
// reference raises as panics surface as std::runtime_error carrying mi_last_error().
#pragma once
#include <stdexcept>
#include <string>
#include <vector>
#include "geometry.hpp"

namespace cs397 {

enum class CameraProjectionMode { Orthographic = MI_PROJ_ORTHOGRAPHIC, Perspective = MI_PROJ_PERSPECTIVE };   // tracing.rs:27-30
enum class ShadingMode { Phong = MI_SHADE_PHONG, PathTrace = MI_SHADE_PATHTRACE };                             // tracing.rs:32-35

struct Camera {                                        // tracing.rs:138-155, same fields
    Vec3 eyepoint{0.0f, 2.0f, 5.5f}; Vec3 view_dir{0.0f, 0.0f, -1.0f}; Vec3 up{0.0f, 1.0f, 0.0f};
    CameraProjectionMode projection_mode = CameraProjectionMode::Perspective;
    ShadingMode shading_mode = ShadingMode::PathTrace;
    uint32_t path_depth = 10, path_samples = 1, screen_width = 100, screen_height = 100;
    float focal_length = 0.6f, focus_dist = 5.0f, lens_radius = 0.0f;
    uint32_t aa_sample_count = 100;
    float max_trace_dist = 100.0f, gamma = 2.0f;

    mi_camera_desc flatten() const {
        mi_camera_desc c{};
        for (int i = 0; i < 3; i++) { c.eyepoint[i] = eyepoint[i]; c.view_dir[i] = view_dir[i]; c.up[i] = up[i]; }
        c.projection_mode = (int)projection_mode; c.shading_mode = (int)shading_mode;
        c.path_depth = path_depth; c.path_samples = path_samples; c.screen_width = screen_width; c.screen_height = screen_height;
        c.focal_length = focal_length; c.focus_dist = focus_dist; c.lens_radius = lens_radius;
        c.aa_sample_count = aa_sample_count; c.max_trace_dist = max_trace_dist; c.gamma = gamma;
        return c;
    }
};

struct RgbImage { uint32_t width = 0, height = 0; std::vector<uint8_t> data; };   // image::RgbImage byte layout

inline void mi_check(int rc) { if (rc != MI_OK) throw std::runtime_error(std::string("mi_rt: ") + mi_last_error()); }

struct Scene : Intersectable {                         // tracing.rs:213-218; `impl Intersectable for Scene` :326
    Camera camera;
    std::vector<IntersectableRef> objects;
    const std::vector<IntersectableRef>* scene_objects() const override { return &objects; }
    void flatten(SceneBuilder& sb) const override { for (auto& o : objects) o->flatten(sb); }
    Vec3 point_light_pos{0.0f, 1.0f, 5.0f};            // read by ShadingMode::Phong only (tracing.rs:282,288)
    Vec3 ambient{0.1f, 0.1f, 0.1f};                    // Phong only (:292)

    // Scene::render_to_image (tracing.rs:221-263).  seed: the reference RNG is unseeded; device: HIP ordinal.
    // n_gpus > 0 renders on the first n_gpus devices of the node through mi_multi_* (tiles t % n, one RCCL fan-in per
    // frame inside the library) — the drop-in for rayon's row parallelism (tracing.rs:228); the image is the same.
    RgbImage render_to_image(uint32_t seed = 1, int device = 0, mi_stats* stats = nullptr, std::vector<float>* linear = nullptr,
                             int n_gpus = 0) const {
        SceneBuilder sb;
        for (auto& o : objects) o->flatten(sb);
        mi_scene_desc d = sb.desc();
        for (int k = 0; k < 3; k++) { d.point_light_pos[k] = point_light_pos[k]; d.ambient[k] = ambient[k]; }
        mi_camera_desc cam = camera.flatten();
        RgbImage img; img.width = camera.screen_width; img.height = camera.screen_height;
        img.data.resize((size_t)img.width * img.height * 3);
        if (linear) linear->resize(img.data.size());
        mi_render_opts opts{}; opts.seed = seed; opts.rank = 0; opts.world = 1;
        int rc; std::string err;
        if (n_gpus > 0) {
            mi_multi* m = nullptr;
            mi_check(mi_multi_create(n_gpus, nullptr, &m));
            rc = mi_multi_scene_upload(m, &d);
            if (rc == MI_OK) rc = mi_multi_render(m, &cam, &opts, linear ? linear->data() : nullptr, img.data.data(), nullptr, stats);
            err = rc == MI_OK ? "" : mi_last_error();
            mi_multi_destroy(m);
        } else {
            mi_ctx* ctx = nullptr;
            mi_check(mi_ctx_create(device, &ctx));
            rc = mi_scene_upload(ctx, &d);
            if (rc == MI_OK) rc = mi_render(ctx, &cam, &opts, linear ? linear->data() : nullptr, img.data.data(), nullptr, stats);
            err = rc == MI_OK ? "" : mi_last_error();
            mi_ctx_destroy(ctx);
        }
        if (rc != MI_OK) throw std::runtime_error("mi_rt: " + err);
        return img;
    }
};

}  // namespace cs397
