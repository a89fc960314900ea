// tracing_flatten.rs — pasted into src/util/tracing.rs by `include!("tracing_flatten.rs");` (last line of that file).
// Camera -> mi_camera_desc (tracing.rs:138-155, field for field), Scene as an Intersectable (tracing.rs:326: only ever the
// boundary of a ConvexVolume), and the new body of Scene::render_to_image (tracing.rs:221): flatten -> ONE FFI call -> RgbImage.
// Needs: `pub trait Intersectable: super::mi_rt::FlattenObject` (tracing.rs:42) and the reference's own render_to_image
// renamed to render_to_image_cpu (tracing.rs:221).  UNVERIFIED by a compiler (see mi_rt.rs).

use super::mi_rt::{self, FlattenObject, SceneBuilder};

impl Camera {
    pub fn flatten(&self) -> mi_rt::mi_camera_desc {
        mi_rt::mi_camera_desc {
            eyepoint: self.eyepoint.into(), view_dir: self.view_dir.into(), up: self.up.into(),
            projection_mode: match self.projection_mode {
                CameraProjectionMode::Orthographic => mi_rt::MI_PROJ_ORTHOGRAPHIC,
                CameraProjectionMode::Perspective => mi_rt::MI_PROJ_PERSPECTIVE,
            },
            shading_mode: match self.shading_mode {
                ShadingMode::Phong => mi_rt::MI_SHADE_PHONG,
                ShadingMode::PathTrace => mi_rt::MI_SHADE_PATHTRACE,
            },
            path_depth: self.path_depth, path_samples: self.path_samples,
            screen_width: self.screen_width, screen_height: self.screen_height,
            focal_length: self.focal_length, focus_dist: self.focus_dist, lens_radius: self.lens_radius,
            aa_sample_count: self.aa_sample_count, max_trace_dist: self.max_trace_dist, gamma: self.gamma,
        }
    }
}

// A Scene used as an Intersectable is its closest hit over `objects`, first entry wins ties (tracing.rs:330-344): flattened
// as its entries in order.  At top level that IS Scene.objects; inside a ConvexVolume boundary the entries are diverted
// (SceneBuilder::begin_boundary) and become a run of mi_scene_desc.boundary_objects.
impl FlattenObject for Scene {
    fn flatten(&self, out: &mut SceneBuilder) {
        for o in self.objects.iter() { o.flatten(out); }
    }
}

impl Scene {
    /// Everything `render_to_image` needs from `self`, as PODs.  The builder borrows the meshes' arrays from `self`.
    pub fn flatten_scene(&self) -> SceneBuilder {
        let mut sb = SceneBuilder::default();
        for o in self.objects.iter() { o.flatten(&mut sb); }          // Scene.objects order is preserved
        sb
    }

    /// Scene::render_to_image (tracing.rs:221-263) on one MI355X: same signature, same RgbImage layout (tracing.rs:226,254-256).
    /// Panics where the reference panics (tracing.rs:546 unwraps): a scene the GPU path does not take is reported, not
    /// silently rendered some other way — call `render_to_image_cpu()` (the reference's own loop) for those.
    pub fn render_to_image(&self) -> RgbImage {
        self.render_to_image_seeded(rand::random::<u32>(), 1)
    }

    /// `seed` makes the image reproducible (the reference's thread_rng is not); `n_gpus` > 1 cuts the image into 32x32 tiles
    /// over the GPUs of this node (mi_multi_*: replaces rayon's row split, tracing.rs:228) — the same image for every n_gpus.
    pub fn render_to_image_seeded(&self, seed: u32, n_gpus: i32) -> RgbImage {
        let sb = self.flatten_scene();
        if let Some(why) = &sb.unsupported { panic!("mi_rt: this scene cannot run on the GPU path: {}", why); }
        let cam = self.camera.flatten();
        let mut desc = sb.desc();
        desc.point_light_pos = self.point_light_pos.into();          // Scene fields read by ShadingMode::Phong (tracing.rs:282,288,292)
        desc.ambient = self.ambient.into();
        let opts = mi_rt::mi_render_opts { seed: seed, rank: 0, world: 1, ..Default::default() };    // flags 0, max_state_bytes 0 = automatic
        let mut img = RgbImage::new(self.camera.screen_width, self.camera.screen_height);
        let null_f32 = std::ptr::null_mut::<f32>();
        let null_u32 = std::ptr::null_mut::<u32>();
        let null_stats = std::ptr::null_mut::<mi_rt::mi_stats>();
        unsafe {
            assert_eq!(mi_rt::mi_abi_version(), mi_rt::MI_RT_ABI_VERSION, "libmi_rt.so and mi_rt.rs disagree on the ABI");
            let rc;
            if n_gpus <= 1 {
                let mut ctx = std::ptr::null_mut();
                assert_eq!(mi_rt::mi_ctx_create(0, &mut ctx), 0, "{}", mi_rt::last_error());
                let up = mi_rt::mi_scene_upload(ctx, &desc);
                rc = if up != 0 { up } else {
                    mi_rt::mi_render(ctx, &cam, &opts, null_f32, img.as_mut_ptr(), null_u32, null_stats)
                };
                let msg = mi_rt::last_error();                         // before destroy: the message is per thread, not per ctx
                mi_rt::mi_ctx_destroy(ctx);
                assert_eq!(rc, 0, "{}", msg);
            } else {
                let mut m = std::ptr::null_mut();
                assert_eq!(mi_rt::mi_multi_create(n_gpus, std::ptr::null(), &mut m), 0, "{}", mi_rt::last_error());
                let up = mi_rt::mi_multi_scene_upload(m, &desc);
                rc = if up != 0 { up } else {
                    mi_rt::mi_multi_render(m, &cam, &opts, null_f32, img.as_mut_ptr(), null_u32, null_stats)
                };
                let msg = mi_rt::last_error();
                mi_rt::mi_multi_destroy(m);
                assert_eq!(rc, 0, "{}", msg);
            }
        }
        img
    }
}
