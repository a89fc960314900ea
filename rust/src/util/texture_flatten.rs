// texture_flatten.rs — pasted into src/util/texture.rs by `include!("texture_flatten.rs");` (last line of that file; `img` is a
// private field, texture.rs:13, so this has to live in that module).  UNVERIFIED by a compiler (see mi_rt.rs).
//
// Texture::sample (texture.rs:26-32) reads `self.img.get_pixel(x, y).to_rgb()`: whatever the file's colour type, the texel the
// path sees is its RGB8 conversion.  `DynamicImage::to_rgb8()` (image 0.23.14) performs that conversion for the whole image with
// the same per-pixel `to_rgb()`, rows top to bottom — exactly the layout mi_texture.rgb wants — so a JPEG decoded by the `image`
// crate reaches the GPU with the crate's own bytes (no second decoder, no +-2 LSB IDCT difference).

impl Texture {
    /// Pushes this texture's RGB8 bytes into the builder; returns its index for mi_mesh.textures.
    pub fn flatten(&self, out: &mut super::mi_rt::SceneBuilder) -> i32 {
        let rgb = self.img.to_rgb8();
        let (w, h) = (rgb.width(), rgb.height());
        out.texture_rgb8(w, h, rgb.into_raw())
    }
}
