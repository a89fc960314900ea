/*
 * orc_texture.c — TEST INFRASTRUCTURE (oracle).  Restates Texture::sample of
 * src/util/texture.rs:26-32.  Texture::load_from_file (:16-25, image::open) is
 * load-time work outside the path: textures arrive pre-decoded as RGB8.
 */
#include "orc_internal.h"

v3 orc_texture_sample_v(const mi_texture* t, v2 uv, orc_path* p) {
    if (p && p->cnt) p->cnt->texel_fetches++;
    uint32_t W = (uint32_t)t->width, H = (uint32_t)t->height;
    /* `as u32` truncates toward zero and saturates; the operand is in [0, W) here */
    uint32_t x = (uint32_t)(orc_clampf(uv.x, 0.0f, 0.999f) * (float)W);            /* :28 */
    if (x > W - 1u) x = W - 1u;
    uint32_t y = (uint32_t)((1.0f - orc_clampf(uv.y, 0.0f, 0.999f)) * (float)H);   /* :29 */
    if (y > H - 1u) y = H - 1u;
    const uint8_t* px = t->rgb + ((size_t)y * W + x) * 3;                          /* :30 */
    return v3_make((float)px[0] / 255.0f, (float)px[1] / 255.0f, (float)px[2] / 255.0f);   /* :31 */
}

void orc_texture_sample(const mi_texture* t, float u, float v, float out[3]) {
    v3 c = orc_texture_sample_v(t, v2_make(u, v), NULL);
    out[0] = c.x; out[1] = c.y; out[2] = c.z;
}
