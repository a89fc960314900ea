// texture.hpp — C++ host mirror of src/util/texture.rs: `Texture` holds a decoded RGB8 image.
// Texture::sample (texture.rs:26-32) runs on the GPU.  Texture::load_from_file (texture.rs:16-25)
// is load-time work: this mirror reads binary PPM (P6) only and returns an empty optional on any
// failure, as the reference returns None; other formats are decoded by the caller.
#pragma once
#include <cstdint>
#include <cstdio>
#include <optional>
#include <string>
#include <vector>

namespace cs397 {

struct Texture {                                       // texture.rs:12-14
    int width = 0, height = 0;
    std::vector<uint8_t> rgb;                          // width*height*3, row 0 = top

    static std::optional<Texture> load_from_file(const std::string& file_name) {
        FILE* f = fopen(file_name.c_str(), "rb");
        if (!f) return std::nullopt;
        Texture t; int maxv = 0; char magic[3] = {0, 0, 0};
        bool ok = fscanf(f, "%2s %d %d %d", magic, &t.width, &t.height, &maxv) == 4 && magic[0] == 'P' && magic[1] == '6' &&
                  maxv == 255 && t.width > 0 && t.height > 0;
        if (ok) {
            fgetc(f);
            t.rgb.resize((size_t)t.width * t.height * 3);
            ok = fread(t.rgb.data(), 1, t.rgb.size(), f) == t.rgb.size();
        }
        fclose(f);
        if (!ok) return std::nullopt;
        return t;
    }
};

}  // namespace cs397
