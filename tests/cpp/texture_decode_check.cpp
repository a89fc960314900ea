// texture_decode_check.cpp — test driver for cs397::Texture::load_from_file (host/texture.hpp): decodes <in> and writes
// "<width> <height>\n" + RGB8 bytes to <out>; exit code 3 when the loader returns nullopt (the reference's `None`).
#include <cstdio>
#include "../../cs397raytracingsp22_amd/host/texture.hpp"
int main(int argc, char** argv) {
    if (argc < 3) return 2;
    auto t = cs397::Texture::load_from_file(argv[1]);
    if (!t) return 3;
    FILE* f = fopen(argv[2], "wb");
    if (!f) return 2;
    fprintf(f, "%d %d\n", t->width, t->height);
    fwrite(t->rgb.data(), 1, t->rgb.size(), f);
    fclose(f);
    return 0;
}
