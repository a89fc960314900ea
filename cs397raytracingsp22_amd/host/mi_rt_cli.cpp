// mi_rt_cli — C++ caller of the C ABI through the host mirror (tracing.hpp): builds the Cornell
// box (+ optional OBJ mesh) from the reference's types, renders on GPU 0 and writes a binary PPM.
//   mi_rt_cli <out.ppm> <width> <height> <spp> <depth> [mesh.obj | -] [n_gpus] [albedo | -] [normal | -]
//   n_gpus > 0: render through mi_multi_*.  With an albedo and / or a normal map (PNG / TGA / JPEG files, decoded by host/texture.hpp)
//   the mesh is placed and textured like the reference's cube (tracing.rs:385-394: material None, albedo + normal map).
// Replaces the reference's run() (tracing.rs:354-548) as the compiled driver of the path.
#include <cstdio>
#include <cstdlib>
#include "tracing.hpp"

using namespace cs397;

static void quad(std::vector<IntersectableRef>& o, Vec3 p0, Vec3 p1, Vec3 p2, Vec3 p3, MaterialRef m) {
    o.push_back(std::make_shared<Triangle>(p0, p1, p2, m));
    o.push_back(std::make_shared<Triangle>(p0, p2, p3, m));
}

int main(int argc, char** argv) {
    if (argc < 6) { fprintf(stderr, "usage: %s out.ppm width height spp depth [mesh.obj | -] [n_gpus] [albedo | -] [normal | -]\n", argv[0]); return 2; }
    const char* albedo = argc > 8 && std::string(argv[8]) != "-" ? argv[8] : nullptr;
    const char* normal = argc > 9 && std::string(argv[9]) != "-" ? argv[9] : nullptr;
    const int n_gpus = argc > 7 ? atoi(argv[7]) : 0;
    Scene sc;
    sc.camera.eyepoint = {0.0f, 3.0f, 6.6f};
    sc.camera.screen_width = (uint32_t)atoi(argv[2]); sc.camera.screen_height = (uint32_t)atoi(argv[3]);
    sc.camera.aa_sample_count = (uint32_t)atoi(argv[4]); sc.camera.path_depth = (uint32_t)atoi(argv[5]);
    auto grey = std::make_shared<Lambertian>(Color{0.73f, 0.73f, 0.73f});
    auto red = std::make_shared<Lambertian>(Color{0.65f, 0.05f, 0.05f});
    auto green = std::make_shared<Lambertian>(Color{0.12f, 0.45f, 0.15f});
    auto light = std::make_shared<Lambertian>(Color{0.73f, 0.73f, 0.73f}, Color{4.0f, 4.0f, 4.0f});
    const float x0 = -3, x1 = 3, y0 = 0, y1 = 6, z0 = -3, z1 = 3;
    quad(sc.objects, {x0, y0, z1}, {x1, y0, z1}, {x1, y0, z0}, {x0, y0, z0}, grey);
    quad(sc.objects, {x0, y0, z0}, {x1, y0, z0}, {x1, y1, z0}, {x0, y1, z0}, grey);
    quad(sc.objects, {x0, y0, z1}, {x0, y0, z0}, {x0, y1, z0}, {x0, y1, z1}, red);
    quad(sc.objects, {x1, y0, z0}, {x1, y0, z1}, {x1, y1, z1}, {x1, y1, z0}, green);
    quad(sc.objects, {x0, y1, z0}, {x1, y1, z0}, {x1, y1, z1}, {x0, y1, z1}, light);
    sc.objects.push_back(std::make_shared<Sphere>(Vec3{-1.4f, 1.0f, -0.5f}, 1.0f, std::make_shared<Metal>(Color{0.8f, 0.8f, 0.8f}, Color{0, 0, 0}, 0.1f)));
    sc.objects.push_back(std::make_shared<Sphere>(Vec3{1.4f, 1.0f, 0.8f}, 1.0f, std::make_shared<Dielectric>(1.5f)));
    try {
        if (argc > 6 && std::string(argv[6]) != "-") {
            if (albedo || normal)           // tracing.rs:385-394
                sc.objects.push_back(StaticMesh::load_from_file(argv[6], albedo, nullptr, nullptr, nullptr, normal, nullptr,
                                                                Matrix4::from_translation({-1.7f, 0.5f, 2.7f}) * Matrix4::from_angle_y(45.0f) * Matrix4::from_scale(0.4f)));
            else
                sc.objects.push_back(StaticMesh::load_from_file(argv[6], nullptr, nullptr, nullptr, nullptr, nullptr,
                                                                std::make_shared<Lambertian>(Color{0.5f, 0.02f, 0.5f}),
                                                                Matrix4::from_translation({0.0f, 0.9f, 0.0f}) * Matrix4::from_angle_x(-90.0f) * Matrix4::from_scale(2.2f)));
        }
        mi_stats st{};
        RgbImage img = sc.render_to_image(1, 0, &st, nullptr, n_gpus);
        FILE* f = fopen(argv[1], "wb");
        if (!f) { perror(argv[1]); return 1; }
        fprintf(f, "P6\n%u %u\n255\n", img.width, img.height);
        fwrite(img.data.data(), 1, img.data.size(), f);
        fclose(f);
        printf("caller=c++ gpus=%d samples=%llu kernel_ms=%.3f msamples_per_s=%.1f\n", n_gpus > 0 ? n_gpus : 1, (unsigned long long)st.samples, st.kernel_ms,
               st.samples / (st.kernel_ms * 1e3));
    } catch (const std::exception& e) { fprintf(stderr, "error: %s\n", e.what()); return 1; }
    return 0;
}
