// geometry.hpp — C++ host mirror of src/util/geometry.rs: the `Intersectable` implementors with
// the reference's names and fields.  intersect_ray() runs on the GPU (csrc/pt_kernels.hip); each
// type gains one additive method, flatten(), that appends its POD to a SceneBuilder.
#pragma once
#include <array>
#include <cmath>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <optional>
#include <sstream>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include "materials.hpp"
#include "texture.hpp"

namespace cs397 {

// cgmath Matrix4<f32>, column-major: m[col*4 + row]
struct Matrix4 {
    float m[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    static Matrix4 from_translation(Vec3 v) { Matrix4 r; r.m[12] = v[0]; r.m[13] = v[1]; r.m[14] = v[2]; return r; }
    static Matrix4 from_scale(float s) { Matrix4 r; r.m[0] = r.m[5] = r.m[10] = s; return r; }
    // Deg -> (sin, cos).  Radians as cgmath 0.18 converts them (`impl From<Deg<f32>> for Rad<f32>`: deg * (PI / 180 as f32), one f32
    // product of the f32-rounded constant); sine and cosine of THAT in f64 rounded to f32, i.e. the correctly rounded f32 values —
    // Rust's f32::sin_cos is the platform's sinf / cosf (< 1 ulp, almost always the same bits; not pinnable here).  Same definition
    // as the Python mirror (cgmath.py _sc), so both mirrors build the same matrices bit for bit
    static void sincos_deg(float deg, float& s, float& c) {
        const float r = deg * (float)(3.14159265358979323846 / 180.0);
        s = (float)sin((double)r); c = (float)cos((double)r);
    }
    static Matrix4 from_angle_x(float deg) {
        float s, c; sincos_deg(deg, s, c);
        Matrix4 r; r.m[5] = c; r.m[6] = s; r.m[9] = -s; r.m[10] = c; return r;
    }
    static Matrix4 from_angle_y(float deg) {
        float s, c; sincos_deg(deg, s, c);
        Matrix4 r; r.m[0] = c; r.m[2] = -s; r.m[8] = s; r.m[10] = c; return r;
    }
    static Matrix4 from_angle_z(float deg) {
        float s, c; sincos_deg(deg, s, c);
        Matrix4 r; r.m[0] = c; r.m[1] = s; r.m[4] = -s; r.m[5] = c; return r;
    }
    // cgmath's Matrix4 * Matrix4: column c of the product is self * rhs[c] = ((col0 * x + col1 * y) + col2 * z) + col3 * w, every
    // operation rounded to f32 on its own (volatile-free: compiled without FMA contraction, csrc/build.sh)
    Matrix4 operator*(const Matrix4& o) const {
        Matrix4 r;
        for (int c = 0; c < 4; c++) {
            const float x = o.m[c * 4 + 0], y = o.m[c * 4 + 1], z = o.m[c * 4 + 2], w = o.m[c * 4 + 3];
            for (int rw = 0; rw < 4; rw++) {
                const float a = m[0 * 4 + rw] * x, b = m[1 * 4 + rw] * y, cc = m[2 * 4 + rw] * z, dd = m[3 * 4 + rw] * w;
                r.m[c * 4 + rw] = ((a + b) + cc) + dd;
            }
        }
        return r;
    }
    // Matrix4::inverse_transform (geometry.rs:168): general inverse (Gauss-Jordan with partial pivoting in f64, rounded to f32;
    // operation for operation what cgmath.py inverse_transform does)
    std::optional<Matrix4> inverse_transform() const {
        double a[4][8];
        for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) { a[r][c] = m[c * 4 + r]; a[r][4 + c] = r == c; }
        for (int i = 0; i < 4; i++) {
            int p = i; for (int r = i + 1; r < 4; r++) if (fabs(a[r][i]) > fabs(a[p][i])) p = r;
            if (fabs(a[p][i]) < 1e-30) return std::nullopt;
            for (int c = 0; c < 8; c++) std::swap(a[i][c], a[p][c]);
            double d = a[i][i]; for (int c = 0; c < 8; c++) a[i][c] /= d;
            for (int r = 0; r < 4; r++) if (r != i) { double f = a[r][i]; for (int c = 0; c < 8; c++) a[r][c] -= f * a[i][c]; }
        }
        Matrix4 o; for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) o.m[c * 4 + r] = (float)a[r][4 + c];
        return o;
    }
};

// tobj::Mesh with single_index = true (geometry.rs:140-148)
struct Mesh { std::vector<float> positions, normals, texcoords; std::vector<uint32_t> indices; };

// tobj 3.2.0 load_obj{single_index, triangulate}: fan triangulation (0,k,k+1); first model only (geometry.rs:157)
inline std::vector<Mesh> load_obj(const std::string& path) {
    std::ifstream in(path);
    if (!in) throw std::runtime_error("Failed to load OBJ file");           // geometry.rs:149-150
    std::vector<std::array<float, 3>> pos, nrm; std::vector<std::array<float, 2>> tex;
    std::vector<std::vector<std::tuple<int, int, int>>> faces; std::vector<Mesh> models;
    auto flush = [&]() {
        if (faces.empty()) return;
        Mesh me; std::map<std::tuple<int, int, int>, uint32_t> index_map;
        for (auto& poly : faces) for (size_t k = 1; k + 1 < poly.size(); k++) for (auto key : {poly[0], poly[k], poly[k + 1]}) {
            auto it = index_map.find(key);
            if (it == index_map.end()) {
                uint32_t i = (uint32_t)index_map.size(); index_map[key] = i;
                auto [v, vt, vn] = key;
                me.positions.insert(me.positions.end(), pos[v].begin(), pos[v].end());
                if (!tex.empty() && vt >= 0) me.texcoords.insert(me.texcoords.end(), tex[vt].begin(), tex[vt].end());
                if (!nrm.empty() && vn >= 0) me.normals.insert(me.normals.end(), nrm[vn].begin(), nrm[vn].end());
                me.indices.push_back(i);
            } else me.indices.push_back(it->second);
        }
        models.push_back(std::move(me)); faces.clear();
    };
    std::string line;
    while (std::getline(in, line)) {
        std::istringstream ss(line); std::string tag; ss >> tag;
        if (tag == "v") { std::array<float, 3> p{}; ss >> p[0] >> p[1] >> p[2]; pos.push_back(p); }
        else if (tag == "vt") { std::array<float, 2> p{}; ss >> p[0] >> p[1]; tex.push_back(p); }
        else if (tag == "vn") { std::array<float, 3> p{}; ss >> p[0] >> p[1] >> p[2]; nrm.push_back(p); }
        else if (tag == "f") {
            std::vector<std::tuple<int, int, int>> poly; std::string tok;
            while (ss >> tok) {
                int idx[3] = {0, 0, 0}; size_t a = 0;
                for (int k = 0; k < 3 && a <= tok.size(); k++) {
                    size_t b = tok.find('/', a); std::string part = tok.substr(a, b == std::string::npos ? std::string::npos : b - a);
                    idx[k] = part.empty() ? 0 : std::stoi(part);
                    if (b == std::string::npos) break;
                    a = b + 1;
                }
                auto fix = [](int i, size_t n) { return i > 0 ? i - 1 : (i < 0 ? (int)n + i : -1); };
                poly.emplace_back(fix(idx[0], pos.size()), fix(idx[1], tex.size()), fix(idx[2], nrm.size()));
            }
            if (poly.size() >= 3) faces.push_back(std::move(poly));
        } else if (tag == "o" || tag == "g") flush();
    }
    flush();
    return models;
}

struct SceneBuilder;

// trait Intersectable (tracing.rs:42-47) + the additive `flatten`
struct Intersectable {
    virtual ~Intersectable() = default;
    virtual void flatten(SceneBuilder& sb) const = 0;
    // the entries of a nested Scene (tracing.rs:326, `impl Intersectable for Scene`); nullptr for every other implementor
    virtual const std::vector<std::shared_ptr<const Intersectable>>* scene_objects() const { return nullptr; }
};
using IntersectableRef = std::shared_ptr<const Intersectable>;   // Arc<dyn Intersectable + Send + Sync>

// Owns every array an mi_scene_desc points into (borrowed by the library per call).
struct SceneBuilder {
    std::vector<mi_object> objects; std::vector<mi_sphere> spheres; std::vector<mi_triangle> triangles;
    std::vector<mi_plane> planes; std::vector<mi_volume> volumes; std::vector<mi_mesh> meshes;
    std::vector<mi_material> materials; std::vector<mi_texture> textures;
    std::vector<mi_object> boundary_objects;           // entries of nested Scenes used as ConvexVolume boundaries
    std::map<const Material*, int> mat_ids; std::map<const Texture*, int> tex_ids; std::map<const void*, int> mesh_ids;
    std::vector<std::shared_ptr<const void>> keep;

    int material(const MaterialRef& m) {
        auto it = mat_ids.find(m.get());
        if (it != mat_ids.end()) return it->second;
        int id = (int)materials.size(); materials.push_back(m->flatten()); mat_ids[m.get()] = id; keep.push_back(m);
        return id;
    }
    int texture(const std::shared_ptr<const Texture>& t) {
        auto it = tex_ids.find(t.get());
        if (it != tex_ids.end()) return it->second;
        mi_texture p{}; p.width = t->width; p.height = t->height; p.rgb = t->rgb.data();
        int id = (int)textures.size(); textures.push_back(p); tex_ids[t.get()] = id; keep.push_back(t);
        return id;
    }
    void add(int kind, int index) { mi_object o{}; o.kind = kind; o.index = index; objects.push_back(o); }
    // flatten `obj` into its typed array WITHOUT listing it in Scene.objects (a ConvexVolume's boundary)
    mi_object detached(const Intersectable& obj);
    mi_scene_desc desc() const {
        mi_scene_desc d{};
        d.objects = objects.data(); d.n_objects = (int)objects.size();
        d.spheres = spheres.data(); d.n_spheres = (int)spheres.size();
        d.triangles = triangles.data(); d.n_triangles = (int)triangles.size();
        d.planes = planes.data(); d.n_planes = (int)planes.size();
        d.volumes = volumes.data(); d.n_volumes = (int)volumes.size();
        d.meshes = meshes.data(); d.n_meshes = (int)meshes.size();
        d.materials = materials.data(); d.n_materials = (int)materials.size();
        d.textures = textures.data(); d.n_textures = (int)textures.size();
        d.boundary_objects = boundary_objects.data(); d.n_boundary_objects = (int)boundary_objects.size();
        return d;
    }
};

struct Sphere : Intersectable {                        // geometry.rs:389-393
    Vec3 center; float radius; MaterialRef material;
    Sphere(Vec3 c, float r, MaterialRef m) : center(c), radius(r), material(std::move(m)) {}
    void flatten(SceneBuilder& sb) const override {
        mi_sphere s{}; memcpy(s.center, center.data(), 12); s.radius = radius; s.material = sb.material(material);
        sb.add(MI_OBJ_SPHERE, (int)sb.spheres.size()); sb.spheres.push_back(s);
    }
};
struct Triangle : Intersectable {                      // geometry.rs:424-429
    Vec3 a, b, c; MaterialRef material;
    Triangle(Vec3 a_, Vec3 b_, Vec3 c_, MaterialRef m) : a(a_), b(b_), c(c_), material(std::move(m)) {}
    void flatten(SceneBuilder& sb) const override {
        mi_triangle t{}; memcpy(t.a, a.data(), 12); memcpy(t.b, b.data(), 12); memcpy(t.c, c.data(), 12);
        t.material = sb.material(material);
        sb.add(MI_OBJ_TRIANGLE, (int)sb.triangles.size()); sb.triangles.push_back(t);
    }
};
struct Plane : Intersectable {                         // geometry.rs:468-472
    Vec3 point, normal; MaterialRef material;
    Plane(Vec3 p, Vec3 n, MaterialRef m) : point(p), normal(n), material(std::move(m)) {}
    void flatten(SceneBuilder& sb) const override {
        mi_plane p{}; memcpy(p.point, point.data(), 12); memcpy(p.normal, normal.data(), 12); p.material = sb.material(material);
        sb.add(MI_OBJ_PLANE, (int)sb.planes.size()); sb.planes.push_back(p);
    }
};
struct StaticMesh;
struct ConvexVolume : Intersectable {                  // geometry.rs:495-500
    IntersectableRef boundary;                         // Arc<dyn Intersectable>: a Sphere in every use of the reference (tracing.rs:499-516)
    MaterialRef phase_function; float density;
    ConvexVolume(IntersectableRef b, MaterialRef p, float d) : boundary(std::move(b)), phase_function(std::move(p)), density(d) {}
    void flatten(SceneBuilder& sb) const override;
};
struct StaticMesh : Intersectable {                    // geometry.rs:127-134
    std::shared_ptr<const Mesh> mesh; MaterialRef material;                 // material may be null (None)
    std::array<std::shared_ptr<const Texture>, 5> textures;                 // 0 albedo 1 emission 2 metallic 3 roughness 4 normal
    Matrix4 transform, inv_transform;

    // StaticMesh::load_from_file (geometry.rs:138-172), same argument order; nullptr = None.
    static std::shared_ptr<StaticMesh> load_from_file(const char* file_name, const char* albedo_path, const char* emission_path,
                                                      const char* metallic_path, const char* roughness_path, const char* normal_path,
                                                      MaterialRef material, Matrix4 transform) {
        auto models = load_obj(file_name);
        if (models.empty()) throw std::runtime_error("Failed to load OBJ file");
        auto sm = std::make_shared<StaticMesh>();
        sm->mesh = std::make_shared<Mesh>(std::move(models[0]));                // models.remove(0) :157
        sm->material = std::move(material);
        const char* paths[5] = {albedo_path, emission_path, metallic_path, roughness_path, normal_path};
        for (int i = 0; i < 5; i++) if (paths[i]) if (auto t = Texture::load_from_file(paths[i])) sm->textures[i] = std::make_shared<Texture>(std::move(*t));
        sm->transform = transform;
        auto inv = transform.inverse_transform();                               // :168 `.unwrap()`
        if (!inv) throw std::runtime_error("transform is not invertible");
        sm->inv_transform = *inv;
        return sm;                                                              // BVH (:170) is built by mi_scene_upload
    }
    void flatten(SceneBuilder& sb) const override {
        auto seen = sb.mesh_ids.find(this);           // the same StaticMesh again (Arc sharing, tracing.rs:215): one mi_mesh, another entry
        if (seen != sb.mesh_ids.end()) { sb.add(MI_OBJ_MESH, seen->second); return; }
        sb.mesh_ids[this] = (int)sb.meshes.size();
        mi_mesh m{};
        m.positions = mesh->positions.data(); m.normals = mesh->normals.data(); m.texcoords = mesh->texcoords.data();
        m.indices = mesh->indices.data();
        m.n_vertices = (int)(mesh->positions.size() / 3); m.n_triangles = (int)(mesh->indices.size() / 3);
        if (mesh->normals.size() != mesh->positions.size() || mesh->texcoords.size() * 3 != mesh->positions.size() * 2)
            throw std::runtime_error("mesh needs per-vertex normals and texcoords (geometry.rs:350,355)");
        memcpy(m.transform, transform.m, 64); memcpy(m.inv_transform, inv_transform.m, 64);
        m.material = material ? sb.material(material) : -1;
        for (int i = 0; i < 5; i++) m.textures[i] = textures[i] ? sb.texture(textures[i]) : -1;
        sb.keep.push_back(mesh);
        sb.add(MI_OBJ_MESH, (int)sb.meshes.size()); sb.meshes.push_back(m);
    }
};

inline mi_object SceneBuilder::detached(const Intersectable& obj) {
    obj.flatten(*this);
    mi_object o = objects.back(); objects.pop_back();
    return o;
}
inline void ConvexVolume::flatten(SceneBuilder& sb) const {
    mi_volume v{};
    if (auto sp = dynamic_cast<const Sphere*>(boundary.get())) {       // inline
        v.boundary_kind = MI_OBJ_SPHERE; memcpy(v.boundary_center, sp->center.data(), 12); v.boundary_radius = sp->radius;
    } else if (auto entries = boundary->scene_objects()) {              // a nested Scene
        std::vector<mi_object> es;
        for (auto& e : *entries) {
            if (dynamic_cast<const ConvexVolume*>(e.get()) || e->scene_objects()) throw std::runtime_error("a ConvexVolume or a Scene inside a ConvexVolume boundary is not supported");
            es.push_back(sb.detached(*e));
        }
        v.boundary_kind = MI_OBJ_SCENE; v.boundary_index = (int)sb.boundary_objects.size(); v.boundary_count = (int)es.size();
        sb.boundary_objects.insert(sb.boundary_objects.end(), es.begin(), es.end());
    } else if (dynamic_cast<const ConvexVolume*>(boundary.get())) {
        throw std::runtime_error("a ConvexVolume inside a ConvexVolume boundary is not supported");
    } else {                                                            // Triangle / Plane / StaticMesh
        const mi_object o = sb.detached(*boundary);
        v.boundary_kind = o.kind; v.boundary_index = o.index;
    }
    sb.keep.push_back(boundary);
    v.density = density; v.phase_material = sb.material(phase_function);
    sb.add(MI_OBJ_VOLUME, (int)sb.volumes.size()); sb.volumes.push_back(v);
}

}  // namespace cs397
