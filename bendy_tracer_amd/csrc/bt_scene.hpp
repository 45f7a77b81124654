// bt_scene.hpp -- host-side scene model (mirrors the reference's Scene / Object / Data,
// scene/mod.rs:84-90, object/mod.rs:33-41, data/mod.rs:17-51) and its flattening into the
// device tables of bt_types.h.  No HIP dependency: this part also builds with plain g++.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "bt_types.h"

namespace bt {

struct Error {
    int code;
    std::string message;
};

enum ObjectKind { OBJ_EMPTY = 0, OBJ_CAMERA, OBJ_SPHERE, OBJ_RECT, OBJ_CUBOID };
enum DataKind { DATA_FLAT = 0, DATA_DIFFUSE, DATA_METALLIC, DATA_GLASS, DATA_EMISSIVE, DATA_VOLUME };

struct Affine { BtV3 cx, cy, cz, t; };     // glam::Affine3A, column-major

struct Rect {                               // rect.rs:11-19
    uint64_t material = 0;
    float half_width = 0, half_height = 0;
    BtV3 x{}, y{}, z{};
};

struct Object {                             // object/mod.rs:33-41
    uint64_t object_ref = 0;
    bool has_tag = false;
    std::string tag;
    uint32_t flags = 0;                     // ObjectFlags::LIGHT = 0x1 (object/mod.rs:17-21)
    Affine world{};                         // transform_world (transform.rs:10-15)
    int kind = OBJ_EMPTY;
    // Camera (camera.rs:3-10)
    float sensor_size = 0, focal_length = 0, aspect_ratio = 0, fstop = 0, focus = 0;
    bool has_focus = false;
    // Sphere (sphere.rs:11-16)
    uint64_t material = 0, volume = 0;
    bool has_volume = false;
    float radius = 0;
    Rect rect;                              // Rect
    BtV3 face_offset[6]{};                  // Cuboid (cuboid.rs:12-15)
    Rect faces[6];
};

struct Data {                               // data/mod.rs:17-51
    uint64_t data_ref = 0;
    int kind = DATA_FLAT;
    BtV3 albedo{};
    float roughness = 0, ior = 1, intensity = 0;
    int32_t width = 0, height = 0, depth = 0;   // DensityMap (volume.rs:75-82)
    BtV3 size{};
    std::vector<float> buffer;
};

struct Scene {                              // scene/mod.rs:84-90
    std::vector<Object> objects;            // ascending object_ref
    std::vector<Data> data;                 // ascending data_ref
    uint64_t root_material = 0;

    int object_index(uint64_t ref) const;   // -1 if absent ("invalid object ref", scene/mod.rs:132)
    int data_index(uint64_t ref) const;     // -1 if absent ("invalid data ref", scene/mod.rs:136)
};

// Parses a decompressed scene.json document.  Throws bt::Error.
Scene parse_scene(const char *json, size_t len);
// Reads a file; gunzips it when the path ends in ".gz" (main.rs:93-102).  Throws bt::Error.
std::string read_scene_file(const std::string &path);

// Device tables (bt_types.h) built from a Scene.  Throws bt::Error for invalid refs, non-material
// data behind a material ref, or a Diffuse material with no LIGHT object.
struct FlatScene {
    std::vector<BtPrim> prims;
    std::vector<BtMaterial> materials;
    std::vector<BtVolume> volumes;
    std::vector<BtLight> lights;
    std::vector<BtLightFace> light_faces;
    std::vector<float> density;
    // rect scenes: the BT_PRIM_RECT_AAN rows grouped by normal axis + the rows of every other kind (bt_types.h BtRectAAN)
    std::vector<BtRectAAN> aan_rows;
    std::vector<BtRectLA> la_rows;
    std::vector<int32_t> other_rows;
    int32_t n_aan[3] = {0, 0, 0};
    BtV3 root_color{}, root_albedo{};
    int root_has_albedo = 0;
    size_t lds_bytes() const;
};
FlatScene flatten_scene(const Scene &scene);

// ---- bt_io.cpp: the callers' side ----
std::string format_f32(float v);
// pretty JSON of `source` with the scene's camera aspect ratios patched in (main.rs:299-313)
std::string scene_to_pretty_json(const Scene &scene, const std::string &source);
void write_text_file(const std::string &path, const std::string &text);   // gzip when the path ends in .gz
std::string default_scene_json();                                           // main.rs:107-214
void write_png(const std::string &path, const uint8_t *rgba, uint32_t w, uint32_t h);

// rand 0.8.5 UniformFloat::new / new_inclusive scale
float uniform_scale(float lo, float hi, bool inclusive);
// glam Vec3::any_orthonormal_pair
void orthonormal_pair(BtV3 n, BtV3 &t1, BtV3 &t2);

} // namespace bt
