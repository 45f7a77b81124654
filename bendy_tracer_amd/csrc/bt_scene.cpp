// bt_scene.cpp -- scene.json[.gz] loader and flattening into device tables.
// Reference: main.rs:93-102 (gzip + serde_json), scene/mod.rs:16-20,84-90, object/mod.rs:23-41,
// 247-256, data/mod.rs:12-51, material.rs:22-44, volume.rs:75-82 (on-disk schema).
#include "bt_scene.hpp"

#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>

#include "../../include/bendy_hip.h"
#include "bt_json.hpp"

#pragma STDC FP_CONTRACT OFF

namespace bt {

namespace {

[[noreturn]] void fail(int code, const std::string &msg) { throw Error{code, msg}; }

// ---- arithmetic shared with the kernels' conventions (numerics contract N2-N4) ----
inline BtV3 v3(float x, float y, float z) { BtV3 r; r.x = x; r.y = y; r.z = z; return r; }
inline BtV3 add(BtV3 a, BtV3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline BtV3 scale(BtV3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
inline BtV3 neg(BtV3 a) { return v3(-a.x, -a.y, -a.z); }
inline float dot(BtV3 a, BtV3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline BtV3 cross(BtV3 a, BtV3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
inline BtV3 xf_vector(const Affine &m, BtV3 v) { return add(add(scale(m.cx, v.x), scale(m.cy, v.y)), scale(m.cz, v.z)); }
inline BtV3 xf_point(const Affine &m, BtV3 p) { return add(xf_vector(m, p), m.t); }

// glam Affine3A::inverse (Mat3A::inverse via cross products, t' = -(Minv * t)); rect.rs:134
Affine inverse(const Affine &a) {
    BtV3 t0 = cross(a.cy, a.cz), t1 = cross(a.cz, a.cx), t2 = cross(a.cx, a.cy);
    float det = dot(a.cz, t2);
    float inv_det = 1.0f / det;
    BtV3 r0 = scale(t0, inv_det), r1 = scale(t1, inv_det), r2 = scale(t2, inv_det);
    Affine o;
    o.cx = v3(r0.x, r1.x, r2.x);
    o.cy = v3(r0.y, r1.y, r2.y);
    o.cz = v3(r0.z, r1.z, r2.z);
    o.t = v3(0, 0, 0);
    o.t = neg(xf_vector(o, a.t));
    return o;
}
// *transform * Affine3A::from_translation(offset) (cuboid.rs:39,52,68,78,95)
Affine translate(const Affine &m, BtV3 offset) {
    Affine r = m;
    r.t = xf_point(m, offset);
    return r;
}

// ---- JSON -> model ----
BtV3 json_v3(const btjson::Value &v) {
    if (v.kind != btjson::Value::Array || v.items.size() != 3) throw std::runtime_error("expected [f32; 3]");
    return v3(v.at(0).as_f32(), v.at(1).as_f32(), v.at(2).as_f32());
}
Affine json_affine(const btjson::Value &v) {
    if (v.kind != btjson::Value::Array || v.items.size() != 12) throw std::runtime_error("expected Affine3A as [f32; 12]");
    float f[12];
    for (int i = 0; i < 12; ++i) f[i] = v.at(i).as_f32();
    Affine a;
    a.cx = v3(f[0], f[1], f[2]);
    a.cy = v3(f[3], f[4], f[5]);
    a.cz = v3(f[6], f[7], f[8]);
    a.t = v3(f[9], f[10], f[11]);
    return a;
}
BtV3 json_rgb(const btjson::Value &v) { return v3(v.at("r").as_f32(), v.at("g").as_f32(), v.at("b").as_f32()); }
Rect json_rect(const btjson::Value &v) {
    Rect r;
    r.material = v.at("material").as_u64();
    r.half_width = v.at("half_width").as_f32();
    r.half_height = v.at("half_height").as_f32();
    r.x = json_v3(v.at("x"));
    r.y = json_v3(v.at("y"));
    r.z = json_v3(v.at("z"));
    return r;
}
// externally tagged enum: {"Variant": body}
const btjson::Value &variant(const btjson::Value &v, std::string &name) {
    if (v.kind != btjson::Value::Object || v.members.size() != 1) throw std::runtime_error("expected an enum variant");
    name = v.members[0].first;
    return *v.members[0].second;
}

} // namespace

int Scene::object_index(uint64_t ref) const {
    for (size_t i = 0; i < objects.size(); ++i)
        if (objects[i].object_ref == ref) return (int)i;
    return -1;
}
int Scene::data_index(uint64_t ref) const {
    for (size_t i = 0; i < data.size(); ++i)
        if (data[i].data_ref == ref) return (int)i;
    return -1;
}

Scene parse_scene(const char *json, size_t len) {
    Scene sc;
    try {
        btjson::ValuePtr doc = btjson::parse(json, len);
        sc.root_material = doc->at("root_material").as_u64();
        doc->at("roots"); // present in every file, never read at render time (scene/mod.rs:86)

        const btjson::Value &objs = doc->at("objects").at("collection");
        if (objs.kind != btjson::Value::Object) throw std::runtime_error("objects.collection must be a map");
        for (auto &m : objs.members) {
            const btjson::Value &src = *m.second;
            Object o;
            o.object_ref = std::strtoull(m.first.c_str(), nullptr, 10);
            const btjson::Value &tag = src.at("tag");
            if (!tag.is_null()) {
                o.has_tag = true;
                o.tag = tag.as_string();
            }
            o.flags = (uint32_t)src.at("flags").at("bits").as_u64();
            o.world = json_affine(src.at("transform").at("transform_world"));
            const btjson::Value &inner = src.at("inner");
            if (inner.kind == btjson::Value::String) {
                if (inner.text != "Empty") throw std::runtime_error("unknown unit ObjectKind `" + inner.text + "`");
                o.kind = OBJ_EMPTY;
            } else {
                std::string name;
                const btjson::Value &body = variant(inner, name);
                if (name == "Camera") {
                    o.kind = OBJ_CAMERA;
                    o.sensor_size = body.at("sensor_size").as_f32();
                    o.focal_length = body.at("focal_length").as_f32();
                    o.aspect_ratio = body.at("aspect_ratio").as_f32();
                    o.fstop = body.at("fstop").as_f32();
                    const btjson::Value &focus = body.at("focus");
                    o.has_focus = !focus.is_null();
                    o.focus = o.has_focus ? focus.as_f32() : 0.0f;
                } else if (name == "Sphere") {
                    o.kind = OBJ_SPHERE;
                    o.material = body.at("material").as_u64();
                    const btjson::Value &vol = body.at("volume");
                    o.has_volume = !vol.is_null();
                    o.volume = o.has_volume ? vol.as_u64() : 0;
                    o.radius = body.at("radius").as_f32();
                } else if (name == "Rect") {
                    o.kind = OBJ_RECT;
                    o.rect = json_rect(body);
                } else if (name == "Cuboid") {
                    o.kind = OBJ_CUBOID;
                    const btjson::Value &faces = body.at("faces");
                    if (faces.kind != btjson::Value::Array || faces.items.size() != 6)
                        throw std::runtime_error("Cuboid.faces must have 6 entries");
                    for (int f = 0; f < 6; ++f) {
                        o.face_offset[f] = json_v3(faces.at(f).at(0));
                        o.faces[f] = json_rect(faces.at(f).at(1));
                    }
                } else {
                    throw std::runtime_error("unknown ObjectKind `" + name + "`");
                }
            }
            sc.objects.push_back(std::move(o));
        }

        const btjson::Value &datas = doc->at("data").at("collection");
        if (datas.kind != btjson::Value::Object) throw std::runtime_error("data.collection must be a map");
        for (auto &m : datas.members) {
            Data d;
            d.data_ref = std::strtoull(m.first.c_str(), nullptr, 10);
            std::string name;
            const btjson::Value &body = variant(m.second->at("inner"), name);
            if (name == "Material") {
                std::string mname;
                const btjson::Value &mb = variant(body, mname);
                d.albedo = json_rgb(mb.at("albedo"));
                if (mname == "Flat") d.kind = DATA_FLAT;
                else if (mname == "Diffuse") { d.kind = DATA_DIFFUSE; d.roughness = mb.at("roughness").as_f32(); }
                else if (mname == "Metallic") { d.kind = DATA_METALLIC; d.roughness = mb.at("roughness").as_f32(); }
                else if (mname == "Glass") {
                    d.kind = DATA_GLASS;
                    d.roughness = mb.at("roughness").as_f32();
                    d.ior = mb.at("ior").as_f32();
                } else if (mname == "Emissive") { d.kind = DATA_EMISSIVE; d.intensity = mb.at("intensity").as_f32(); }
                else throw std::runtime_error("unknown Material `" + mname + "`");
            } else if (name == "Volume") {
                std::string vname;
                const btjson::Value &vb = variant(body, vname);
                if (vname != "DensityMap") throw std::runtime_error("unknown Volume `" + vname + "`");
                d.kind = DATA_VOLUME;
                d.width = (int32_t)vb.at("width").as_u64();
                d.height = (int32_t)vb.at("height").as_u64();
                d.depth = (int32_t)vb.at("depth").as_u64();
                d.size = json_v3(vb.at("size"));
                const btjson::Value &buf = vb.at("buffer");
                if (buf.kind != btjson::Value::Array) throw std::runtime_error("DensityMap.buffer must be an array");
                d.buffer.reserve(buf.items.size());
                for (auto &x : buf.items) d.buffer.push_back(x->as_f32());
                if (d.buffer.size() != (size_t)d.width * d.height * d.depth)
                    throw std::runtime_error("DensityMap.buffer length != width*height*depth");
            } else {
                throw std::runtime_error("unknown DataKind `" + name + "`");
            }
            sc.data.push_back(std::move(d));
        }
    } catch (const std::exception &e) {
        fail(BT_ERR_PARSE, e.what());
    }
    std::sort(sc.objects.begin(), sc.objects.end(), [](const Object &a, const Object &b) { return a.object_ref < b.object_ref; });
    std::sort(sc.data.begin(), sc.data.end(), [](const Data &a, const Data &b) { return a.data_ref < b.data_ref; });
    return sc;
}

std::string read_scene_file(const std::string &path) {
    const bool gz = path.size() >= 3 && path.compare(path.size() - 3, 3, ".gz") == 0; // main.rs:97
    std::string out;
    if (gz) {
        gzFile f = gzopen(path.c_str(), "rb");
        if (!f) fail(BT_ERR_IO, "cannot open " + path);
        char buf[1 << 16];
        for (;;) {
            int n = gzread(f, buf, sizeof buf);
            if (n < 0) {
                gzclose(f);
                fail(BT_ERR_IO, "gzip error in " + path);
            }
            if (n == 0) break;
            out.append(buf, (size_t)n);
        }
        gzclose(f);
    } else {
        FILE *f = std::fopen(path.c_str(), "rb");
        if (!f) fail(BT_ERR_IO, "cannot open " + path);
        char buf[1 << 16];
        size_t n;
        while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) out.append(buf, n);
        std::fclose(f);
    }
    return out;
}

float uniform_scale(float lo, float hi, bool inclusive) {
    const float max_rand = 1.0f - 1.1920928955078125e-7f;
    float scale = inclusive ? (hi - lo) / max_rand : (hi - lo);
    for (int guard = 0; guard < 64; ++guard) {
        float top = scale * max_rand + lo;
        bool bad = inclusive ? (top > hi) : (top >= hi);
        if (!bad) break;
        uint32_t b;
        std::memcpy(&b, &scale, 4);
        b -= 1;
        std::memcpy(&scale, &b, 4);
    }
    return scale;
}

void orthonormal_pair(BtV3 n, BtV3 &t1, BtV3 &t2) {
    float sign = std::copysign(1.0f, n.z);
    float a = -1.0f / (sign + n.z);
    float b = n.x * n.y * a;
    t1 = v3(1.0f + sign * n.x * n.x * a, sign * b, -sign * n.x);
    t2 = v3(b, sign + n.y * n.y * a, -n.y);
}

size_t FlatScene::lds_bytes() const {
    size_t n = sizeof(BtPrimLite) * prims.size() + sizeof(BtMaterial) * materials.size() +
               sizeof(BtVolume) * volumes.size() + sizeof(BtLight) * lights.size() +
               sizeof(BtLightFace) * light_faces.size();
    if (!density.empty() && density.size() <= BT_DENSITY_LDS_MAX) n += sizeof(float) * density.size();
    return (n + 15) & ~(size_t)15;
}

FlatScene flatten_scene(const Scene &sc) {
    FlatScene fs;
    // data -> materials / volumes
    std::vector<int> mat_of(sc.data.size(), -1), vol_of(sc.data.size(), -1);
    bool any_diffuse_used = false;
    for (size_t i = 0; i < sc.data.size(); ++i) {
        const Data &d = sc.data[i];
        if (d.kind == DATA_VOLUME) {
            BtVolume v{};
            v.width = d.width; v.height = d.height; v.depth = d.depth;
            v.offset = (int32_t)fs.density.size();
            v.size = d.size;
            fs.density.insert(fs.density.end(), d.buffer.begin(), d.buffer.end());
            vol_of[i] = (int)fs.volumes.size();
            fs.volumes.push_back(v);
        } else {
            BtMaterial m{};
            m.kind = d.kind;       // DATA_* and BT_MAT_* share values 0..4
            m.albedo = d.albedo;
            m.roughness = d.roughness;
            m.ior = d.ior;
            m.inv_ior = 1.0f / d.ior;                                   // ior.recip(), material.rs:244
            if (d.kind == DATA_FLAT) m.emitted = d.albedo;              // material.rs:76
            else if (d.kind == DATA_EMISSIVE) m.emitted = scale(d.albedo, d.intensity); // :77
            else m.emitted = v3(0, 0, 0);
            mat_of[i] = (int)fs.materials.size();
            fs.materials.push_back(m);
        }
    }
    auto material_index = [&](uint64_t ref) -> int {
        int di = sc.data_index(ref);
        if (di < 0) fail(BT_ERR_INVALID_REF, "invalid data ref " + std::to_string(ref));
        if (mat_of[di] < 0) fail(BT_ERR_NOT_MATERIAL, "expected material data at ref " + std::to_string(ref));
        if (sc.data[di].kind == DATA_DIFFUSE) any_diffuse_used = true;
        return mat_of[di];
    };
    auto volume_index = [&](uint64_t ref) -> int {
        int di = sc.data_index(ref);
        if (di < 0) fail(BT_ERR_INVALID_REF, "invalid data ref " + std::to_string(ref));
        if (vol_of[di] < 0) fail(BT_ERR_NOT_MATERIAL, "expected volume data at ref " + std::to_string(ref));
        return vol_of[di];
    };
    // exactly one component is +-1, the others +-0 -> its index, else -1
    auto unit_axis = [](BtV3 a) -> int {
        const float c[3] = {a.x, a.y, a.z};
        int idx = -1;
        for (int i = 0; i < 3; ++i) {
            if (c[i] == 1.0f || c[i] == -1.0f) {
                if (idx >= 0) return -1;
                idx = i;
            } else if (c[i] != 0.0f) {
                return -1;
            }
        }
        return idx;
    };
    auto rect_prim = [&](const Rect &r, const Affine &tf, int object, bool strict) {
        BtPrim p{};
        p.object = object;
        p.material = material_index(r.material);
        p.volume = -1;
        p.c = xf_vector(tf, r.z);                     // rect.rs:119
        p.t = tf.t;                                   // rect.rs:118
        Affine inv = inverse(tf);                     // rect.rs:134
        p.icx = inv.cx; p.icy = inv.cy; p.icz = inv.cz; p.it = inv.t;
        p.ax = r.x; p.ay = r.y;
        p.w_sqr = r.half_width * r.half_width;        // rect.rs:77-78
        p.h_sqr = r.half_height * r.half_height;
        const bool identity = tf.cx.x == 1.0f && tf.cx.y == 0.0f && tf.cx.z == 0.0f && tf.cy.x == 0.0f &&
                              tf.cy.y == 1.0f && tf.cy.z == 0.0f && tf.cz.x == 0.0f && tf.cz.y == 0.0f && tf.cz.z == 1.0f;
        const int au = unit_axis(r.x), av = unit_axis(r.y);
        int shape = BT_PRIM_RECT;
        if (au >= 0 && av >= 0 && au != av && !identity) {
            shape = BT_PRIM_RECT_LA;
            p.aa_u = au;
            p.aa_v = av;
            auto comp = [](BtV3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); };
            p.ax = v3(comp(inv.cx, au), comp(inv.cy, au), comp(inv.cz, au)); p.ax_w = comp(inv.t, au);
            p.ay = v3(comp(inv.cx, av), comp(inv.cy, av), comp(inv.cz, av)); p.ay_w = comp(inv.t, av);
        }
        if (identity && au >= 0 && av >= 0) {
            shape = BT_PRIM_RECT_AA;
            p.aa_u = au;
            p.aa_v = av;
            const int aw = unit_axis(p.c);
            if (au != av && aw >= 0 && aw != au && aw != av) {
                shape = BT_PRIM_RECT_AAN;
                p.aa_w = aw;
                auto comp = [](BtV3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); };
                const int a = aw == 0 ? 1 : 0, b = aw == 2 ? 1 : 2;          // in-plane axes, ascending
                p.ax = v3(comp(p.t, aw), comp(p.it, a), comp(p.it, b));
                p.ax_w = au == a ? p.w_sqr : p.h_sqr;
                p.ay = v3(au == a ? p.h_sqr : p.w_sqr, comp(p.c, aw), 0.0f);
                p.ay_w = 0.0f;
            }
        }
        p.kind = shape | (strict ? BT_PRIM_STRICT : 0);
        fs.prims.push_back(p);
    };
    auto light_face = [&](const Rect &r, const Affine &tf) {
        BtLightFace f{};
        f.mcx = tf.cx; f.mcy = tf.cy; f.mcz = tf.cz; f.mt = tf.t;
        f.ax = r.x; f.ay = r.y;
        f.half_width = r.half_width;
        f.half_height = r.half_height;
        f.scale_x = uniform_scale(-r.half_width, r.half_width, true);     // rect.rs:83
        f.scale_y = uniform_scale(-r.half_height, r.half_height, true);   // rect.rs:84
        f.area = 4.0f * r.half_width * r.half_height;                     // rect.rs:88-90
        fs.light_faces.push_back(f);
    };

    for (size_t oi = 0; oi < sc.objects.size(); ++oi) {
        const Object &o = sc.objects[oi];
        const int prim_first = (int)fs.prims.size();
        if (o.kind == OBJ_SPHERE) {
            BtPrim p{};
            p.kind = BT_PRIM_SPHERE;
            p.object = (int)oi;
            p.material = material_index(o.material);
            p.volume = o.has_volume ? volume_index(o.volume) : -1;
            p.c = o.world.t;                          // object/mod.rs:170-172
            p.radius = o.radius;
            fs.prims.push_back(p);
        } else if (o.kind == OBJ_RECT) {
            rect_prim(o.rect, o.world, (int)oi, false);
        } else if (o.kind == OBJ_CUBOID) {
            for (int f = 0; f < 6; ++f) rect_prim(o.faces[f], translate(o.world, o.face_offset[f]), (int)oi, true);
        }
        if (o.flags & 1u) {                           // ObjectFlags::LIGHT, material.rs:106-119
            BtLight l{};
            l.prim_first = prim_first;
            l.prim_count = (int)fs.prims.size() - prim_first;
            l.face_first = (int)fs.light_faces.size();
            l.centre = o.world.t;                     // object/mod.rs:150
            if (o.kind == OBJ_SPHERE) {
                l.kind = BT_LIGHT_SPHERE;
                l.radius = o.radius;
                l.shadow = 3.14159265358979323846f * o.radius * o.radius;   // sphere.rs:53-54
            } else if (o.kind == OBJ_RECT) {
                l.kind = BT_LIGHT_RECT;
                light_face(o.rect, o.world);
            } else if (o.kind == OBJ_CUBOID) {
                l.kind = BT_LIGHT_CUBOID;
                float total = 0.0f;
                for (int f = 0; f < 6; ++f) {
                    light_face(o.faces[f], translate(o.world, o.face_offset[f]));
                    total += fs.light_faces.back().area;
                    if (f < 5) l.cum[f] = total;      // WeightedIndex cumulative weights, cuboid.rs:49
                }
                l.total_scale = uniform_scale(0.0f, total, false);
            } else {
                l.kind = BT_LIGHT_POINT;              // object/mod.rs:150: `_ => translation`, pdf None
            }
            fs.lights.push_back(l);
        }
    }

    // root material: the ColorData sample_root returns (mod.rs:429-452)
    {
        int di = sc.data_index(sc.root_material);
        if (di < 0) fail(BT_ERR_INVALID_REF, "invalid data ref (root_material) " + std::to_string(sc.root_material));
        if (mat_of[di] < 0) fail(BT_ERR_NOT_MATERIAL, "expected root material to be a material"); // scene/mod.rs:117
        const Data &d = sc.data[di];
        const BtMaterial &m = fs.materials[mat_of[di]];
        fs.root_has_albedo = d.kind != DATA_EMISSIVE;
        BtV3 shade_color = v3(0, 0, 0);
        if (d.kind == DATA_DIFFUSE || d.kind == DATA_METALLIC || d.kind == DATA_GLASS) {
            shade_color = d.albedo;
            if (d.kind == DATA_DIFFUSE) any_diffuse_used = true;
        }
        fs.root_albedo = shade_color;
        fs.root_color = add(shade_color, m.emitted);  // color_data.color += emitted (mod.rs:450)
    }
    // sorted view of the table for bt_device.hpp intersect_sorted()
    // Rect::contains_point tests `x * x <= L` (rect.rs:74-80).  x -> fl(x * x) is monotone in |x| (rounding and underflow
    // included), so the set of x that pass is |x| <= s for the largest float s with fl(s * s) <= L: found here by bisection
    // over the bit patterns, with the same IEEE multiplication the kernel would perform.  The sorted rows carry s; the
    // kernel compares |x| with it and saves the multiplication.  (L negative or NaN: nothing passes, s = -1.)
    auto abs_limit = [](float L) -> float {
        if (!(L >= 0.0f)) return -1.0f;
        uint32_t lo = 0u, hi = 0x7f800000u;                 // fl(0 * 0) = 0 <= L holds
        auto passes = [L](uint32_t bits) {
            float x;
            std::memcpy(&x, &bits, 4);
            volatile float sq = x * x;
            return sq <= L;
        };
        if (passes(hi)) return std::numeric_limits<float>::infinity();
        while (hi - lo > 1u) {                              // passes(lo) && !passes(hi)
            const uint32_t mid = lo + (hi - lo) / 2u;
            if (passes(mid)) lo = mid; else hi = mid;
        }
        float s;
        std::memcpy(&s, &lo, 4);
        return s;
    };
    for (int axis = 0; axis < 3; ++axis)
        for (size_t i = 0; i < fs.prims.size(); ++i) {
            const BtPrim &p = fs.prims[i];
            if ((p.kind & BT_PRIM_SHAPE_MASK) != BT_PRIM_RECT_AAN || p.aa_w != axis) continue;
            BtRectAAN r{};
            r.it_a = p.ax.y; r.it_b = p.ax.z; r.lim_a = abs_limit(p.ax_w); r.lim_b = abs_limit(p.ay.x); r.t_w = p.ax.x;
            r.sgn_mask = std::signbit(p.ay.y) ? 0x80000000u : 0u;
            r.prio = (p.kind & BT_PRIM_STRICT) ? 0xfffeu - (uint32_t)i : 0x10000u + (uint32_t)i;
            fs.aan_rows.push_back(r);
            fs.n_aan[axis] += 1;
        }
    {
        // LA rows; a row whose normal equals (bit for bit) that of a row already emitted goes right behind it
        std::vector<size_t> la;
        for (size_t i = 0; i < fs.prims.size(); ++i)
            if ((fs.prims[i].kind & BT_PRIM_SHAPE_MASK) == BT_PRIM_RECT_LA) la.push_back(i);
        std::vector<bool> done(la.size(), false);
        auto same = [](const BtV3 &a, const BtV3 &b) { return std::memcmp(&a, &b, sizeof a) == 0; };
        for (size_t a = 0; a < la.size(); ++a) {
            if (done[a]) continue;
            for (size_t b = a; b < la.size(); ++b) {
                if (done[b] || !same(fs.prims[la[a]].c, fs.prims[la[b]].c)) continue;
                const BtPrim &p = fs.prims[la[b]];
                BtRectLA r{};
                r.n = p.c;
                r.first_of_normal = b == a ? 1u : 0u;
                r.t = p.t;
                r.prio = (p.kind & BT_PRIM_STRICT) ? 0xfffeu - (uint32_t)la[b] : 0x10000u + (uint32_t)la[b];
                r.a_x[0] = p.ax.x; r.a_x[1] = p.ay.x; r.a_y[0] = p.ax.y; r.a_y[1] = p.ay.y; r.a_z[0] = p.ax.z; r.a_z[1] = p.ay.z;
                r.a_w[0] = p.ax_w; r.a_w[1] = p.ay_w;
                r.lim[0] = abs_limit(p.w_sqr); r.lim[1] = abs_limit(p.h_sqr);
                fs.la_rows.push_back(r);
                done[b] = true;
            }
        }
    }
    for (size_t i = 0; i < fs.prims.size(); ++i) {
        const int shape = fs.prims[i].kind & BT_PRIM_SHAPE_MASK;
        if (shape != BT_PRIM_RECT_AAN && shape != BT_PRIM_RECT_LA) fs.other_rows.push_back((int32_t)i);
    }
    if (any_diffuse_used && fs.lights.empty())
        fail(BT_ERR_NO_LIGHT, "scene has a Diffuse material but no LIGHT object (Uniform::new(0, 0) would panic, material.rs:112)");
    return fs;
}

} // namespace bt
