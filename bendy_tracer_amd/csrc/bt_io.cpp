// bt_io.cpp -- the callers' side of the path: scene save (serde_json::to_writer_pretty
// + optional gzip, main.rs:299-313), the built-in default scene (main.rs:107-214) and the PNG
// screenshot of the 8-bit preview (main.rs:275-298).  No GPU work here.
#include <zlib.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/bendy_hip.h"
#include "bt_json.hpp"
#include "bt_scene.hpp"

#pragma STDC FP_CONTRACT OFF

namespace bt {

// shortest decimal that round-trips through strtof, always with a '.' or exponent (serde_json / ryu style)
std::string format_f32(float v) {
    if (!std::isfinite(v)) return "null";       // serde_json writes non-finite floats as null
    char buf[64];
    for (int p = 1; p <= 9; ++p) {
        std::snprintf(buf, sizeof buf, "%.*g", p, (double)v);
        if (std::strtof(buf, nullptr) == v) break;
    }
    std::string s(buf);
    if (s.find_first_of(".eEn") == std::string::npos) s += ".0";
    return s;
}

namespace {

void escape_into(std::string &out, const std::string &s) {
    out += '"';
    for (unsigned char c : s) {
        switch (c) {
        case '"': out += "\\\""; break;
        case '\\': out += "\\\\"; break;
        case '\n': out += "\\n"; break;
        case '\r': out += "\\r"; break;
        case '\t': out += "\\t"; break;
        case '\b': out += "\\b"; break;
        case '\f': out += "\\f"; break;
        default:
            if (c < 0x20) {
                char b[8];
                std::snprintf(b, sizeof b, "\\u%04x", c);
                out += b;
            } else {
                out += (char)c;
            }
        }
    }
    out += '"';
}

// serde_json::ser::PrettyFormatter: two-space indent, "key": value, [] and {} for empty containers
void pretty(std::string &out, const btjson::Value &v, int depth) {
    auto indent = [&](int d) { out.append((size_t)d * 2, ' '); };
    switch (v.kind) {
    case btjson::Value::Null: out += "null"; break;
    case btjson::Value::Bool: out += v.b ? "true" : "false"; break;
    case btjson::Value::Number: out += v.text; break;
    case btjson::Value::String: escape_into(out, v.text); break;
    case btjson::Value::Array:
        if (v.items.empty()) { out += "[]"; break; }
        out += "[\n";
        for (size_t i = 0; i < v.items.size(); ++i) {
            indent(depth + 1);
            pretty(out, *v.items[i], depth + 1);
            out += i + 1 < v.items.size() ? ",\n" : "\n";
        }
        indent(depth);
        out += ']';
        break;
    case btjson::Value::Object:
        if (v.members.empty()) { out += "{}"; break; }
        out += "{\n";
        for (size_t i = 0; i < v.members.size(); ++i) {
            indent(depth + 1);
            escape_into(out, v.members[i].first);
            out += ": ";
            pretty(out, *v.members[i].second, depth + 1);
            out += i + 1 < v.members.size() ? ",\n" : "\n";
        }
        indent(depth);
        out += '}';
        break;
    }
}

btjson::Value *member(btjson::Value &v, const std::string &key) {
    if (v.kind != btjson::Value::Object) return nullptr;
    for (auto &m : v.members)
        if (m.first == key) return m.second.get();
    return nullptr;
}

struct V { float x, y, z; };
inline V vneg(V a) { return {0.0f - a.x, 0.0f - a.y, 0.0f - a.z}; }   // +0.0 for zero components, as the saved scenes show
inline float vlen(V a) { return std::sqrt((a.x * a.x + a.y * a.y) + a.z * a.z); }
inline V vdiv(V a, float s) { return {a.x / s, a.y / s, a.z / s}; }
inline V vcross(V a, V b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

std::string j3(V a) { return "[" + format_f32(a.x) + "," + format_f32(a.y) + "," + format_f32(a.z) + "]"; }

// Rect::new (rect.rs:22-36)
std::string rect_json(int material, V x, V y) {
    float hw = vlen(x), hh = vlen(y);
    V xn = vdiv(x, hw), yn = vdiv(y, hh), z = vcross(xn, yn);
    return "{\"material\":" + std::to_string(material) + ",\"half_width\":" + format_f32(hw) + ",\"half_height\":" +
           format_f32(hh) + ",\"x\":" + j3(xn) + ",\"y\":" + j3(yn) + ",\"z\":" + j3(z) + "}";
}
// Cuboid::new (cuboid.rs:19-30)
std::string cuboid_json(int material, V x, V y, V z) {
    struct F { V off; V a, b; } f[6] = {{vneg(z), x, y}, {z, vneg(x), y}, {vneg(x), z, y},
                                        {x, vneg(z), y}, {vneg(y), x, z}, {y, x, vneg(z)}};
    std::string s = "{\"Cuboid\":{\"faces\":[";
    for (int i = 0; i < 6; ++i) s += std::string(i ? "," : "") + "[" + j3(f[i].off) + "," + rect_json(material, f[i].a, f[i].b) + "]";
    return s + "]}}";
}
std::string affine_json(const float m[9], V t) {
    std::string s = "[";
    for (int i = 0; i < 9; ++i) s += format_f32(m[i]) + ",";
    return s + format_f32(t.x) + "," + format_f32(t.y) + "," + format_f32(t.z) + "]";
}
std::string object_json(int ref, const char *tag, unsigned flags, const float m[9], V t, const std::string &inner) {
    std::string a = affine_json(m, t);
    return "\"" + std::to_string(ref) + "\":{\"object_ref\":" + std::to_string(ref) + ",\"tag\":" +
           (tag ? std::string("\"") + tag + "\"" : std::string("null")) + ",\"flags\":{\"bits\":" + std::to_string(flags) +
           "},\"transform\":{\"transform_world\":" + a + ",\"transform_local\":" + a + ",\"transform_parent\":null},\"inner\":" +
           inner + ",\"children\":null}";
}
std::string material_json(int ref, const std::string &body) {
    return "\"" + std::to_string(ref) + "\":{\"inner\":{\"Material\":" + body + "}}";
}
std::string rgb(float r, float g, float b) {
    return "{\"r\":" + format_f32(r) + ",\"g\":" + format_f32(g) + ",\"b\":" + format_f32(b) + "}";
}

} // namespace

// Re-serialises `source` (the document the scene was parsed from) with the camera aspect ratios of
// `scene` patched in -- the only mutation this API offers (main.rs:218-223) -- as pretty JSON.
std::string scene_to_pretty_json(const Scene &scene, const std::string &source) {
    btjson::ValuePtr doc = btjson::parse(source.data(), source.size());
    btjson::Value *coll = member(*doc, "objects");
    coll = coll ? member(*coll, "collection") : nullptr;
    if (coll && coll->kind == btjson::Value::Object) {
        for (auto &m : coll->members) {
            uint64_t ref = std::strtoull(m.first.c_str(), nullptr, 10);
            int oi = scene.object_index(ref);
            if (oi < 0 || scene.objects[oi].kind != OBJ_CAMERA) continue;
            btjson::Value *inner = member(*m.second, "inner");
            btjson::Value *cam = inner ? member(*inner, "Camera") : nullptr;
            btjson::Value *ar = cam ? member(*cam, "aspect_ratio") : nullptr;
            if (ar && ar->kind == btjson::Value::Number && ar->as_f32() != scene.objects[oi].aspect_ratio)
                ar->text = format_f32(scene.objects[oi].aspect_ratio);
        }
    }
    std::string out;
    pretty(out, *doc, 0);
    return out;
}

void write_text_file(const std::string &path, const std::string &text) {
    const bool gz = path.size() >= 3 && path.compare(path.size() - 3, 3, ".gz") == 0;   // main.rs:305
    if (gz) {
        gzFile f = gzopen(path.c_str(), "wb6");                   // Compression::default() = level 6
        if (!f) throw Error{BT_ERR_IO, "cannot create " + path};
        const bool ok = gzwrite(f, text.data(), (unsigned)text.size()) == (int)text.size();
        if (gzclose(f) != Z_OK || !ok) throw Error{BT_ERR_IO, "gzip write error on " + path};
    } else {
        FILE *f = std::fopen(path.c_str(), "wb");
        if (!f) throw Error{BT_ERR_IO, "cannot create " + path};
        const bool ok = std::fwrite(text.data(), 1, text.size(), f) == text.size();
        if (std::fclose(f) != 0 || !ok) throw Error{BT_ERR_IO, "write error on " + path};
    }
}

// The scene main.rs builds when the --scene file does not exist (main.rs:107-214): Cornell box with
// a metallic tall box.  Equal to the bundled cornell2.json.gz up to the last bit of the tall box's
// rotation (glam's quaternion path is restated here with libm sinf/cosf).
std::string default_scene_json() {
    const float I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    // Scene::new adds Flat black as data 0 (scene/mod.rs:93-104); then main.rs:110-120
    std::string data = material_json(0, "{\"Flat\":{\"albedo\":" + rgb(0, 0, 0) + "}}") + "," +
                       material_json(1, "{\"Emissive\":{\"albedo\":" + rgb(1, 1, 1) + ",\"intensity\":20.0}}") + "," +
                       material_json(2, "{\"Diffuse\":{\"albedo\":" + rgb(0.73f, 0.73f, 0.73f) + ",\"roughness\":1.0}}") + "," +
                       material_json(3, "{\"Metallic\":{\"albedo\":" + rgb(0.55f, 0.55f, 0.55f) + ",\"roughness\":0.01}}") + "," +
                       material_json(4, "{\"Diffuse\":{\"albedo\":" + rgb(0.7f, 0.1f, 0.1f) + ",\"roughness\":0.5}}") + "," +
                       material_json(5, "{\"Diffuse\":{\"albedo\":" + rgb(0.2f, 0.7f, 0.4f) + ",\"roughness\":0.8}}");
    const int light = 1, white = 2, metal = 3, red = 4, green = 5;
    auto rect = [&](int mat, V x, V y) { return "{\"Rect\":" + rect_json(mat, x, y) + "}"; };
    // Quat::from_euler(YXZ, 20 deg, 0, 0) -> Mat3 (glam): q = (0, sin(a/2), 0, cos(a/2))
    const float angle = 20.0f * (3.14159265358979323846f / 180.0f);
    const float qy = std::sin(angle * 0.5f), qw = std::cos(angle * 0.5f);
    const float y2 = qy + qy, yy = qy * y2, wy = qw * y2;
    const float R[9] = {1.0f - yy, 0.0f, -wy, 0.0f, 1.0f, 0.0f, wy, 0.0f, 1.0f - yy};
    std::string objects =
        object_json(0, "camera", 0, I, {0.0f, 2.5f, 10.0f},
                    "{\"Camera\":{\"sensor_size\":0.024,\"focal_length\":0.05,\"aspect_ratio\":1.5,\"fstop\":1.4,\"focus\":12.5}}") + "," +
        object_json(1, nullptr, 0, I, {-2.5f, 2.5f, -2.5f}, rect(green, {0, 0, -2.5f}, {0, 2.5f, 0})) + "," +   // left
        object_json(2, nullptr, 0, I, {2.5f, 2.5f, -2.5f}, rect(red, {0, 0, 2.5f}, {0, 2.5f, 0})) + "," +       // right
        object_json(3, nullptr, 0, I, {0.0f, 2.5f, -5.0f}, rect(white, {2.5f, 0, 0}, {0, 2.5f, 0})) + "," +     // back
        object_json(4, nullptr, 0, I, {0.0f, 0.0f, -2.5f}, rect(white, {2.5f, 0, 0}, {0, 0, -2.5f})) + "," +    // floor
        object_json(5, nullptr, 0, I, {0.0f, 5.0f, -2.5f}, rect(white, {2.5f, 0, 0}, {0, 0, 2.5f})) + "," +     // ceiling
        object_json(6, nullptr, 1, I, {0.0f, 4.999f, -2.5f}, rect(light, {0.5f, 0, 0}, {0, 0, 0.5f})) + "," +   // light
        object_json(7, nullptr, 0, R, {-1.2f, 1.0f, -3.2f}, cuboid_json(metal, {0.5f, 0, 0}, {0, 1.0f, 0}, {0, 0, 0.4f})) + "," +
        object_json(8, nullptr, 0, I, {1.0f, 0.6f, -1.4f}, cuboid_json(white, {0.5f, 0, 0}, {0, 0.6f, 0}, {0, 0, 0.5f}));
    return "{\"roots\":[],\"root_material\":0,\"objects\":{\"collection\":{" + objects + "},\"next_key\":9},\"data\":{\"collection\":{" +
           data + "},\"next_key\":6}}";
}

// RGBA8 PNG (colour type 6), one IDAT, filter 0 on every row: image::RgbaImage::save (main.rs:294)
void write_png(const std::string &path, const uint8_t *rgba, uint32_t w, uint32_t h) {
    std::vector<uint8_t> raw((size_t)h * (1 + (size_t)w * 4));
    for (uint32_t y = 0; y < h; ++y) {
        raw[(size_t)y * (1 + (size_t)w * 4)] = 0;
        std::memcpy(&raw[(size_t)y * (1 + (size_t)w * 4) + 1], rgba + (size_t)y * w * 4, (size_t)w * 4);
    }
    uLongf zlen = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(zlen);
    if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6) != Z_OK) throw Error{BT_ERR_IO, "deflate failed"};
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) throw Error{BT_ERR_IO, "cannot create " + path};
    auto be32 = [](uint8_t *p, uint32_t v) { p[0] = v >> 24; p[1] = v >> 16; p[2] = v >> 8; p[3] = v; };
    auto chunk = [&](const char *type, const uint8_t *data, uint32_t len) {
        uint8_t hdr[8];
        be32(hdr, len);
        std::memcpy(hdr + 4, type, 4);
        uint32_t crc = (uint32_t)crc32(0L, hdr + 4, 4);
        if (len) crc = (uint32_t)crc32(crc, data, len);
        uint8_t tail[4];
        be32(tail, crc);
        std::fwrite(hdr, 1, 8, f);
        if (len) std::fwrite(data, 1, len, f);
        std::fwrite(tail, 1, 4, f);
    };
    const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    std::fwrite(sig, 1, 8, f);
    uint8_t ihdr[13];
    be32(ihdr, w);
    be32(ihdr + 4, h);
    ihdr[8] = 8; ihdr[9] = 6; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
    chunk("IHDR", ihdr, 13);
    chunk("IDAT", z.data(), (uint32_t)zlen);
    chunk("IEND", nullptr, 0);
    if (std::fclose(f) != 0) throw Error{BT_ERR_IO, "write error on " + path};
}

} // namespace bt
