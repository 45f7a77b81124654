// bendy_tracer.hpp -- header-only C++ mirror of the reference's library API over the C ABI of
// libbendy_hip.so (include/bendy_hip.h).  The reference is a Rust crate; its public surface used by
// src/main.rs is `Scene`, `Tracer`, `Config`, `RenderConfig`, `Subsample`, `Output`, `Status`,
// `Buffer`, `ColorSpace` (tracer/mod.rs:16-203, tracer/buffer.rs:11-179, scene/mod.rs:84-146).
// Same names, same defaults, same call shapes; reference panics surface as bendy::Error.
//
//     bendy::Scene scene = bendy::Scene::load("scene.json.gz");
//     auto camera = scene.find_by_tag("camera").value();
//     scene.set_camera_aspect(camera, 1920.f / 1080.f);                       // main.rs:218-223
//     bendy::Tracer tracer = bendy::Tracer::with_config({.chunks_x = 8, .chunks_y = 4});
//     bendy::Buffer buffer(1920, 1080, bendy::ColorSpace::SRgb);              // host-resident sums
//     while (buffer.samples() < 64)
//         tracer.render(scene, camera, bendy::RenderConfig::with_samples_subsample(1, bendy::Subsample::subpixel(2)), buffer);
//     auto rgba8 = buffer.preview();                                          // Buffer::preview
#pragma once
#include <cstdint>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "bendy_hip.h"

namespace bendy {

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};
inline int check(int rc) {
    if (rc < 0) throw Error(rc, bt_last_error());
    return rc;
}

enum class Output { Full = BT_OUTPUT_FULL, Albedo = BT_OUTPUT_ALBEDO, Normal = BT_OUTPUT_NORMAL, Depth = BT_OUTPUT_DEPTH };   // mod.rs:108-115
enum class ColorSpace { None = BT_COLOR_NONE, Normal = BT_COLOR_NORMAL, Linear = BT_COLOR_LINEAR, SRgb = BT_COLOR_SRGB };   // buffer.rs:11-17
enum class Status { Done = BT_DONE, InProgress = BT_IN_PROGRESS };                                                             // mod.rs:159-163
using ObjectRef = std::uint64_t;

struct Subsample {                                      // mod.rs:47-68
    unsigned n = 0;                                     // 0 = None, n = Subpixel(n)
    static Subsample none() { return {0}; }
    static Subsample subpixel(unsigned n) { return {n}; }
    float subpixel_size() const { return n == 0 ? 1.0f : 1.0f / (float)n; }
    unsigned subpixel_count() const { return n == 0 ? 1 : n * n; }
};

struct Config {                                         // mod.rs:16-45
    unsigned max_bounces = 8, max_volume_bounces = 32;
    float clip_min = 0.01f, clip_max = 1000.0f, volume_step = 0.1f;
    unsigned chunks_x = 4, chunks_y = 2;
    Output output = Output::Full;
};

struct RenderConfig {                                   // mod.rs:117-157
    Subsample subsample{};
    unsigned samples = 64;
    std::optional<Output> output;
    std::optional<unsigned> max_bounces, max_volume_bounces;
    std::optional<float> volume_step;
    static RenderConfig with_samples(unsigned s) { RenderConfig r; r.samples = s; return r; }
    static RenderConfig with_samples_subsample(unsigned s, Subsample sub) { RenderConfig r; r.samples = s; r.subsample = sub; return r; }
};

class Scene {                                           // scene/mod.rs:84-146
  public:
    static Scene load(const std::string &path) { return Scene(bt_scene_load(path.c_str())); }            // main.rs:93-102
    static Scene from_json(const std::string &json) { return Scene(bt_scene_from_json(json.data(), json.size())); }
    static Scene default_scene() { return Scene(bt_scene_default()); }                                    // main.rs:107-214
    Scene(Scene &&o) noexcept : h_(o.h_) { o.h_ = nullptr; }
    Scene &operator=(Scene &&o) noexcept { if (this != &o) { bt_scene_free(h_); h_ = o.h_; o.h_ = nullptr; } return *this; }
    Scene(const Scene &) = delete;
    Scene &operator=(const Scene &) = delete;
    ~Scene() { bt_scene_free(h_); }

    std::optional<ObjectRef> find_by_tag(const std::string &tag) const {                                  // scene/mod.rs:124-129
        ObjectRef r = 0;
        return bt_scene_find_by_tag(h_, tag.c_str(), &r) == 0 ? std::optional<ObjectRef>(r) : std::nullopt;
    }
    void set_camera_aspect(ObjectRef camera, float aspect) { check(bt_scene_set_camera_aspect(h_, camera, aspect)); }
    void save(const std::string &path) const { check(bt_scene_save(h_, path.c_str())); }                  // main.rs:299-313
    std::string to_json() const {
        int n = check(bt_scene_to_json(h_, nullptr, 0));
        std::string s((size_t)n + 1, '\0');
        check(bt_scene_to_json(h_, s.data(), s.size()));
        s.resize((size_t)n);
        return s;
    }
    bt_stats last_stats() const { bt_stats st{}; check(bt_scene_last_stats(h_, &st)); return st; }
    void trim() { check(bt_scene_trim(h_)); }           // returns the handle's scratch / cached frame to the device (not in the reference)
    bt_scene *handle() const { return h_; }

  private:
    explicit Scene(bt_scene *h) : h_(h) { if (!h_) throw Error(bt_last_error_code(), bt_last_error()); }
    bt_scene *h_;
};

class Buffer {                                          // buffer.rs:32-179 (host-resident RGBA32F sums)
  public:
    Buffer(unsigned width, unsigned height, ColorSpace cs = ColorSpace::SRgb) : w_(width), h_(height), cs_(cs) { clear(); }
    unsigned width() const { return w_; }
    unsigned height() const { return h_; }
    unsigned samples() const { return samples_; }
    float pixel_width() const { return 2.0f * (1.0f / (float)w_); }                                       // buffer.rs:68-71
    float pixel_height() const { return 2.0f * (1.0f / (float)h_); }
    void clear() {                                                                                        // buffer.rs:82-87
        data_.assign((size_t)w_ * h_ * 4, 0.0f);
        for (size_t i = 3; i < data_.size(); i += 4) data_[i] = 1.0f;
        samples_ = 0;
    }
    void resize(unsigned width, unsigned height) { w_ = width; h_ = height; clear(); }                   // buffer.rs:89-100
    void inc_samples(unsigned n) { samples_ += n; }                                                       // buffer.rs:155-157
    float *data() { return data_.data(); }
    const float *data() const { return data_.data(); }
    std::vector<std::uint8_t> preview() const {                                                           // buffer.rs:117-138
        std::vector<std::uint8_t> out((size_t)w_ * h_ * 4);
        check(bt_preview(data_.data(), out.data(), w_, h_, samples_ ? samples_ : 1, (int)cs_));
        return out;
    }
    void save_png(const std::string &path) const { auto p = preview(); check(bt_write_png(path.c_str(), p.data(), w_, h_)); }   // main.rs:294

  private:
    unsigned w_, h_, samples_ = 0;
    ColorSpace cs_;
    std::vector<float> data_;
};

class Tracer {                                          // mod.rs:165-203
  public:
    Config config;
    static Tracer with_config(Config c) { Tracer t; t.config = c; return t; }
    // Tracer::render (mod.rs:179-202).  `seed` stands in for SmallRng::from_entropy() (mod.rs:239-242).
    Status render(const Scene &scene, ObjectRef camera, const RenderConfig &rc, Buffer &buffer, std::uint64_t seed = 0x5EED) const {
        bt_config c{config.max_bounces, config.max_volume_bounces, config.clip_min, config.clip_max, config.volume_step,
                    config.chunks_x, config.chunks_y, (int)config.output};
        bt_render_config r{};
        r.subsample_n = rc.subsample.n;
        r.samples = rc.samples;
        r.has_output = rc.output.has_value();
        r.output = rc.output ? (int)*rc.output : 0;
        r.has_max_bounces = rc.max_bounces.has_value();
        r.max_bounces = rc.max_bounces.value_or(0);
        r.has_max_volume_bounces = rc.max_volume_bounces.has_value();
        r.max_volume_bounces = rc.max_volume_bounces.value_or(0);
        r.has_volume_step = rc.volume_step.has_value();
        r.volume_step = rc.volume_step.value_or(0.0f);
        r.sample_base = buffer.samples() / rc.subsample.subpixel_count();
        int st = check(bt_render(scene.handle(), camera, &c, &r, buffer.data(), buffer.width(), buffer.height(), seed));
        if (st == BT_IN_PROGRESS) buffer.inc_samples(rc.samples * rc.subsample.subpixel_count());        // mod.rs:199
        return (Status)st;
    }
};

} // namespace bendy
