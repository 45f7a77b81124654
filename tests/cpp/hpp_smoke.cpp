// Exercises include/bendy_tracer.hpp (C++ mirror of the reference API).  Usage: hpp_smoke <scene> <out.bin>
// Without a GPU the render throws bendy::Error(BT_ERR_DEVICE); with one, the running sums are written out.
#include <cstdio>
#include <cstring>

#include "bendy_tracer.hpp"

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    bendy::Config def;
    if (def.max_bounces != 8 || def.max_volume_bounces != 32 || def.chunks_x != 4 || def.chunks_y != 2) return 3;
    if (bendy::RenderConfig().samples != 64 || bendy::Subsample::subpixel(3).subpixel_count() != 9) return 3;
    try {
        bendy::Scene::load("/nonexistent/scene.json");
        return 4;
    } catch (const bendy::Error &e) {
        if (e.code != BT_ERR_IO) return 4;
    }
    bendy::Scene scene = bendy::Scene::load(argv[1]);
    auto camera = scene.find_by_tag("camera");
    if (!camera || scene.find_by_tag("nope")) return 5;
    const unsigned w = 48, h = 27;
    scene.set_camera_aspect(*camera, (float)w / h);
    if (scene.to_json().find("\"roots\"") == std::string::npos) return 6;
    scene.trim();                                        // nothing held yet: a no-op that must not fail (with or without a GPU)
    bendy::Tracer tracer = bendy::Tracer::with_config({.chunks_x = 8, .chunks_y = 4});
    bendy::Buffer buffer(w, h, bendy::ColorSpace::SRgb);
    if (tracer.render(scene, *camera, bendy::RenderConfig::with_samples(0), buffer) != bendy::Status::Done) return 7;
    try {
        while (buffer.samples() < 8)
            if (tracer.render(scene, *camera, bendy::RenderConfig::with_samples_subsample(1, bendy::Subsample::subpixel(2)), buffer, 99) !=
                bendy::Status::InProgress)
                return 8;
    } catch (const bendy::Error &e) {
        std::printf("render error %d: %s\n", e.code, e.what());
        return e.code == BT_ERR_DEVICE ? 42 : 9;
    }
    FILE *f = std::fopen(argv[2], "wb");
    std::fwrite(buffer.data(), sizeof(float), (size_t)w * h * 4, f);
    auto p = buffer.preview();
    std::fwrite(p.data(), 1, p.size(), f);
    std::fclose(f);
    scene.trim();                                        // returns the scratch of the renders above
    std::printf("ok samples=%u\n", buffer.samples());
    return 0;
}
