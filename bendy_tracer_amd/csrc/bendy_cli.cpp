// bendy_cli.cpp -- headless counterpart of the reference's viewer (src/main.rs), over the C ABI.
//
// Same flags as the clap `Cli` (main.rs:49-72): --width 768 --height 512 --output <full|albedo|normal>
// (required) --samples 64 --subsample 2 --screenshot screenshots/render.png --scene scene.json.
// The reference opens a minifb window unconditionally (main.rs:79-87) and reacts to Ctrl+P / Ctrl+K;
// without a display this program runs the same progressive loop (one `Tracer::render` call of
// 1 x subsample^2 rays per pixel per iteration until --samples is reached, main.rs:245-254), keeps
// the frame in HBM, prints what the window title would show (main.rs:352-388) and then does what
// Ctrl+P does (preview -> PNG, main.rs:275-298) and, with --save-scene, what Ctrl+K does
// (pretty JSON, gzip for .gz, main.rs:299-313).  Extra flags: --seed, --save-scene, --device, --quiet, and
// --lens x,y,z,rs,step,radius[,max_steps] for the gravitational-lens EXTENSION (not in the reference; bt_lens).
#include <hip/hip_runtime.h>
#include <sys/stat.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/bendy_hip.h"

namespace {

[[noreturn]] void die(const std::string &msg) {
    std::fprintf(stderr, "error: %s\n", msg.c_str());
    std::exit(1);
}
void check(int rc, const char *what) {
    if (rc < 0) die(std::string(what) + ": " + bt_last_error());
}
void hip_check(hipError_t e, const char *what) {
    if (e != hipSuccess) die(std::string(what) + ": " + hipGetErrorString(e));
}
bool exists(const std::string &p) {
    struct stat st;
    return ::stat(p.c_str(), &st) == 0;
}
bool is_dir(const std::string &p) {
    struct stat st;
    return ::stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode);
}
void create_dir_all(const std::string &dir) {            // fs::create_dir_all, main.rs:292
    std::string cur;
    for (size_t i = 0; i <= dir.size(); ++i) {
        if (i == dir.size() || dir[i] == '/') {
            if (!cur.empty() && !exists(cur) && ::mkdir(cur.c_str(), 0777) != 0) die("cannot create directory " + cur);
        }
        if (i < dir.size()) cur += dir[i];
    }
}
std::string fmt_duration(double seconds) {               // main.rs:364-386
    long total_ms = (long)(seconds * 1000.0);
    long s = total_ms / 1000, ms = total_ms % 1000;
    char buf[64];
    if (s == 0) std::snprintf(buf, sizeof buf, "%ldms", ms);
    else std::snprintf(buf, sizeof buf, "%lds %ldms", s, ms);
    return buf;
}

struct Args {
    unsigned width = 768, height = 512;                  // main.rs:51-55
    std::string output;                                  // required, main.rs:57-58
    unsigned samples = 64, subsample = 2;                // main.rs:60-64
    std::string screenshot = "screenshots/render.png";   // main.rs:66-67
    std::string scene = "scene.json";                    // main.rs:69-70
    unsigned long long seed = 0x5EED;
    std::string save_scene;
    int device = 0;
    bool quiet = false;
    unsigned samples_per_call = 1;                       // main.rs:248 renders 1 sample per displayed frame
    bool no_screenshot = false;
    std::string stats_json;                              // per-call kernel times / segment counts (profiling harness)
    unsigned shard_rank = 0, shard_world = 1;            // --shard r,w: render rank r's tile shard of a w-rank job (measurement
                                                         // harness for bench.py --gpus N; no screenshot, the shard is tile-major)
    bool has_lens = false;
    bt_lens lens{};
};

void usage() {
    std::fprintf(stderr,
                 "usage: bendy-tracer-hip --output <full|albedo|normal> [--width 768] [--height 512] [--samples 64]\n"
                 "       [--subsample 2] [--screenshot screenshots/render.png] [--scene scene.json]\n"
                 "       [--seed N] [--save-scene PATH] [--device N] [--quiet]\n"
                 "       [--samples-per-call 1] [--no-screenshot] [--stats-json PATH] [--shard rank,world]   (measurement harness)\n"
                 "       [--lens x,y,z,rs,step,radius[,max_steps]]   (extension: not in the reference)\n");
}

Args parse(int argc, char **argv) {
    Args a;
    for (int i = 1; i < argc; ++i) {
        std::string k = argv[i], v;
        size_t eq = k.find('=');
        bool has_v = false;
        if (eq != std::string::npos) { v = k.substr(eq + 1); k = k.substr(0, eq); has_v = true; }
        auto val = [&]() -> std::string {
            if (has_v) return v;
            if (i + 1 >= argc) { usage(); die("missing value for " + k); }
            return argv[++i];
        };
        if (k == "--width") a.width = (unsigned)std::strtoul(val().c_str(), nullptr, 10);
        else if (k == "--height") a.height = (unsigned)std::strtoul(val().c_str(), nullptr, 10);
        else if (k == "--output") a.output = val();
        else if (k == "--samples") a.samples = (unsigned)std::strtoul(val().c_str(), nullptr, 10);
        else if (k == "--subsample") a.subsample = (unsigned)std::strtoul(val().c_str(), nullptr, 10);
        else if (k == "--screenshot") a.screenshot = val();
        else if (k == "--scene") a.scene = val();
        else if (k == "--seed") a.seed = std::strtoull(val().c_str(), nullptr, 0);
        else if (k == "--save-scene") a.save_scene = val();
        else if (k == "--device") a.device = std::atoi(val().c_str());
        else if (k == "--quiet") a.quiet = true;
        else if (k == "--samples-per-call") a.samples_per_call = std::max(1u, (unsigned)std::strtoul(val().c_str(), nullptr, 10));
        else if (k == "--no-screenshot") a.no_screenshot = true;
        else if (k == "--stats-json") a.stats_json = val();
        else if (k == "--shard") {
            const std::string spec = val();
            if (std::sscanf(spec.c_str(), "%u,%u", &a.shard_rank, &a.shard_world) != 2 || a.shard_world == 0 || a.shard_rank >= a.shard_world)
                die("--shard expects rank,world with rank < world");
            a.no_screenshot = true;
        }
        else if (k == "--lens") {
            const std::string spec = val();
            float f[7] = {0, 0, 0, 0, 0, 0, 4096};
            int n = std::sscanf(spec.c_str(), "%f,%f,%f,%f,%f,%f,%f", &f[0], &f[1], &f[2], &f[3], &f[4], &f[5], &f[6]);
            if (n < 6) die("--lens expects x,y,z,rs,step,radius[,max_steps]");
            a.lens.centre[0] = f[0]; a.lens.centre[1] = f[1]; a.lens.centre[2] = f[2];
            a.lens.rs = f[3]; a.lens.step = f[4]; a.lens.radius = f[5];
            a.lens.max_steps = (uint32_t)f[6];
            a.has_lens = true;
        }
        else if (k == "--help" || k == "-h") { usage(); std::exit(0); }
        else { usage(); die("unknown argument " + k); }
    }
    if (a.output.empty()) { usage(); die("the following required arguments were not provided: --output <OUTPUT>"); }
    if (a.width == 0 || a.height == 0) die("width and height must be positive");
    return a;
}

} // namespace

int main(int argc, char **argv) {
    const Args args = parse(argc, argv);
    // main.rs:23-47: CLI Output -> tracer Output + ColorSpace
    int output, color_space;
    if (args.output == "full") { output = BT_OUTPUT_FULL; color_space = BT_COLOR_SRGB; }
    else if (args.output == "albedo") { output = BT_OUTPUT_ALBEDO; color_space = BT_COLOR_SRGB; }
    else if (args.output == "normal") { output = BT_OUTPUT_NORMAL; color_space = BT_COLOR_NORMAL; }
    else die("invalid value '" + args.output + "' for '--output <OUTPUT>' [possible values: full, albedo, normal]");

    hip_check(hipSetDevice(args.device), "hipSetDevice");

    // main.rs:93-214: load the scene if the file exists, else the built-in Cornell scene
    bt_scene *scene;
    if (exists(args.scene)) {
        scene = bt_scene_load(args.scene.c_str());
        if (!scene) die(bt_last_error());
        std::fprintf(stderr, "loaded scene from %s\n", args.scene.c_str());
    } else {
        scene = bt_scene_default();
        if (!scene) die(bt_last_error());
    }
    uint64_t camera = 0;
    check(bt_scene_find_by_tag(scene, "camera", &camera), "find_by_tag(\"camera\")");          // main.rs:216
    check(bt_scene_set_camera_aspect(scene, camera, (float)args.width / (float)args.height), "aspect");  // :218-223
    if (args.has_lens) check(bt_scene_set_lens(scene, &args.lens), "bt_scene_set_lens");
    {
        // measurement harness: launch-shape knobs from the environment (the LIBRARY never reads it; this tool does, like
        // tools/*.py through Scene.tuning_from_env)
        bt_tuning t;
        bt_tuning_default(&t);
        bool any = false;
        auto env = [&](const char *name, auto &field) {
            if (const char *e = std::getenv(name)) { field = (std::remove_reference_t<decltype(field)>)std::strtoll(e, nullptr, 10); any = true; }
        };
        env("BT_SLICES", t.slices); env("BT_PHASE_VOTE", t.phase_vote); env("BT_SCRATCH_CAP", t.scratch_cap_bytes); env("BT_PACKED", t.packed);
        if (any) check(bt_scene_set_tuning(scene, &t), "bt_scene_set_tuning");
    }

    bt_config cfg;
    bt_config_default(&cfg);
    cfg.output = output;
    cfg.chunks_x = 8;                                      // main.rs:225-230
    cfg.chunks_y = 4;
    bt_render_config rc;
    bt_render_config_default(&rc);
    rc.subsample_n = args.subsample <= 1 ? 0 : args.subsample;   // main.rs:234-237
    const unsigned nn = rc.subsample_n ? rc.subsample_n * rc.subsample_n : 1;

    // Buffer::new (buffer.rs:41-50), resident in HBM
    const bool sharded = args.shard_world > 1;
    const size_t n_px = sharded ? bt_shard_floats(args.width, args.height, args.shard_world) / 4 : (size_t)args.width * args.height;
    std::vector<float> init(n_px * 4, 0.0f);
    for (size_t i = 0; i < n_px; ++i) init[4 * i + 3] = 1.0f;
    float *d_frame = nullptr;
    uint8_t *d_rgba8 = nullptr;
    hip_check(hipMalloc((void **)&d_frame, n_px * 16), "hipMalloc");
    hip_check(hipMalloc((void **)&d_rgba8, n_px * 4), "hipMalloc");
    hip_check(hipMemcpy(d_frame, init.data(), n_px * 16, hipMemcpyHostToDevice), "hipMemcpy");

    // main.rs:245-254: one sample per call while buffer.samples() < max_samples
    unsigned buffer_samples = 0;
    double sum_delta = 0.0;
    const auto start = std::chrono::steady_clock::now();
    std::string per_call;                                  // --stats-json rows
    while (buffer_samples < args.samples) {
        rc.samples = std::min(args.samples_per_call, std::max(1u, (args.samples - buffer_samples) / nn));
        rc.sample_base = (buffer_samples + nn - 1) / nn;
        const auto t0 = std::chrono::steady_clock::now();
        int st = sharded ? bt_render_shard_device(scene, camera, &cfg, &rc, d_frame, args.width, args.height, args.shard_rank,
                                                  args.shard_world, args.seed, nullptr)
                         : bt_render_device(scene, camera, &cfg, &rc, d_frame, args.width, args.height, args.seed, nullptr);
        check(st, "bt_render_device");
        hip_check(hipDeviceSynchronize(), "hipDeviceSynchronize");
        const double delta = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        sum_delta += delta;
        buffer_samples += rc.samples * nn;                 // Buffer::inc_samples, mod.rs:199
        if (!args.stats_json.empty()) {
            bt_stats cs{};
            bt_scene_last_stats(scene, &cs);
            char row[256];
            std::snprintf(row, sizeof row, "%s{\"kernel_ms\": %.5f, \"segments\": %llu, \"samples\": %llu, \"pixels\": %llu, \"slices\": %u, \"packed\": %u}",
                          per_call.empty() ? "" : ", ", cs.kernel_ms, (unsigned long long)cs.segments,
                          (unsigned long long)cs.samples, (unsigned long long)cs.pixels, cs.slices, cs.packed);
            per_call += row;
        }
        if (!args.quiet)
            std::fprintf(stderr, "bendy tracer; samples: %u/%u; delta t: %s\n", buffer_samples, args.samples,
                         fmt_duration(delta / (rc.samples * nn)).c_str());
    }
    const double total = std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();
    bt_stats stats{};
    bt_scene_last_stats(scene, &stats);
    std::fprintf(stderr, "bendy tracer; samples: %u/%u; avg t per sample: %s; total t: %s\n", buffer_samples, args.samples,
                 fmt_duration(buffer_samples ? sum_delta / buffer_samples : 0.0).c_str(), fmt_duration(total).c_str());
    std::fprintf(stderr, "%.1f Msamples/s (render calls only)\n",
                 sum_delta > 0 ? (double)n_px * buffer_samples / sum_delta / 1e6 : 0.0);

    if (!args.stats_json.empty()) {
        FILE *f = std::fopen(args.stats_json.c_str(), "w");
        if (!f) die("cannot write " + args.stats_json);
        std::fprintf(f, "{\"width\": %u, \"height\": %u, \"samples_per_call\": %u, \"subsample\": %u, \"calls\": [%s]}\n", args.width,
                     args.height, args.samples_per_call, args.subsample, per_call.c_str());
        std::fclose(f);
    }

    // Ctrl+P (main.rs:275-298)
    std::string shot = args.screenshot;
    if (!args.no_screenshot) {
        size_t slash = shot.find_last_of('/');
        std::string file = slash == std::string::npos ? shot : shot.substr(slash + 1);
        if (file.find('.') == std::string::npos)           // no extension -> with_file_name("render.png")
            shot = (slash == std::string::npos ? std::string() : shot.substr(0, slash + 1)) + "render.png";
        slash = shot.find_last_of('/');
        if (slash != std::string::npos && slash > 0) {
            std::string dir = shot.substr(0, slash);
            if (exists(dir)) {
                if (!is_dir(dir)) die(dir + " is not a directory");
            } else {
                create_dir_all(dir);
            }
        }
    }
    if (!args.no_screenshot) {
        check(bt_preview_device(d_frame, d_rgba8, args.width, args.height, buffer_samples ? buffer_samples : 1, color_space, nullptr),
              "bt_preview_device");
        std::vector<uint8_t> rgba8(n_px * 4);
        hip_check(hipMemcpy(rgba8.data(), d_rgba8, n_px * 4, hipMemcpyDeviceToHost), "hipMemcpy");
        check(bt_write_png(shot.c_str(), rgba8.data(), args.width, args.height), "bt_write_png");
        std::fprintf(stderr, "saved screenshot to %s\n", shot.c_str());
    }

    // Ctrl+K (main.rs:299-313)
    if (!args.save_scene.empty()) {
        check(bt_scene_save(scene, args.save_scene.c_str()), "bt_scene_save");
        std::fprintf(stderr, "saved scene to %s\n", args.save_scene.c_str());
    }

    (void)hipFree(d_frame);
    (void)hipFree(d_rgba8);
    bt_scene_free(scene);
    return 0;
}
