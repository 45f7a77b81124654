// bt_api.cpp -- the C ABI of libbendy_hip.so (include/bendy_hip.h): scene handles, parameter
// preparation for Tracer::render (reference tracer/mod.rs:179-320) and kernel launches.
// There is no CPU fallback: without a HIP device every render entry point returns BT_ERR_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <memory>
#include <string>

#include "../../include/bendy_hip.h"
#include "bt_scene.hpp"
#include "bt_types.h"

#pragma STDC FP_CONTRACT OFF

extern "C" hipError_t bt_launch_render(const BtLaunch *P, int output, unsigned grid, size_t lds_bytes, hipStream_t stream);
extern "C" hipError_t bt_launch_unshard(const float *gathered, float *frame, uint32_t width, uint32_t height,
                                        uint32_t tiles_x, uint32_t tiles_y, uint32_t world, uint32_t tiles_per_rank,
                                        hipStream_t stream);
extern "C" hipError_t bt_launch_preview(const float *rgba, uint8_t *out, uint32_t n, uint32_t samples, int color_space,
                                        hipStream_t stream);

static_assert(BT_TILE == BT_TILE_DIM, "public and device tile sizes must agree");

#ifndef BT_POOL_RECORDS
#define BT_POOL_RECORDS 128        // PathRec records per workgroup for the drain of a packed rect launch (at most 256)
#endif
namespace {

thread_local std::string g_error;
thread_local int g_error_code = 0;
constexpr uint64_t kDefaultScratchCap = 2ull << 30;   // parked sample values per launch; deeper renders are split
constexpr uint32_t kCounterSlots = 64;                // work counters: one memset per 64 renders instead of one per render
constexpr uint32_t kScratchShrinkAfter = 8;           // renders in a row that need < 1/4 of the scratch before it shrinks
int set_error(int code, const std::string &msg) {
    g_error = msg;
    g_error_code = code;
    return code;
}

template <class T> struct DeviceArray {
    T *ptr = nullptr;
    size_t count = 0;
    ~DeviceArray() { release(); }
    void release() {
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        count = 0;
    }
    hipError_t upload(const std::vector<T> &src) {
        release();
        count = src.size();
        size_t bytes = sizeof(T) * (count ? count : 1);
        hipError_t e = hipMalloc((void **)&ptr, bytes);
        if (e != hipSuccess) return e;
        if (count) e = hipMemcpy(ptr, src.data(), sizeof(T) * count, hipMemcpyHostToDevice);
        return e;
    }
};

} // namespace

struct bt_scene {
    bt::Scene scene;
    std::string source;            // the JSON document the scene was parsed from (for bt_scene_save)
    bt::FlatScene flat;
    bool flat_valid = false;
    bool device_valid = false;
    int device = -1;
    DeviceArray<BtPrim> d_prims;
    DeviceArray<BtMaterial> d_materials;
    DeviceArray<BtVolume> d_volumes;
    DeviceArray<BtLight> d_lights;
    DeviceArray<BtLightFace> d_light_faces;
    DeviceArray<BtSpherePair> d_sphere_pairs;
    DeviceArray<BtSphereRow> d_sphere_rows;
    DeviceArray<BtRectAAN> d_aan_rows;
    DeviceArray<BtRectLA> d_la_rows;
    DeviceArray<int32_t> d_other_rows;
    DeviceArray<float> d_density;
    DeviceArray<int32_t> d_lens_prims;     // lens extension: rows of d_prims near the sphere of influence
    bt_lens lens_prims_for{};              // the lens d_lens_prims was built for
    bool lens_prims_valid = false;
    unsigned long long *d_counters = nullptr;   // kCounterSlots x 16 words: render number n counts into slot n mod kCounterSlots
    uint32_t render_seq = 0;       // renders issued on this handle (selects the counter slot)
    uint32_t last_slot = 0;
    float *d_scratch = nullptr;    // parked sample values of sliced renders
    size_t scratch_bytes = 0;
    uint32_t scratch_small_streak = 0;   // consecutive renders that needed less than a quarter of the scratch held
    float *d_host_frame = nullptr; // device copy of the caller's host buffer (bt_render), kept between calls
    size_t host_frame_bytes = 0;
    int n_cu = 0;                  // hipDeviceProp_t::multiProcessorCount of `device`
    bt_tuning tuning{};            // bt_scene_set_tuning; zero / negative fields = automatic
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    bt_stats last{};
    bool lens_on = false;          // lens extension (not in the reference), bt_scene_set_lens
    bt_lens lens{};
    bool stats_pending = false;

    bt_scene() { bt_tuning_default(&tuning); }
    // everything that lives on `device` besides the scene tables (which upload() replaces)
    void release_device_state() {
        if (d_counters) (void)hipFree(d_counters);
        if (d_scratch) (void)hipFree(d_scratch);
        if (d_host_frame) (void)hipFree(d_host_frame);
        if (ev_start) (void)hipEventDestroy(ev_start);
        if (ev_stop) (void)hipEventDestroy(ev_stop);
        d_counters = nullptr;
        d_scratch = nullptr;
        d_host_frame = nullptr;
        scratch_bytes = host_frame_bytes = 0;
        scratch_small_streak = 0;
        ev_start = ev_stop = nullptr;
        stats_pending = false;
    }
    ~bt_scene() { release_device_state(); }
};

namespace {

#define BT_HIP(expr)                                                                                     \
    do {                                                                                                 \
        hipError_t _e = (expr);                                                                          \
        if (_e != hipSuccess)                                                                            \
            return set_error(BT_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(_e));          \
    } while (0)

int ensure_flat(bt_scene *s) {
    if (s->flat_valid) return 0;
    try {
        s->flat = bt::flatten_scene(s->scene);
    } catch (const bt::Error &e) {
        return set_error(e.code, e.message);
    }
    s->flat_valid = true;
    s->device_valid = false;
    return 0;
}

int ensure_device(bt_scene *s) {
    int rc = ensure_flat(s);
    if (rc) return rc;
    int dev = -1;
    BT_HIP(hipGetDevice(&dev));
    if (s->device_valid && s->device == dev) return 0;
    if (s->device >= 0 && s->device != dev) {
        // the handle moves to another GPU: counters, scratch, the cached host frame and the events belong to the old
        // one (a kernel on `dev` must not write into them) -- free them there and start afresh here
        (void)hipSetDevice(s->device);
        s->release_device_state();
        BT_HIP(hipSetDevice(dev));
    }
    {
        hipDeviceProp_t prop;
        BT_HIP(hipGetDeviceProperties(&prop, dev));
        s->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    BT_HIP(s->d_prims.upload(s->flat.prims));
    BT_HIP(s->d_materials.upload(s->flat.materials));
    BT_HIP(s->d_volumes.upload(s->flat.volumes));
    BT_HIP(s->d_lights.upload(s->flat.lights));
    BT_HIP(s->d_light_faces.upload(s->flat.light_faces));
    {
        std::vector<BtSpherePair> pairs;
        const std::vector<BtPrim> &pr = s->flat.prims;
        bool spheres_only = true;
        for (const BtPrim &R : pr) spheres_only = spheres_only && (R.kind & BT_PRIM_SHAPE_MASK) == BT_PRIM_SPHERE;
        if (spheres_only)
            for (size_t i = 0; i < pr.size(); i += 2) {
                const BtPrim &A = pr[i], &B = pr[i + 1 < pr.size() ? i + 1 : i];
                BtSpherePair q{};
                q.cx[0] = A.c.x; q.cy[0] = A.c.y; q.cz[0] = A.c.z; q.radius[0] = A.radius; q.object[0] = A.object;
                q.cx[1] = B.c.x; q.cy[1] = B.c.y; q.cz[1] = B.c.z; q.radius[1] = B.radius; q.object[1] = B.object;
                pairs.push_back(q);
            }
        BT_HIP(s->d_sphere_pairs.upload(pairs));
        std::vector<BtSphereRow> rows;
        if (spheres_only)
            for (const BtPrim &R : pr) rows.push_back(BtSphereRow{R.c.x, R.c.y, R.c.z, R.radius * R.radius});
        if (rows.size() & 1) rows.push_back(rows.back());           // (never visited: keeps an x8 load of the last pair inside the table)
        BT_HIP(s->d_sphere_rows.upload(rows));
    }
    BT_HIP(s->d_density.upload(s->flat.density));
    BT_HIP(s->d_aan_rows.upload(s->flat.aan_rows));
    BT_HIP(s->d_la_rows.upload(s->flat.la_rows));
    BT_HIP(s->d_other_rows.upload(s->flat.other_rows));
    if (!s->d_counters) {
        BT_HIP(hipMalloc((void **)&s->d_counters, kCounterSlots * 16 * sizeof(unsigned long long)));
        s->render_seq = 0;
    }
    if (!s->ev_start) BT_HIP(hipEventCreate(&s->ev_start));
    if (!s->ev_stop) BT_HIP(hipEventCreate(&s->ev_stop));
    s->device = dev;
    s->device_valid = true;
    s->lens_prims_valid = false;
    return 0;
}

// Lens extension: rows of the primitive table whose surface can come within `reach` of the lens centre.
// Conservative by construction (double arithmetic, bounding spheres for rects, 0.1 % slack): a row that is left
// out cannot be touched by any point within `reach` of the centre.
std::vector<int32_t> lens_candidates(const std::vector<BtPrim> &prims, const bt_lens &lens, double reach) {
    std::vector<int32_t> rows;
    const double cx = lens.centre[0], cy = lens.centre[1], cz = lens.centre[2];
    reach *= 1.001;
    for (size_t i = 0; i < prims.size(); ++i) {
        const BtPrim &R = prims[i];
        bool near_ = true;
        if ((R.kind & BT_PRIM_SHAPE_MASK) == BT_PRIM_SPHERE) {
            const double d = std::sqrt((R.c.x - cx) * (R.c.x - cx) + (R.c.y - cy) * (R.c.y - cy) + (R.c.z - cz) * (R.c.z - cz));
            near_ = std::fabs(d - (double)R.radius) <= reach + 1e-3 * (double)R.radius;    // the SURFACE is what is hit
        } else {
            // forward matrix = inverse of (icx, icy, icz); corners = t + M * (+-hw * ax +- hh * ay)
            const double a[3][3] = {{R.icx.x, R.icy.x, R.icz.x}, {R.icx.y, R.icy.y, R.icz.y}, {R.icx.z, R.icy.z, R.icz.z}};
            const double det = a[0][0] * (a[1][1] * a[2][2] - a[1][2] * a[2][1]) - a[0][1] * (a[1][0] * a[2][2] - a[1][2] * a[2][0]) +
                               a[0][2] * (a[1][0] * a[2][1] - a[1][1] * a[2][0]);
            if (std::isfinite(det) && std::fabs(det) > 1e-30) {
                double m[3][3];
                for (int r = 0; r < 3; ++r)
                    for (int c = 0; c < 3; ++c) {
                        const int r1 = (c + 1) % 3, r2 = (c + 2) % 3, c1 = (r + 1) % 3, c2 = (r + 2) % 3;
                        m[r][c] = (a[r1][c1] * a[r2][c2] - a[r1][c2] * a[r2][c1]) / det;
                    }
                const double hw = std::sqrt((double)R.w_sqr), hh = std::sqrt((double)R.h_sqr);
                // local axes: Rect.x / Rect.y, which BT_PRIM_RECT_LA / _AAN rows replace by other constants (bt_types.h)
                double lx[3] = {R.ax.x, R.ax.y, R.ax.z}, ly[3] = {R.ay.x, R.ay.y, R.ay.z};
                if ((R.kind & BT_PRIM_SHAPE_MASK) == BT_PRIM_RECT_LA || (R.kind & BT_PRIM_SHAPE_MASK) == BT_PRIM_RECT_AAN)
                    for (int i = 0; i < 3; ++i) {
                        lx[i] = i == R.aa_u ? 1.0 : 0.0;
                        ly[i] = i == R.aa_v ? 1.0 : 0.0;
                    }
                double bound = 0.0;
                for (int sx = -1; sx <= 1; sx += 2)
                    for (int sy = -1; sy <= 1; sy += 2) {
                        const double l[3] = {sx * hw * lx[0] + sy * hh * ly[0], sx * hw * lx[1] + sy * hh * ly[1],
                                             sx * hw * lx[2] + sy * hh * ly[2]};
                        double wv[3];
                        for (int r = 0; r < 3; ++r) wv[r] = m[r][0] * l[0] + m[r][1] * l[1] + m[r][2] * l[2];
                        bound = std::max(bound, std::sqrt(wv[0] * wv[0] + wv[1] * wv[1] + wv[2] * wv[2]));
                    }
                const double d = std::sqrt((R.t.x - cx) * (R.t.x - cx) + (R.t.y - cy) * (R.t.y - cy) + (R.t.z - cz) * (R.t.z - cz));
                near_ = d - bound * 1.001 <= reach;
            }
        }
        if (near_) rows.push_back((int32_t)i);
    }
    return rows;
}

// ChunkConfig::with_configs (mod.rs:217-229) + camera setup (mod.rs:244-267)
int fill_launch(bt_scene *s, uint64_t camera_ref, const bt_config *cfg, const bt_render_config *rc, uint32_t width,
                uint32_t height, uint64_t seed, BtLaunch &P, int &output) {
    if (!s || !cfg || !rc) return set_error(BT_ERR_INVALID_ARG, "null argument");
    if (width == 0 || height == 0) return set_error(BT_ERR_INVALID_ARG, "zero-sized buffer");
    int ci = s->scene.object_index(camera_ref);
    if (ci < 0) return set_error(BT_ERR_INVALID_REF, "invalid object ref " + std::to_string(camera_ref));
    const bt::Object &cam = s->scene.objects[ci];
    if (cam.kind != bt::OBJ_CAMERA) return set_error(BT_ERR_NOT_CAMERA, "expected a camera object");

    std::memset(&P, 0, sizeof P);
    const bt::FlatScene &f = s->flat;
    P.prims = s->d_prims.ptr;
    P.materials = s->d_materials.ptr;
    P.volumes = s->d_volumes.ptr;
    P.lights = s->d_lights.ptr;
    P.light_faces = s->d_light_faces.ptr;
    P.sphere_pairs = s->d_sphere_pairs.count ? s->d_sphere_pairs.ptr : nullptr;
    P.sphere_rows = s->d_sphere_rows.count ? s->d_sphere_rows.ptr : nullptr;
    P.density = s->d_density.ptr;
    P.n_prims = (int32_t)f.prims.size();
    P.n_materials = (int32_t)f.materials.size();
    P.n_volumes = (int32_t)f.volumes.size();
    P.n_lights = (int32_t)f.lights.size();
    P.n_light_faces = (int32_t)f.light_faces.size();
    P.n_density = (int32_t)f.density.size();
    P.any_rects = 0;
    P.any_volumes = 0;
    for (const BtPrim &R : f.prims) {
        if ((R.kind & BT_PRIM_SHAPE_MASK) != BT_PRIM_SPHERE) P.any_rects = 1;
        if (R.volume >= 0) P.any_volumes = 1;
    }
    P.aan_rows = s->d_aan_rows.ptr;
    P.la_rows = s->d_la_rows.ptr;
    P.n_la = (int32_t)f.la_rows.size();
    P.other_rows = s->d_other_rows.ptr;
    P.n_aan[0] = f.n_aan[0]; P.n_aan[1] = f.n_aan[1]; P.n_aan[2] = f.n_aan[2];
    P.n_other = (int32_t)f.other_rows.size();
    P.root_color = f.root_color;
    P.root_albedo = f.root_albedo;
    P.root_has_albedo = f.root_has_albedo;

    P.cam_cx = cam.world.cx; P.cam_cy = cam.world.cy; P.cam_cz = cam.world.cz; P.cam_t = cam.world.t;
    P.yfov = 2.0f * atan2f(cam.sensor_size, 2.0f * cam.focal_length);  // mod.rs:248
    P.xfov = P.yfov * cam.aspect_ratio;                                // mod.rs:249
    P.pixel_width = 2.0f * (1.0f / (float)width);                      // buffer.rs:68-71
    P.pixel_height = 2.0f * (1.0f / (float)height);                    // buffer.rs:73-76
    const uint32_t n = rc->subsample_n >= 2 ? rc->subsample_n : 1;     // mod.rs:47-67, main.rs:234-237
    const float subpixel_scale = rc->subsample_n >= 2 ? 1.0f / (float)rc->subsample_n : 1.0f;
    const float umin = -0.5f * P.pixel_width * subpixel_scale, umax = 0.5f * P.pixel_width * subpixel_scale;
    const float vmin = -0.5f * P.pixel_height * subpixel_scale, vmax = 0.5f * P.pixel_height * subpixel_scale;
    P.jitter_u_lo = umin;
    P.jitter_u_scale = bt::uniform_scale(umin, umax, false);           // mod.rs:255-259
    P.jitter_v_lo = vmin;
    P.jitter_v_scale = bt::uniform_scale(vmin, vmax, false);           // mod.rs:261-265
    P.has_focus = cam.has_focus ? 1 : 0;
    P.focus = cam.focus;
    P.aperture = 0.5f * cam.focal_length / cam.fstop;                  // mod.rs:289
    BtV3 neg_z; neg_z.x = 0.0f; neg_z.y = 0.0f; neg_z.z = -1.0f;      // UnitDisk::new(Vec3::NEG_Z), mod.rs:267
    bt::orthonormal_pair(neg_z, P.disk_x, P.disk_y);
    P.tau_scale = bt::uniform_scale(0.0f, 6.28318530717958647692f, true);
    P.one_scale = bt::uniform_scale(0.0f, 1.0f, true);

    output = rc->has_output ? rc->output : cfg->output;                // mod.rs:220
    if (output < 0 || output > 3) return set_error(BT_ERR_INVALID_ARG, "invalid output mode");
    P.max_bounces = (int32_t)(rc->has_max_bounces ? rc->max_bounces : cfg->max_bounces);                 // :223
    P.max_volume_bounces = (int32_t)(rc->has_max_bounces ? rc->max_bounces : cfg->max_volume_bounces);   // :224 (Q1)
    P.clip_min = cfg->clip_min;
    P.clip_max = cfg->clip_max;
    P.volume_step = rc->has_volume_step ? rc->volume_step : cfg->volume_step;                            // :227
    // The build for rect scenes without volumes walks the sorted tables with a division whose range handling is hoisted
    // out (bt_device.hpp div_refined): valid for 2^-30 <= clip_min, clip_max <= 2^60 and row ranks that fit 15 bits.  Any
    // other rect scene runs the generic loop, which lives in the rects + volumes build.
    if (P.any_rects && !P.any_volumes &&
        !(P.clip_min >= 0x1p-30f && P.clip_max <= 0x1p60f && f.prims.size() < 0x7fffu))
        P.any_volumes = 1;
    if (rc->samples > 0x7fffffffu / (n * n)) return set_error(BT_ERR_INVALID_ARG, "samples * n^2 overflows");
    P.samples = (int32_t)rc->samples;
    P.subsample_n = (int32_t)n;
    P.sample_base = rc->sample_base;
    P.seed_lo = (uint32_t)seed;
    P.seed_hi = (uint32_t)(seed >> 32);
    P.width = width;
    P.height = height;
    P.tiles_x = (width + BT_TILE_DIM - 1) / BT_TILE_DIM;
    P.tiles_x_magic = P.tiles_x == 1 ? 0xffffffffu : (uint32_t)(0x100000000ull / P.tiles_x);   // kernels: tile / tiles_x by umulhi + one fix-up
    P.tiles_y = (height + BT_TILE_DIM - 1) / BT_TILE_DIM;
    P.rank = 0;
    P.world = 1;
    P.sharded = 0;
    P.counters = s->d_counters;                 // (render_common points it at this render's slot)
    P.lens_on = s->lens_on ? 1 : 0;
    P.lens_c.x = s->lens.centre[0]; P.lens_c.y = s->lens.centre[1]; P.lens_c.z = s->lens.centre[2];
    P.lens_rs = s->lens.rs;
    P.lens_step = s->lens.step;
    P.lens_radius = s->lens.radius;
    P.lens_max_steps = (int32_t)s->lens.max_steps;
    P.lens_prims = nullptr;
    P.n_lens_prims = 0;
    P.lens_margin = 0.0f;
    if (s->lens_on) {
        // |v| <= sqrt(1 + rs h^2 / r^3) <= 2.8 along any geodesic that started with |v| = 1 (h^2 <= 6.75 rs^2 for the
        // captured ones, r >= rs), so a chord is at most ~2.8 steps long; longer ones (never seen) fall back to the
        // full table in the kernel
        P.lens_margin = 3.0f * s->lens.step;
        if (!s->lens_prims_valid || std::memcmp(&s->lens_prims_for, &s->lens, sizeof(bt_lens)) != 0) {
            BT_HIP(s->d_lens_prims.upload(lens_candidates(f.prims, s->lens, (double)s->lens.radius + (double)P.lens_margin)));
            s->lens_prims_for = s->lens;
            s->lens_prims_valid = true;
        }
        P.lens_prims = s->d_lens_prims.ptr;
        P.n_lens_prims = (int32_t)s->d_lens_prims.count;
    }
    return 0;
}

int render_common(bt_scene *s, uint64_t camera_ref, const bt_config *cfg, const bt_render_config *rc, float *out_device,
                  uint32_t width, uint32_t height, uint32_t rank, uint32_t world, bool sharded, uint64_t seed,
                  hipStream_t stream) {
    if (!s || !cfg || !rc || !out_device) return set_error(BT_ERR_INVALID_ARG, "null argument");
    if (rc->samples == 0) return BT_DONE;                              // mod.rs:186-188
    if (world == 0 || rank >= world) return set_error(BT_ERR_INVALID_ARG, "rank/world out of range");
    int rcode = ensure_device(s);
    if (rcode) return rcode;
    BtLaunch P;
    int output = 0;
    rcode = fill_launch(s, camera_ref, cfg, rc, width, height, seed, P, output);
    if (rcode) return rcode;
    P.rank = rank;
    P.world = world;
    P.sharded = sharded ? 1 : 0;
    P.out = out_device;
    const uint32_t n_tiles = P.tiles_x * P.tiles_y;
    const uint32_t grid = sharded ? (n_tiles + world - 1) / world : n_tiles;

    // Shape of the launch.  A workgroup owns a block of 256 / S pixels and deals their samples to its lanes,
    // every sample's value is parked in `scratch` (12 B per sample of the launch).  S is chosen so that a workgroup holds
    // ~16 samples per lane (4 with the lens on, whose paths differ far more in length; down to 4 as well when the launch has
    // too few pixels to fill the GPU).  A render whose scratch would exceed the cap is issued as several launches over
    // consecutive sample ranges (k launches of m samples == one launch of k * m samples).  bt_tuning
    // (bt_scene_set_tuning) pins any of these for tests and A/B tools.
    const bt_tuning &tune = s->tuning;
    const uint32_t nn = (uint32_t)(P.subsample_n * P.subsample_n);
    const uint64_t px_launch = (uint64_t)grid * BT_TILE_DIM * BT_TILE_DIM;
    uint32_t chunk = (uint32_t)P.samples;                         // samples per launch
    P.slices = 1;
    P.scratch = nullptr;
    P.table_lds_bytes = (uint32_t)s->flat.lds_bytes();
    size_t lds_bytes = s->flat.lds_bytes();
    {
        // scenes with volumes: the BtVolBox table behind the scene tables; density maps whose bounds tests cannot fire
        bool real_volumes = false, safe = true;
        for (const BtPrim &R : s->flat.prims) real_volumes = real_volumes || R.volume >= 0;
        for (const BtVolume &v : s->flat.volumes)
            safe = safe && v.width >= 1 && v.height >= 1 && v.depth >= 1 && v.size.x >= 0.0f && v.size.y >= 0.0f && v.size.z >= 0.0f &&
                   std::ceil(v.size.x) <= (float)(v.width - 1) && std::ceil(v.size.y) <= (float)(v.height - 1) &&
                   std::ceil(v.size.z) <= (float)(v.depth - 1);
        safe = safe && s->flat.density.size() < (1u << 24);      // density_sample_safe() indexes with 24-bit multiply-adds
        P.vols_safe = safe ? 1 : 0;
        P.vbox_lds_bytes = real_volumes ? (uint32_t)(sizeof(BtVolBox) * s->flat.prims.size()) : 0u;
        if (lds_bytes + P.vbox_lds_bytes > 32 * 1024) P.vbox_lds_bytes = 0;          // big scenes keep the per-step arithmetic
        lds_bytes += P.vbox_lds_bytes;
    }
#ifdef BT_LDS_PAD                                   // developer build: unused LDS per workgroup, to time lower occupancies
    lds_bytes += BT_LDS_PAD;
#endif
    // the launch should hold >= 4 x 20 waves per CU (tuned on the MI355X's 256 CUs as "4 * 5120 waves", round 1d)
    const uint64_t wave_slots = (uint64_t)s->n_cu * 20;
    auto ensure_scratch = [&](uint64_t need) -> bool {
        // Grow when too small.  Give memory back only after kScratchShrinkAfter consecutive renders that each needed less
        // than a quarter of what is held: a caller that alternates deep renders with shallow previews on one handle keeps
        // its scratch (no hipFree / hipMalloc -- a device-wide synchronisation -- per call); bt_scene_trim() returns it at once.
        if (s->scratch_bytes >= need) {
            if (s->scratch_bytes / 4 <= need) { s->scratch_small_streak = 0; return true; }
            if (++s->scratch_small_streak < kScratchShrinkAfter) return true;
        }
        s->scratch_small_streak = 0;
        if (s->d_scratch) {
            if (hipStreamSynchronize(stream) != hipSuccess) return false;   // an earlier launch on this stream may still read it
            (void)hipFree(s->d_scratch);
        }
        s->d_scratch = nullptr;
        s->scratch_bytes = 0;
        if (need == 0) return true;
        if (hipMalloc((void **)&s->d_scratch, need) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        s->scratch_bytes = need;
        return true;
    };
    const uint64_t T_all = (uint64_t)P.samples * nn;
    const uint64_t per_sample = px_launch * nn * 3 * sizeof(float);          // 12 B per parked sample value
    const uint64_t cap = tune.scratch_cap_bytes ? tune.scratch_cap_bytes : kDefaultScratchCap;
    if (per_sample * chunk > cap) chunk = (uint32_t)std::max<uint64_t>(1, cap / per_sample);
    // the parked values need device memory; when it cannot be had, render fewer samples per launch (there is no path that
    // does without: a lane that owned a pixel and summed in a register lost every measurement and left in round 3)
    while (!ensure_scratch(per_sample * chunk)) {
        if (chunk == 1) return set_error(BT_ERR_DEVICE, "no device memory for the parked samples (" + std::to_string(per_sample) + " bytes per sample)");
        chunk = (chunk + 1) / 2;
    }
    P.scratch = s->d_scratch;
    auto pick = [&](uint64_t T) -> uint32_t {
        if (tune.slices) return tune.slices;
        uint32_t S = 1;
        if (P.lens_on) {
            while (S < 32 && T / (2 * S) >= 4) S *= 2;       // lens paths differ far more in length: ~4 samples per lane
        } else {
            // Measured on 1080p and 512 x 512 frames, T = 1 ... 128 rays per pixel per launch, and on the shards of 2 / 4 / 8
            // ranks with 128 / 256 / 512 rays (profiles/r02z/time_shallow_before.log, time_shallow_512_before.log, time_shard.log).
            // A launch wants ~21 rounds of workgroups over the GPU (tiles x S ~ 32 000 on 256 CUs: S = 4 for a full 1080p
            // frame, 8 / 16 / 32 for the shards) with >= 8 samples per lane; below half of that, 4 samples per lane are
            // enough; and a frame that cannot even fill the wave slots twice is cut down to one sample per lane.
            const uint64_t target = 21ull * (uint64_t)s->n_cu * 6;
            while (S < 32 && (uint64_t)grid * (2 * S) * 4 <= 5 * target && T / (2 * S) >= 8) S *= 2;
            while (S < 32 && (uint64_t)grid * S * 2 < target && T / (2 * S) >= 4) S *= 2;
            while (S < 32 && (uint64_t)grid * 4 * S < 2 * wave_slots && T / (2 * S) >= 1) S *= 2;
            // Sphere-only scenes (cheaper items, eight waves per SIMD) want more, smaller workgroups on small frames than the rect
            // builds: up to ~7 500 of them while a lane still gets a whole item (profiles/r04j: scene.json 768 x 512 with the
            // reference CLI's 1 sample x Subpixel(2): S = 2 -> 4, 0.129 -> 0.119 ms; 8 rays: 0.197 -> 0.167; 1280 x 720 x 4:
            // 0.212 -> 0.184; the Cornell boxes lose with the same change)
            if (!P.any_rects)
                while (S < 32 && (uint64_t)grid * (2 * S) <= 30ull * (uint64_t)s->n_cu && 256 * T / (2 * S) >= 256) S *= 2;
        }
        return S;
    };
    P.slices = (int32_t)pick((uint64_t)chunk * nn);
    // a workgroup counts its path segments in 32 bits (bt_stats.segments): keep its work items x the longest possible path
    // below 2^32 -- only a pinned launch shape (bt_tuning.slices with an enormous scratch cap) can get near
    const uint64_t longest = ((uint64_t)P.max_bounces + 2) * ((uint64_t)P.max_volume_bounces + 3) + (P.lens_on ? 2 : 0);
    const uint64_t items_max = std::max<uint64_t>(1, 0xffffffffull / longest);
    {
        const uint64_t pxb = 256u / (uint32_t)P.slices;
        if (pxb * chunk * nn > items_max) chunk = (uint32_t)std::max<uint64_t>(1, items_max / (pxb * nn));
    }
    // Packed launch: when the whole render is one launch of at most a few generations of workgroups, ONE generation -- a workgroup
    // per workgroup slot of the GPU, each owning every n_workgroups-th small pixel block behind one queue -- ends with one drain
    // of its longest paths instead of one per generation.
    P.wg_blocks = 1;
    P.wg_blocks_rem = 0;
    P.n_workgroups = 0;
    P.log_rows = 0;
    P.row_mask = 0xffffffffu;
    P.pool_records = 0;
    P.pool_lds_offset = 0;
    // The drain of a packed launch compacts the paths in flight through LDS records (bt_kernels.hip PathRec, 80 B): room for 128
    // behind the tables, where that does not cost a workgroup slot and the packed record fields are wide enough.
    const size_t pool_offset = (lds_bytes + 15) & ~(size_t)15, pool_bytes = BT_POOL_RECORDS * 80;
    bool pool_ok = tune.packed != 1 && P.any_rects && !P.any_volumes && output == 0 &&   // (the Full-output rect build is the one that has the code)
                   P.max_bounces < 0xfff0 && P.max_volume_bounces < 0xfff0 && s->flat.prims.size() < 0xfffff0u;
    // workgroup slots of the GPU: 7 per CU by the builds' __launch_bounds__ (72 VGPRs), fewer when the scene tables are large
    // (160 KB of LDS per CU, allocated in 2 KB steps here to stay on the safe side)
    auto slots_per_cu = [](size_t lds) { return std::max(1u, std::min(7u, 160u * 1024u / (uint32_t)((lds + 64 + 2047) & ~(size_t)2047))); };
    pool_ok = pool_ok && slots_per_cu(pool_offset + pool_bytes) == slots_per_cu(lds_bytes);
    const uint32_t wg_slots = (uint32_t)s->n_cu * slots_per_cu(lds_bytes);
    const uint64_t T_launch = (uint64_t)chunk * nn;
    // Measured (profiles/r04t: 256 x 256 ... 1920 x 1080 frames, 1 ... 64 rays per pixel, three scene classes): packing pays
    // between ~1 and ~24 work items per lane of the GPU (scene.json 768 x 512 x 4: 0.121 -> 0.09 ms); deeper launches overlap
    // their drains with other workgroups' work and lose 5 - 20 % when packed, emptier ones do not fill the slots
    const uint64_t items_all = px_launch * T_launch, lanes_all = (uint64_t)wg_slots * 256;
    const bool pack = tune.packed > 0 || (tune.packed < 0 && items_all > lanes_all && items_all <= 24 * lanes_all);
    if (pack && chunk == (uint32_t)P.samples && !P.lens_on) {                     // (the lens extension has no packed builds)
        // blocks of ~64 / T pixels: one wave's take from the queue is one block's samples (coherent camera rays)
        uint32_t S = 4;
        while (S < 32 && S < 4 * T_launch) S *= 2;
        if (tune.slices) S = tune.slices;
        const uint64_t blocks = (uint64_t)grid * S, per_wg = (blocks + wg_slots - 1) / wg_slots;
        uint32_t log_rows = 0;                                   // T padded to a power of two: rows of a block in the queue
        while ((1ull << log_rows) < T_launch) log_rows += 1;
        const uint64_t wg_items = (per_wg * (256u / S)) << log_rows;
        const uint64_t need = (uint64_t)wg_slots * wg_items * 3 * sizeof(float);
        if (blocks > wg_slots && blocks <= 0x7fffffffu && wg_items <= items_max &&
            (need <= s->scratch_bytes || ensure_scratch(need))) {
            P.slices = (int32_t)S;
            P.n_workgroups = wg_slots;
            P.wg_blocks = (uint32_t)per_wg;
            P.wg_blocks_rem = (uint32_t)(blocks - (per_wg - 1) * wg_slots);
            P.log_rows = log_rows;
            P.row_mask = (1u << log_rows) - 1u;
            if (pool_ok) {
                P.pool_records = BT_POOL_RECORDS;
                P.pool_lds_offset = (uint32_t)pool_offset;
                lds_bytes = pool_offset + pool_bytes;
            }
        } else if (!s->d_scratch && !ensure_scratch(per_sample * chunk)) {
            return set_error(BT_ERR_DEVICE, "no device memory for the parked samples");
        }
        P.scratch = s->d_scratch;
    }
    const uint64_t parked_bytes = px_launch * T_all * 3 * sizeof(float);

    if (lds_bytes > 158 * 1024)
        return set_error(BT_ERR_INVALID_ARG, "scene tables (" + std::to_string(s->flat.lds_bytes()) +
                                                 " bytes) exceed the 160 KB of LDS of a gfx950 CU");
    // longest wait in iterations (0 = no voting); measured best: 3 on scene.json, 4 on the volume scenes
    // (profiles/r01f/ab_phase_vote.log, profiles/r01g/ab_vote_both.log)
    P.phase_vote = tune.phase_vote >= 0 ? tune.phase_vote : (P.any_volumes ? 4 : 3);
    uint32_t launches = 0;
    // work counters: a ring of slots, zeroed all at once when the ring wraps -- the interactive loop (main.rs:245-254, one
    // render per displayed frame) then pays one memset per 64 frames instead of one per frame in front of a 0.1 ms kernel
    s->last_slot = s->render_seq % kCounterSlots;
    if (s->last_slot == 0) BT_HIP(hipMemsetAsync(s->d_counters, 0, kCounterSlots * 16 * sizeof(unsigned long long), stream));
    s->render_seq += 1;
    P.counters = s->d_counters + (size_t)s->last_slot * 16;
    BT_HIP(hipEventRecord(s->ev_start, stream));
    {
        const uint32_t all = (uint32_t)P.samples, base = P.sample_base;
        for (uint32_t done = 0; done < all; done += chunk) {
            P.samples = (int32_t)std::min(chunk, all - done);
            P.sample_base = base + done;
            BT_HIP(bt_launch_render(&P, output, grid, lds_bytes, stream));
            launches += 1;
        }
        P.samples = (int32_t)all;
        P.sample_base = base;
    }
    BT_HIP(hipEventRecord(s->ev_stop, stream));

    // pixels actually owned by this rank
    uint64_t pixels = 0;
    for (uint32_t t = rank; t < n_tiles; t += world) {
        uint32_t tx = t % P.tiles_x, ty = t / P.tiles_x;
        uint32_t w = std::min<uint32_t>(BT_TILE_DIM, width - tx * BT_TILE_DIM);
        uint32_t h = std::min<uint32_t>(BT_TILE_DIM, height - ty * BT_TILE_DIM);
        pixels += (uint64_t)w * h;
    }
    s->last.pixels = pixels;
    s->last.samples = pixels * (uint64_t)P.samples * (uint64_t)(P.subsample_n * P.subsample_n);
    s->last.segments = 0;
    s->last.kernel_ms = 0.0f;
    s->last.slices = (uint32_t)P.slices;
    s->last.launches = launches;
    s->last.packed = P.wg_blocks > 1 ? (P.pool_records ? 2u : 1u) : 0u;
    s->last.workgroups = P.wg_blocks > 1 ? P.n_workgroups : grid * (uint32_t)P.slices;
    s->last.scratch_bytes = s->scratch_bytes;
    s->last.parked_bytes = parked_bytes;
    s->stats_pending = true;
    return BT_IN_PROGRESS;                                             // mod.rs:201
}

} // namespace

extern "C" {

void bt_config_default(bt_config *c) {
    if (!c) return;
    c->max_bounces = 8;
    c->max_volume_bounces = 32;
    c->clip_min = 0.01f;
    c->clip_max = 1000.0f;
    c->volume_step = 0.1f;
    c->chunks_x = 4;
    c->chunks_y = 2;
    c->output = BT_OUTPUT_FULL;
}

void bt_render_config_default(bt_render_config *r) {
    if (!r) return;
    std::memset(r, 0, sizeof *r);
    r->samples = 64;
}

const char *bt_last_error(void) { return g_error.c_str(); }
int bt_set_error_internal(int code, const char *msg) { return set_error(code, msg ? msg : ""); }   // for bt_comm.cpp; not in the header
int bt_last_error_code(void) { return g_error_code; }

void bt_tuning_default(bt_tuning *t) {
    if (!t) return;
    std::memset(t, 0, sizeof *t);
    t->phase_vote = -1;
    t->packed = -1;
}

int bt_scene_set_tuning(bt_scene *scene, const bt_tuning *t) {
    if (!scene) return set_error(BT_ERR_INVALID_ARG, "null scene");
    if (!t) {
        bt_tuning_default(&scene->tuning);
        return 0;
    }
    const uint32_t S = t->slices;
    if (!(S == 0 || S == 1 || S == 2 || S == 4 || S == 8 || S == 16 || S == 32))
        return set_error(BT_ERR_INVALID_ARG, "bt_tuning.slices must be 0 (auto), 1, 2, 4, 8, 16 or 32");
    if (t->phase_vote < -1 || t->phase_vote > 64)
        return set_error(BT_ERR_INVALID_ARG, "bt_tuning.phase_vote must be -1 .. 64");
    if (t->packed < -1 || t->packed > 2) return set_error(BT_ERR_INVALID_ARG, "bt_tuning.packed must be -1, 0, 1 or 2");
    scene->tuning = *t;
    return 0;
}

int bt_scene_get_tuning(const bt_scene *scene, bt_tuning *out) {
    if (!scene || !out) return set_error(BT_ERR_INVALID_ARG, "null argument");
    *out = scene->tuning;
    return 0;
}

const char *bt_version(void) { return "bendy-hip 0.1 (gfx950)"; }

bt_scene *bt_scene_from_json(const char *json, size_t len) {
    if (!json) {
        set_error(BT_ERR_INVALID_ARG, "null json");
        return nullptr;
    }
    try {
        std::unique_ptr<bt_scene> s(new bt_scene());
        s->scene = bt::parse_scene(json, len);
        s->source.assign(json, len);
        return s.release();
    } catch (const bt::Error &e) {
        set_error(e.code, e.message);
    } catch (const std::exception &e) {
        set_error(BT_ERR_PARSE, e.what());
    }
    return nullptr;
}

bt_scene *bt_scene_load(const char *path) {
    if (!path) {
        set_error(BT_ERR_INVALID_ARG, "null path");
        return nullptr;
    }
    try {
        std::string text = bt::read_scene_file(path);
        return bt_scene_from_json(text.data(), text.size());
    } catch (const bt::Error &e) {
        set_error(e.code, e.message);
    } catch (const std::exception &e) {
        set_error(BT_ERR_IO, e.what());
    }
    return nullptr;
}

bt_scene *bt_scene_default(void) {
    std::string text = bt::default_scene_json();
    return bt_scene_from_json(text.data(), text.size());
}

void bt_scene_free(bt_scene *scene) { delete scene; }

int bt_scene_to_json(const bt_scene *scene, char *out, size_t cap) {
    if (!scene) return set_error(BT_ERR_INVALID_ARG, "null scene");
    try {
        std::string text = bt::scene_to_pretty_json(scene->scene, scene->source);
        if (out && cap > 0) {
            size_t n = std::min(cap - 1, text.size());
            std::memcpy(out, text.data(), n);
            out[n] = 0;
        }
        return (int)text.size();
    } catch (const bt::Error &e) {
        return set_error(e.code, e.message);
    } catch (const std::exception &e) {
        return set_error(BT_ERR_PARSE, e.what());
    }
}

int bt_scene_save(const bt_scene *scene, const char *path) {
    if (!scene || !path) return set_error(BT_ERR_INVALID_ARG, "null argument");
    try {
        bt::write_text_file(path, bt::scene_to_pretty_json(scene->scene, scene->source));
        return 0;
    } catch (const bt::Error &e) {
        return set_error(e.code, e.message);
    } catch (const std::exception &e) {
        return set_error(BT_ERR_IO, e.what());
    }
}

int bt_write_png(const char *path, const uint8_t *rgba8, uint32_t width, uint32_t height) {
    if (!path || !rgba8 || width == 0 || height == 0) return set_error(BT_ERR_INVALID_ARG, "invalid argument");
    try {
        bt::write_png(path, rgba8, width, height);
        return 0;
    } catch (const bt::Error &e) {
        return set_error(e.code, e.message);
    }
}

int bt_scene_find_by_tag(const bt_scene *scene, const char *tag, uint64_t *object_ref) {
    if (!scene || !tag || !object_ref) return set_error(BT_ERR_INVALID_ARG, "null argument");
    for (const bt::Object &o : scene->scene.objects)
        if (o.has_tag && o.tag == tag) {
            *object_ref = o.object_ref;
            return 0;
        }
    return set_error(BT_ERR_INVALID_REF, std::string("no object tagged `") + tag + "`");
}

int bt_scene_set_camera_aspect(bt_scene *scene, uint64_t camera_ref, float aspect_ratio) {
    if (!scene) return set_error(BT_ERR_INVALID_ARG, "null scene");
    int i = scene->scene.object_index(camera_ref);
    if (i < 0) return set_error(BT_ERR_INVALID_REF, "invalid object ref " + std::to_string(camera_ref));
    if (scene->scene.objects[i].kind != bt::OBJ_CAMERA) return set_error(BT_ERR_NOT_CAMERA, "expected a camera object");
    scene->scene.objects[i].aspect_ratio = aspect_ratio;   // read at launch time; device tables unaffected
    return 0;
}

int bt_scene_set_lens(bt_scene *scene, const bt_lens *lens) {
    if (!scene) return set_error(BT_ERR_INVALID_ARG, "null scene");
    if (!lens) {
        scene->lens_on = false;
        return 0;
    }
    if (!(lens->rs >= 0.0f) || !(lens->step > 0.0f) || !(lens->radius > lens->rs) || lens->max_steps == 0 ||
        lens->max_steps > 0x7fffffffu)
        return set_error(BT_ERR_INVALID_ARG, "lens needs rs >= 0, step > 0, radius > rs, max_steps > 0");
    scene->lens = *lens;
    scene->lens_on = true;
    return 0;
}

int bt_scene_object_count(const bt_scene *scene) { return scene ? (int)scene->scene.objects.size() : 0; }
int bt_scene_data_count(const bt_scene *scene) { return scene ? (int)scene->scene.data.size() : 0; }

int bt_scene_export_prims(const bt_scene *scene, float *out, int cap) {
    if (!scene) return set_error(BT_ERR_INVALID_ARG, "null scene");
    bt_scene *s = const_cast<bt_scene *>(scene);
    int rc = ensure_flat(s);
    if (rc) return rc;
    const size_t words = sizeof(BtPrim) / 4;
    const int total = (int)(s->flat.prims.size() * words);
    if (out && cap > 0) std::memcpy(out, s->flat.prims.data(), sizeof(float) * (size_t)std::min(cap, total));
    return total;
}

int bt_render_device(bt_scene *scene, uint64_t camera_ref, const bt_config *config, const bt_render_config *render,
                     float *rgba_device, uint32_t width, uint32_t height, uint64_t seed, void *stream) {
    return render_common(scene, camera_ref, config, render, rgba_device, width, height, 0, 1, false, seed,
                         (hipStream_t)stream);
}

int bt_render(bt_scene *scene, uint64_t camera_ref, const bt_config *config, const bt_render_config *render,
              float *rgba_host, uint32_t width, uint32_t height, uint64_t seed) {
    if (!scene) return set_error(BT_ERR_INVALID_ARG, "null scene");
    if (!rgba_host) return set_error(BT_ERR_INVALID_ARG, "null buffer");
    if (render && render->samples == 0) return BT_DONE;
    if (width == 0 || height == 0) return set_error(BT_ERR_INVALID_ARG, "zero-sized buffer");
    int rc = ensure_device(scene);                        // binds the handle (and its cached frame) to the current device
    if (rc) return rc;
    const size_t bytes = (size_t)width * height * 4 * sizeof(float);
    // the device copy of the caller's buffer lives on the handle: no hipMalloc / hipFree per displayed frame
    if (scene->host_frame_bytes < bytes || scene->host_frame_bytes / 4 > bytes) {
        if (scene->d_host_frame) {
            BT_HIP(hipDeviceSynchronize());
            (void)hipFree(scene->d_host_frame);
        }
        scene->d_host_frame = nullptr;
        scene->host_frame_bytes = 0;
        BT_HIP(hipMalloc((void **)&scene->d_host_frame, bytes));
        scene->host_frame_bytes = bytes;
    }
    float *d = scene->d_host_frame;
    BT_HIP(hipMemcpyAsync(d, rgba_host, bytes, hipMemcpyHostToDevice, nullptr));
    rc = bt_render_device(scene, camera_ref, config, render, d, width, height, seed, nullptr);
    if (rc < 0) return rc;
    BT_HIP(hipMemcpy(rgba_host, d, bytes, hipMemcpyDeviceToHost));      // stream-ordered behind the kernel, blocks the host
    return rc;
}

size_t bt_shard_floats(uint32_t width, uint32_t height, uint32_t world) {
    if (world == 0) return 0;
    size_t tiles = (size_t)((width + BT_TILE - 1) / BT_TILE) * ((height + BT_TILE - 1) / BT_TILE);
    size_t per_rank = (tiles + world - 1) / world;
    return per_rank * BT_TILE * BT_TILE * 4;
}

int bt_render_shard_device(bt_scene *scene, uint64_t camera_ref, const bt_config *config,
                           const bt_render_config *render, float *shard_device, uint32_t width, uint32_t height,
                           uint32_t rank, uint32_t world, uint64_t seed, void *stream) {
    return render_common(scene, camera_ref, config, render, shard_device, width, height, rank, world, true, seed,
                         (hipStream_t)stream);
}

int bt_unshard_device(const float *gathered_device, float *rgba_device, uint32_t width, uint32_t height, uint32_t world,
                      void *stream) {
    if (!gathered_device || !rgba_device || world == 0 || width == 0 || height == 0)
        return set_error(BT_ERR_INVALID_ARG, "invalid argument");
    uint32_t tiles_x = (width + BT_TILE - 1) / BT_TILE, tiles_y = (height + BT_TILE - 1) / BT_TILE;
    uint32_t per_rank = (tiles_x * tiles_y + world - 1) / world;
    BT_HIP(bt_launch_unshard(gathered_device, rgba_device, width, height, tiles_x, tiles_y, world, per_rank,
                             (hipStream_t)stream));
    return 0;
}

int bt_preview_device(const float *rgba_device, uint8_t *rgba8_device, uint32_t width, uint32_t height, uint32_t samples,
                      int32_t color_space, void *stream) {
    if (!rgba_device || !rgba8_device || width == 0 || height == 0)
        return set_error(BT_ERR_INVALID_ARG, "invalid argument");
    BT_HIP(bt_launch_preview(rgba_device, rgba8_device, width * height, samples, color_space, (hipStream_t)stream));
    return 0;
}

int bt_preview(const float *rgba_host, uint8_t *rgba8_host, uint32_t width, uint32_t height, uint32_t samples,
               int32_t color_space) {
    if (!rgba_host || !rgba8_host || width == 0 || height == 0) return set_error(BT_ERR_INVALID_ARG, "invalid argument");
    const size_t n = (size_t)width * height;
    float *d_in = nullptr;
    uint8_t *d_out = nullptr;
    BT_HIP(hipMalloc((void **)&d_in, n * 16));
    hipError_t e = hipMalloc((void **)&d_out, n * 4);
    int rc = 0;
    if (e == hipSuccess) e = hipMemcpy(d_in, rgba_host, n * 16, hipMemcpyHostToDevice);
    if (e == hipSuccess) rc = bt_preview_device(d_in, d_out, width, height, samples, color_space, nullptr);
    if (e == hipSuccess && rc == 0) e = hipMemcpy(rgba8_host, d_out, n * 4, hipMemcpyDeviceToHost);
    (void)hipFree(d_in);
    if (d_out) (void)hipFree(d_out);
    if (e != hipSuccess) return set_error(BT_ERR_DEVICE, hipGetErrorString(e));
    return rc;
}

int bt_scene_trim(bt_scene *scene) {
    if (!scene) return set_error(BT_ERR_INVALID_ARG, "null scene");
    if (scene->device < 0 || (!scene->d_scratch && !scene->d_host_frame)) return 0;
    int cur = -1;
    BT_HIP(hipGetDevice(&cur));
    if (cur != scene->device) BT_HIP(hipSetDevice(scene->device));
    BT_HIP(hipDeviceSynchronize());                       // launches that still read the scratch / the cached frame
    if (scene->d_scratch) (void)hipFree(scene->d_scratch);
    if (scene->d_host_frame) (void)hipFree(scene->d_host_frame);
    scene->d_scratch = nullptr;
    scene->d_host_frame = nullptr;
    scene->scratch_bytes = scene->host_frame_bytes = 0;
    scene->scratch_small_streak = 0;
    if (cur != scene->device) BT_HIP(hipSetDevice(cur));
    return 0;
}

int bt_scene_last_stats(bt_scene *scene, bt_stats *out) {
    if (!scene || !out) return set_error(BT_ERR_INVALID_ARG, "null argument");
    if (scene->stats_pending) {
        BT_HIP(hipEventSynchronize(scene->ev_stop));
        unsigned long long c[16] = {0, 0};
        BT_HIP(hipMemcpy(c, scene->d_counters + (size_t)scene->last_slot * 16, sizeof c, hipMemcpyDeviceToHost));
#ifdef BT_LANESTAT
        // developer build (-DBT_LANESTAT): what the lanes of a wave do per iteration, see bt_kernels.hip
        if (c[2]) {
            const double it = (double)c[2] * 64.0;
            fprintf(stderr, "[bt lanes] wave-iterations %llu; of 64 lanes per iteration: trace %.1f%% | camera %.1f%% diffuse %.1f%% metallic %.1f%% glass %.1f%% volume %.1f%% | waiting for phase %.1f%% | left the loop %.1f%%\n",
                    c[2], 100.0 * c[3] / it, 100.0 * c[4] / it, 100.0 * c[5] / it, 100.0 * c[6] / it, 100.0 * c[7] / it,
                    100.0 * c[8] / it, 100.0 * c[9] / it, 100.0 * c[10] / it);
        }
#elif defined(BT_PROFILE)
        // developer build (-DBT_PROFILE): wave cycles per section of the render loop, see bt_kernels.hip
        {
            unsigned long long tot = 0;
            for (int i = 2; i < BT_N_COUNTERS; ++i) tot += c[i];
            fprintf(stderr, "[bt profile] section share of wave cycles:");
            for (int i = 2; i < BT_N_COUNTERS; ++i) fprintf(stderr, " s%d=%.1f%%", i - 2, tot ? 100.0 * (double)c[i] / (double)tot : 0.0);
            fprintf(stderr, "\n");
        }
#endif
        float ms = 0.0f;
        BT_HIP(hipEventElapsedTime(&ms, scene->ev_start, scene->ev_stop));
        scene->last.segments = c[0];
        scene->last.lens_steps = c[1];
        scene->last.kernel_ms = ms;
        scene->stats_pending = false;
    }
    *out = scene->last;
    return 0;
}

} // extern "C"
