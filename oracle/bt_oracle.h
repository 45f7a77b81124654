/*
 * bt_oracle.h -- CPU ORACLE for the bendy-tracer hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the reference's per-pixel / per-sample path
 * (soycan-sim/bendy-tracer @ v1: src/tracer/mod.rs, src/tracer/ray.rs,
 * src/scene/object/{sphere,rect,cuboid}.rs, src/scene/data/{material,volume}.rs,
 * src/math/{mod,distr}.rs).  Every function cites the reference lines it follows.
 *
 * PARITY STATUS: "parity unpinned" against the Rust binary.  The reference has no
 * golden vectors, no asserting tests, cannot be built here (no cargo/rustc), and
 * seeds its RNG from OS entropy (src/tracer/mod.rs:239-242), so no pixel-exact
 * comparison with it is definable.  This oracle is pinned instead by analytic
 * known-answer tests derived from the reference source (tests/test_oracle_kat.py)
 * and by the numerics contract in DESIGN.md (counter-based Philox RNG, own
 * sin/cos, explicit operation order).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  The product (bendy_tracer_amd/) never links or calls it.
 */
#ifndef BT_ORACLE_H
#define BT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float x, y, z; } bto_v3;

/* glam::Affine3A: matrix3 columns + translation (12 floats, column-major). */
typedef struct { bto_v3 cx, cy, cz, t; } bto_affine;

enum { BTO_EMPTY = 0, BTO_CAMERA = 1, BTO_SPHERE = 2, BTO_RECT = 3, BTO_CUBOID = 4 };
enum { BTO_FLAT = 0, BTO_DIFFUSE = 1, BTO_METALLIC = 2, BTO_GLASS = 3, BTO_EMISSIVE = 4, BTO_VOLUME = 5 };
enum { BTO_OUT_FULL = 0, BTO_OUT_ALBEDO = 1, BTO_OUT_NORMAL = 2, BTO_OUT_DEPTH = 3 };
enum { BTO_FLAG_LIGHT = 1 };

/* scene/object/rect.rs:11-19 */
typedef struct {
    int32_t material;          /* index into data[] */
    float half_width, half_height;
    bto_v3 x, y, z;
} bto_rect;

/* scene/object/mod.rs:33-41 + the ObjectKind payloads (camera.rs, sphere.rs, rect.rs, cuboid.rs) */
typedef struct {
    uint64_t object_ref;
    int32_t kind;
    uint32_t flags;
    bto_affine world;          /* transform_world only (object/mod.rs:118-120) */
    /* Camera (camera.rs:3-10) */
    float sensor_size, focal_length, aspect_ratio, fstop, focus;
    int32_t has_focus;
    /* Sphere (sphere.rs:11-16) */
    int32_t material;          /* index into data[] */
    int32_t volume;            /* index into data[] or -1 */
    float radius;
    /* Rect */
    bto_rect rect;
    /* Cuboid (cuboid.rs:12-15): faces: [(offset, Rect); 6] */
    bto_v3 face_offset[6];
    bto_rect faces[6];
} bto_object;

/* scene/data/material.rs:22-44, volume.rs:75-82 */
typedef struct {
    uint64_t data_ref;
    int32_t kind;
    float albedo[3];
    float roughness, ior, intensity;
    int32_t width, height, depth;
    float size[3];
    int64_t buffer_offset;     /* into density[] */
} bto_data;

typedef struct {
    int32_t n_objects;         /* ascending object_ref (iteration order) */
    int32_t n_data;
    const bto_object *objects;
    const bto_data *data;
    const float *density;
    int32_t root_material;     /* index into data[] */
} bto_scene;

/* tracer/mod.rs:16-45 (Config) merged with :117-135 (RenderConfig) as in :217-229 */
typedef struct {
    int32_t max_bounces;
    int32_t max_volume_bounces;
    float clip_min, clip_max, volume_step;
    int32_t chunks_x, chunks_y;
    int32_t output;
    int32_t samples;
    int32_t subsample_n;       /* 0/1 = Subsample::None, n>=2 = Subpixel(n) */
    uint32_t sample_base;      /* index of this call's first sample (progressive calls) */
    int32_t recursive;         /* 1 = recursive evaluation exactly as the reference nests it;
                                  0 = iterative throughput form */
    /* EXTENSION, NOT IN THE REFERENCE ("parity unpinned", default off): a point mass
     * bends every non-marching path segment.  Inside the sphere of influence the photon is stepped with
     * fixed-step RK4 along the Schwarzschild null geodesic (x'' = -1.5 rs h^2 x / r^5, h = |x x v|); each
     * step's chord is intersected like a volume-march step.  Outside the sphere rays are straight. */
    int32_t lens_on;
    float lens_centre[3];
    float lens_rs;             /* Schwarzschild radius, scene units */
    float lens_step;           /* RK4 step (affine length) */
    float lens_radius;         /* sphere of influence */
    int32_t lens_max_steps;    /* RK4 steps per path segment before the segment is abandoned (treated as a miss) */
} bto_config;

/* Lens extension, free space: integrates one ray and reports where it ends.  Returns 0 escaped (out = final
 * position, direction), 1 captured by the horizon, 2 step / length budget exhausted. */
int bto_lens_trace_free(const bto_config *cfg, const float *origin3, const float *dir3, float *pos_out3, float *dir_out3);

/* Render `samples * n^2` rays per pixel and ADD them into rgba (row-major,
 * 4 floats per pixel, alpha untouched): Tracer::render, tracer/mod.rs:179-202.
 * Returns 0 = Done (samples == 0), 1 = InProgress, <0 error.
 * nthreads <= 1: single thread.  Otherwise the chunks_x*chunks_y tiles of
 * Buffer::chunks (buffer.rs:102-115) are handed to nthreads workers.
 * segments_out (optional): number of path segments traced. */
int bto_render(const bto_scene *scene, int32_t camera_index, const bto_config *cfg,
               float *rgba, uint32_t width, uint32_t height, uint64_t seed,
               int32_t nthreads, uint64_t *segments_out);

/* Trace ONE sample of ONE pixel; out = color(3) albedo(3) normal(3) depth(1). */
int bto_trace_one(const bto_scene *scene, int32_t camera_index, const bto_config *cfg,
                  uint32_t width, uint32_t height, uint32_t px, uint32_t py,
                  uint32_t sample_index, uint64_t seed, float *out10);

/* --- building blocks exported for the known-answer tests --- */
void bto_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
void bto_sincos(float x, float *s, float *c);
float bto_uniform_scale(float lo, float hi, int inclusive);
void bto_ray_with_frustum(float yfov, float xfov, float u, float v, float *dir3);
/* returns face (0 Front,1 Back,2 Volume,3 VolumeFront,4 VolumeBack) or -1 for a miss */
int bto_object_hit(const bto_scene *scene, int32_t object_index, const float *origin3,
                   const float *dir3, float clip_min, float clip_max, int volumetric,
                   float *t_out, float *pos3, float *normal3);
float bto_object_pdf(const bto_scene *scene, int32_t object_index, const float *origin3,
                     const float *dir3, float clip_min, float clip_max);
void bto_reflect(const float *v3, const float *n3, float *out3);
void bto_refract(const float *v3, const float *n3, float ior, float *out3);
float bto_fresnel(const float *v3, const float *n3, float ior);
float bto_density_sample(const bto_scene *scene, int32_t data_index, const float *coord3);
void bto_orthonormal_pair(const float *n3, float *t1, float *t2);
void bto_affine_inverse(const bto_affine *a, bto_affine *out);
/* resolve: mean -> colour space -> u8 (buffer.rs:117-138, color.rs:14-24) */
void bto_preview(const float *rgba, uint32_t n_pixels, uint32_t samples, int32_t color_space,
                 uint8_t *out_rgba8);
void bto_chunk_bounds(uint32_t width, uint32_t height, int32_t chunks_x, int32_t chunks_y,
                      int32_t *n_chunks, uint32_t *bounds /* [n][4] = minx,miny,maxx,maxy */);

#ifdef __cplusplus
}
#endif
#endif
