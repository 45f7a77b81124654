/*
 * bendy_hip.h -- C ABI of libbendy_hip.so, the MI355X (gfx950) implementation of
 * bendy-tracer's per-pixel / per-sample hot path.
 *
 * The reference (soycan-sim/bendy-tracer @ v1) has no FFI surface; its boundary is
 * the Rust library API that src/main.rs uses.  Each entry point below names the
 * reference interface it replaces (paths relative to the reference tree).
 * INTEGRATION.md shows the Rust `extern "C"` binding a maintainer would add.
 *
 * Conventions: plain pointers and sizes only; no exceptions cross the ABI; every
 * function that can fail returns a negative bt_status and records a message that
 * bt_last_error() returns (thread-local).  A bt_scene handle is not thread-safe
 * for concurrent bt_render* calls (same as `&mut Buffer` in the reference).
 */
#ifndef BENDY_HIP_H
#define BENDY_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* tracer/mod.rs:159-163 `Status` (>= 0) and error codes (< 0; the reference panics instead). */
typedef enum {
    BT_DONE = 0,               /* Status::Done: samples == 0 (mod.rs:186-188) */
    BT_IN_PROGRESS = 1,        /* Status::InProgress (mod.rs:201) */
    BT_ERR_INVALID_ARG = -1,
    BT_ERR_IO = -2,            /* file missing / gzip error (main.rs:93-102) */
    BT_ERR_PARSE = -3,         /* malformed JSON / unexpected schema (serde error in the reference) */
    BT_ERR_INVALID_REF = -4,   /* "invalid object ref" / "invalid data ref" (scene/mod.rs:132,136) */
    BT_ERR_NOT_CAMERA = -5,    /* "expected a camera object" (tracer/mod.rs:246) */
    BT_ERR_NOT_MATERIAL = -6,  /* "expected material data" / "expected volume data" (mod.rs:464,499) */
    BT_ERR_NO_LIGHT = -7,      /* Diffuse material but no LIGHT object: Uniform::new(0,0) panics (material.rs:112) */
    BT_ERR_DEVICE = -8,        /* HIP runtime error / no gfx950 device */
    BT_ERR_UNSUPPORTED = -9
} bt_status;

/* tracer/mod.rs:108-115 `Output` */
typedef enum { BT_OUTPUT_FULL = 0, BT_OUTPUT_ALBEDO = 1, BT_OUTPUT_NORMAL = 2, BT_OUTPUT_DEPTH = 3 } bt_output;

/* tracer/buffer.rs:11-17 `ColorSpace` */
typedef enum { BT_COLOR_NONE = 0, BT_COLOR_NORMAL = 1, BT_COLOR_LINEAR = 2, BT_COLOR_SRGB = 3 } bt_color_space;

/* tracer/mod.rs:16-26 `Config`; bt_config_default() = Config::DEFAULT (:29-38). */
typedef struct {
    uint32_t max_bounces;
    uint32_t max_volume_bounces;
    float clip_min;
    float clip_max;
    float volume_step;
    uint32_t chunks_x;         /* kept for API fidelity; the GPU grid does its own tiling */
    uint32_t chunks_y;
    int32_t output;            /* bt_output */
} bt_config;

/* tracer/mod.rs:117-125 `RenderConfig` (Option<T> -> has_* flag + value);
 * bt_render_config_default() = RenderConfig::DEFAULT (:128-135). */
typedef struct {
    uint32_t subsample_n;      /* Subsample: 0 or 1 = None, n >= 2 = Subpixel(n) (:47-52, main.rs:234-237) */
    uint32_t samples;
    int32_t has_output;
    int32_t output;
    int32_t has_max_bounces;
    uint32_t max_bounces;
    int32_t has_max_volume_bounces;
    uint32_t max_volume_bounces; /* accepted but ignored, exactly like the reference (quirk Q1, mod.rs:224) */
    int32_t has_volume_step;
    float volume_step;
    /* Not in the reference (it seeds from OS entropy, mod.rs:239-242): index of this
     * call's first sample, so that k progressive calls of 1 sample equal one call of
     * k samples bit for bit.  Callers normally pass Buffer::samples() / n^2. */
    uint32_t sample_base;
} bt_render_config;

/* Work counters of the last render on a scene handle (for the roofline model). */
typedef struct {
    uint64_t samples;          /* rays per pixel * pixels rendered by this rank */
    uint64_t segments;         /* try_hit / try_hit_volume calls (mod.rs:389-427) */
    uint64_t pixels;
    float kernel_ms;           /* HIP-event time of the render kernel(s), 0 if not measured */
    uint64_t lens_steps;       /* RK4 steps taken by the lens extension (0 when it is off) */
    uint32_t slices;           /* S of the last launch: pixel blocks of 256/S pixels whose samples are dealt to the lanes of a
                                * workgroup through a queue; DESIGN.md 5.3 */
    uint32_t launches;         /* kernel launches the render was split into (deep renders under the scratch cap) */
    uint64_t scratch_bytes;    /* HBM the handle holds for parked sample values after this render */
    uint64_t parked_bytes;     /* bytes of sample values the render parked in HBM (12 per sample, edge tiles padded) */
    uint32_t workgroups;       /* workgroups of the last launch */
    uint32_t packed;           /* 1 / 2: the last launch was packed (bt_tuning.packed; 2 = with the compacting drain): `workgroups` = the
                                * GPU's workgroup slots, each owning every workgroups-th pixel block behind one queue */
} bt_stats;

/* Launch-shape knobs of a scene handle.  Every field's zero / negative value means "let the library decide" (what
 * bt_tuning_default() fills in); the library itself never reads environment variables.  Tests and the A/B tools under
 * tools/ set these to pin a shape; none of them can change a pixel (tests/test_gpu_parity.py renders every setting). */
typedef struct {
    uint32_t slices;           /* 0 = auto; 1, 2, 4, 8, 16, 32: pixel blocks of 256 / slices pixels */
    int32_t phase_vote;        /* -1 = auto; 0 = off; n = longest wait of the phase vote in iterations */
    uint64_t scratch_cap_bytes;/* 0 = the default 2 GiB: most parked sample values per launch; deeper renders are split into
                                * several launches over consecutive sample ranges */
    int32_t packed;            /* -1 = auto; 0 = one pixel block per workgroup; 1 / 2 = packed launch where one fits: one workgroup per
                                * workgroup slot of the GPU, each owning every n-th pixel block behind one queue --
                                * 2 (what auto uses): its drain compacts the paths in flight into fewer waves through LDS records */
    int32_t reserved;
} bt_tuning;

/* EXTENSION -- NOT IN THE REFERENCE.  bendy-tracer v1 traces straight rays only (`Ray::at` is
 * origin + t * direction, tracer/ray.rs:115-117); "gravitational lensing" exists in its README as an
 * aspiration.  This optional mode (off unless set) bends every non-marching path segment around one point
 * mass: inside `radius` the photon follows the Schwarzschild null geodesic, integrated with fixed-step RK4
 * (x'' = -1.5 rs h^2 x / r^5, h = |x x v|), and each step's chord is intersected like a volume-march step;
 * outside `radius` rays are straight; r <= rs swallows the path (black).  Light-sampling pdfs
 * (material.rs:313-316) still assume straight visibility.  There is no reference behaviour to match:
 * validation is analytic (weak-field deflection 2 rs / b, capture below b = 3 sqrt(3)/2 rs) plus GPU == CPU
 * oracle; with the lens unset every code path and every pixel is exactly what it is without this extension. */
typedef struct {
    float centre[3];
    float rs;                  /* Schwarzschild radius, scene units */
    float step;                /* RK4 step (affine length) */
    float radius;              /* sphere of influence */
    uint32_t max_steps;        /* RK4 steps per path segment before the segment is abandoned as a miss */
} bt_lens;

typedef struct bt_scene bt_scene; /* opaque; replaces `Scene` (scene/mod.rs:84-90) */

void bt_config_default(bt_config *out);                 /* Config::default(), mod.rs:41-45 */
void bt_render_config_default(bt_render_config *out);   /* RenderConfig::default(), mod.rs:153-157 */
const char *bt_last_error(void);
int bt_last_error_code(void);      /* bt_status of the last failure on this thread (for NULL-returning constructors) */
const char *bt_version(void);

/* --- Scene: serde_json::from_reader(GzDecoder) in main.rs:93-102 ------------------- */
/* `path` ends in .gz -> gzip, else plain JSON (main.rs:97-102).  NULL on error. */
bt_scene *bt_scene_load(const char *path);
/* serde_json::from_slice on an in-memory, already decompressed document. */
bt_scene *bt_scene_from_json(const char *json, size_t len);
/* The scene main.rs builds when the --scene file does not exist (main.rs:107-214). */
bt_scene *bt_scene_default(void);
void bt_scene_free(bt_scene *scene);
/* serde_json::to_writer_pretty(&scene) (main.rs:299-313): writes up to cap-1 bytes + NUL into `out`
 * (may be NULL) and returns the full length.  bt_scene_save gzips when `path` ends in .gz. */
int bt_scene_to_json(const bt_scene *scene, char *out, size_t cap);
int bt_scene_save(const bt_scene *scene, const char *path);
/* buffer.preview().save(path) (main.rs:275-298): RGBA8 PNG. */
int bt_write_png(const char *path, const uint8_t *rgba8, uint32_t width, uint32_t height);
/* Scene::find_by_tag (scene/mod.rs:124-129).  Writes the ObjectRef; returns 0, or
 * BT_ERR_INVALID_REF if no object carries the tag.  When several objects share a tag
 * the lowest ObjectRef wins (the reference's hash-map order is unspecified). */
int bt_scene_find_by_tag(const bt_scene *scene, const char *tag, uint64_t *object_ref);
/* object.as_camera_mut().unwrap().aspect_ratio = a (main.rs:218-223, quirk Q12). */
int bt_scene_set_camera_aspect(bt_scene *scene, uint64_t camera_ref, float aspect_ratio);
/* Lens extension (see bt_lens): NULL switches it off again. */
int bt_scene_set_lens(bt_scene *scene, const bt_lens *lens);
int bt_scene_object_count(const bt_scene *scene);
int bt_scene_data_count(const bt_scene *scene);
/* Flattened primitive table as uploaded to the GPU, for loader cross-checks:
 * writes up to `cap` floats, returns the number available. */
int bt_scene_export_prims(const bt_scene *scene, float *out, int cap);

/* --- Tracer::render (tracer/mod.rs:179-202) ----------------------------------------
 * Adds `samples * n^2` radiance samples per pixel into the RGB channels of `rgba`
 * (row-major, 4 floats per pixel, alpha untouched: buffer.rs:159-178).  The caller
 * tracks Buffer::samples += samples * n^2 (mod.rs:199).  `seed` replaces
 * SmallRng::from_entropy() (mod.rs:239-242).  Returns BT_DONE when samples == 0,
 * BT_IN_PROGRESS otherwise, < 0 on error.  Nothing is retained after return. */

/* Host buffer: copies H2D into a device frame cached on the scene handle, renders on the current device, copies D2H
 * (two PCIe transfers of w*h*16 bytes per call).  A caller that renders once per displayed frame (main.rs:245-254)
 * should keep the frame on the device and use bt_render_device + bt_preview_device instead. */
int bt_render(bt_scene *scene, uint64_t camera_ref, const bt_config *config, const bt_render_config *render,
              float *rgba_host, uint32_t width, uint32_t height, uint64_t seed);

/* Device-resident buffer; `stream` is a hipStream_t (NULL = default stream).  The call
 * enqueues work and returns without synchronising. */
int bt_render_device(bt_scene *scene, uint64_t camera_ref, const bt_config *config, const bt_render_config *render,
                     float *rgba_device, uint32_t width, uint32_t height, uint64_t seed, void *stream);

/* --- Multi-GPU pixel-tile sharding (new; the reference's only parallelism is rayon
 * tiles inside one process, tracer/mod.rs:190-197) ---------------------------------
 * The frame is cut into BT_TILE x BT_TILE pixel tiles, numbered row-major; rank r of
 * `world` owns tiles r, r+world, r+2*world, ...  A shard holds the rank's tiles
 * back to back, each tile BT_TILE*BT_TILE*4 floats (padded tiles included), so every
 * rank's shard has bt_shard_floats() elements and an all-gather concatenates them. */
#define BT_TILE 16
size_t bt_shard_floats(uint32_t width, uint32_t height, uint32_t world);
/* Renders this rank's tiles into `shard_device`, which must hold the rank's
 * running sums in shard layout (zero + alpha 1 for a fresh frame). */
int bt_render_shard_device(bt_scene *scene, uint64_t camera_ref, const bt_config *config,
                           const bt_render_config *render, float *shard_device, uint32_t width, uint32_t height,
                           uint32_t rank, uint32_t world, uint64_t seed, void *stream);
/* gathered = `world` shards back to back (the all-gather result) -> row-major frame. */
int bt_unshard_device(const float *gathered_device, float *rgba_device, uint32_t width, uint32_t height,
                      uint32_t world, void *stream);

/* The exchange step: one RCCL all-gather over xGMI of every rank's shard, one process per GPU.  RCCL is bound at run
 * time (librccl.so.1; a copy the process already holds, e.g. PyTorch's, is shared).  Rank 0 obtains a unique id and
 * hands it to the other ranks by any host-side channel (a file, a socket, MPI); bt_comm_init is collective. */
#define BT_COMM_ID_BYTES 128
typedef struct bt_comm bt_comm;
/* ncclGetUniqueId: writes BT_COMM_ID_BYTES bytes, returns that count, < 0 on error. */
int bt_comm_unique_id(void *id_out, size_t cap);
/* ncclCommInitRank on the current device.  NULL on error (bt_last_error). */
bt_comm *bt_comm_init(uint32_t rank, uint32_t world, const void *unique_id, size_t id_bytes);
void bt_comm_free(bt_comm *comm);
int bt_comm_rank(const bt_comm *comm);
int bt_comm_world(const bt_comm *comm);
/* ncclAllGather(shard -> gathered, bt_shard_floats(width, height, world) floats per rank) on `stream`. */
int bt_allgather_shards_device(bt_comm *comm, const float *shard_device, float *gathered_device, uint32_t width,
                               uint32_t height, void *stream);
/* All-gather + bt_unshard_device: every rank ends up with the row-major frame of running sums in `rgba_device`
 * (north_star: "RCCL all-gather over xGMI of the final framebuffer").  `gathered_device` is world * shard floats of
 * staging the caller owns. */
int bt_exchange_frame_device(bt_comm *comm, const float *shard_device, float *gathered_device, float *rgba_device,
                             uint32_t width, uint32_t height, void *stream);

/* --- Buffer::preview (tracer/buffer.rs:117-138): sum/samples -> colour space -> RGBA8 */
int bt_preview_device(const float *rgba_device, uint8_t *rgba8_device, uint32_t width, uint32_t height,
                      uint32_t samples, int32_t color_space, void *stream);
int bt_preview(const float *rgba_host, uint8_t *rgba8_host, uint32_t width, uint32_t height, uint32_t samples,
               int32_t color_space);

void bt_tuning_default(bt_tuning *out);
/* NULL restores the defaults.  Returns BT_ERR_INVALID_ARG for a value outside the sets above. */
int bt_scene_set_tuning(bt_scene *scene, const bt_tuning *tuning);
int bt_scene_get_tuning(const bt_scene *scene, bt_tuning *out);

/* Returns the device memory a handle keeps between calls -- the parked sample values of the work queue (bt_stats.scratch_bytes;
 * it otherwise shrinks only after eight consecutive renders that needed less than a quarter of it) and bt_render's cached copy
 * of the caller's frame -- to the device (synchronises it).  The reference holds no such state: `Tracer::render` borrows the
 * scene and the buffer for the call (tracer/mod.rs:179-185).  A handle serves ONE stream at a time, like `&mut Buffer`. */
int bt_scene_trim(bt_scene *scene);

/* Work counters of the most recent bt_render* call on this handle (synchronises). */
int bt_scene_last_stats(bt_scene *scene, bt_stats *out);

#ifdef __cplusplus
}
#endif
#endif
