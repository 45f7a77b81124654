// bt_types.h -- POD layouts shared by the host scene preparation (bt_scene.cpp) and the
// HIP kernels (bt_kernels.hip).  Everything here is what actually sits in HBM / LDS.
#pragma once
#include <stdint.h>

#define BT_TILE_DIM 16   // pixels per tile edge (== BT_TILE in include/bendy_hip.h)
#if defined(BT_PROFILE) || defined(BT_LANESTAT)
#define BT_N_COUNTERS 12 // developer build: [2..] = wave cycles per section of the render loop
#else
#define BT_N_COUNTERS 2
#endif
#define BT_DENSITY_LDS_MAX 8192   // density maps up to this many cells are staged in LDS (32 KB)

struct BtV3 { float x, y, z; };

// One row per analytic primitive after flattening: a Cuboid (cuboid.rs:12-15) becomes six
// rects with the face transform `M * translate(offset)` (cuboid.rs:95) baked in, and every
// rect carries the inverse of its transform (recomputed per call by the reference,
// rect.rs:134).  144 bytes, read with wave-uniform (scalar) loads in the intersection loop.
// kind = shape | BT_PRIM_STRICT.  BT_PRIM_RECT_AA is a rect whose transform matrix is exactly the
// identity and whose x / y axes are signed unit basis vectors: for it `M^-1 * pos + t'` and the two
// projections of Rect::contains_point (rect.rs:74-80) reduce, bit for bit, to two component adds
// and two squares (DESIGN.md "Kernel").  BT_PRIM_RECT_AAN is such a rect whose world normal `c` is a signed unit
// basis vector as well (component aa_w): then dot(d, n) = +-d[w] and dot(t - o, n) = +-(t[w] - o[w]) exactly, and
// p / q = (t[w] - o[w]) / d[w] bit for bit -- the two dot products of rect.rs:120-124 reduce to one subtraction.
// BT_PRIM_RECT_LA is a rect under any transform whose LOCAL axes Rect.x / Rect.y are signed unit basis vectors (every
// cuboid face, cuboid.rs:19-30; the rotated box of the Cornell scenes): `local.project_onto_normalized(x)` then has
// the squared length local[aa_u]^2 exactly, so only two components of `M^-1 * pos + t'` are formed and the two
// projections (rect.rs:74-80) reduce to two squares.
// BT_PRIM_STRICT marks cuboid faces (`manifold.t < t`, cuboid.rs:96).
enum { BT_PRIM_SPHERE = 0, BT_PRIM_RECT = 1, BT_PRIM_RECT_AA = 2, BT_PRIM_RECT_AAN = 3, BT_PRIM_RECT_LA = 4, BT_PRIM_SHAPE_MASK = 7, BT_PRIM_STRICT = 8 };
struct BtPrim {
    int32_t kind;
    int32_t object;     // object index (ascending ObjectRef) -- `last_object` test, mod.rs:415
    int32_t material;   // index into materials
    int32_t volume;     // index into volumes or -1 (sphere.rs:14)
    // sphere: c = centre (transform.translation), radius.   rect: c = n = M*z (rect.rs:119)
    BtV3 c;
    float radius;
    BtV3 t;             // rect: transform.translation (rect.rs:118)
    float w_sqr;        // half_width^2  (rect.rs:77)
    BtV3 icx; float h_sqr;   // inverse transform columns (rect.rs:134) ; half_height^2
    BtV3 icy; int32_t aa_u;     // BT_PRIM_RECT_AA: component index of Rect.x
    BtV3 icz; int32_t aa_v;     // BT_PRIM_RECT_AA: component index of Rect.y
    BtV3 it;  int32_t aa_w;     // BT_PRIM_RECT_AAN: component index of the normal
    // Rect.x, Rect.y (rect.rs:17-18); for BT_PRIM_RECT_LA rows instead the two rows of `M^-1 | t'` that the test
    // needs: (ax, ax_w) = (icx[u], icy[u], icz[u], it[u]), (ay, ay_w) likewise for v; for BT_PRIM_RECT_AAN rows the six
    // constants of rect_aan_t side by side (one scalar load): ax = (t[w], it[a], it[b]), ax_w = limit of a,
    // ay = (limit of b, c[w] = +-1, 0), with a < b the two in-plane axes
    BtV3 ax;  float ax_w;
    BtV3 ay;  float ay_w;
};
static_assert(sizeof(BtPrim) == 144, "BtPrim must be 144 bytes");

// Sphere-only scenes: the same centres / radii again, two spheres side by side, so that the part of Sphere::hit that
// does not depend on the running clip (oc, half_b, c, discriminant -- sphere.rs:122-127) is evaluated for two
// spheres at once with packed FP32 instructions (bt_device.hpp intersect_spheres).  Row p = primitives 2p, 2p + 1;
// an odd tail repeats the last sphere in slot 1 (never tested).
struct BtSpherePair {
    float cx[2], cy[2], cz[2], radius[2];
    int32_t object[2];
    int32_t pad[2];
};
static_assert(sizeof(BtSpherePair) == 48, "BtSpherePair must be 48 bytes");
// Sphere-only scenes WITHOUT volumes (scene.json): one 16-byte row per sphere -- a pair of spheres is one s_load_dwordx8, an odd
// table's last sphere one s_load_dwordx4 (bt_device.hpp intersect_spheres_plain; BtSpherePair costs three loads per pair).
struct BtSphereRow {
    float cx, cy, cz;
    float r2;               // radius * radius (sphere.rs:130), multiplied once on the host: the same IEEE product
};
static_assert(sizeof(BtSphereRow) == 16, "BtSphereRow must be 16 bytes");

// Rect scenes without volumes: the BT_PRIM_RECT_AAN rows once more, 32 bytes each, grouped by the axis of the normal
// (all x-normal rows, then y, then z; ascending row inside a group) -- bt_device.hpp intersect_sorted() walks a group
// with one refined reciprocal of the ray direction's component for the whole group.  `prio` ranks the row for exact ties
// in t, so that any processing order gives try_hit's result (mod.rs:389-402: rows in ascending order, a plain rect or
// sphere accepts t <= clip.max, a cuboid face only t < clip.max): 0x10000 + row for the former, 0xfffe - row for the
// latter (0xffff = no hit yet); the hit with the smallest t wins, among equal t the largest prio.
struct BtRectAAN {
    float it_a, it_b;       // it[a], it[b] of the inverse transform (a < b the in-plane axes): an aligned SGPR pair
    float lim_a, lim_b;     // largest |x| whose square passes `x * x <= w_sqr` / `<= h_sqr` (bt_scene.cpp abs_limit)
    float t_w;              // t[w]
    uint32_t sgn_mask;      // sign bit of c[w] = +-1: p = (t[w] - o[w]) * c[w] has the sign of (t[w] - o[w]) ^ sgn_mask
    uint32_t prio;
    uint32_t pad;
};
static_assert(sizeof(BtRectAAN) == 32, "BtRectAAN must be 32 bytes");
// ... and the BT_PRIM_RECT_LA rows (every cuboid face under a rotation), rows with bitwise equal normals next to each
// other: opposite faces of a cuboid share the normal (cuboid.rs:19-30), hence q = dot(d, n) and its reciprocal.  The two
// rows of `M^-1 | t'` that the containment test needs sit side by side as pairs for packed arithmetic.
struct BtRectLA {
    BtV3 n;                 // world normal c = M * z
    uint32_t first_of_normal; // 1: q and its reciprocal have to be formed for this row, 0: same normal as the row before
    BtV3 t;                 // transform.translation
    uint32_t prio;
    float a_x[2], a_y[2], a_z[2], a_w[2];   // (ax, ay) component pairs; a_w = (ax_w, ay_w)
    float lim[2];           // abs_limit(w_sqr), abs_limit(h_sqr)
    uint32_t pad[2];
};
static_assert(sizeof(BtRectLA) == 80, "BtRectLA must be 80 bytes");

struct BtVolBox {       // LDS only
    BtV3 bmin; float ok;    // bbox.min; ok != 0: size components within [2^-20, 2^20] (div_refined's range)
    BtV3 size; float pad0;  // bbox.max - bbox.min
    BtV3 rcp;  float pad1;  // refined_rcp(size)
};
static_assert(sizeof(BtVolBox) == 48, "BtVolBox must be 48 bytes");

// Per-lane (divergent) lookups after the loop read this 32-byte digest from LDS.
struct BtPrimLite {
    BtV3 c;             // sphere centre | rect world normal
    float radius;
    int32_t kind_object;  // shape | object << 8
    int32_t material;
    int32_t volume;
    float rcp_radius;     // bt_kernels.hip only: refined_rcp(radius) for spheres with 2^-20 <= radius <= 2^20, else 0 (bt_device.hpp
                          // div_refined: the three divisions of Sphere's normal, sphere.rs:95-99, share it)
};

enum { BT_MAT_FLAT = 0, BT_MAT_DIFFUSE = 1, BT_MAT_METALLIC = 2, BT_MAT_GLASS = 3, BT_MAT_EMISSIVE = 4 };
struct BtMaterial {     // material.rs:22-44; 48 bytes
    int32_t kind;
    BtV3 albedo;
    float roughness, ior, inv_ior, pad;
    BtV3 emitted;       // material.rs:71-79 (albedo or albedo*intensity or 0)
    float pad1;
};

struct BtVolume {       // volume.rs:75-82
    int32_t width, height, depth, offset;   // offset into the density buffer
    BtV3 size;
    float pad;
};

// One entry per LIGHT object (object/mod.rs:17-21), ascending ObjectRef.
enum { BT_LIGHT_SPHERE = 0, BT_LIGHT_RECT = 1, BT_LIGHT_CUBOID = 2, BT_LIGHT_POINT = 3 };
struct BtLightFace {    // a rect that can be sampled: rect.rs:82-86
    BtV3 mcx, mcy, mcz, mt;   // world transform (face transform for cuboids)
    BtV3 ax, ay;              // Rect.x, Rect.y
    float half_width, half_height;
    float scale_x, scale_y;   // Uniform::new_inclusive(-hw,hw) / (-hh,hh) scales
    float area;               // rect.rs:88-90
    float pad;
};
struct BtLight {
    int32_t kind;
    int32_t prim_first, prim_count;   // rows of BtPrim that make up the object
    int32_t face_first;               // rows of BtLightFace
    BtV3 centre;                      // sphere / point fallback: transform.translation
    float radius;
    float shadow;                     // sphere: PI*r*r (sphere.rs:54)
    float cum[5];                     // cuboid WeightedIndex cumulative areas (cuboid.rs:49)
    float total_scale;                // Uniform::new(0, total) scale
    float pad;
};

// Kernel launch parameters (by value; they live in the kernarg segment / SGPRs).
struct BtLaunch {
    // scene tables
    const BtPrim *prims;
    const BtSpherePair *sphere_pairs; // sphere-only scenes (any_rects == 0), else null
    const BtMaterial *materials;
    const BtVolume *volumes;
    const BtLight *lights;
    const BtLightFace *light_faces;
    const float *density;
    int32_t n_prims, n_materials, n_volumes, n_lights, n_light_faces, n_density;
    int32_t any_rects;                // 0: every row of `prims` is a sphere (selects the build without rect code)
    int32_t any_volumes;              // 0: no row carries a volume (selects the build without the march); the host also sets it
                                      // when a rect scene cannot use the sorted tables below (the generic loop lives in that build)
    // rect scenes without volumes (bt_device.hpp intersect_sorted): AAN rows grouped by normal axis, then every other row
    const BtRectAAN *aan_rows;
    const BtRectLA *la_rows;
    const int32_t *other_rows;        // rows of `prims` that are neither BT_PRIM_RECT_AAN nor BT_PRIM_RECT_LA, ascending
    int32_t n_aan[3];                 // x-, y-, z-normal rows of aan_rows (back to back)
    int32_t n_la, n_other;
    // root material (mod.rs:429-452), precomputed ColorData of sample_root
    BtV3 root_color, root_albedo;
    int32_t root_has_albedo;
    // camera (mod.rs:248-302)
    BtV3 cam_cx, cam_cy, cam_cz, cam_t;
    float yfov, xfov, pixel_width, pixel_height;
    float jitter_u_lo, jitter_u_scale, jitter_v_lo, jitter_v_scale;
    int32_t has_focus;
    float focus, aperture;
    BtV3 disk_x, disk_y;              // UnitDisk::new(-Z) frame (distr.rs:111-116)
    float tau_scale, one_scale;       // Uniform::new_inclusive(0,TAU) / (0,1) scales
    // config (mod.rs:205-230)
    int32_t max_bounces, max_volume_bounces;
    float clip_min, clip_max, volume_step;
    int32_t samples, subsample_n;     // subsample_n >= 1
    uint32_t sample_base;
    uint32_t seed_lo, seed_hi;
    // frame / sharding
    uint32_t width, height, tiles_x, tiles_y;
    uint32_t rank, world;
    int32_t sharded;                  // 0: out = row-major frame; 1: out = this rank's shard
    float *out;
    unsigned long long *counters;     // [0] path segments, [1] lens RK4 steps
    // The work queue (bt_api.cpp decides its shape).  A 16x16 tile is cut into `slices` = S in {1,2,4,8,16,32} pixel blocks
    // of 256/S pixels; a block's (pixel, sample) pairs are dealt to the lanes of a workgroup, every sample's value is parked
    // in scratch[(block * T + k) * (256/S) + pixel] (12 bytes, T = samples * n^2) and added to the frame in sample order --
    // the additions a lane that owned the pixel would perform, in the same order, hence the same bits.
    int32_t slices;
    float *scratch;
    uint32_t tiles_x_magic;           // floor(2^32 / tiles_x) (0xffffffff for tiles_x = 1): tile / tiles_x = umulhi(tile, magic), fixed up by one step
    uint32_t table_lds_bytes;         // bytes of the scene tables at the start of dynamic LDS
    // Scenes with volumes: behind the tables one BtVolBox per primitive (48 B; filled by the kernel's prologue for the
    // spheres that carry a volume): the bounding box Volume::shade divides by (volume.rs:26-35, sphere.rs:35-38) and the
    // refined reciprocal of its size, so that the three divisions of every march step share div_refined()'s reciprocal.
    uint32_t vbox_lds_bytes;          // 48 * n_prims, or 0
    int32_t vols_safe;                // 1: every density map has dims >= 1 and 0 <= ceil(size) <= dim - 1: DensityMap::index's
                                      // bounds tests (volume.rs:119-134) cannot fire for a clamped coordinate
    // Phase voting (bt_kernels.hip, sphere-only builds): every iteration the wave runs EITHER the camera event OR the
    // scatter / volume events, whichever more of its lanes want; the others keep what they have (no ray yet, or their
    // hit) for the next iteration, at most phase_vote iterations in a row (0 = off).  Scheduling only: every lane
    // performs the same operations in the same order.
    int32_t phase_vote;
    // lens EXTENSION (not in the reference, default off; include/bendy_hip.h bt_lens)
    int32_t lens_on;
    BtV3 lens_c;
    float lens_rs, lens_step, lens_radius;
    int32_t lens_max_steps;
    // primitives that can be met inside the lens' sphere of influence: those whose surface comes within
    // lens_radius + lens_margin of lens_c (ascending rows of `prims`; bt_api.cpp lens_candidates).  A chord of the
    // RK4 march that starts inside the sphere and is no longer than lens_margin cannot touch any other primitive,
    // so only these rows are tested for it -- an optimisation that cannot change a result.
    const int32_t *lens_prims;
    int32_t n_lens_prims;
    float lens_margin;
    const BtSphereRow *sphere_rows;   // sphere-only scenes without volumes, else null
    // Packed launch (small frames / few samples, bt_api.cpp): n_workgroups workgroups -- one per workgroup slot of the GPU --
    // share the launch's blocks, workgroup w owning blocks w, w + n_workgroups, ... (wg_blocks of them, one less from workgroup
    // wg_blocks_rem on) behind one queue.  wg_blocks = 1: one block per workgroup.
    uint32_t wg_blocks, wg_blocks_rem, n_workgroups;
    uint32_t pool_records, pool_lds_offset;   // packed: PathRec records behind the tables in dynamic LDS for the drain's compaction
                                      // rounds (bt_kernels.hip), 0 = none: lanes leave the loop when the queue is empty
    uint32_t log_rows, row_mask;      // packed: a block's T = samples * n^2 samples are padded to 2^log_rows rows in the queue and in
                                      // scratch, row_mask = 2^log_rows - 1; not packed: row_mask = 0xffffffff
};
