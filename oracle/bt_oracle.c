/*
 * bt_oracle.c -- CPU ORACLE (test infrastructure, see bt_oracle.h).
 *
 * Plain-C restatement of the reference hot path.  Citations are relative to the
 * reference tree (soycan-sim/bendy-tracer @ v1).  PARITY UNPINNED against the
 * Rust binary (no toolchain, entropy-seeded RNG, no golden vectors): pinned by
 * the analytic KATs in tests/test_oracle_kat.py and by DESIGN.md's numerics
 * contract, which this file and the HIP kernels implement independently.
 *
 * Numerics contract (DESIGN.md "Numerics contract"), restated:
 *   N1 all arithmetic IEEE binary32, round-to-nearest, no contraction
 *      (build with -ffp-contract=off); fmaf only where written.
 *   N2 dot(a,b) = (ax*bx + ay*by) + az*bz.
 *   N3 normalize(a) = a * (1 / sqrt(dot(a,a))).
 *   N4 M*v = (cx*vx + cy*vy) + cz*vz  (columns), point = M*v + t.
 *   N5 sin/cos: 3-term Cody-Waite reduction by pi/2 with fmaf + degree-7/8
 *      polynomials (bto_sincos).
 *   N6 random numbers: Philox4x32-10, key = seed, counter =
 *      (pixel index, sample index, event index, block); one block of four
 *      u32 per random event with fixed slot assignment (Appendix B of SURVEY.md).
 *   N7 u32 -> float as rand 0.8.5 does (23 mantissa bits, value*scale+low).
 */
#include "bt_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

typedef bto_v3 v3;

/* ------------------------------------------------------------------ vectors */
static inline v3 V3(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 vadd(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vmul(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 vdiv(v3 a, v3 b) { return V3(a.x / b.x, a.y / b.y, a.z / b.z); }
static inline v3 vscale(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline v3 vdivs(v3 a, float s) { return V3(a.x / s, a.y / s, a.z / s); }
static inline v3 vneg(v3 a) { return V3(-a.x, -a.y, -a.z); }
static inline float vdot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; } /* N2 */
static inline float vlen2(v3 a) { return vdot(a, a); }
static inline v3 vcross(v3 a, v3 b) {
    return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline v3 vnormalize(v3 a) { /* N3 */
    float rl = 1.0f / sqrtf(vlen2(a));
    return vscale(a, rl);
}
/* glam normalize_or_zero (ray.rs:133) */
static inline v3 vnormalize_or_zero(v3 a) {
    float rl = 1.0f / sqrtf(vlen2(a));
    if (isfinite(rl) && rl > 0.0f) return vscale(a, rl);
    return V3(0.0f, 0.0f, 0.0f);
}
/* glam Affine3A::transform_vector3a / transform_point3a (N4) */
static inline v3 xf_vector(const bto_affine *m, v3 v) {
    return vadd(vadd(vscale(m->cx, v.x), vscale(m->cy, v.y)), vscale(m->cz, v.z));
}
static inline v3 xf_point(const bto_affine *m, v3 p) { return vadd(xf_vector(m, p), m->t); }

/* glam Affine3A::inverse = Mat3A::inverse (cross products / det) + -(Minv * t);
 * used per call by Rect::hit (rect.rs:134). */
void bto_affine_inverse(const bto_affine *a, bto_affine *out) {
    v3 t0 = vcross(a->cy, a->cz);
    v3 t1 = vcross(a->cz, a->cx);
    v3 t2 = vcross(a->cx, a->cy);
    float det = vdot(a->cz, t2);
    float inv_det = 1.0f / det;
    v3 r0 = vscale(t0, inv_det), r1 = vscale(t1, inv_det), r2 = vscale(t2, inv_det);
    /* rows r0,r1,r2 -> transpose into columns */
    out->cx = V3(r0.x, r1.x, r2.x);
    out->cy = V3(r0.y, r1.y, r2.y);
    out->cz = V3(r0.z, r1.z, r2.z);
    out->t = V3(0, 0, 0);
    out->t = vneg(xf_vector(out, a->t));
}

/* glam Affine3A * Affine3A::from_translation(offset) (cuboid.rs:39,52,68,78,95):
 * matrix3 unchanged, translation = M*offset + t. */
static inline bto_affine xf_translate(const bto_affine *m, v3 offset) {
    bto_affine r = *m;
    r.t = xf_point(m, offset);
    return r;
}

/* ------------------------------------------------------------ math (N5, math/mod.rs) */
void bto_sincos(float x, float *s, float *c) {
    float k = rintf(x * 0.636619772f);
    float r = fmaf(k, -1.5703125f, x);
    r = fmaf(k, -4.837512969970703125e-4f, r);
    r = fmaf(k, -7.54978995489188e-8f, r);
    float r2 = r * r;
    float ps = fmaf(fmaf(-1.9515295891e-4f, r2, 8.3321608736e-3f), r2, -1.6666654611e-1f);
    float sn = fmaf(ps * r2, r, r);
    float pc = fmaf(fmaf(2.443315711809948e-5f, r2, -1.388731625493765e-3f), r2, 4.166664568298827e-2f);
    float cs = fmaf(pc, r2 * r2, fmaf(-0.5f, r2, 1.0f));
    int q = ((int)k) & 3;
    float so = (q & 1) ? cs : sn;
    float co = (q & 1) ? sn : cs;
    if (q == 2 || q == 3) so = -so;
    if (q == 1 || q == 2) co = -co;
    *s = so;
    *c = co;
}

/* math/mod.rs:9-13 */
static inline float lerpf(float a, float b, float f) { return a + (b - a) * f; }
/* math/mod.rs:41-43 */
static inline v3 reflect(v3 v, v3 n) { return vsub(v, vscale(n, 2.0f * vdot(v, n))); }
/* math/mod.rs:45-50 */
static inline v3 refract(v3 v, v3 n, float ior) {
    float cos_theta = fminf(vdot(vneg(v), n), 1.0f);
    v3 perp = vscale(vadd(vscale(n, cos_theta), v), ior);
    v3 parallel = vscale(n, -sqrtf(fabsf(1.0f - vlen2(perp))));
    return vadd(perp, parallel);
}
/* math/mod.rs:52-57; powi(5) fixed as ((x*x)*(x*x))*x */
static inline float fresnel(v3 v, v3 n, float ior) {
    float cos_theta = fminf(vdot(vneg(v), n), 1.0f);
    float r0 = (1.0f - ior) / (1.0f + ior);
    r0 = r0 * r0;
    float x = 1.0f - cos_theta;
    float x2 = x * x;
    return r0 + (1.0f - r0) * ((x2 * x2) * x);
}
void bto_reflect(const float *v, const float *n, float *o) {
    v3 r = reflect(V3(v[0], v[1], v[2]), V3(n[0], n[1], n[2]));
    o[0] = r.x; o[1] = r.y; o[2] = r.z;
}
void bto_refract(const float *v, const float *n, float ior, float *o) {
    v3 r = refract(V3(v[0], v[1], v[2]), V3(n[0], n[1], n[2]), ior);
    o[0] = r.x; o[1] = r.y; o[2] = r.z;
}
float bto_fresnel(const float *v, const float *n, float ior) {
    return fresnel(V3(v[0], v[1], v[2]), V3(n[0], n[1], n[2]), ior);
}

/* glam Vec3::any_orthonormal_pair (distr.rs:39,77,114): Duff et al. 2017 */
static inline void orthonormal_pair(v3 n, v3 *t1, v3 *t2) {
    float sign = copysignf(1.0f, n.z);
    float a = -1.0f / (sign + n.z);
    float b = n.x * n.y * a;
    *t1 = V3(1.0f + sign * n.x * n.x * a, sign * b, -sign * n.x);
    *t2 = V3(b, sign + n.y * n.y * a, -n.y);
}
void bto_orthonormal_pair(const float *n, float *t1, float *t2) {
    v3 a, b;
    orthonormal_pair(V3(n[0], n[1], n[2]), &a, &b);
    t1[0] = a.x; t1[1] = a.y; t1[2] = a.z;
    t2[0] = b.x; t2[1] = b.y; t2[2] = b.z;
}

/* ------------------------------------------------------------------ RNG (N6, N7) */
void bto_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int i = 0; i < 10; ++i) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static inline float u23(uint32_t x) { /* rand 0.8.5 UniformFloat::sample: [1,2) - 1 */
    uint32_t b = (x >> 9) | 0x3F800000u;
    float f;
    memcpy(&f, &b, 4);
    return f - 1.0f;
}
static inline float u24(uint32_t x) { return (float)(x >> 8) * 5.9604644775390625e-8f; } /* Standard */
static inline int bernoulli(uint32_t x, float p) { return u24(x) < p; }

/* rand 0.8.5 UniformFloat::new / new_inclusive: the scale is nudged down until the
 * largest sample stays inside the range (recalled semantics). */
float bto_uniform_scale(float lo, float hi, int inclusive) {
    const float max_rand = 1.0f - 1.1920928955078125e-7f;
    float scale = inclusive ? (hi - lo) / max_rand : (hi - lo);
    for (int guard = 0; guard < 64; ++guard) {
        float top = scale * max_rand + lo;
        int bad = inclusive ? (top > hi) : (top >= hi);
        if (!bad) break;
        uint32_t b;
        memcpy(&b, &scale, 4);
        b -= 1;
        memcpy(&scale, &b, 4);
    }
    return scale;
}
static inline float uniform_sample(uint32_t x, float lo, float scale) { return u23(x) * scale + lo; }

typedef struct {
    uint32_t key[2];
    uint32_t pixel, sample, event;
    uint64_t segments;
} rng_t;

/* one random event = one Philox block; `event` then advances */
static inline void rng_event(rng_t *r, uint32_t out[4]) {
    uint32_t ctr[4] = {r->pixel, r->sample, r->event, 0};
    bto_philox4x32_10(ctr, r->key, out);
    r->event += 1;
}
/* overflow block of the event just drawn (cuboid light: face pick) */
static inline void rng_event_extra(const rng_t *r, uint32_t out[4]) {
    uint32_t ctr[4] = {r->pixel, r->sample, r->event - 1, 1};
    bto_philox4x32_10(ctr, r->key, out);
}

/* ------------------------------------------------------------------ tracer types (ray.rs) */
enum { FACE_FRONT = 0, FACE_BACK = 1, FACE_VOLUME = 2, FACE_VOLUME_FRONT = 3, FACE_VOLUME_BACK = 4 };
static inline int face_is_surface(int f) { return f == FACE_FRONT || f == FACE_BACK; }           /* ray.rs:26-28 */
static inline int face_is_front(int f) { return f == FACE_FRONT || f == FACE_VOLUME_FRONT; }     /* ray.rs:18-20 */

typedef struct { v3 origin, direction; } ray_t;
typedef struct { float min, max; } clip_t;
typedef struct { /* ray.rs:35-47 */
    v3 position, normal, bbox_min, bbox_max;
    int face;
    float t;
    ray_t ray;
    int object, mat, vol; /* indices or -1 */
    float depth_extra;    /* lens extension: path length before the chord that hit; 0 otherwise */
} manifold_t;
typedef struct { v3 color, albedo, normal; float depth; } colordata_t; /* ray.rs:49-55 */

static inline colordata_t colordata_default(void) { /* ray.rs:67-76 */
    colordata_t c = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, INFINITY};
    return c;
}
static inline colordata_t colordata_from_emitted(v3 e) { /* ray.rs:58-64 */
    colordata_t c = colordata_default();
    c.color = e;
    c.albedo = e;
    return c;
}
static inline ray_t ray_new(v3 o, v3 d) { ray_t r = {o, vnormalize(d)}; return r; } /* ray.rs:96-101 */
static inline v3 ray_at(const ray_t *r, float t) { return vadd(r->origin, vscale(r->direction, t)); } /* ray.rs:115-117 */

/* ray.rs:103-113: Quat::from_euler(YXZ, yrot, xrot, 0) * -Z in closed form:
 * Ry(a)*Rx(b)*(0,0,-1) = (-cos b sin a, sin b, -cos b cos a). */
static inline v3 frustum_direction(float yfov, float xfov, float u, float v) {
    float yrot = xfov * 0.5f * -u;
    float xrot = yfov * 0.5f * -v;
    float sy, cy, sx, cx;
    bto_sincos(yrot, &sy, &cy);
    bto_sincos(xrot, &sx, &cx);
    return V3(-(cx * sy), sx, -(cx * cy));
}
void bto_ray_with_frustum(float yfov, float xfov, float u, float v, float *d) {
    v3 r = frustum_direction(yfov, xfov, u, v);
    d[0] = r.x; d[1] = r.y; d[2] = r.z;
}
/* ray.rs:126-137: origin = translation + origin (NOT a point transform, Q2);
 * direction = normalize(normalize_or_zero(M*d)) */
static inline ray_t affine_mul_ray(const bto_affine *m, const ray_t *r) {
    v3 o = vadd(m->t, r->origin);
    v3 d = vnormalize_or_zero(xf_vector(m, r->direction));
    return ray_new(o, d);
}

/* ------------------------------------------------------------------ distributions (math/distr.rs) */
typedef struct { float tau_scale, one_scale; } distr_t;
static distr_t g_distr;
static pthread_once_t g_distr_once = PTHREAD_ONCE_INIT;
static void distr_init(void) {
    g_distr.tau_scale = bto_uniform_scale(0.0f, 6.28318530717958647692f, 1);
    g_distr.one_scale = bto_uniform_scale(0.0f, 1.0f, 1);
}
static inline float draw_tau(uint32_t x) { return uniform_sample(x, 0.0f, g_distr.tau_scale); }
static inline float draw_one(uint32_t x) { return uniform_sample(x, 0.0f, g_distr.one_scale); }

/* distr.rs:10-21 */
static inline v3 unit_sphere(uint32_t x1, uint32_t x2) {
    float r1 = draw_tau(x1), r2 = draw_one(x2);
    float s, c;
    bto_sincos(r1, &s, &c);
    float x = c * 2.0f * sqrtf(r2 * (1.0f - r2));
    float y = s * 2.0f * sqrtf(r2 * (1.0f - r2));
    float z = 1.0f - 2.0f * r2;
    return V3(x, y, z);
}
/* distr.rs:36-45,48-59 ("hemisphere": z = 1 - r2, Q3) */
static inline v3 unit_hemisphere(v3 normal, uint32_t x1, uint32_t x2) {
    v3 z_axis = vnormalize(normal), x_axis, y_axis;
    orthonormal_pair(z_axis, &x_axis, &y_axis);
    float r1 = draw_tau(x1), r2 = draw_one(x2);
    float s, c;
    bto_sincos(r1, &s, &c);
    float x = c * 2.0f * sqrtf(r2 * (1.0f - r2));
    float y = s * 2.0f * sqrtf(r2 * (1.0f - r2));
    float z = 1.0f - r2;
    return vadd(vadd(vscale(x_axis, x), vscale(y_axis, y)), vscale(z_axis, z));
}
/* distr.rs:74-83,86-97 */
static inline v3 cosine(v3 normal, uint32_t x1, uint32_t x2) {
    v3 z_axis = vnormalize(normal), x_axis, y_axis;
    orthonormal_pair(z_axis, &x_axis, &y_axis);
    float r1 = draw_tau(x1), r2 = draw_one(x2);
    float s, c;
    bto_sincos(r1, &s, &c);
    float x = c * sqrtf(r2);
    float y = s * sqrtf(r2);
    float z = sqrtf(1.0f - r2);
    return vadd(vadd(vscale(x_axis, x), vscale(y_axis, y)), vscale(z_axis, z));
}
/* distr.rs:111-116,119-132 (radius not sqrt'd, Q4) */
static inline v3 unit_disk(v3 normal, uint32_t x1, uint32_t x2) {
    v3 n = vnormalize(normal), x_axis, y_axis;
    orthonormal_pair(n, &x_axis, &y_axis);
    float angle = draw_tau(x1), r = draw_one(x2);
    float s, c;
    bto_sincos(angle, &s, &c);
    return vscale(vadd(vscale(x_axis, c), vscale(y_axis, s)), r);
}

/* ------------------------------------------------------------------ primitives */
/* sphere.rs:35-38 */
static inline void sphere_bbox(float radius, v3 tr, v3 *mn, v3 *mx) {
    v3 h = V3(radius, radius, radius);
    *mn = vsub(tr, h);
    *mx = vadd(tr, h);
}
/* sphere.rs:85-119 */
static manifold_t sphere_surface_manifold(const bto_object *o, int oi, v3 tr, ray_t ray, float t) {
    int front_face = o->volume >= 0 ? FACE_VOLUME_FRONT : FACE_FRONT;
    int back_face = o->volume >= 0 ? FACE_VOLUME_BACK : FACE_BACK;
    manifold_t m;
    m.position = ray_at(&ray, t);
    v3 normal = vdivs(vsub(m.position, tr), o->radius);
    if (vdot(ray.direction, normal) < 0.0f) {
        m.normal = normal;
        m.face = front_face;
    } else {
        m.normal = vneg(normal);
        m.face = back_face;
    }
    sphere_bbox(o->radius, tr, &m.bbox_min, &m.bbox_max);
    m.t = t;
    m.ray = ray;
    m.object = oi;
    m.mat = o->material;
    m.vol = o->volume;
    m.depth_extra = 0.0f;
    return m;
}
/* sphere.rs:121-148.  `discriminant.is_sign_negative()` is restated as !(d >= 0):
 * identical for every non-NaN value (hb*hb - c cannot be -0.0); a NaN discriminant
 * is a miss here (the sign of an invalid-operation NaN is platform-defined). */
static int sphere_hit(const bto_object *o, int oi, v3 tr, const ray_t *ray, const clip_t *clip, manifold_t *out) {
    v3 oc = vsub(ray->origin, tr);
    float half_b = vdot(oc, ray->direction);
    float c = vlen2(oc) - o->radius * o->radius;
    float disc = half_b * half_b - c;
    if (!(disc >= 0.0f)) return 0;
    float sqrtd = sqrtf(disc);
    float t = -half_b - sqrtd;
    if (t < clip->min || t > clip->max) {
        t = -half_b + sqrtd;
        if (t < clip->min || t > clip->max) return 0;
    }
    *out = sphere_surface_manifold(o, oi, tr, *ray, t);
    return 1;
}
/* sphere.rs:150-166 (+ :63-83) */
static int sphere_hit_volumetric(const bto_object *o, int oi, v3 tr, const ray_t *ray, const clip_t *clip, manifold_t *out) {
    float t = clip->max;
    v3 d = vsub(ray_at(ray, t), tr);
    float dist_sqr = vlen2(d);
    float r_sqr = o->radius * o->radius;
    if (dist_sqr <= r_sqr) {
        manifold_t m;
        m.position = ray_at(ray, t);
        m.normal = V3(0, 0, 0);
        sphere_bbox(o->radius, tr, &m.bbox_min, &m.bbox_max);
        m.face = FACE_VOLUME;
        m.t = t;
        m.ray = *ray;
        m.object = oi;
        m.mat = o->material;
        m.vol = o->volume;
        m.depth_extra = 0.0f;
        *out = m;
        return 1;
    }
    return sphere_hit(o, oi, tr, ray, clip, out);
}

/* rect.rs:74-80: project_onto_normalized(n) = n * dot(p, n) */
static inline int rect_contains_point(const bto_rect *r, v3 p) {
    v3 x = vscale(r->x, vdot(p, r->x));
    v3 y = vscale(r->y, vdot(p, r->y));
    float w_sqr = r->half_width * r->half_width;
    float h_sqr = r->half_height * r->half_height;
    return vlen2(x) <= w_sqr && vlen2(y) <= h_sqr;
}
/* rect.rs:38-56 */
static void rect_bbox(const bto_rect *r, const bto_affine *tf, v3 *mn, v3 *mx) {
    v3 xw = vscale(r->x, r->half_width), yh = vscale(r->y, r->half_height);
    v3 nxw = vscale(vneg(r->x), r->half_width);
    v3 p[4] = {xf_point(tf, vadd(xw, yh)), xf_point(tf, vsub(xw, yh)), xf_point(tf, vadd(nxw, yh)),
               xf_point(tf, vsub(nxw, yh))};
    *mn = V3(INFINITY, INFINITY, INFINITY);
    *mx = V3(-INFINITY, -INFINITY, -INFINITY);
    for (int i = 0; i < 4; ++i) {
        mn->x = fminf(mn->x, p[i].x); mn->y = fminf(mn->y, p[i].y); mn->z = fminf(mn->z, p[i].z);
        mx->x = fmaxf(mx->x, p[i].x); mx->y = fmaxf(mx->y, p[i].y); mx->z = fmaxf(mx->z, p[i].z);
    }
}
/* rect.rs:110-155 */
static int rect_hit(const bto_rect *r, int oi, const bto_affine *tf, const ray_t *ray, const clip_t *clip, manifold_t *out) {
    v3 translation = tf->t;
    v3 normal = xf_vector(tf, r->z);
    float q = vdot(ray->direction, normal);
    if (fabsf(q) <= 1e-5f) return 0;
    float p = vdot(vsub(translation, ray->origin), normal);
    float t = p / q;
    if (t < clip->min || t > clip->max) return 0;
    v3 position = ray_at(ray, t);
    bto_affine inv;
    bto_affine_inverse(tf, &inv);
    if (!rect_contains_point(r, xf_point(&inv, position))) return 0;
    manifold_t m;
    m.position = position;
    if (p < 0.0f) {
        m.normal = normal;
        m.face = FACE_FRONT;
    } else {
        m.normal = vneg(normal);
        m.face = FACE_BACK;
    }
    rect_bbox(r, tf, &m.bbox_min, &m.bbox_max);
    m.t = t;
    m.ray = *ray;
    m.object = oi;
    m.mat = r->material;
    m.vol = -1;
    m.depth_extra = 0.0f;
    *out = m;
    return 1;
}
/* rect.rs:88-90 */
static inline float rect_area(const bto_rect *r) { return 4.0f * r->half_width * r->half_height; }
/* rect.rs:92-108 */
static int rect_pdf(const bto_rect *r, int oi, const bto_affine *tf, const ray_t *ray, const clip_t *clip, float *pdf) {
    manifold_t m;
    if (!rect_hit(r, oi, tf, ray, clip, &m)) return 0;
    float shadow = rect_area(r) * fabsf(vdot(ray->direction, m.normal));
    float dist_sqr = m.t * m.t;
    *pdf = dist_sqr / shadow;
    return 1;
}
/* rect.rs:82-86: Uniform::new_inclusive(-hw, hw), (-hh, hh) */
static v3 rect_random_point(const bto_rect *r, const bto_affine *tf, uint32_t x1, uint32_t x2) {
    float sx = bto_uniform_scale(-r->half_width, r->half_width, 1);
    float sy = bto_uniform_scale(-r->half_height, r->half_height, 1);
    float x = uniform_sample(x1, -r->half_width, sx);
    float y = uniform_sample(x2, -r->half_height, sy);
    return xf_point(tf, vadd(vscale(r->x, x), vscale(r->y, y)));
}

/* cuboid.rs:83-105 */
static int cuboid_hit(const bto_object *o, int oi, const ray_t *ray, const clip_t *clip, manifold_t *out) {
    float t = clip->max;
    int found = 0;
    for (int f = 0; f < 6; ++f) {
        bto_affine tf = xf_translate(&o->world, o->face_offset[f]);
        manifold_t m;
        if (rect_hit(&o->faces[f], oi, &tf, ray, clip, &m)) {
            if (m.t < t) {
                t = m.t;
                *out = m;
                found = 1;
            }
        }
    }
    return found;
}
/* cuboid.rs:56-81 */
static int cuboid_pdf(const bto_object *o, int oi, const ray_t *ray, const clip_t *clip, float *pdf) {
    float t = clip->max;
    int best = -1;
    for (int f = 0; f < 6; ++f) {
        bto_affine tf = xf_translate(&o->world, o->face_offset[f]);
        manifold_t m;
        if (rect_hit(&o->faces[f], oi, &tf, ray, clip, &m)) {
            if (m.t < t) {
                t = m.t;
                best = f;
            }
        }
    }
    if (best < 0) return 0;
    bto_affine tf = xf_translate(&o->world, o->face_offset[best]);
    return rect_pdf(&o->faces[best], oi, &tf, ray, clip, pdf);
}
/* cuboid.rs:47-54: WeightedIndex over the face areas (rand 0.8.5: cumulative sums,
 * one Uniform(0,total) draw, index = number of cumulative weights <= draw) */
static v3 cuboid_random_point(const bto_object *o, uint32_t xface, uint32_t x1, uint32_t x2) {
    float cum[5], total = 0.0f;
    for (int f = 0; f < 6; ++f) {
        total += rect_area(&o->faces[f]);
        if (f < 5) cum[f] = total;
    }
    float scale = bto_uniform_scale(0.0f, total, 0);
    float chosen = uniform_sample(xface, 0.0f, scale);
    int index = 0;
    for (int f = 0; f < 5; ++f)
        if (cum[f] <= chosen) index = f + 1;
    bto_affine tf = xf_translate(&o->world, o->face_offset[index]);
    return rect_random_point(&o->faces[index], &tf, x1, x2);
}

/* object/mod.rs:168-180 */
static int object_hit(const bto_scene *sc, int oi, const ray_t *ray, const clip_t *clip, manifold_t *out) {
    const bto_object *o = &sc->objects[oi];
    switch (o->kind) {
    case BTO_SPHERE: return sphere_hit(o, oi, o->world.t, ray, clip, out);
    case BTO_RECT: return rect_hit(&o->rect, oi, &o->world, ray, clip, out);
    case BTO_CUBOID: return cuboid_hit(o, oi, ray, clip, out);
    default: return 0;
    }
}
/* object/mod.rs:182-198 */
static int object_hit_volumetric(const bto_scene *sc, int oi, const ray_t *ray, const clip_t *clip, manifold_t *out) {
    const bto_object *o = &sc->objects[oi];
    if (o->kind == BTO_SPHERE) return sphere_hit_volumetric(o, oi, o->world.t, ray, clip, out);
    return 0;
}
/* object/mod.rs:154-166 (+ sphere.rs:44-61) */
static int object_pdf(const bto_scene *sc, int oi, const ray_t *ray, const clip_t *clip, float *pdf) {
    const bto_object *o = &sc->objects[oi];
    switch (o->kind) {
    case BTO_SPHERE: {
        manifold_t m;
        if (!sphere_hit(o, oi, o->world.t, ray, clip, &m)) return 0;
        float r = o->radius;
        float shadow = 3.14159265358979323846f * r * r;
        float dist_sqr = m.t * m.t;
        *pdf = dist_sqr / shadow;
        return 1;
    }
    case BTO_RECT: return rect_pdf(&o->rect, oi, &o->world, ray, clip, pdf);
    case BTO_CUBOID: return cuboid_pdf(o, oi, ray, clip, pdf);
    default: return 0;
    }
}
/* object/mod.rs:145-152; u[0..3] = the event block, extra = overflow block */
static v3 object_random_point(const bto_scene *sc, int oi, const uint32_t u[4], const rng_t *rng) {
    const bto_object *o = &sc->objects[oi];
    switch (o->kind) {
    case BTO_SPHERE: return vadd(o->world.t, vscale(unit_sphere(u[2], u[3]), o->radius)); /* sphere.rs:40-42 */
    case BTO_RECT: return rect_random_point(&o->rect, &o->world, u[2], u[3]);
    case BTO_CUBOID: {
        uint32_t e[4];
        rng_event_extra(rng, e);
        return cuboid_random_point(o, e[0], u[2], u[3]);
    }
    default: return o->world.t;
    }
}

int bto_object_hit(const bto_scene *sc, int32_t oi, const float *o, const float *d, float cmin, float cmax,
                   int volumetric, float *t_out, float *pos, float *nrm) {
    ray_t ray = {V3(o[0], o[1], o[2]), V3(d[0], d[1], d[2])};
    clip_t clip = {cmin, cmax};
    manifold_t m;
    int hit = volumetric ? object_hit_volumetric(sc, oi, &ray, &clip, &m) : object_hit(sc, oi, &ray, &clip, &m);
    if (!hit) return -1;
    *t_out = m.t;
    pos[0] = m.position.x; pos[1] = m.position.y; pos[2] = m.position.z;
    nrm[0] = m.normal.x; nrm[1] = m.normal.y; nrm[2] = m.normal.z;
    return m.face;
}
float bto_object_pdf(const bto_scene *sc, int32_t oi, const float *o, const float *d, float cmin, float cmax) {
    ray_t ray = {V3(o[0], o[1], o[2]), V3(d[0], d[1], d[2])};
    clip_t clip = {cmin, cmax};
    float pdf = 0.0f;
    if (!object_pdf(sc, oi, &ray, &clip, &pdf)) return -1.0f;
    return pdf;
}

/* ------------------------------------------------------------------ volume (volume.rs) */
/* volume.rs:119-138 */
static inline float density_index(const bto_scene *sc, const bto_data *d, float fx, float fy, float fz) {
    if (d->width == 0 || d->height == 0 || d->depth == 0) return 0.0f;
    int64_t x = (int64_t)fx, y = (int64_t)fy, z = (int64_t)fz;
    if (x < 0) x = 0;
    if (y < 0) y = 0;
    if (z < 0) z = 0;
    if (x >= d->width || y >= d->height || z >= d->depth) return 0.0f; /* reference asserts */
    int64_t idx = z * d->height * d->width + y * d->width + x;
    return sc->density[d->buffer_offset + idx];
}
/* volume.rs:140-167 (Trilinear) */
static float density_sample(const bto_scene *sc, const bto_data *d, v3 coord) {
    float cx = fminf(fmaxf(coord.x, 0.0f), 1.0f) * d->size[0];
    float cy = fminf(fmaxf(coord.y, 0.0f), 1.0f) * d->size[1];
    float cz = fminf(fmaxf(coord.z, 0.0f), 1.0f) * d->size[2];
    float fx = floorf(cx), fy = floorf(cy), fz = floorf(cz);
    float ux = ceilf(cx), uy = ceilf(cy), uz = ceilf(cz);
    float tx = cx - truncf(cx), ty = cy - truncf(cy), tz = cz - truncf(cz);
    float x0 = density_index(sc, d, fx, fy, fz);
    float x1 = density_index(sc, d, ux, fy, fz);
    float y0 = lerpf(x0, x1, tx);
    x0 = density_index(sc, d, fx, uy, fz);
    x1 = density_index(sc, d, ux, uy, fz);
    float y1 = lerpf(x0, x1, tx);
    float z0 = lerpf(y0, y1, ty);
    x0 = density_index(sc, d, fx, fy, uz);
    x1 = density_index(sc, d, ux, fy, uz);
    y0 = lerpf(x0, x1, tx);
    x0 = density_index(sc, d, fx, uy, uz);
    x1 = density_index(sc, d, ux, uy, uz);
    y1 = lerpf(x0, x1, tx);
    float z1 = lerpf(y0, y1, ty);
    return lerpf(z0, z1, tz);
}
float bto_density_sample(const bto_scene *sc, int32_t di, const float *c) {
    return density_sample(sc, &sc->data[di], V3(c[0], c[1], c[2]));
}
/* volume.rs:26-60.  Slots: [0] Bernoulli(density), [1] Standard f32, [2],[3] UnitSphere. */
static int volume_shade(const bto_scene *sc, const bto_data *vol, rng_t *rng, const manifold_t *m, float step,
                        ray_t *ray_out, colordata_t *cd_out) {
    uint32_t u[4];
    rng_event(rng, u);
    v3 offset = m->bbox_min;
    v3 size = vsub(m->bbox_max, m->bbox_min);
    v3 coord = vdiv(vsub(m->position, offset), size);
    float density = step * density_sample(sc, vol, coord);
    if (density >= 1.0f || bernoulli(u[0], density)) {
        v3 origin = m->position;
        if (m->face == FACE_VOLUME) origin = vsub(origin, vscale(vscale(m->ray.direction, step), u24(u[1])));
        v3 direction = unit_sphere(u[2], u[3]);
        *ray_out = ray_new(origin, direction);
        cd_out->color = V3(0.8f, 0.8f, 0.8f);
        cd_out->albedo = V3(0.8f, 0.8f, 0.8f);
        cd_out->normal = m->normal;
        cd_out->depth = m->t + m->depth_extra;
        return 1;
    }
    *ray_out = ray_new(m->position, m->ray.direction);
    return 0;
}

/* ------------------------------------------------------------------ materials (material.rs) */
static inline v3 albedo_of(const bto_data *d) { return V3(d->albedo[0], d->albedo[1], d->albedo[2]); }
/* material.rs:71-79 */
static inline v3 material_emitted(const bto_data *d) {
    if (d->kind == BTO_FLAT) return albedo_of(d);
    if (d->kind == BTO_EMISSIVE) return vscale(albedo_of(d), d->intensity);
    return V3(0, 0, 0);
}
/* material.rs:301-303 */
static inline float diffuse_pdf(const ray_t *ray, const manifold_t *m) {
    return vdot(m->normal, ray->direction) * 0.318309886183790671538f;
}
/* material.rs:201-209 */
static inline float material_pdf(const bto_data *d, const manifold_t *m, const ray_t *ray) {
    return d->kind == BTO_DIFFUSE ? diffuse_pdf(ray, m) : 1.0f;
}
/* material.rs:313-316 */
static inline float light_pdf(const bto_scene *sc, int light, const ray_t *ray, const clip_t *clip) {
    float p = 0.0f;
    if (!object_pdf(sc, light, ray, clip, &p)) return 0.0f;
    return p;
}

typedef struct { int has_scatter; ray_t scatter; int has_albedo; colordata_t albedo; float pdf; } shader_t;

/* material.rs:81-199 with Pdf::{scatter,pdf} (:221-299) folded in */
static shader_t material_shade(const bto_scene *sc, const bto_data *d, rng_t *rng, const manifold_t *m, const clip_t *clip) {
    shader_t sh;
    memset(&sh, 0, sizeof sh);
    sh.pdf = 1.0f;
    if (d->kind == BTO_EMISSIVE) return sh; /* :193-197 */
    sh.has_albedo = 1;
    sh.albedo.normal = m->normal;
    sh.albedo.depth = m->t + m->depth_extra;
    if (d->kind == BTO_FLAT) { /* :88-97 */
        sh.albedo.color = V3(0, 0, 0);
        sh.albedo.albedo = V3(0, 0, 0);
        return sh;
    }
    sh.albedo.color = albedo_of(d);
    sh.albedo.albedo = albedo_of(d);

    uint32_t u[4];
    rng_event(rng, u);
    ray_t ray;
    float p;
    if (d->kind == BTO_DIFFUSE) {
        /* :106-119: count LIGHT objects, pick one uniformly.  Slot [0]. */
        int count = 0;
        for (int i = 0; i < sc->n_objects; ++i)
            if (sc->objects[i].flags & BTO_FLAG_LIGHT) ++count;
        int index = (int)(((uint64_t)u[0] * (uint64_t)count) >> 32);
        int light = -1;
        for (int i = 0, k = 0; i < sc->n_objects; ++i)
            if (sc->objects[i].flags & BTO_FLAG_LIGHT) {
                if (k == index) { light = i; break; }
                ++k;
            }
        /* :122-123, :269-275: Mix(Diffuse, Light, 0.5): gen_bool(0.5) ? Light : Diffuse.  Slot [1]. */
        if (bernoulli(u[1], 0.5f)) {
            v3 origin = m->position; /* :262-268 */
            v3 direction = vsub(object_random_point(sc, light, u, rng), origin);
            ray = ray_new(origin, direction);
        } else {
            ray = ray_new(m->position, cosine(m->normal, u[2], u[3])); /* :224-230 */
        }
        /* :294-296 */
        p = lerpf(diffuse_pdf(&ray, m), light_pdf(sc, light, &ray, clip), 0.5f);
    } else if (d->kind == BTO_METALLIC) { /* :231-239; slots [0],[1] */
        v3 direction = reflect(m->ray.direction, m->normal);
        v3 fuzz = vscale(unit_hemisphere(m->normal, u[0], u[1]), d->roughness);
        ray = ray_new(m->position, vadd(direction, fuzz));
        p = 1.0f;
    } else { /* Glass :240-261; slots [0] Bernoulli(fresnel), [1],[2] hemisphere */
        float ior = face_is_front(m->face) ? 1.0f / d->ior : d->ior;
        float cos_theta = fminf(vdot(vneg(m->ray.direction), m->normal), 1.0f);
        float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
        float fr = fresnel(m->ray.direction, m->normal, ior);
        v3 direction;
        if (ior * sin_theta > 1.0f || bernoulli(u[0], fr))
            direction = reflect(m->ray.direction, m->normal);
        else
            direction = refract(m->ray.direction, m->normal, ior);
        v3 fuzz = vscale(unit_hemisphere(m->normal, u[1], u[2]), d->roughness);
        ray = ray_new(m->position, vadd(direction, fuzz));
        p = 1.0f;
    }
    /* :279-286: abs_diff_eq(p, 0, 1e-5) -> no scatter, pdf 1 */
    if (fabsf(p - 0.0f) <= 1e-5f) return sh;
    sh.has_scatter = 1;
    sh.scatter = ray;
    sh.pdf = p;
    return sh;
}

/* ------------------------------------------------------------------ integrator (tracer/mod.rs) */
typedef struct {
    const bto_scene *scene;
    bto_config cfg;
    rng_t rng;
} chunk_state_t;

static inline clip_t clip_of(const chunk_state_t *cs) { clip_t c = {cs->cfg.clip_min, cs->cfg.clip_max}; return c; } /* :375-380 */
static inline clip_t clip_volumetric(const chunk_state_t *cs) { clip_t c = {0.0f, cs->cfg.volume_step}; return c; }    /* :382-387 */

/* the object loop of try_hit (:394-399) over a given clip */
static int scan_objects(chunk_state_t *cs, const ray_t *ray, float cmin, float cmax, manifold_t *out) {
    int found = 0;
    clip_t clip = {cmin, cmax};
    for (int i = 0; i < cs->scene->n_objects; ++i) {
        manifold_t m;
        if (object_hit(cs->scene, i, ray, &clip, &m)) {
            clip.max = m.t;
            *out = m;
            found = 1;
        }
    }
    if (found) out->depth_extra = 0.0f;
    return found;
}

/* ---- lens extension (NOT in the reference; see bt_oracle.h) ---- */
typedef struct { v3 c; float rs, step, radius; int max_steps; } lens_t;
static inline lens_t lens_of(const bto_config *cfg) {
    lens_t ln = {V3(cfg->lens_centre[0], cfg->lens_centre[1], cfg->lens_centre[2]), cfg->lens_rs, cfg->lens_step,
                 cfg->lens_radius, cfg->lens_max_steps};
    return ln;
}
/* 1/sqrt(x) of the lens march: integer seed + three Newton steps with explicit fmaf -- no divider, no sqrt unit, the
 * same bits on CPU and GPU; relative error <= ~2e-7 for normal x (the march is an RK4 scheme with truncation error
 * orders of magnitude above that).  The reference has no such code: this extension defines its own arithmetic. */
static inline float lens_rsqrt(float x) {
    union { float f; uint32_t u; } b;
    b.f = x;
    b.u = 0x5f375a86u - (b.u >> 1);
    float y = b.f;
    const float hx = 0.5f * x;
    for (int i = 0; i < 3; ++i) {
        float t = hx * y;
        float e = fmaf(-t, y, 1.5f);
        y = y * e;
    }
    return y;
}
static inline v3 lens_accel(const lens_t *ln, v3 x, float h2) {
    v3 rel = vsub(x, ln->c);
    float r2 = vlen2(rel);
    float y = lens_rsqrt(r2);
    float y2 = y * y;
    float y5 = (y2 * y2) * y;                     /* r^-5 */
    float k = (-1.5f * ln->rs * h2) * y5;
    return vscale(rel, k);
}
static inline void lens_rk4(const lens_t *ln, float h2, v3 x, v3 v, v3 *x1, v3 *v1) {
    float dt = ln->step, hdt = 0.5f * dt, w = dt / 6.0f;
    v3 k1x = v, k1v = lens_accel(ln, x, h2);
    v3 k2x = vadd(v, vscale(k1v, hdt)), k2v = lens_accel(ln, vadd(x, vscale(k1x, hdt)), h2);
    v3 k3x = vadd(v, vscale(k2v, hdt)), k3v = lens_accel(ln, vadd(x, vscale(k2x, hdt)), h2);
    v3 k4x = vadd(v, vscale(k3v, dt)), k4v = lens_accel(ln, vadd(x, vscale(k3x, dt)), h2);
    v3 sx = vadd(vadd(k1x, vscale(vadd(k2x, k3x), 2.0f)), k4x);
    v3 sv = vadd(vadd(k1v, vscale(vadd(k2v, k3v), 2.0f)), k4v);
    *x1 = vadd(x, vscale(sx, w));
    *v1 = vadd(v, vscale(sv, w));
}
/* One bent path segment.  1 = hit (out->ray is the chord that hit, out->t its local t), 0 = miss (root),
 * -1 = captured by the horizon.  `cs` may be NULL (free-space integration); last = the final straight ray. */
static int lens_trace(chunk_state_t *cs, const bto_config *cfg, const ray_t *ray, manifold_t *out, ray_t *last, int *status) {
    const lens_t ln = lens_of(cfg);
    const float R2 = ln.radius * ln.radius, rs2 = ln.rs * ln.rs;
    v3 x = ray->origin, v = ray->direction;
    float remaining = cfg->clip_max, travelled = 0.0f;
    int first = 1, steps_left = ln.max_steps;
    if (status) *status = 0;
    for (;;) {
        v3 rel = vsub(x, ln.c);
        float r2 = vlen2(rel);
        if (!(r2 <= R2)) {
            /* straight flight to the sphere of influence (or to the end of the clip) */
            float hb = vdot(rel, v), cc = r2 - R2, disc = hb * hb - cc;
            float t_enter = INFINITY;
            if (disc >= 0.0f) {
                float te = -hb - sqrtf(disc);
                if (te > 0.0f) t_enter = te;
            }
            float seg = fminf(t_enter, remaining);
            ray_t sr = {x, v};
            *last = sr;
            if (cs && scan_objects(cs, &sr, first ? cfg->clip_min : 0.0f, seg, out)) {
                out->depth_extra = travelled;
                return 1;
            }
            if (!(t_enter < remaining)) return 0;
            x = vadd(x, vscale(v, t_enter));
            remaining -= t_enter;
            travelled += t_enter;
            first = 0;
        }
        float h2 = vlen2(vcross(vsub(x, ln.c), v));
        for (;;) {
            if (steps_left-- <= 0) {               /* step budget exhausted: the segment is abandoned as a miss */
                ray_t lr = {x, vnormalize(v)};
                *last = lr;
                if (status) *status = 2;
                return 0;
            }
            v3 x1, v1;
            lens_rk4(&ln, h2, x, v, &x1, &v1);
            v3 chord = vsub(x1, x);
            float l2 = vlen2(chord), rl = lens_rsqrt(l2);
            float len = l2 * rl;
            ray_t sr = {x, vscale(chord, rl)};
            *last = sr;
            float seg = fminf(len, remaining);
            if (cs && scan_objects(cs, &sr, first ? cfg->clip_min : 0.0f, seg, out)) {
                out->depth_extra = travelled;
                return 1;
            }
            if (!(len < remaining)) { if (status) *status = 2; return 0; }
            remaining -= len;
            travelled += len;
            first = 0;
            x = x1;
            v = v1;
            rel = vsub(x, ln.c);
            r2 = vlen2(rel);
            if (r2 <= rs2) { if (status) *status = 1; return -1; }
            if (r2 > R2 && vdot(rel, v) > 0.0f) {
                v = vnormalize(v);
                break;
            }
        }
    }
}
int bto_lens_trace_free(const bto_config *cfg, const float *o, const float *d, float *pos_out, float *dir_out) {
    ray_t ray = {V3(o[0], o[1], o[2]), vnormalize(V3(d[0], d[1], d[2]))}, last = ray;
    manifold_t m;
    int status = 0;
    lens_trace(NULL, cfg, &ray, &m, &last, &status);
    pos_out[0] = last.origin.x; pos_out[1] = last.origin.y; pos_out[2] = last.origin.z;
    dir_out[0] = last.direction.x; dir_out[1] = last.direction.y; dir_out[2] = last.direction.z;
    return status;
}

/* :389-402.  With the lens extension on: 1 hit, 0 miss, -1 captured; `*last` is the ray that reaches the root. */
static int try_hit(chunk_state_t *cs, const ray_t *ray, manifold_t *out, ray_t *last) {
    cs->rng.segments += 1;
    *last = *ray;
    if (cs->cfg.lens_on) return lens_trace(cs, &cs->cfg, ray, out, last, NULL);
    return scan_objects(cs, ray, cs->cfg.clip_min, cs->cfg.clip_max, out);
}
/* :404-427 */
static int try_hit_volume(chunk_state_t *cs, const ray_t *ray, int last_object, manifold_t *out) {
    int found = 0;
    clip_t clip = clip_volumetric(cs);
    cs->rng.segments += 1;
    for (int i = 0; i < cs->scene->n_objects; ++i) {
        manifold_t m;
        int hit = (i == last_object) ? object_hit_volumetric(cs->scene, i, ray, &clip, &m)
                                     : object_hit(cs->scene, i, ray, &clip, &m);
        if (hit) {
            clip.max = m.t;
            *out = m;
            found = 1;
        }
    }
    if (found) out->depth_extra = 0.0f;
    return found;
}

static colordata_t sample(chunk_state_t *cs, const ray_t *ray, int bounce);
static colordata_t sample_volumetric(chunk_state_t *cs, const ray_t *ray, int last_object, int bounce, int volume_bounce);

/* :429-452 */
static colordata_t sample_root(chunk_state_t *cs, const ray_t *ray) {
    const bto_data *material = &cs->scene->data[cs->scene->root_material];
    manifold_t m;
    m.position = ray_at(ray, cs->cfg.clip_max);
    m.normal = vneg(ray->direction);
    m.bbox_min = V3(-INFINITY, -INFINITY, -INFINITY);
    m.bbox_max = V3(INFINITY, INFINITY, INFINITY);
    m.face = FACE_VOLUME;
    m.t = cs->cfg.clip_max;
    m.ray = *ray;
    m.object = m.mat = m.vol = -1;
    m.depth_extra = 0.0f;
    clip_t clip = clip_of(cs);
    v3 emitted = material_emitted(material);
    shader_t data = material_shade(cs->scene, material, &cs->rng, &m, &clip);
    colordata_t cd = data.has_albedo ? data.albedo : colordata_default();
    cd.color = vadd(cd.color, emitted);
    return cd;
}
/* :454-486 */
static colordata_t sample_surface(chunk_state_t *cs, const manifold_t *m, int mat, int bounce) {
    const bto_data *material = &cs->scene->data[mat];
    clip_t clip = clip_of(cs);
    v3 emitted = material_emitted(material);
    shader_t data = material_shade(cs->scene, material, &cs->rng, m, &clip);
    if (data.has_scatter) {
        colordata_t reflected = sample(cs, &data.scatter, bounce + 1);
        colordata_t cd;
        if (data.has_albedo) {
            cd = data.albedo;
            cd.color = vscale(cd.color, material_pdf(material, m, &data.scatter));
            cd.color = vmul(cd.color, vdivs(reflected.color, data.pdf));
        } else {
            cd = reflected;
        }
        cd.color = vadd(cd.color, emitted);
        return cd;
    }
    return colordata_from_emitted(emitted);
}
/* :488-523 */
static colordata_t sample_volume(chunk_state_t *cs, const manifold_t *m, int vol, int bounce, int volume_bounce) {
    const bto_data *volume = &cs->scene->data[vol];
    ray_t ray;
    colordata_t att;
    int has_att = volume_shade(cs->scene, volume, &cs->rng, m, cs->cfg.volume_step, &ray, &att);
    colordata_t reflected;
    if (m->face == FACE_VOLUME_BACK)
        reflected = sample(cs, &ray, bounce + 1);
    else
        reflected = sample_volumetric(cs, &ray, m->object, bounce, volume_bounce + 1);
    if (has_att) {
        att.color = vmul(att.color, reflected.color);
        return att;
    }
    return reflected;
}
/* :322-342 */
static colordata_t sample(chunk_state_t *cs, const ray_t *ray, int bounce) {
    if (bounce > cs->cfg.max_bounces) return colordata_default();
    manifold_t m;
    ray_t last;
    int hit = try_hit(cs, ray, &m, &last);
    if (hit < 0) return colordata_default();      /* lens extension: swallowed by the horizon */
    if (hit) {
        if (face_is_surface(m.face)) {
            if (m.mat >= 0) return sample_surface(cs, &m, m.mat, bounce);
            return colordata_default();
        }
        if (m.vol >= 0) return sample_volume(cs, &m, m.vol, bounce, 0);
        return colordata_default();
    }
    return sample_root(cs, &last);
}
/* :344-373 */
static colordata_t sample_volumetric(chunk_state_t *cs, const ray_t *ray, int last_object, int bounce, int volume_bounce) {
    if (volume_bounce > cs->cfg.max_volume_bounces) return colordata_default();
    manifold_t m;
    if (try_hit_volume(cs, ray, last_object, &m)) {
        if (face_is_surface(m.face)) {
            if (m.mat >= 0) return sample_surface(cs, &m, m.mat, bounce);
            return colordata_default();
        }
        if (m.vol >= 0) return sample_volume(cs, &m, m.vol, bounce, volume_bounce);
        return colordata_default();
    }
    return sample_root(cs, ray);
}

/* the same estimator unrolled into a loop (throughput beta, radiance L);
 * albedo/normal/depth come from the first non-pass-through ColorData. */
static colordata_t sample_iterative(chunk_state_t *cs, const ray_t *ray0) {
    ray_t ray = *ray0;
    v3 beta = V3(1, 1, 1), L = V3(0, 0, 0);
    colordata_t first = colordata_default();
    int have_first = 0;
    int bounce = 0, volume_bounce = 0, marching = 0, last_object = -1;
    for (;;) {
        manifold_t m;
        int hit;
        if (!marching) {
            if (bounce > cs->cfg.max_bounces) break;
            ray_t last;
            hit = try_hit(cs, &ray, &m, &last);
            if (hit < 0) break;                   /* lens extension: swallowed by the horizon */
            ray = last;
            volume_bounce = 0;
        } else {
            if (volume_bounce > cs->cfg.max_volume_bounces) break;
            hit = try_hit_volume(cs, &ray, last_object, &m);
        }
        if (!hit) {
            colordata_t root = sample_root(cs, &ray);
            L = vadd(L, vmul(beta, root.color));
            if (!have_first) { first = root; have_first = 1; }
            break;
        }
        if (face_is_surface(m.face)) {
            if (m.mat < 0) break;
            const bto_data *material = &cs->scene->data[m.mat];
            clip_t clip = clip_of(cs);
            v3 emitted = material_emitted(material);
            shader_t data = material_shade(cs->scene, material, &cs->rng, &m, &clip);
            L = vadd(L, vmul(beta, emitted));
            if (!data.has_scatter) {
                if (!have_first) { first = colordata_from_emitted(emitted); have_first = 1; }
                break;
            }
            if (data.has_albedo) {
                float weight = material_pdf(material, &m, &data.scatter) / data.pdf;
                beta = vmul(beta, vscale(data.albedo.color, weight));
                if (!have_first) { first = data.albedo; have_first = 1; }
            }
            ray = data.scatter;
            bounce += 1;
            marching = 0;
        } else {
            if (m.vol < 0) break;
            ray_t next;
            colordata_t att;
            int has_att = volume_shade(cs->scene, &cs->scene->data[m.vol], &cs->rng, &m, cs->cfg.volume_step, &next, &att);
            if (has_att) {
                beta = vmul(beta, att.color);
                if (!have_first) { first = att; have_first = 1; }
            }
            ray = next;
            if (m.face == FACE_VOLUME_BACK) {
                bounce += 1;
                marching = 0;
            } else {
                marching = 1;
                last_object = m.object;
                volume_bounce += 1;
            }
        }
    }
    first.color = L;
    return first;
}

/* camera ray of one sample: mod.rs:248-302.  Slots: [0] jitter u, [1] jitter v,
 * [2] disk angle, [3] disk radius. */
typedef struct {
    float yfov, xfov, pixel_width, pixel_height, subpixel_scale;
    float jitter_u_lo, jitter_u_scale, jitter_v_lo, jitter_v_scale;
    int n;
} camera_setup_t;

static void camera_setup(const bto_object *cam, const bto_config *cfg, uint32_t width, uint32_t height, camera_setup_t *cs) {
    cs->yfov = 2.0f * atan2f(cam->sensor_size, 2.0f * cam->focal_length); /* :248 */
    cs->xfov = cs->yfov * cam->aspect_ratio;                              /* :249 */
    cs->pixel_width = 2.0f * (1.0f / (float)width);                       /* buffer.rs:68-71 */
    cs->pixel_height = 2.0f * (1.0f / (float)height);                     /* buffer.rs:73-76 */
    cs->n = cfg->subsample_n >= 2 ? cfg->subsample_n : 1;
    cs->subpixel_scale = cfg->subsample_n >= 2 ? 1.0f / (float)cfg->subsample_n : 1.0f; /* :55-60 */
    float umin = -0.5f * cs->pixel_width * cs->subpixel_scale, umax = 0.5f * cs->pixel_width * cs->subpixel_scale;
    float vmin = -0.5f * cs->pixel_height * cs->subpixel_scale, vmax = 0.5f * cs->pixel_height * cs->subpixel_scale;
    cs->jitter_u_lo = umin;
    cs->jitter_u_scale = bto_uniform_scale(umin, umax, 0); /* :255-259 */
    cs->jitter_v_lo = vmin;
    cs->jitter_v_scale = bto_uniform_scale(vmin, vmax, 0); /* :261-265 */
}

static colordata_t trace_sample(chunk_state_t *st, const bto_object *cam, const camera_setup_t *cs, uint32_t x, uint32_t y,
                                uint32_t width, uint32_t sample_index) {
    st->rng.pixel = y * width + x;
    st->rng.sample = sample_index;
    st->rng.event = 0;
    uint32_t sub = sample_index % (uint32_t)(cs->n * cs->n);
    float width_sub = 1.0f / (float)cs->n; /* :97 */
    float u_sub = cs->n > 1 ? (float)(sub % (uint32_t)cs->n) * width_sub : 0.0f; /* :98-101 */
    float v_sub = cs->n > 1 ? (float)(sub / (uint32_t)cs->n) * width_sub : 0.0f;
    uint32_t r[4];
    rng_event(&st->rng, r);
    float v0 = (float)y * cs->pixel_height - 1.0f; /* :272 */
    float u0 = (float)x * cs->pixel_width - 1.0f;  /* :275 */
    float u_offset = u_sub * cs->pixel_width + uniform_sample(r[0], cs->jitter_u_lo, cs->jitter_u_scale);  /* :279 */
    float v_offset = v_sub * cs->pixel_height + uniform_sample(r[1], cs->jitter_v_lo, cs->jitter_v_scale); /* :280 */
    float u = u0 + u_offset, v = v0 + v_offset;
    ray_t ray = {V3(0, 0, 0), frustum_direction(cs->yfov, cs->xfov, u, v)}; /* :285 */
    if (cam->has_focus) { /* :286-299 */
        v3 defocus = unit_disk(V3(0, 0, -1), r[2], r[3]);
        float aperture = 0.5f * cam->focal_length / cam->fstop;
        v3 defocus_offset = xf_vector(&cam->world, vscale(defocus, aperture));
        float frac_f_z = cam->focus / fabsf(ray.direction.z);
        ray = affine_mul_ray(&cam->world, &ray);
        ray.origin = vadd(ray.origin, defocus_offset);
        ray.direction = vnormalize(vsub(vscale(ray.direction, frac_f_z), defocus_offset));
    } else {
        ray = affine_mul_ray(&cam->world, &ray); /* :301 */
    }
    return st->cfg.recursive ? sample(st, &ray, 0) : sample_iterative(st, &ray); /* :304 */
}

/* buffer.rs:102-115, 293-326 */
void bto_chunk_bounds(uint32_t width, uint32_t height, int32_t chunks_x, int32_t chunks_y, int32_t *n_chunks, uint32_t *bounds) {
    uint32_t cw = width % chunks_x == 0 ? width / chunks_x : width / chunks_x + 1;
    uint32_t ch = height % chunks_y == 0 ? height / chunks_y : height / chunks_y + 1;
    uint32_t ox = 0, oy = 0;
    int n = 0, done = (width == 0 || height == 0);
    while (!done) {
        uint32_t w = cw < width - ox ? cw : width - ox;
        uint32_t h = ch < height - oy ? ch : height - oy;
        if (bounds) {
            bounds[4 * n + 0] = ox; bounds[4 * n + 1] = oy;
            bounds[4 * n + 2] = ox + w; bounds[4 * n + 3] = oy + h;
        }
        ++n;
        ox += w;
        if (ox == width) { ox = 0; oy += h; }
        if (oy == height) done = 1;
    }
    *n_chunks = n;
}

typedef struct {
    const bto_scene *scene;
    const bto_object *cam;
    bto_config cfg;
    camera_setup_t cam_setup;
    float *rgba;
    uint32_t width, height;
    uint64_t seed;
    int n_chunks;
    uint32_t *bounds;
    int next_chunk;
    pthread_mutex_t lock;
    uint64_t segments;
} job_t;

/* ChunkState::render_samples, mod.rs:244-320 */
static void render_chunk(job_t *job, const uint32_t *b, uint64_t *segments) {
    chunk_state_t st;
    st.scene = job->scene;
    st.cfg = job->cfg;
    st.rng.key[0] = (uint32_t)job->seed;
    st.rng.key[1] = (uint32_t)(job->seed >> 32);
    st.rng.segments = 0;
    int nn = job->cam_setup.n * job->cam_setup.n;
    for (uint32_t y = b[1]; y < b[3]; ++y)
        for (uint32_t x = b[0]; x < b[2]; ++x) {
            float *px = job->rgba + 4 * ((size_t)y * job->width + x);
            for (int s = 0; s < job->cfg.samples; ++s)
                for (int j = 0; j < nn; ++j) {
                    uint32_t sample_index = (job->cfg.sample_base + (uint32_t)s) * (uint32_t)nn + (uint32_t)j;
                    colordata_t cd = trace_sample(&st, job->cam, &job->cam_setup, x, y, job->width, sample_index);
                    float depth = (cd.depth - job->cfg.clip_min) / (job->cfg.clip_max - job->cfg.clip_min); /* :306-308 */
                    depth = fminf(fmaxf(depth, 0.0f), 1.0f);
                    switch (job->cfg.output) { /* :310-315 -> buffer.rs:159-178 */
                    case BTO_OUT_FULL: px[0] += cd.color.x; px[1] += cd.color.y; px[2] += cd.color.z; break;
                    case BTO_OUT_ALBEDO: px[0] += cd.albedo.x; px[1] += cd.albedo.y; px[2] += cd.albedo.z; break;
                    case BTO_OUT_NORMAL: px[0] += cd.normal.x; px[1] += cd.normal.y; px[2] += cd.normal.z; break;
                    default: px[0] += depth; px[1] += depth; px[2] += depth; break;
                    }
                }
        }
    *segments = st.rng.segments;
}

static void *worker(void *arg) {
    job_t *job = (job_t *)arg;
    uint64_t total = 0;
    for (;;) {
        pthread_mutex_lock(&job->lock);
        int c = job->next_chunk++;
        pthread_mutex_unlock(&job->lock);
        if (c >= job->n_chunks) break;
        uint64_t seg = 0;
        render_chunk(job, job->bounds + 4 * c, &seg);
        total += seg;
    }
    pthread_mutex_lock(&job->lock);
    job->segments += total;
    pthread_mutex_unlock(&job->lock);
    return NULL;
}

static int check_scene(const bto_scene *scene, int32_t camera_index) {
    if (!scene || camera_index < 0 || camera_index >= scene->n_objects) return -1;
    if (scene->objects[camera_index].kind != BTO_CAMERA) return -2; /* mod.rs:246 */
    if (scene->root_material < 0 || scene->root_material >= scene->n_data) return -3;
    return 0;
}

/* Tracer::render, mod.rs:179-202 */
int bto_render(const bto_scene *scene, int32_t camera_index, const bto_config *cfg, float *rgba, uint32_t width,
               uint32_t height, uint64_t seed, int32_t nthreads, uint64_t *segments_out) {
    if (segments_out) *segments_out = 0;
    int rc = check_scene(scene, camera_index);
    if (rc) return rc;
    if (cfg->samples == 0) return 0; /* :186-188 */
    if (width == 0 || height == 0) return -4;
    pthread_once(&g_distr_once, distr_init);

    job_t job;
    memset(&job, 0, sizeof job);
    job.scene = scene;
    job.cam = &scene->objects[camera_index];
    job.cfg = *cfg;
    camera_setup(job.cam, cfg, width, height, &job.cam_setup);
    job.rgba = rgba;
    job.width = width;
    job.height = height;
    job.seed = seed;
    int cx = cfg->chunks_x > 0 ? cfg->chunks_x : 1, cy = cfg->chunks_y > 0 ? cfg->chunks_y : 1;
    bto_chunk_bounds(width, height, cx, cy, &job.n_chunks, NULL);
    job.bounds = (uint32_t *)malloc(sizeof(uint32_t) * 4 * (size_t)job.n_chunks);
    bto_chunk_bounds(width, height, cx, cy, &job.n_chunks, job.bounds);
    pthread_mutex_init(&job.lock, NULL);

    if (nthreads <= 1) {
        worker(&job);
    } else {
        pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)nthreads);
        for (int i = 0; i < nthreads; ++i) pthread_create(&th[i], NULL, worker, &job);
        for (int i = 0; i < nthreads; ++i) pthread_join(th[i], NULL);
        free(th);
    }
    if (segments_out) *segments_out = job.segments;
    pthread_mutex_destroy(&job.lock);
    free(job.bounds);
    return 1; /* :201 Status::InProgress */
}

int bto_trace_one(const bto_scene *scene, int32_t camera_index, const bto_config *cfg, uint32_t width, uint32_t height,
                  uint32_t px, uint32_t py, uint32_t sample_index, uint64_t seed, float *out10) {
    int rc = check_scene(scene, camera_index);
    if (rc) return rc;
    pthread_once(&g_distr_once, distr_init);
    chunk_state_t st;
    st.scene = scene;
    st.cfg = *cfg;
    st.rng.key[0] = (uint32_t)seed;
    st.rng.key[1] = (uint32_t)(seed >> 32);
    st.rng.segments = 0;
    camera_setup_t cs;
    camera_setup(&scene->objects[camera_index], cfg, width, height, &cs);
    colordata_t cd = trace_sample(&st, &scene->objects[camera_index], &cs, px, py, width, sample_index);
    out10[0] = cd.color.x; out10[1] = cd.color.y; out10[2] = cd.color.z;
    out10[3] = cd.albedo.x; out10[4] = cd.albedo.y; out10[5] = cd.albedo.z;
    out10[6] = cd.normal.x; out10[7] = cd.normal.y; out10[8] = cd.normal.z;
    out10[9] = cd.depth;
    return 1;
}

/* ------------------------------------------------------------------ resolve (buffer.rs:117-138) */
/* Numerics contract N9: x^(1/2.4) = exp2(log2(x) / 2.4) with the polynomials below (explicit fmaf),
 * identical on the GPU; max relative error 2.6e-7 on (0.0031308, 1].  The reference calls libm powf
 * (color.rs:18); the two agree to 4 ulp, i.e. to the same u8 except at a handful of values. */
static inline float bt_log2f(float x) {
    uint32_t xi;
    memcpy(&xi, &x, 4);
    int e = (int)((xi >> 23) & 0xffu) - 127;
    uint32_t mi = (xi & 0x7fffffu) | 0x3f800000u;
    float m;
    memcpy(&m, &mi, 4);
    if (m > 1.41421356f) { m = m * 0.5f; e += 1; }
    float t = m - 1.0f;
    float s = t / (2.0f + t), s2 = s * s;            /* ln(1+t) = 2 atanh(s) */
    float p = fmaf(s2, 0.0909090909f, 0.1111111111f);
    p = fmaf(p, s2, 0.1428571429f);
    p = fmaf(p, s2, 0.2f);
    p = fmaf(p, s2, 0.3333333333f);
    float ln = fmaf(p * s2, s, s) * 2.0f;
    return fmaf(ln, 1.4426950408889634f, (float)e);
}
static inline float bt_exp2f(float y) {
    float k = rintf(y), r = y - k;
    float z = r * 0.6931471805599453f;
    float p = fmaf(z, 1.984126984e-4f, 1.388888889e-3f);
    p = fmaf(p, z, 8.333333333e-3f);
    p = fmaf(p, z, 4.166666667e-2f);
    p = fmaf(p, z, 1.666666667e-1f);
    p = fmaf(p, z, 0.5f);
    p = fmaf(p, z, 1.0f);
    p = fmaf(p, z, 1.0f);
    int ki = (int)k;
    if (ki < -126) return 0.0f;
    if (ki > 127) return INFINITY;
    uint32_t bits = (uint32_t)(ki + 127) << 23;
    float scale;
    memcpy(&scale, &bits, 4);
    return p * scale;
}
/* color.rs:14-20 */
static inline float linear_to_srgb(float x) {
    if (x <= 0.0031308f) return 12.92f * x;
    if (!(x < 3.0e38f)) return x;                    /* +inf / NaN pass through */
    return 1.055f * bt_exp2f(bt_log2f(x) * (1.0f / 2.4f)) - 0.055f;
}
/* color.rs:22-24: `(x * 255.0) as u8` saturating, NaN -> 0 */
static inline uint8_t f32_to_u8(float x) {
    float v = x * 255.0f;
    if (!(v > 0.0f)) return 0;
    if (v >= 255.0f) return 255;
    return (uint8_t)v;
}
/* color_space: 0 None, 1 Normal, 2 Linear, 3 SRgb (buffer.rs:11-30) */
void bto_preview(const float *rgba, uint32_t n_pixels, uint32_t samples, int32_t color_space, uint8_t *out) {
    float recip = 1.0f / (float)samples;
    for (uint32_t i = 0; i < n_pixels; ++i) {
        v3 rgb = vscale(V3(rgba[4 * i], rgba[4 * i + 1], rgba[4 * i + 2]), recip);
        if (color_space == 1) {
            v3 n = vnormalize(rgb);
            rgb = vscale(vadd(n, V3(1, 1, 1)), 0.5f);
        } else if (color_space == 3) {
            rgb = V3(linear_to_srgb(rgb.x), linear_to_srgb(rgb.y), linear_to_srgb(rgb.z));
        }
        out[4 * i + 0] = f32_to_u8(rgb.x);
        out[4 * i + 1] = f32_to_u8(rgb.y);
        out[4 * i + 2] = f32_to_u8(rgb.z);
        out[4 * i + 3] = f32_to_u8(rgba[4 * i + 3]);
    }
}
