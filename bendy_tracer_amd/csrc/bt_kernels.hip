// bt_kernels.hip -- gfx950 (MI355X / CDNA4) kernels for bendy-tracer's per-sample hot path.
//
// One launch of bt_render_kernel computes Tracer::render (reference tracer/mod.rs:179-202)
// for every pixel of the frame (or of this rank's tile shard):
//   camera ray (mod.rs:271-302) -> iterative form of sample / sample_surface /
//   sample_volume / sample_volumetric / sample_root (mod.rs:322-523) with
//   try_hit / try_hit_volume over the flattened primitive table (mod.rs:389-427,
//   sphere.rs, rect.rs, cuboid.rs), material shading (material.rs) and the density-map
//   march (volume.rs) -> `+=` into the RGBA32F accumulator (buffer.rs:159-178).
//
// Mapping to the hardware (DESIGN.md "Kernel"):
//   * one lane owns one pixel and walks that pixel's samples in order, so the per-pixel
//     float sum has the reference's order and needs no atomics;
//   * a wave is an 8x8 pixel tile; lanes regenerate a camera ray as soon as their path
//     ends (per-lane sample counter), so a wave never idles on its longest path;
//   * the primitive table is read with wave-uniform indices -> scalar (SMEM) loads that
//     broadcast through SGPRs; per-lane lookups (hit primitive, material, light, density)
//     go to tables staged in LDS;
//   * no MFMA: there is no dense contraction on this path.
//
// Arithmetic follows DESIGN.md's numerics contract so that geometry decisions are
// bit-identical to the CPU oracle: no FMA contraction, explicit fmaf only in sincos,
// correctly rounded sqrt / divide (hipcc default), fixed operation order.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bt_types.h"

#pragma clang fp contract(off)

#define BT_DEV static __device__ __forceinline__

namespace {

struct V3 { float x, y, z; };
BT_DEV V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
BT_DEV V3 mk(const BtV3 &a) { return mk(a.x, a.y, a.z); }
BT_DEV V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
BT_DEV V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
BT_DEV V3 operator*(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
BT_DEV V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
BT_DEV V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
BT_DEV float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
BT_DEV float len2(V3 a) { return dot(a, a); }
BT_DEV V3 normalize(V3 a) { float rl = 1.0f / sqrtf(len2(a)); return a * rl; }
BT_DEV V3 normalize_or_zero(V3 a) {
    float rl = 1.0f / sqrtf(len2(a));
    bool ok = (rl > 0.0f) && (rl < __builtin_inff());
    return ok ? a * rl : mk(0.0f, 0.0f, 0.0f);
}
// M*v with columns cx,cy,cz (glam Affine3A::transform_vector3a)
BT_DEV V3 xf_vector(V3 cx, V3 cy, V3 cz, V3 v) { return (cx * v.x + cy * v.y) + cz * v.z; }

// ---- sin/cos (numerics contract N5) ------------------------------------------------
BT_DEV void sincos_bt(float x, float &s, float &c) {
    float k = __builtin_rintf(x * 0.636619772f);
    float r = __builtin_fmaf(k, -1.5703125f, x);
    r = __builtin_fmaf(k, -4.837512969970703125e-4f, r);
    r = __builtin_fmaf(k, -7.54978995489188e-8f, r);
    float r2 = r * r;
    float ps = __builtin_fmaf(__builtin_fmaf(-1.9515295891e-4f, r2, 8.3321608736e-3f), r2, -1.6666654611e-1f);
    float sn = __builtin_fmaf(ps * r2, r, r);
    float pc = __builtin_fmaf(__builtin_fmaf(2.443315711809948e-5f, r2, -1.388731625493765e-3f), r2,
                              4.166664568298827e-2f);
    float cs = __builtin_fmaf(pc, r2 * r2, __builtin_fmaf(-0.5f, r2, 1.0f));
    int q = ((int)k) & 3;
    float so = (q & 1) ? cs : sn;
    float co = (q & 1) ? sn : cs;
    s = (q & 2) ? -so : so;
    c = ((q + 1) & 2) ? -co : co;
}

// ---- Philox4x32-10 (numerics contract N6) --------------------------------------------
struct U4 { uint32_t x, y, z, w; };
BT_DEV U4 philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        c1 = (uint32_t)p1;
        c3 = (uint32_t)p0;
        c0 = n0;
        c2 = n2;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    U4 r; r.x = c0; r.y = c1; r.z = c2; r.w = c3;
    return r;
}
BT_DEV float u23(uint32_t x) { return __uint_as_float((x >> 9) | 0x3F800000u) - 1.0f; }
BT_DEV float u24(uint32_t x) { return (float)(x >> 8) * 5.9604644775390625e-8f; }
BT_DEV bool bernoulli(uint32_t x, float p) { return u24(x) < p; }
BT_DEV float uniform_sample(uint32_t x, float lo, float scale) { return u23(x) * scale + lo; }

// ---- math/mod.rs ----------------------------------------------------------------------
BT_DEV float lerpf(float a, float b, float f) { return a + (b - a) * f; }
BT_DEV V3 reflect(V3 v, V3 n) { return v - n * (2.0f * dot(v, n)); }
BT_DEV V3 refract(V3 v, V3 n, float ior) {
    float cos_theta = fminf(dot(-v, n), 1.0f);
    V3 perp = (n * cos_theta + v) * ior;
    V3 parallel = n * -sqrtf(fabsf(1.0f - len2(perp)));
    return perp + parallel;
}
BT_DEV float fresnel(V3 v, V3 n, float ior) {
    float cos_theta = fminf(dot(-v, n), 1.0f);
    float r0 = (1.0f - ior) / (1.0f + ior);
    r0 = r0 * r0;
    float x = 1.0f - cos_theta;
    float x2 = x * x;
    return r0 + (1.0f - r0) * ((x2 * x2) * x);
}
// glam any_orthonormal_pair (Duff et al.)
BT_DEV void orthonormal_pair(V3 n, V3 &t1, V3 &t2) {
    float sign = __builtin_copysignf(1.0f, n.z);
    float a = -1.0f / (sign + n.z);
    float b = n.x * n.y * a;
    t1 = mk(1.0f + sign * n.x * n.x * a, sign * b, -sign * n.x);
    t2 = mk(b, sign + n.y * n.y * a, -n.y);
}

// ---- math/distr.rs ---------------------------------------------------------------------
BT_DEV V3 unit_sphere(const BtLaunch &P, uint32_t x1, uint32_t x2) {
    float r1 = uniform_sample(x1, 0.0f, P.tau_scale), r2 = uniform_sample(x2, 0.0f, P.one_scale);
    float s, c;
    sincos_bt(r1, s, c);
    float x = c * 2.0f * sqrtf(r2 * (1.0f - r2));
    float y = s * 2.0f * sqrtf(r2 * (1.0f - r2));
    float z = 1.0f - 2.0f * r2;
    return mk(x, y, z);
}
BT_DEV V3 unit_hemisphere(const BtLaunch &P, V3 normal, uint32_t x1, uint32_t x2) {
    V3 z_axis = normalize(normal), x_axis, y_axis;
    orthonormal_pair(z_axis, x_axis, y_axis);
    float r1 = uniform_sample(x1, 0.0f, P.tau_scale), r2 = uniform_sample(x2, 0.0f, P.one_scale);
    float s, c;
    sincos_bt(r1, s, c);
    float x = c * 2.0f * sqrtf(r2 * (1.0f - r2));
    float y = s * 2.0f * sqrtf(r2 * (1.0f - r2));
    float z = 1.0f - r2;
    return (x_axis * x + y_axis * y) + z_axis * z;
}
BT_DEV V3 cosine(const BtLaunch &P, V3 normal, uint32_t x1, uint32_t x2) {
    V3 z_axis = normalize(normal), x_axis, y_axis;
    orthonormal_pair(z_axis, x_axis, y_axis);
    float r1 = uniform_sample(x1, 0.0f, P.tau_scale), r2 = uniform_sample(x2, 0.0f, P.one_scale);
    float s, c;
    sincos_bt(r1, s, c);
    float x = c * sqrtf(r2);
    float y = s * sqrtf(r2);
    float z = sqrtf(1.0f - r2);
    return (x_axis * x + y_axis * y) + z_axis * z;
}

// ---- LDS scene tables --------------------------------------------------------------------
struct SceneLds {
    const BtPrimLite *lite;
    const BtMaterial *materials;
    const BtVolume *volumes;
    const BtLight *lights;
    const BtLightFace *faces;
    const float *density;     // LDS copy, or the global buffer when it does not fit
};

// ---- intersection --------------------------------------------------------------------------
// Sphere::hit's t selection (sphere.rs:129-145) against the running clip.
BT_DEV bool sphere_t(V3 o, V3 d, V3 c, float radius, float tmin, float tmax, float &t_out) {
    V3 oc = o - c;
    float half_b = dot(oc, d);
    float cc = len2(oc) - radius * radius;
    float disc = half_b * half_b - cc;
    if (!(disc >= 0.0f)) return false;
    float sqrtd = sqrtf(disc);
    float t = -half_b - sqrtd;
    if (t < tmin || t > tmax) {
        t = -half_b + sqrtd;
        if (t < tmin || t > tmax) return false;
    }
    t_out = t;
    return true;
}
// Rect::hit up to the containment test (rect.rs:110-137); q and p returned for pdf / face.
BT_DEV bool rect_t(V3 o, V3 d, const BtPrim &R, float tmin, float tmax, bool strict, float &t_out, float &q_out,
                   float &p_out) {
    V3 n = mk(R.c);
    float q = dot(d, n);
    if (fabsf(q) <= 1e-5f) return false;
    float p = dot(mk(R.t) - o, n);
    float t = p / q;
    if (t < tmin || t > tmax) return false;
    if (strict && !(t < tmax)) return false;     // Cuboid::hit keeps `manifold.t < t` (cuboid.rs:96)
    V3 pos = o + d * t;
    V3 local = xf_vector(mk(R.icx), mk(R.icy), mk(R.icz), pos) + mk(R.it);
    V3 ax = mk(R.ax), ay = mk(R.ay);
    V3 px = ax * dot(local, ax);
    V3 py = ay * dot(local, ay);
    if (!(len2(px) <= R.w_sqr && len2(py) <= R.h_sqr)) return false;
    t_out = t;
    q_out = q;
    p_out = p;
    return true;
}

struct HitRec {
    float t;
    int prim;          // -1 = miss
    bool inside;       // Face::Volume manifold from hit_volumetric (sphere.rs:158-163)
    bool p_neg;        // rect: p < 0 -> Face::Front (rect.rs:138-142)
};

// try_hit (mod.rs:389-402) and try_hit_volume (mod.rs:404-427) in one loop: in normal mode
// last_object is -1 and the clip is [clip_min, clip_max]; while marching it is the marched
// object and the clip is [0, volume_step].
BT_DEV HitRec intersect(const BtLaunch &P, V3 o, V3 d, float tmin, float tmax, int last_object) {
    HitRec h;
    h.t = tmax;
    h.prim = -1;
    h.inside = false;
    h.p_neg = false;
    const int n = P.n_prims;
    for (int i = 0; i < n; ++i) {
        const BtPrim &R = P.prims[i];           // wave-uniform index -> scalar loads
        if (R.kind == BT_PRIM_SPHERE) {
            V3 c = mk(R.c);
            bool taken = false;
            if (R.object == last_object) {      // Sphere::hit_volumetric (sphere.rs:150-166)
                V3 e = (o + d * h.t) - c;
                if (len2(e) <= R.radius * R.radius) {
                    h.prim = i;
                    h.inside = true;
                    taken = true;
                }
            }
            if (!taken) {
                float t;
                if (sphere_t(o, d, c, R.radius, tmin, h.t, t)) {
                    h.t = t;
                    h.prim = i;
                    h.inside = false;
                }
            }
        } else {
            float t, q, p;
            if (rect_t(o, d, R, tmin, h.t, R.kind == BT_PRIM_CUBOID_FACE, t, q, p)) {
                h.t = t;
                h.prim = i;
                h.inside = false;
                h.p_neg = p < 0.0f;
            }
        }
    }
    return h;
}

// Object::pdf of a light (object/mod.rs:154-166; sphere.rs:44-61, rect.rs:92-108,
// cuboid.rs:56-81); 0 when the ray misses it (material.rs:313-316 unwrap_or_default).
BT_DEV float light_pdf(const BtLaunch &P, const BtLight &Lt, const SceneLds &S, V3 o, V3 d) {
    if (Lt.kind == BT_LIGHT_SPHERE) {
        float t;
        if (!sphere_t(o, d, mk(Lt.centre), Lt.radius, P.clip_min, P.clip_max, t)) return 0.0f;
        return (t * t) / Lt.shadow;
    }
    if (Lt.kind == BT_LIGHT_RECT) {
        float t, q, p;
        if (!rect_t(o, d, P.prims[Lt.prim_first], P.clip_min, P.clip_max, false, t, q, p)) return 0.0f;
        float shadow = S.faces[Lt.face_first].area * fabsf(q);
        return (t * t) / shadow;
    }
    if (Lt.kind == BT_LIGHT_CUBOID) {
        float best_t = P.clip_max, best_q = 0.0f;
        int best = -1;
        for (int f = 0; f < Lt.prim_count; ++f) {
            float t, q, p;
            // rect.hit with the object-level clip, then `manifold.t < t` (cuboid.rs:63-75)
            if (rect_t(o, d, P.prims[Lt.prim_first + f], P.clip_min, P.clip_max, false, t, q, p) && t < best_t) {
                best_t = t;
                best_q = q;
                best = f;
            }
        }
        if (best < 0) return 0.0f;
        float shadow = S.faces[Lt.face_first + best].area * fabsf(best_q);
        return (best_t * best_t) / shadow;
    }
    return 0.0f;
}

// Rect::random_point (rect.rs:82-86) on a light face
BT_DEV V3 face_random_point(const BtLightFace &F, uint32_t x1, uint32_t x2) {
    float x = uniform_sample(x1, -F.half_width, F.scale_x);
    float y = uniform_sample(x2, -F.half_height, F.scale_y);
    V3 local = mk(F.ax) * x + mk(F.ay) * y;
    return xf_vector(mk(F.mcx), mk(F.mcy), mk(F.mcz), local) + mk(F.mt);
}

// DensityMap::sample, Trilinear (volume.rs:119-167)
BT_DEV float density_at(const BtVolume &vol, const float *density, float fx, float fy, float fz) {
    if (vol.width == 0 || vol.height == 0 || vol.depth == 0) return 0.0f;
    int x = (int)fx, y = (int)fy, z = (int)fz;
    x = x < 0 ? 0 : x; y = y < 0 ? 0 : y; z = z < 0 ? 0 : z;
    if (x >= vol.width || y >= vol.height || z >= vol.depth) return 0.0f;
    return density[vol.offset + (z * vol.height + y) * vol.width + x];
}
BT_DEV float density_sample(const BtVolume &vol, const float *density, V3 coord) {
    float cx = fminf(fmaxf(coord.x, 0.0f), 1.0f) * vol.size.x;
    float cy = fminf(fmaxf(coord.y, 0.0f), 1.0f) * vol.size.y;
    float cz = fminf(fmaxf(coord.z, 0.0f), 1.0f) * vol.size.z;
    float fx = floorf(cx), fy = floorf(cy), fz = floorf(cz);
    float ux = ceilf(cx), uy = ceilf(cy), uz = ceilf(cz);
    float tx = cx - truncf(cx), ty = cy - truncf(cy), tz = cz - truncf(cz);
    float x0 = density_at(vol, density, fx, fy, fz);
    float x1 = density_at(vol, density, ux, fy, fz);
    float y0 = lerpf(x0, x1, tx);
    x0 = density_at(vol, density, fx, uy, fz);
    x1 = density_at(vol, density, ux, uy, fz);
    float y1 = lerpf(x0, x1, tx);
    float z0 = lerpf(y0, y1, ty);
    x0 = density_at(vol, density, fx, fy, uz);
    x1 = density_at(vol, density, ux, fy, uz);
    y0 = lerpf(x0, x1, tx);
    x0 = density_at(vol, density, fx, uy, uz);
    x1 = density_at(vol, density, ux, uy, uz);
    y1 = lerpf(x0, x1, tx);
    float z1 = lerpf(y0, y1, ty);
    return lerpf(z0, z1, tz);
}

// ---- camera ray (mod.rs:271-302, ray.rs:103-113,126-137) ----------------------------------
BT_DEV void camera_ray(const BtLaunch &P, uint32_t px, uint32_t py, uint32_t k, U4 r, V3 &origin, V3 &dir) {
    float u_sub = 0.0f, v_sub = 0.0f;
    if (P.subsample_n > 1) {
        uint32_t n = (uint32_t)P.subsample_n;
        uint32_t sub = k % (n * n);
        float width_sub = 1.0f / (float)n;
        u_sub = (float)(sub % n) * width_sub;
        v_sub = (float)(sub / n) * width_sub;
    }
    float v0 = (float)py * P.pixel_height - 1.0f;
    float u0 = (float)px * P.pixel_width - 1.0f;
    float u_offset = u_sub * P.pixel_width + uniform_sample(r.x, P.jitter_u_lo, P.jitter_u_scale);
    float v_offset = v_sub * P.pixel_height + uniform_sample(r.y, P.jitter_v_lo, P.jitter_v_scale);
    float u = u0 + u_offset, v = v0 + v_offset;
    float yrot = P.xfov * 0.5f * -u;
    float xrot = P.yfov * 0.5f * -v;
    float sy, cy, sx, cx;
    sincos_bt(yrot, sy, cy);
    sincos_bt(xrot, sx, cx);
    V3 d_cam = mk(-(cx * sy), sx, -(cx * cy));
    V3 mcx = mk(P.cam_cx), mcy = mk(P.cam_cy), mcz = mk(P.cam_cz), mt = mk(P.cam_t);
    // Affine3A * Ray: origin = translation + 0, direction = normalize(normalize_or_zero(M*d))
    V3 o = mt + mk(0.0f, 0.0f, 0.0f);
    V3 d = normalize(normalize_or_zero(xf_vector(mcx, mcy, mcz, d_cam)));
    if (P.has_focus) {
        float angle = uniform_sample(r.z, 0.0f, P.tau_scale), rad = uniform_sample(r.w, 0.0f, P.one_scale);
        float s, c;
        sincos_bt(angle, s, c);
        V3 defocus = (mk(P.disk_x) * c + mk(P.disk_y) * s) * rad;
        V3 defocus_offset = xf_vector(mcx, mcy, mcz, defocus * P.aperture);
        float frac_f_z = P.focus / fabsf(d_cam.z);
        o = o + defocus_offset;
        d = normalize(d * frac_f_z - defocus_offset);
    }
    origin = o;
    dir = d;
}

BT_DEV unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

} // namespace

// --------------------------------------------------------------------------------------------
// The render kernel.  OUTPUT: 0 Full, 1 Albedo, 2 Normal, 3 Depth (tracer/mod.rs:108-115).
// Block = 256 threads = one 16x16 pixel tile (BT_TILE); wave w covers the 8x8 quadrant w.
#ifndef BT_WAVES_PER_SIMD
#define BT_WAVES_PER_SIMD 1
#endif
template <int OUTPUT>
__global__ __launch_bounds__(256, BT_WAVES_PER_SIMD) void bt_render_kernel(BtLaunch P) {
    extern __shared__ __align__(16) unsigned char smem[];

    // ---- stage the per-lane lookup tables in LDS ----
    SceneLds S;
    {
        unsigned char *p = smem;
        BtPrimLite *lite = (BtPrimLite *)p;        p += sizeof(BtPrimLite) * P.n_prims;
        BtMaterial *mats = (BtMaterial *)p;        p += sizeof(BtMaterial) * P.n_materials;
        BtVolume *vols = (BtVolume *)p;            p += sizeof(BtVolume) * P.n_volumes;
        BtLight *lights = (BtLight *)p;            p += sizeof(BtLight) * P.n_lights;
        BtLightFace *faces = (BtLightFace *)p;     p += sizeof(BtLightFace) * P.n_light_faces;
        float *dens = (float *)p;
        for (int i = threadIdx.x; i < P.n_prims; i += blockDim.x) {
            const BtPrim &R = P.prims[i];
            BtPrimLite l;
            l.c = R.c;
            l.radius = R.radius;
            l.kind_object = R.kind | (R.object << 8);
            l.material = R.material;
            l.volume = R.volume;
            l.pad = 0;
            lite[i] = l;
        }
        for (int i = threadIdx.x; i < P.n_materials; i += blockDim.x) mats[i] = P.materials[i];
        for (int i = threadIdx.x; i < P.n_volumes; i += blockDim.x) vols[i] = P.volumes[i];
        for (int i = threadIdx.x; i < P.n_lights; i += blockDim.x) lights[i] = P.lights[i];
        for (int i = threadIdx.x; i < P.n_light_faces; i += blockDim.x) faces[i] = P.light_faces[i];
        const bool dens_lds = P.n_density > 0 && P.n_density <= 8192;
        if (dens_lds)
            for (int i = threadIdx.x; i < P.n_density; i += blockDim.x) dens[i] = P.density[i];
        S.lite = lite; S.materials = mats; S.volumes = vols; S.lights = lights; S.faces = faces;
        S.density = dens_lds ? dens : P.density;
        __syncthreads();
    }

    // ---- tile / pixel mapping ----
    const uint32_t tile = P.sharded ? (blockIdx.x * P.world + P.rank) : blockIdx.x;
    const uint32_t tx = tile % P.tiles_x, ty = tile / P.tiles_x;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t lx = ((wave & 1) << 3) | (lane & 7), ly = ((wave >> 1) << 3) | (lane >> 3);
    const uint32_t px = tx * BT_TILE_DIM + lx, py = ty * BT_TILE_DIM + ly;
    const bool in_frame = (ty < P.tiles_y) && (px < P.width) && (py < P.height);
    float *out_px = P.sharded ? P.out + ((size_t)blockIdx.x * (BT_TILE_DIM * BT_TILE_DIM) + ly * BT_TILE_DIM + lx) * 4
                              : P.out + ((size_t)py * P.width + px) * 4;

    const uint32_t pixel_index = py * P.width + px;
    const uint32_t nn = (uint32_t)(P.subsample_n * P.subsample_n);
    const uint32_t total = in_frame ? (uint32_t)P.samples * nn : 0u;
    const uint32_t sample0 = P.sample_base * nn;

    V3 acc = mk(0.0f, 0.0f, 0.0f);
    if (in_frame) acc = mk(out_px[0], out_px[1], out_px[2]);   // `*r += pixel.r` (buffer.rs:159-164)

    // per-lane path state
    V3 ro = mk(0, 0, 0), rd = mk(0, 0, -1), beta = mk(1, 1, 1), L = mk(0, 0, 0);
    V3 first = mk(0, 0, 0);            // first non-pass-through albedo / normal (AOV outputs)
    float first_depth = __builtin_inff();
    bool have_first = false;
    int bounce = 0, vbounce = 0, last_object = -1;
    uint32_t event = 0, k = 0;
    bool fresh = true;
    unsigned long long segments = 0;

    while (k < total) {
        const uint32_t sample_index = sample0 + k;
        if (fresh) {
            U4 r = philox(pixel_index, sample_index, 0u, 0u, P.seed_lo, P.seed_hi);
            camera_ray(P, px, py, k, r, ro, rd);
            beta = mk(1, 1, 1);
            L = mk(0, 0, 0);
            bounce = 0; vbounce = 0; last_object = -1;
            event = 1;
            have_first = false;
            first = mk(0, 0, 0);
            first_depth = __builtin_inff();
            fresh = false;
        }
        const bool marching = last_object >= 0;
        if (!marching) vbounce = 0;                                   // sample() -> sample_volume(.., 0), mod.rs:335
        bool done = marching ? (vbounce > P.max_volume_bounces)       // mod.rs:352-354
                             : (bounce > P.max_bounces);              // mod.rs:323-325
        if (!done) {
            segments += 1;
            const float tmin = marching ? 0.0f : P.clip_min;
            const float tmax = marching ? P.volume_step : P.clip_max;
            HitRec h = intersect(P, ro, rd, tmin, tmax, last_object);
            if (h.prim < 0) {
                // sample_root (mod.rs:429-452)
                L = L + beta * mk(P.root_color);
                if (OUTPUT != 0 && !have_first) {
                    have_first = true;
                    if (OUTPUT == 1) first = mk(P.root_albedo);
                    if (OUTPUT == 2) first = P.root_has_albedo ? -rd : mk(0, 0, 0);
                    if (OUTPUT == 3) first_depth = P.root_has_albedo ? P.clip_max : __builtin_inff();
                }
                done = true;
            } else {
                const BtPrimLite &pl = S.lite[h.prim];
                const int pkind = pl.kind_object & 0xff, pobject = pl.kind_object >> 8;
                const V3 pos = ro + rd * h.t;
                V3 normal;
                bool front = false, vol_face = false, vol_back = false;
                if (h.inside) {                       // generate_volume_manifold (sphere.rs:63-83)
                    normal = mk(0, 0, 0);
                    vol_face = true;
                } else if (pkind == BT_PRIM_SPHERE) { // generate_surface_manifold (sphere.rs:85-119)
                    V3 c = mk(pl.c);
                    V3 nrm = pos - c;
                    nrm = mk(nrm.x / pl.radius, nrm.y / pl.radius, nrm.z / pl.radius);
                    front = dot(rd, nrm) < 0.0f;
                    normal = front ? nrm : -nrm;
                    vol_face = pl.volume >= 0;
                    vol_back = vol_face && !front;
                } else {                              // rect.rs:138-142
                    V3 nrm = mk(pl.c);
                    front = h.p_neg;
                    normal = front ? nrm : -nrm;
                }

                if (!vol_face) {
                    // ---- sample_surface (mod.rs:454-486) + Material::shade (material.rs:81-199) ----
                    const BtMaterial &M = S.materials[pl.material];
                    const V3 albedo = mk(M.albedo);
                    L = L + beta * mk(M.emitted);
                    bool scatter = false;
                    V3 nd = mk(0, 0, 0);
                    float weight = 1.0f;              // material.pdf / shade.pdf
                    if (M.kind == BT_MAT_DIFFUSE || M.kind == BT_MAT_METALLIC || M.kind == BT_MAT_GLASS) {
                        U4 u = philox(pixel_index, sample_index, event, 0u, P.seed_lo, P.seed_hi);
                        float p = 1.0f, mp = 1.0f;
                        if (M.kind == BT_MAT_DIFFUSE) {
                            const int li = (int)__umulhi(u.x, (uint32_t)P.n_lights);   // material.rs:106-119
                            const BtLight &Lt = S.lights[li];
                            V3 dir;
                            if (bernoulli(u.y, 0.5f)) {                                 // Pdf::Mix -> Light (:269-275)
                                V3 point;
                                if (Lt.kind == BT_LIGHT_SPHERE) {
                                    point = mk(Lt.centre) + unit_sphere(P, u.z, u.w) * Lt.radius;
                                } else if (Lt.kind == BT_LIGHT_RECT) {
                                    point = face_random_point(S.faces[Lt.face_first], u.z, u.w);
                                } else if (Lt.kind == BT_LIGHT_CUBOID) {               // cuboid.rs:47-54
                                    U4 e = philox(pixel_index, sample_index, event, 1u, P.seed_lo, P.seed_hi);
                                    float chosen = uniform_sample(e.x, 0.0f, Lt.total_scale);
                                    int index = 0;
#pragma unroll
                                    for (int f = 0; f < 5; ++f)
                                        if (Lt.cum[f] <= chosen) index = f + 1;
                                    point = face_random_point(S.faces[Lt.face_first + index], u.z, u.w);
                                } else {
                                    point = mk(Lt.centre);
                                }
                                dir = point - pos;
                            } else {
                                dir = cosine(P, normal, u.z, u.w);                      // Pdf::Diffuse (:224-230)
                            }
                            nd = normalize(dir);
                            const float pd = dot(normal, nd) * 0.318309886183790671538f; // diffuse_pdf (:301-303)
                            const float plight = light_pdf(P, Lt, S, pos, nd);
                            p = lerpf(pd, plight, 0.5f);                                // :294-296
                            mp = pd;                                                    // Material::pdf (:204)
                        } else if (M.kind == BT_MAT_METALLIC) {                         // :231-239
                            V3 dir = reflect(rd, normal);
                            V3 fuzz = unit_hemisphere(P, normal, u.x, u.y) * M.roughness;
                            nd = normalize(dir + fuzz);
                        } else {                                                        // Glass :240-261
                            const float ior = front ? M.inv_ior : M.ior;
                            const float cos_theta = fminf(dot(-rd, normal), 1.0f);
                            const float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
                            const float fr = fresnel(rd, normal, ior);
                            V3 dir;
                            if (ior * sin_theta > 1.0f || bernoulli(u.x, fr))
                                dir = reflect(rd, normal);
                            else
                                dir = refract(rd, normal, ior);
                            V3 fuzz = unit_hemisphere(P, normal, u.y, u.z) * M.roughness;
                            nd = normalize(dir + fuzz);
                        }
                        event += 1;
                        scatter = !(fabsf(p) <= 1e-5f);                                 // Pdf::pdf (:279-286)
                        weight = mp / p;
                    }
                    if (OUTPUT != 0 && !have_first) {
                        have_first = true;
                        if (scatter) {      // data.albedo ColorData (material.rs:99-104,140-145,169-174)
                            if (OUTPUT == 1) first = albedo;
                            if (OUTPUT == 2) first = normal;
                            if (OUTPUT == 3) first_depth = h.t;
                        } else {            // ColorData::from_emitted(emitted) (mod.rs:483-485)
                            if (OUTPUT == 1) first = mk(M.emitted);
                        }
                    }
                    if (scatter) {
                        beta = beta * (albedo * weight);
                        ro = pos;
                        rd = nd;
                        bounce += 1;
                        last_object = -1;
                    } else {
                        done = true;
                    }
                } else {
                    // ---- sample_volume (mod.rs:488-523) + Volume::shade (volume.rs:26-60) ----
                    if (pl.volume < 0) {
                        done = true;    // unreachable: vol_face implies a volume
                    } else {
                        const BtVolume &vol = S.volumes[pl.volume];
                        U4 u = philox(pixel_index, sample_index, event, 0u, P.seed_lo, P.seed_hi);
                        event += 1;
                        const V3 c = mk(pl.c);
                        const V3 hsz = mk(pl.radius, pl.radius, pl.radius);
                        const V3 bmin = c - hsz, bmax = c + hsz;        // sphere.rs:35-38
                        const V3 size = bmax - bmin;
                        const V3 rel = pos - bmin;
                        const V3 coord = mk(rel.x / size.x, rel.y / size.y, rel.z / size.z);
                        const float density = P.volume_step * density_sample(vol, S.density, coord);
                        if (density >= 1.0f || bernoulli(u.x, density)) {
                            V3 origin = pos;
                            if (h.inside) origin = origin - (rd * P.volume_step) * u24(u.y);
                            ro = origin;
                            rd = normalize(unit_sphere(P, u.z, u.w));
                            beta = beta * mk(0.8f, 0.8f, 0.8f);
                            if (OUTPUT != 0 && !have_first) {
                                have_first = true;
                                if (OUTPUT == 1) first = mk(0.8f, 0.8f, 0.8f);
                                if (OUTPUT == 2) first = normal;
                                if (OUTPUT == 3) first_depth = h.t;
                            }
                        } else {
                            ro = pos;
                            rd = normalize(rd);     // Ray::new(origin, direction) (volume.rs:54-57)
                        }
                        if (vol_back) {             // mod.rs:504-505
                            bounce += 1;
                            last_object = -1;
                        } else {                    // mod.rs:507-513
                            last_object = pobject;
                            vbounce += 1;
                        }
                    }
                }
            }
        }
        if (done) {
            // mod.rs:304-315 -> Chunk::write_* -> Buffer::write_* (buffer.rs:159-178)
            if (OUTPUT == 0) {
                acc = acc + L;
            } else if (OUTPUT == 3) {
                float depth = (first_depth - P.clip_min) / (P.clip_max - P.clip_min);
                depth = fminf(fmaxf(depth, 0.0f), 1.0f);
                acc = acc + mk(depth, depth, depth);
            } else {
                acc = acc + first;
            }
            k += 1;
            fresh = true;
        }
    }

    if (in_frame) {
        out_px[0] = acc.x;
        out_px[1] = acc.y;
        out_px[2] = acc.z;
    }
    if (P.counters) {
        unsigned long long s = wave_sum(segments);
        if (lane == 0 && s) atomicAdd(&P.counters[0], s);
    }
}

// shard (tile-major, `world` ranks back to back) -> row-major frame; rgb AND alpha copied.
__global__ __launch_bounds__(256) void bt_unshard_kernel(const float4 *gathered, float4 *frame, uint32_t width,
                                                         uint32_t height, uint32_t tiles_x, uint32_t tiles_y,
                                                         uint32_t world, uint32_t tiles_per_rank) {
    const uint32_t tile = blockIdx.x;
    if (tile >= tiles_x * tiles_y) return;
    const uint32_t rank = tile % world, slot = tile / world;
    const uint32_t lx = threadIdx.x & 15, ly = threadIdx.x >> 4;
    const uint32_t px = (tile % tiles_x) * BT_TILE_DIM + lx, py = (tile / tiles_x) * BT_TILE_DIM + ly;
    if (px >= width || py >= height) return;
    const size_t src = ((size_t)rank * tiles_per_rank + slot) * (BT_TILE_DIM * BT_TILE_DIM) + ly * BT_TILE_DIM + lx;
    frame[(size_t)py * width + px] = gathered[src];
}

// Buffer::preview (buffer.rs:117-138): mean -> colour space -> (x * 255) as u8.
namespace {
BT_DEV float linear_to_srgb(float x) {              // color.rs:14-20
    if (x <= 0.0031308f) return 12.92f * x;
    return 1.055f * powf(x, 1.0f / 2.4f) - 0.055f;
}
BT_DEV uint32_t f32_to_u8(float x) {                // color.rs:22-24 (saturating `as u8`)
    float v = x * 255.0f;
    if (!(v > 0.0f)) return 0u;
    if (v >= 255.0f) return 255u;
    return (uint32_t)v;
}
} // namespace
__global__ __launch_bounds__(256) void bt_preview_kernel(const float4 *rgba, uint32_t *out, uint32_t n,
                                                         float samples_recip, int color_space) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 s = rgba[i];
    V3 rgb = mk(s.x, s.y, s.z) * samples_recip;
    if (color_space == 1) {
        V3 nrm = normalize(rgb);
        rgb = (nrm + mk(1, 1, 1)) * 0.5f;
    } else if (color_space == 3) {
        rgb = mk(linear_to_srgb(rgb.x), linear_to_srgb(rgb.y), linear_to_srgb(rgb.z));
    }
    out[i] = f32_to_u8(rgb.x) | (f32_to_u8(rgb.y) << 8) | (f32_to_u8(rgb.z) << 16) | (f32_to_u8(s.w) << 24);
}

// ---- host-side launchers (called from bt_api.cpp) ---------------------------------------------
extern "C" hipError_t bt_launch_render(const BtLaunch *P, int output, unsigned grid, size_t lds_bytes,
                                       hipStream_t stream) {
    dim3 g(grid), b(256);
    switch (output) {
    case 0: hipLaunchKernelGGL(bt_render_kernel<0>, g, b, lds_bytes, stream, *P); break;
    case 1: hipLaunchKernelGGL(bt_render_kernel<1>, g, b, lds_bytes, stream, *P); break;
    case 2: hipLaunchKernelGGL(bt_render_kernel<2>, g, b, lds_bytes, stream, *P); break;
    default: hipLaunchKernelGGL(bt_render_kernel<3>, g, b, lds_bytes, stream, *P); break;
    }
    return hipGetLastError();
}
extern "C" hipError_t bt_launch_unshard(const float *gathered, float *frame, uint32_t width, uint32_t height,
                                        uint32_t tiles_x, uint32_t tiles_y, uint32_t world, uint32_t tiles_per_rank,
                                        hipStream_t stream) {
    hipLaunchKernelGGL(bt_unshard_kernel, dim3(tiles_x * tiles_y), dim3(256), 0, stream, (const float4 *)gathered,
                       (float4 *)frame, width, height, tiles_x, tiles_y, world, tiles_per_rank);
    return hipGetLastError();
}
extern "C" hipError_t bt_launch_preview(const float *rgba, uint8_t *out, uint32_t n, uint32_t samples,
                                        int color_space, hipStream_t stream) {
    float recip = 1.0f / (float)samples;
    hipLaunchKernelGGL(bt_preview_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, (const float4 *)rgba,
                       (uint32_t *)out, n, recip, color_space);
    return hipGetLastError();
}
