// bt_device.hpp -- device-side building blocks of the render kernel: vector math, the numerics
// contract's sin/cos and Philox, the samplers of math/distr.rs, and primitive intersection
// (sphere.rs, rect.rs, cuboid.rs) over the flattened BtPrim table.  Included only by
// bt_kernels.hip.  Arithmetic follows DESIGN.md's numerics contract so that geometry decisions
// are bit-identical to the CPU oracle: no FMA contraction, explicit fmaf only in sincos,
// correctly rounded sqrt / divide (hipcc default), fixed operation order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bt_types.h"

#pragma clang fp contract(off)

#define BT_DEV static __device__ __forceinline__

namespace {

struct V3 { float x, y, z; };
BT_DEV V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
BT_DEV V3 mk(const BtV3 &a) { return mk(a.x, a.y, a.z); }
// The primitive table is read through the constant address space (same memory, AS 4): loads from it are
// invariant by definition, so a wave-uniform index always selects scalar (s_load) instructions -- even in
// kernels that also store to global memory inside the loop (the sliced variant's parked samples), where the
// compiler can no longer prove that a plain global load is not clobbered and falls back to per-lane loads.
typedef const __attribute__((address_space(4))) BtPrim BtPrimK;
typedef const __attribute__((address_space(4))) BtV3 BtV3K;
BT_DEV V3 mk(BtV3K &a) { return mk(a.x, a.y, a.z); }
BT_DEV BtPrimK *prim_table(const BtLaunch &P) { return (BtPrimK *)P.prims; }
BT_DEV V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
BT_DEV V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
BT_DEV V3 operator*(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
BT_DEV V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
BT_DEV V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
BT_DEV float dot(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
BT_DEV float len2(V3 a) { return dot(a, a); }
// sqrtf(x) and 1.0f / sqrtf(x), bit for bit, in fewer issue slots.  hipcc expands an IEEE square root into v_sqrt_f32
// (1 ulp), the two neighbours s -/+ 1 ulp tried against the residuals x - s_down s, x - s_up s, and around that a 2^32
// pre-scaling of inputs below 2^-96 plus a pass-through of +-0 / inf (AMDGPU lowerFSQRTF32); an IEEE division by s is
// v_rcp_f32, two Newton steps, quotient, two residual corrections, and around those v_div_scale / v_div_fmas /
// v_div_fixup (refined_rcp() / div_refined() below).  When every live lane's x lies in [2^-96, 2^96) -- one integer
// compare and a wave-uniform branch -- the scaling, the pass-through and the fix-up are no-ops: what is left is the
// core sequence, written out here; any other wave takes the compiler's expansion.  (s in [2^-48, 2^48): no
// intermediate of 1/s leaves the normal range, and 1 * r is exact, so the quotient step drops out.)
#ifndef BT_FAST_SQRT
#define BT_FAST_SQRT 1
#endif
BT_DEV bool sqrt_core_ok(float x) {
    return __builtin_amdgcn_ballot_w64((__float_as_uint(x) - 0x0f800000u) >= (0x6f800000u - 0x0f800000u)) == 0ull;
}
BT_DEV float sqrt_core(float x) {
    float s = __builtin_amdgcn_sqrtf(x);
    const float sd = __uint_as_float(__float_as_uint(s) - 1u), su = __uint_as_float(__float_as_uint(s) + 1u);
    const float rd = __builtin_fmaf(-sd, s, x), ru = __builtin_fmaf(-su, s, x);
    s = rd <= 0.0f ? sd : s;
    return ru > 0.0f ? su : s;
}
BT_DEV float sqrt_bt(float x) {
#if BT_FAST_SQRT
    if (sqrt_core_ok(x)) return sqrt_core(x);
#endif
    return sqrtf(x);
}
BT_DEV float rsqrt_bt(float x) {                    // 1.0f / sqrtf(x)
#if BT_FAST_SQRT
    if (sqrt_core_ok(x)) {
        const float s = sqrt_core(x);
        const float r0 = __builtin_amdgcn_rcpf(s);
        const float r = __builtin_fmaf(__builtin_fmaf(-s, r0, 1.0f), r0, r0);
        const float m2 = __builtin_fmaf(__builtin_fmaf(-s, r, 1.0f), r, r);
        return __builtin_fmaf(__builtin_fmaf(-s, m2, 1.0f), r, m2);
    }
#endif
    return 1.0f / sqrtf(x);
}
BT_DEV V3 normalize(V3 a) { float rl = rsqrt_bt(len2(a)); return a * rl; }
BT_DEV V3 normalize_or_zero(V3 a) {
    float rl = rsqrt_bt(len2(a));
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

// sincos_bt() for a wave whose arguments all reduce with k = 0 (|x| <= pi/4: the camera's frustum angles): the three
// Cody-Waite steps are fmaf(+-0, c, r) = r and the quadrant fix-up is the identity, so they are skipped -- same bits.
// Any lane with k != 0 sends the whole wave through sincos_bt().
BT_DEV void sincos_small_bt(float x, float &s, float &c) {
    const float k = __builtin_rintf(x * 0.636619772f);
    if (__ballot(k != 0.0f) != 0ull) {
        sincos_bt(x, s, c);
        return;
    }
    const float r = x + 0.0f, r2 = r * r;          // fmaf(+-0, c, x) is x, except that it turns x = -0 into +0: so does x + 0
    const float ps = __builtin_fmaf(__builtin_fmaf(-1.9515295891e-4f, r2, 8.3321608736e-3f), r2, -1.6666654611e-1f);
    s = __builtin_fmaf(ps * r2, r, r);
    const float pc = __builtin_fmaf(__builtin_fmaf(2.443315711809948e-5f, r2, -1.388731625493765e-3f), r2,
                                    4.166664568298827e-2f);
    c = __builtin_fmaf(pc, r2 * r2, __builtin_fmaf(-0.5f, r2, 1.0f));
}

// ---- Philox4x32-10 (numerics contract N6) --------------------------------------------
struct U4 { uint32_t x, y, z, w; };
#ifndef BT_PHILOX_ROUNDS
#define BT_PHILOX_ROUNDS 10        // the numerics contract; other values only for timing experiments
#endif
BT_DEV U4 philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int i = 0; i < BT_PHILOX_ROUNDS; ++i) {
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
    V3 parallel = n * -sqrt_bt(fabsf(1.0f - len2(perp)));
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
    float x = c * 2.0f * sqrt_bt(r2 * (1.0f - r2));
    float y = s * 2.0f * sqrt_bt(r2 * (1.0f - r2));
    float z = 1.0f - 2.0f * r2;
    return mk(x, y, z);
}
BT_DEV V3 unit_hemisphere(const BtLaunch &P, V3 normal, uint32_t x1, uint32_t x2) {
    V3 z_axis = normalize(normal), x_axis, y_axis;
    orthonormal_pair(z_axis, x_axis, y_axis);
    float r1 = uniform_sample(x1, 0.0f, P.tau_scale), r2 = uniform_sample(x2, 0.0f, P.one_scale);
    float s, c;
    sincos_bt(r1, s, c);
    float x = c * 2.0f * sqrt_bt(r2 * (1.0f - r2));
    float y = s * 2.0f * sqrt_bt(r2 * (1.0f - r2));
    float z = 1.0f - r2;
    return (x_axis * x + y_axis * y) + z_axis * z;
}
BT_DEV V3 cosine(const BtLaunch &P, V3 normal, uint32_t x1, uint32_t x2) {
    V3 z_axis = normalize(normal), x_axis, y_axis;
    orthonormal_pair(z_axis, x_axis, y_axis);
    float r1 = uniform_sample(x1, 0.0f, P.tau_scale), r2 = uniform_sample(x2, 0.0f, P.one_scale);
    float s, c;
    sincos_bt(r1, s, c);
    float x = c * sqrt_bt(r2);
    float y = s * sqrt_bt(r2);
    float z = sqrt_bt(1.0f - r2);
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
    float sqrtd = sqrt_bt(disc);
    float t = -half_b - sqrtd;
    if (t < tmin || t > tmax) {
        t = -half_b + sqrtd;
        if (t < tmin || t > tmax) return false;
    }
    t_out = t;
    return true;
}
// Rect::hit up to the containment test (rect.rs:110-137); q and p returned for pdf / face.
// Containment test of an axis-aligned rect (BT_PRIM_RECT_AAN) for the plane with normal axis W: A, B are the two
// in-plane axes in ascending order; Rect.x is one of them (aa_u), which decides which extent limits which.
#define BT_COMP(v, i) ((i) == 0 ? (v).x : ((i) == 1 ? (v).y : (v).z))
template <int W, class PrimRef>
BT_DEV bool rect_aan_t(V3 o, V3 d, PrimRef &R, float tmin, float tmax, bool strict, float &t_out, float &q_out, float &p_out) {
    constexpr int A = W == 0 ? 1 : 0, B = W == 2 ? 1 : 2;
    // Branch-free: every test is evaluated and AND-ed (a rejected lane's t may be inf / NaN -- it is never used).
    // Rect scenes issue almost as many scalar as vector instructions (exec-mask bookkeeping of nested early returns,
    // one scalar issue per cycle and CU); the wave almost never skips a block as a whole anyway.
    // the row's six constants sit side by side (bt_types.h): t[w], it[a], it[b], the two limits, c[w] = +-1
    const float r_tw = R.ax.x, r_ia = R.ax.y, r_ib = R.ax.z, lim_a = R.ax_w, lim_b = R.ay.x, sgn = R.ay.y;
    const float dq = BT_COMP(d, W);
    const float dp = r_tw - BT_COMP(o, W);
    const float t = dp / dq;                      // == dot(t - o, n) / dot(d, n), the signs of n cancel exactly
    const float la = (BT_COMP(o, A) + BT_COMP(d, A) * t) + r_ia;
    const float lb = (BT_COMP(o, B) + BT_COMP(d, B) * t) + r_ib;
    const bool ok = !(fabsf(dq) <= 1e-5f) & !(t < tmin || t > tmax) & !(strict && !(t < tmax)) &
                    (la * la <= lim_a) & (lb * lb <= lim_b);
    t_out = t;
    q_out = dq * sgn;
    p_out = dp * sgn;
    return ok;
}
// Rect::hit up to the containment test (rect.rs:110-137); q and p returned for pdf / face.
template <class PrimRef>
BT_DEV bool rect_t(V3 o, V3 d, PrimRef &R, float tmin, float tmax, bool strict, float &t_out, float &q_out,
                   float &p_out) {
    if ((R.kind & BT_PRIM_SHAPE_MASK) == BT_PRIM_RECT_AAN) {
        const int w = R.aa_w;
        if (w == 0) return rect_aan_t<0>(o, d, R, tmin, tmax, strict, t_out, q_out, p_out);
        if (w == 1) return rect_aan_t<1>(o, d, R, tmin, tmax, strict, t_out, q_out, p_out);
        return rect_aan_t<2>(o, d, R, tmin, tmax, strict, t_out, q_out, p_out);
    }
    // general and local-axes rects: branch-free as well (see rect_aan_t)
    V3 n = mk(R.c);
    float q = dot(d, n);
    float p = dot(mk(R.t) - o, n);
    float t = p / q;
    bool ok = !(fabsf(q) <= 1e-5f) & !(t < tmin || t > tmax) & !(strict && !(t < tmax));   // cuboid.rs:96 keeps `manifold.t < t`
    V3 pos = o + d * t;
    if ((R.kind & BT_PRIM_SHAPE_MASK) == BT_PRIM_RECT_AA) {
        // identity matrix, Rect.x = +-e_u, Rect.y = +-e_v: `M^-1*pos + t'` is pos + t' and each
        // projection's squared length is the square of one component -- the same float values
        // the general path below produces (only the sign of exact zeros can differ, and those
        // are squared), at a third of the instructions.
        const int u = R.aa_u, v = R.aa_v;
        float lu = (u == 0 ? pos.x : (u == 1 ? pos.y : pos.z)) + (u == 0 ? R.it.x : (u == 1 ? R.it.y : R.it.z));
        float lv = (v == 0 ? pos.x : (v == 1 ? pos.y : pos.z)) + (v == 0 ? R.it.x : (v == 1 ? R.it.y : R.it.z));
        ok = ok & (lu * lu <= R.w_sqr) & (lv * lv <= R.h_sqr);
    } else if ((R.kind & BT_PRIM_SHAPE_MASK) == BT_PRIM_RECT_LA) {
        // Rect.x = +-e_u, Rect.y = +-e_v in LOCAL space: the projections' squared lengths are local[u]^2, local[v]^2
        // (as above), so only those two components of `M^-1 * pos + t'` are formed -- component by component the
        // expression of xf_vector() + it; the host has put the two rows where Rect.x / Rect.y sit in other rows.
        const float lu = ((R.ax.x * pos.x + R.ax.y * pos.y) + R.ax.z * pos.z) + R.ax_w;     // rows u, v of M^-1 | t'
        const float lv = ((R.ay.x * pos.x + R.ay.y * pos.y) + R.ay.z * pos.z) + R.ay_w;
        ok = ok & (lu * lu <= R.w_sqr) & (lv * lv <= R.h_sqr);
    } else {
        V3 local = xf_vector(mk(R.icx), mk(R.icy), mk(R.icz), pos) + mk(R.it);
        V3 ax = mk(R.ax), ay = mk(R.ay);
        V3 px = ax * dot(local, ax);
        V3 py = ay * dot(local, ay);
        const bool in_u = len2(px) <= R.w_sqr, in_v = len2(py) <= R.h_sqr;
        ok = ok & in_u & in_v;
    }
    t_out = t;
    q_out = q;
    p_out = p;
    return ok;
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
// RECTS = false: the scene holds spheres only (bt_api.cpp checks), all rect code drops out of the build.
// VOLS = false: no sphere carries a volume, so nothing ever marches and the hit_volumetric test drops out.
// short_seg (wave-uniform: the chords of the lens march; tried for waves in which every lane is on a volume-march step
// too -- such waves are too rare to matter, profiles/r01e/ab_march_reject.log): the segment is at most h.t long on entry, so a sphere whose surface is farther than that from the origin in either
// direction cannot be touched -- a two-compare reject (with a 1e-4 relative safety margin against rounding) in front
// of the quadratic; it can only skip tests that would fail.  The marched sphere's hit_volumetric test is never skipped.
template <bool RECTS = true, bool VOLS = true>
BT_DEV void intersect_row(BtPrimK *prims, int i, V3 o, V3 d, float tmin, int last_object, HitRec &h, bool short_seg) {
    BtPrimK &R = prims[i];                      // wave-uniform index -> scalar loads
    if (!RECTS || R.kind == BT_PRIM_SPHERE) {
        V3 c = mk(R.c);
        bool far = false;                       // short_seg is wave-uniform: the reject costs nothing elsewhere
        if (short_seg) {
            const float d2 = len2(o - c), outer = R.radius + h.t, inner = R.radius - h.t;
            far = d2 > (outer * outer) * 1.0001f || (inner > 0.0f && d2 < (inner * inner) * 0.9999f);
        }
        bool taken = false;
        if (VOLS && R.object == last_object) {  // Sphere::hit_volumetric (sphere.rs:150-166)
            V3 e = (o + d * h.t) - c;
            if (len2(e) <= R.radius * R.radius) {
                h.prim = i;
                h.inside = true;
                taken = true;
            }
        }
        if (!taken && !far) {
            float t;
            if (sphere_t(o, d, c, R.radius, tmin, h.t, t)) {
                h.t = t;
                h.prim = i;
                h.inside = false;
            }
        }
    } else {
        float t = 0.0f, q = 0.0f, p = 0.0f;
        const bool hit = rect_t(o, d, R, tmin, h.t, (R.kind & BT_PRIM_STRICT) != 0, t, q, p);
        h.t = hit ? t : h.t;                    // selects, not a branch (scalar-issue pressure, see rect_aan_t)
        h.prim = hit ? i : h.prim;
        h.inside = hit ? false : h.inside;
        h.p_neg = hit ? p < 0.0f : h.p_neg;
    }
}
// Sphere-only scenes: try_hit / try_hit_volume over BtSpherePair rows.  The clip-independent half of Sphere::hit
// (oc, half_b = dot(oc, d), c = |oc|^2 - r^2, discriminant; sphere.rs:122-127) is formed for two spheres at once in
// float2 lanes (v_pk_add_f32 / v_pk_mul_f32, the pair's constants straight from SGPR pairs) -- the same IEEE
// operations in the same order per sphere as sphere_t() -- then each sphere's root selection runs against the running
// clip, first sphere first, exactly as the one-at-a-time loop does.
// (round 1 formed the pair in float2 lanes, v_pk_add_f32 / v_pk_mul_f32; a packed instruction occupies the issue port as
// long as the two scalar ones and needs aligned register pairs: slower since the kernels are register-bound,
// profiles/r03c/ab_nopk.log)
struct f2 { float x, y; };
BT_DEV f2 splat2(float v) { return f2{v, v}; }
BT_DEV f2 operator+(f2 a, f2 b) { return f2{a.x + b.x, a.y + b.y}; }
BT_DEV f2 operator-(f2 a, f2 b) { return f2{a.x - b.x, a.y - b.y}; }
BT_DEV f2 operator*(f2 a, f2 b) { return f2{a.x * b.x, a.y * b.y}; }
BT_DEV f2 operator*(f2 a, float b) { return f2{a.x * b, a.y * b}; }
// The scan of the sphere table as whole pairs: what the volume builds run (volume.json / cloud.json hold four spheres; the form
// below with a peeled odd sphere cost their register allocation 5 - 7 %, profiles/r05d, r05e).
template <bool VOLS>
BT_DEV HitRec intersect_sphere_pairs(const BtLaunch &P, V3 o, V3 d, float tmin, float tmax, int last_object) {
    HitRec h;
    h.t = tmax;
    h.prim = -1;
    h.inside = false;
    h.p_neg = false;
    const int n = P.n_prims;
    typedef const __attribute__((address_space(4))) BtSpherePair PairK;
    PairK *pairs = (PairK *)P.sphere_pairs;
    for (int i = 0; i < n; i += 2) {
        PairK &Q = pairs[i >> 1];               // wave-uniform index -> scalar loads
        const f2 cx = {Q.cx[0], Q.cx[1]}, cy = {Q.cy[0], Q.cy[1]}, cz = {Q.cz[0], Q.cz[1]}, r = {Q.radius[0], Q.radius[1]};
        const f2 ocx = splat2(o.x) - cx, ocy = splat2(o.y) - cy, ocz = splat2(o.z) - cz;
        const f2 half_b = (ocx * d.x + ocy * d.y) + ocz * d.z;
        const f2 cc = ((ocx * ocx + ocy * ocy) + ocz * ocz) - r * r;
        const f2 disc = half_b * half_b - cc;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (s == 1 && i + 1 >= n) break;
            const int row = i + s;
            const float rs = s ? r.y : r.x;
            bool taken = false;
            if (VOLS && (s ? Q.object[1] : Q.object[0]) == last_object) {   // Sphere::hit_volumetric (sphere.rs:150-166)
                const V3 e = (o + d * h.t) - mk(s ? cx.y : cx.x, s ? cy.y : cy.x, s ? cz.y : cz.x);
                if (len2(e) <= rs * rs) {
                    h.prim = row;
                    h.inside = true;
                    taken = true;
                }
            }
            const float ds = s ? disc.y : disc.x, hb = s ? half_b.y : half_b.x;
            if (!taken && ds >= 0.0f) {                                      // sphere_t()'s root selection
                const float sqrtd = sqrt_bt(ds);
                // both roots, then selects (as in intersect_spheres_plain: C4 9.50 -> 9.45 ms, cloud 9.61 -> 9.56, profiles/r05l)
                const float t1 = -hb - sqrtd, t2 = -hb + sqrtd;
                const bool ok1 = !(t1 < tmin || t1 > h.t), ok2 = !(t2 < tmin || t2 > h.t);
                const bool ok = ok1 | ok2;
                h.t = ok ? (ok1 ? t1 : t2) : h.t;
                h.prim = ok ? row : h.prim;
                h.inside = ok ? false : h.inside;
            }
        }
    }
    return h;
}

// The scan for scenes without volumes (scene.json): an odd table's last sphere is visited on its own (no arithmetic for an empty
// second slot: 18 instructions per path segment with scene.json's five spheres), radius^2 comes from the table and a sphere is
// one 16-byte row (BtSphereRow: one scalar load per pair instead of three) -- C3 3.03 -> 2.92 ms on one box, profiles/r05f.
BT_DEV HitRec intersect_spheres_plain(const BtLaunch &P, V3 o, V3 d, float tmin, float tmax) {
    HitRec h;
    h.t = tmax;
    h.prim = -1;
    h.inside = false;
    h.p_neg = false;
    const int n = P.n_prims;
    typedef const __attribute__((address_space(4))) BtSphereRow RowK;
    RowK *rows = (RowK *)P.sphere_rows;         // wave-uniform indices -> scalar loads: x8 for two spheres, x4 for one
    // one sphere's turn in try_hit's scan, from its discriminant on (sphere_t()'s root selection; sphere.rs:121-148)
    auto visit = [&](int row, float ds, float hb) {
        if (ds >= 0.0f) {
            const float sqrtd = sqrt_bt(ds);
            // both roots and their range tests, then selects: the branches of sphere_t()'s "first root, else second" cost the
            // scalar unit more than the three extra instructions cost the vector unit (C3 2.766 -> 2.744 ms, profiles/r05j)
            const float t1 = -hb - sqrtd, t2 = -hb + sqrtd;
            const bool ok1 = !(t1 < tmin || t1 > h.t), ok2 = !(t2 < tmin || t2 > h.t);
            const float t = ok1 ? t1 : t2;
            const bool ok = ok1 | ok2;
            h.t = ok ? t : h.t;
            h.prim = ok ? row : h.prim;
        }
    };
    const int n_paired = n & ~1;
    int i = 0;
    for (; i < n_paired; i += 2) {
        RowK &A = rows[i], &B = rows[i + 1];
        const f2 cx = {A.cx, B.cx}, cy = {A.cy, B.cy}, cz = {A.cz, B.cz}, r2 = {A.r2, B.r2};
        const f2 ocx = splat2(o.x) - cx, ocy = splat2(o.y) - cy, ocz = splat2(o.z) - cz;
        const f2 half_b = (ocx * d.x + ocy * d.y) + ocz * d.z;
        const f2 cc = ((ocx * ocx + ocy * ocy) + ocz * ocz) - r2;
        const f2 disc = half_b * half_b - cc;
        visit(i, disc.x, half_b.x);
        visit(i + 1, disc.y, half_b.y);
    }
    if (i < n) {
        RowK &A = rows[i];
        const float ocx = o.x - A.cx, ocy = o.y - A.cy, ocz = o.z - A.cz;
        const float half_b = (ocx * d.x + ocy * d.y) + ocz * d.z;
        const float cc = ((ocx * ocx + ocy * ocy) + ocz * ocz) - A.r2;
        const float disc = half_b * half_b - cc;
        visit(i, disc, half_b);
    }
    return h;
}
template <bool VOLS>
BT_DEV HitRec intersect_spheres(const BtLaunch &P, V3 o, V3 d, float tmin, float tmax, int last_object) {
    if (VOLS) return intersect_sphere_pairs<VOLS>(P, o, d, tmin, tmax, last_object);
    return intersect_spheres_plain(P, o, d, tmin, tmax);
}

// ---- rect scenes without volumes: rows grouped by kind, one refined reciprocal per group of parallel planes ----------
// try_hit (mod.rs:389-402) keeps the smallest t and resolves exact ties by row order (a later plain rect / sphere
// replaces an equal t, a later cuboid face does not).  The same winner falls out of ANY processing order when every hit
// carries the rank `prio` of its row (bt_types.h BtRectAAN): smallest t, then largest prio -- the row that beats every
// other row of its tie in try_hit's scan.  That frees the order, so the twelve axis-aligned rows of a Cornell box are
// walked as three groups of parallel planes.
struct SortedHit { float t; uint32_t prio; float psgn; };
// The acceptance test of a rect row as ONE block of ISA: every condition narrows EXEC (v_cmpx), the running hit is then
// overwritten with plain moves in the surviving lanes, EXEC is restored.  The compiler's form of the same test (a compare
// into an SGPR pair per condition, s_and_b64 to combine them, v_cndmask per field) spends a dozen scalar instructions
// per row, and the one scalar unit of a CU was the busiest part of the rect builds (profiles/r01g/pmcq_cornell_r01g.log:
// 0.82 scalar instructions per CU-cycle).  ok_mask: lanes whose |q| passed the 1e-5 test (rect.rs:121-124);
// |a| <= lim_a, |b| <= lim_b: Rect::contains_point's `x * x <= w_sqr` (rect.rs:74-80) through the host's abs_limit(), the
// multiplications saved; the hit's p = dp * (+-1) is stored as dp ^ sign mask (same sign, zeros and NaN included); !(t < tmin): Clip (NaN passes, as in the C form, and
// then fails the containment tests); t against the running hit: sorted_accept()'s rule.
#ifndef BT_ACCEPT_ASM
#define BT_ACCEPT_ASM 1
#endif
BT_DEV void sorted_accept_rect(SortedHit &h, unsigned long long ok_mask, float t, float tmin, float a, float lim_a, float b,
                               float lim_b, uint32_t prio, float p, uint32_t sgn_mask) {
#if BT_ACCEPT_ASM
    unsigned long long saved, tmp;
    asm volatile("s_mov_b64 %[sv], exec\n\t"
                 "s_and_b64 exec, exec, %[okm]\n\t"
                 "v_cmpx_ngt_f32 vcc, %[tmin], %[t]\n\t"
                 "v_cmpx_ge_f32_e64 vcc, %[lima], |%[a]|\n\t"
                 "v_cmpx_ge_f32_e64 vcc, %[limb], |%[b]|\n\t"
                 "v_cmpx_le_f32 vcc, %[t], %[ht]\n\t"
                 "v_cmp_eq_f32 vcc, %[t], %[ht]\n\t"
                 "v_cmp_le_u32 %[tmp], %[prio], %[hp]\n\t"
                 "s_and_b64 vcc, vcc, %[tmp]\n\t"
                 "s_andn2_b64 exec, exec, vcc\n\t"
                 "v_mov_b32 %[ht], %[t]\n\t"
                 "v_mov_b32 %[hp], %[prio]\n\t"
                 "v_xor_b32 %[hs], %[sm], %[p]\n\t"
                 "s_mov_b64 exec, %[sv]"
                 : [ht] "+v"(h.t), [hp] "+v"(h.prio), [hs] "+v"(h.psgn), [sv] "=&s"(saved), [tmp] "=&s"(tmp)
                 : [okm] "s"(ok_mask), [tmin] "s"(tmin), [t] "v"(t), [lima] "s"(lim_a), [a] "v"(a), [limb] "s"(lim_b),
                   [b] "v"(b), [prio] "s"(prio), [p] "v"(p), [sm] "s"(sgn_mask)
                 : "vcc", "scc");
#else
    const bool ok = ((ok_mask >> (threadIdx.x & 63)) & 1) & !(t < tmin) & (fabsf(a) <= lim_a) & (fabsf(b) <= lim_b);
    const bool better = ok & ((t < h.t) | ((t == h.t) & (prio > h.prio)));
    h.t = better ? t : h.t;
    h.prio = better ? prio : h.prio;
    h.psgn = better ? __uint_as_float(__float_as_uint(p) ^ sgn_mask) : h.psgn;
#endif
}
BT_DEV void sorted_accept(SortedHit &h, bool ok, float t, uint32_t prio, float psgn) {
    const bool better = ok & ((t < h.t) | ((t == h.t) & (prio > h.prio)));
    h.t = better ? t : h.t;
    h.prio = better ? prio : h.prio;
    h.psgn = better ? psgn : h.psgn;
}
// p / q by the sequence hipcc emits for an IEEE-754 binary32 division (v_rcp_f32, two Newton steps on the reciprocal,
// quotient, two residual corrections; AMDGPU LowerFDIV32) WITHOUT its v_div_scale / v_div_fmas / v_div_fixup range
// handling: identical bits whenever no intermediate leaves the normal range.  Here q = a component of a unit direction
// with |q| > 1e-5 (smaller ones are a miss before t is looked at) and a hit needs clip_min <= t <= clip_max, so every
// accepted t has |p| = |t q| within [clip_min 1e-5, 1.01 clip_max]; the host uses this path only for clip_min >= 2^-30,
// clip_max <= 2^60 (bt_api.cpp).  A quotient outside that window is wrong at worst by being tiny, huge, inf or NaN --
// it fails the clip / containment tests like the exact one.  `r` is the refined reciprocal: it depends on q alone.
BT_DEV float refined_rcp(float q) {
    const float r0 = __builtin_amdgcn_rcpf(q);
    const float e0 = __builtin_fmaf(-q, r0, 1.0f);
    return __builtin_fmaf(e0, r0, r0);
}
BT_DEV float div_refined(float p, float q, float r) {
    const float m = p * r;
    const float e = __builtin_fmaf(-q, m, p);
    const float m2 = __builtin_fmaf(e, r, m);
    const float e2 = __builtin_fmaf(-q, m2, p);
    return __builtin_fmaf(e2, r, m2);
}
typedef const __attribute__((address_space(4))) BtRectAAN BtRectAANK;
// one group of BT_PRIM_RECT_AAN rows with normal axis W: rect_aan_t<W>() with the division's reciprocal hoisted
template <int W>
BT_DEV void aan_group(BtRectAANK *rows, int n, V3 o, V3 d, float tmin, SortedHit &h) {
    constexpr int A = W == 0 ? 1 : 0, B = W == 2 ? 1 : 2;
    const float dq = BT_COMP(d, W), ow = BT_COMP(o, W);
    const float oa = BT_COMP(o, A), da = BT_COMP(d, A), ob = BT_COMP(o, B), db = BT_COMP(d, B);
    const float r = refined_rcp(dq);
    const unsigned long long dq_ok = __builtin_amdgcn_ballot_w64(!(fabsf(dq) <= 1e-5f));
    auto one = [&](BtRectAANK &R) {                 // wave-uniform row -> one s_load_dwordx8
        const float dp = R.t_w - ow;
        const float t = div_refined(dp, dq, r);     // == dp / dq (see above)
        const float la = (oa + da * t) + R.it_a;
        const float lb = (ob + db * t) + R.it_b;
        sorted_accept_rect(h, dq_ok, t, tmin, la, R.lim_a, lb, R.lim_b, R.prio, dp, R.sgn_mask);
    };
    int i = 0;
    for (; i + 1 < n; i += 2) {                     // two rows per trip: half the loop bookkeeping on the scalar unit
        one(rows[i]);
        one(rows[i + 1]);
    }
    if (i < n) one(rows[i]);
}
// the BT_PRIM_RECT_LA rows: rect_t()'s arithmetic with q = dot(d, n) and its reciprocal formed once per normal
typedef const __attribute__((address_space(4))) BtRectLA BtRectLAK;
BT_DEV void la_rows(BtRectLAK *rows, int n, V3 o, V3 d, float tmin, SortedHit &h) {
    float q = 0.0f, r = 0.0f;
    unsigned long long q_ok = 0;
    for (int i = 0; i < n; ++i) {
        BtRectLAK &R = rows[i];                     // wave-uniform index -> scalar loads
        const V3 nrm = mk(R.n);
        if (R.first_of_normal) {                    // wave-uniform
            q = dot(d, nrm);
            r = refined_rcp(q);
            q_ok = __builtin_amdgcn_ballot_w64(!(fabsf(q) <= 1e-5f));
        }
        const float p = dot(mk(R.t) - o, nrm);
        const float t = div_refined(p, q, r);       // == p / q
        const V3 pos = o + d * t;
        const f2 ax = {R.a_x[0], R.a_x[1]}, ay = {R.a_y[0], R.a_y[1]}, az = {R.a_z[0], R.a_z[1]}, aw = {R.a_w[0], R.a_w[1]};
        const f2 l = ((ax * pos.x + ay * pos.y) + az * pos.z) + aw;       // (lu, lv) of rect_t()
        sorted_accept_rect(h, q_ok, t, tmin, l.x, R.lim[0], l.y, R.lim[1], R.prio, p, 0u);
    }
}
BT_DEV HitRec intersect_sorted(const BtLaunch &P, V3 o, V3 d, float tmin, float tmax) {
    SortedHit h;
    h.t = tmax;
    h.prio = 0xffffu;                               // a strict row never replaces it at t == tmax, a plain one does
    h.psgn = 0.0f;
    BtRectAANK *aan = (BtRectAANK *)P.aan_rows;
    if (P.n_aan[0]) aan_group<0>(aan, P.n_aan[0], o, d, tmin, h);
    if (P.n_aan[1]) aan_group<1>(aan + P.n_aan[0], P.n_aan[1], o, d, tmin, h);
    if (P.n_aan[2]) aan_group<2>(aan + P.n_aan[0] + P.n_aan[1], P.n_aan[2], o, d, tmin, h);
    if (P.n_la) la_rows((BtRectLAK *)P.la_rows, P.n_la, o, d, tmin, h);
    BtPrimK *prims = prim_table(P);
    const __attribute__((address_space(4))) int32_t *rows = (const __attribute__((address_space(4))) int32_t *)P.other_rows;
    for (int j = 0; j < P.n_other; ++j) {
        const int row = rows[j];
        BtPrimK &R = prims[row];
        const bool strict = (R.kind & BT_PRIM_STRICT) != 0;
        const uint32_t prio = strict ? 0xfffeu - (uint32_t)row : 0x10000u + (uint32_t)row;
        if ((R.kind & BT_PRIM_SHAPE_MASK) == BT_PRIM_SPHERE) {
            // Sphere::hit (sphere.rs:121-148) against the running clip == the near root unless it lies before tmin, then
            // the far one; accepted like every other hit (a root beyond the running clip loses the comparison)
            const V3 oc = o - mk(R.c);
            const float half_b = dot(oc, d), cc = len2(oc) - R.radius * R.radius;
            const float disc = half_b * half_b - cc, sqrtd = sqrtf(disc);
            const float t1 = -half_b - sqrtd, t2 = -half_b + sqrtd;
            const float t = t1 < tmin ? t2 : t1;
            sorted_accept(h, (disc >= 0.0f) & !(t < tmin), t, prio, 0.0f);
        } else {
            float t = 0.0f, q = 0.0f, p = 0.0f;
            // the row's own clip tests run against [tmin, tmax]; the running clip is the comparison in sorted_accept
            const bool hit = rect_t(o, d, R, tmin, __builtin_inff(), false, t, q, p);
            sorted_accept(h, hit, t, prio, p);
        }
    }
    HitRec out;
    out.t = h.t;
    out.inside = false;
    out.p_neg = h.psgn < 0.0f;
    out.prim = h.prio == 0xffffu ? -1 : (h.prio >= 0x10000u ? (int)(h.prio - 0x10000u) : (int)(0xfffeu - h.prio));
    return out;
}

template <bool RECTS = true, bool VOLS = true, bool SORTED = false>
BT_DEV HitRec intersect(const BtLaunch &P, V3 o, V3 d, float tmin, float tmax, int last_object, bool short_seg = false) {
    if (SORTED) return intersect_sorted(P, o, d, tmin, tmax);
    if (!RECTS && !short_seg) return intersect_spheres<VOLS>(P, o, d, tmin, tmax, last_object);
    HitRec h;
    h.t = tmax;
    h.prim = -1;
    h.inside = false;
    h.p_neg = false;
    const int n = P.n_prims;
    BtPrimK *prims = prim_table(P);
    for (int i = 0; i < n; ++i) intersect_row<RECTS, VOLS>(prims, i, o, d, tmin, last_object, h, short_seg);
    return h;
}
// Lens extension: the same loop over the rows listed in P.lens_prims (ascending, so ties resolve as in intersect()).
template <bool RECTS = true>
BT_DEV HitRec intersect_listed(const BtLaunch &P, V3 o, V3 d, float tmin, float tmax) {
    HitRec h;
    h.t = tmax;
    h.prim = -1;
    h.inside = false;
    h.p_neg = false;
    BtPrimK *prims = prim_table(P);
    const __attribute__((address_space(4))) int32_t *rows = (const __attribute__((address_space(4))) int32_t *)P.lens_prims;
    for (int j = 0; j < P.n_lens_prims; ++j) intersect_row<RECTS, false>(prims, rows[j], o, d, tmin, -1, h, true);
    return h;
}

// Object::pdf of a light (object/mod.rs:154-166; sphere.rs:44-61, rect.rs:92-108,
// cuboid.rs:56-81); 0 when the ray misses it (material.rs:313-316 unwrap_or_default).
// LightRef / PrimTab: the light header and the primitive table through the LDS copy and the global pointer (light index
// per lane), or both through the constant address space (scenes with ONE light, i.e. every bundled scene: the index is
// wave-uniform, the rows arrive by scalar loads instead of per-lane global loads with their latency).
template <bool RECTS, class LightRef, class PrimTab>
BT_DEV float light_pdf_impl(const BtLaunch &P, LightRef &Lt, PrimTab prims, const SceneLds &S, V3 o, V3 d) {
    if (Lt.kind == BT_LIGHT_SPHERE) {
        float t;
        if (!sphere_t(o, d, mk(Lt.centre), Lt.radius, P.clip_min, P.clip_max, t)) return 0.0f;
        return (t * t) / Lt.shadow;
    }
    if (RECTS && Lt.kind == BT_LIGHT_RECT) {
        float t, q, p;
        if (!rect_t(o, d, prims[Lt.prim_first], P.clip_min, P.clip_max, false, t, q, p)) return 0.0f;
        float shadow = S.faces[Lt.face_first].area * fabsf(q);
        return (t * t) / shadow;
    }
    if (RECTS && Lt.kind == BT_LIGHT_CUBOID) {
        float best_t = P.clip_max, best_q = 0.0f;
        int best = -1;
        for (int f = 0; f < Lt.prim_count; ++f) {
            float t, q, p;
            // rect.hit with the object-level clip, then `manifold.t < t` (cuboid.rs:63-75)
            if (rect_t(o, d, prims[Lt.prim_first + f], P.clip_min, P.clip_max, false, t, q, p) && t < best_t) {
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
template <bool RECTS = true>
BT_DEV float light_pdf(const BtLaunch &P, const BtLight &Lt, const SceneLds &S, V3 o, V3 d) {
    return light_pdf_impl<RECTS>(P, Lt, P.prims, S, o, d);
}
typedef const __attribute__((address_space(4))) BtLight BtLightK;
template <bool RECTS = true>
BT_DEV float light_pdf_only_light(const BtLaunch &P, const SceneLds &S, V3 o, V3 d) {
    BtLightK &Lt = *(BtLightK *)P.lights;
    return light_pdf_impl<RECTS>(P, Lt, prim_table(P), S, o, d);
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

// DensityMap::sample for maps whose bounds tests cannot fire (BtLaunch::vols_safe): density_sample() without the clamp of
// negative indices and the width / height / depth tests of density_at() -- cx = clamp(coord, 0, 1) * size lies in
// [0, size] (NaN clamps to 0), so floor and ceil lie in [0, dim - 1].  Same fetches, same lerps, same bits.
// Ptr: the map in LDS (address space 3: ds_read with 32-bit addressing) or wherever `const float *` points (a map beyond the
// LDS budget stays in global memory; through the generic pointer every fetch is a flat load with a 64-bit address).
// Indices are non-negative and the map has fewer than 2^24 cells (vols_safe), so the row arithmetic runs in 24-bit
// multiply-adds (full rate) instead of v_mul_lo_u32.
template <class Ptr>
BT_DEV float density_sample_safe(const BtVolume &vol, Ptr density, V3 coord) {
    const float cx = fminf(fmaxf(coord.x, 0.0f), 1.0f) * vol.size.x;
    const float cy = fminf(fmaxf(coord.y, 0.0f), 1.0f) * vol.size.y;
    const float cz = fminf(fmaxf(coord.z, 0.0f), 1.0f) * vol.size.z;
    const float fx = floorf(cx), fy = floorf(cy), fz = floorf(cz);
    const float tx = cx - truncf(cx), ty = cy - truncf(cy), tz = cz - truncf(cz);
    const uint32_t x0 = (uint32_t)fx, y0 = (uint32_t)fy, z0 = (uint32_t)fz;
    const uint32_t x1 = (uint32_t)ceilf(cx), y1 = (uint32_t)ceilf(cy), z1 = (uint32_t)ceilf(cz);
    const uint32_t W = (uint32_t)vol.width, H = (uint32_t)vol.height;
    Ptr d = density + vol.offset;
    const uint32_t zh0 = __umul24(z0, H), zh1 = __umul24(z1, H);
    const uint32_t r00 = __umul24(zh0 + y0, W), r01 = __umul24(zh0 + y1, W);
    const uint32_t r10 = __umul24(zh1 + y0, W), r11 = __umul24(zh1 + y1, W);
    const float a = lerpf(d[r00 + x0], d[r00 + x1], tx), b = lerpf(d[r01 + x0], d[r01 + x1], tx);
    const float z_lo = lerpf(a, b, ty);
    const float c = lerpf(d[r10 + x0], d[r10 + x1], tx), e = lerpf(d[r11 + x0], d[r11 + x1], tx);
    const float z_hi = lerpf(c, e, ty);
    return lerpf(z_lo, z_hi, tz);
}
// march_density() through the BtVolBox table: rel / size by div_refined() -- the same bits as the IEEE division whenever
// no intermediate leaves the normal range, which holds for 2^-20 <= size <= 2^20 (box.ok) and |rel| within
// [2^-60, 2^60]; a wave in which any lane falls outside takes the exact path below.
BT_DEV float march_density_box(const BtLaunch &P, const SceneLds &S, int vol_index, const BtVolBox &box, V3 pos) {
    const BtVolume &vol = S.volumes[vol_index];
    const V3 rel = pos - mk(box.bmin), size = mk(box.size);
    // (min / max of the three magnitudes: two compares instead of nine; an exact 0 now also takes the exact path, and a
    // NaN component, which fminf / fmaxf skip, gives NaN on either path and is clamped to 0 by the sampler)
    const float ax = fabsf(rel.x), ay = fabsf(rel.y), az = fabsf(rel.z);
    const bool in_range = box.ok != 0.0f && fminf(fminf(ax, ay), az) >= 0x1p-60f && fmaxf(fmaxf(ax, ay), az) <= 0x1p60f;
    V3 coord;
    if (__ballot(!in_range) == 0ull) {
        coord = mk(div_refined(rel.x, size.x, box.rcp.x), div_refined(rel.y, size.y, box.rcp.y), div_refined(rel.z, size.z, box.rcp.z));
    } else {
        coord = mk(rel.x / size.x, rel.y / size.y, rel.z / size.z);
    }
    typedef const __attribute__((address_space(3))) float *LdsF;
    float dens;
    if (P.vols_safe && P.n_density <= BT_DENSITY_LDS_MAX) dens = density_sample_safe(vol, (LdsF)S.density, coord);   // the kernel's dens_lds
    else dens = density_sample(vol, S.density, coord);       // a map in global memory, or one whose bounds tests can fire
    return P.volume_step * dens;
}

// The scatter probability of one march step, Volume::shade's `volume_step * density.sample(coord)` (volume.rs:26-35)
// with coord = (pos - bbox.min) / bbox.size of the sphere's bounding box (sphere.rs:35-38).
BT_DEV float march_density(const BtLaunch &P, const SceneLds &S, int vol_index, V3 prim_c, float prim_radius, V3 pos) {
    const BtVolume &vol = S.volumes[vol_index];
    const V3 hsz = mk(prim_radius, prim_radius, prim_radius);
    const V3 bmin = prim_c - hsz, bmax = prim_c + hsz;                // sphere.rs:35-38
    const V3 size = bmax - bmin;
    const V3 rel = pos - bmin;
    const V3 coord = mk(rel.x / size.x, rel.y / size.y, rel.z / size.z);
    return P.volume_step * density_sample(vol, S.density, coord);
}

// lanes set in a wave64 mask, as two 32-bit scalar popcounts: comparisons of the result stay on the scalar unit (the 64-bit
// popcount's result is compared with a 64-bit VALU compare)
BT_DEV uint32_t popc64(unsigned long long m) { return (uint32_t)__builtin_popcount((uint32_t)m) + (uint32_t)__builtin_popcount((uint32_t)(m >> 32)); }

BT_DEV unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// ---- lens EXTENSION (not in the reference: SURVEY F1, 8 f-4; default off; mirrors oracle/bt_oracle.c) ----
// A point mass of Schwarzschild radius rs bends every non-marching path segment.  Inside the sphere of
// influence the photon follows the Schwarzschild null geodesic, integrated with fixed-step RK4 on
// (x, v): x'' = -1.5 rs h^2 x / r^5 with h = |x x v| (conserved); every step's chord is intersected like
// a volume-march step (clip [0, |chord|]).  Outside the sphere rays are straight.
BT_DEV V3 cross(V3 a, V3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
// 1/sqrt(x) of the lens march: integer seed + three Newton steps with explicit fmaf (12 full-rate instructions instead
// of the correctly rounded sqrt + divide, ~110 cycles); bit-identical to lens_rsqrt() in oracle/bt_oracle.c.
BT_DEV float lens_rsqrt(float x) {
    float y = __uint_as_float(0x5f375a86u - (__float_as_uint(x) >> 1));
    const float hx = 0.5f * x;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float t = hx * y;
        const float e = __builtin_fmaf(-t, y, 1.5f);
        y = y * e;
    }
    return y;
}
BT_DEV V3 lens_accel(const BtLaunch &P, V3 x, float h2) {
    const V3 rel = x - mk(P.lens_c);
    const float r2 = len2(rel);
    const float y = lens_rsqrt(r2);
    const float y2 = y * y;
    const float y5 = (y2 * y2) * y;               // r^-5
    const float k = (-1.5f * P.lens_rs * h2) * y5;
    return rel * k;
}
BT_DEV void lens_rk4(const BtLaunch &P, float h2, V3 x, V3 v, V3 &x1, V3 &v1) {
    const float dt = P.lens_step, hdt = 0.5f * dt, w = dt / 6.0f;
    const V3 k1x = v, k1v = lens_accel(P, x, h2);
    const V3 k2x = v + k1v * hdt, k2v = lens_accel(P, x + k1x * hdt, h2);
    const V3 k3x = v + k2v * hdt, k3v = lens_accel(P, x + k2x * hdt, h2);
    const V3 k4x = v + k3v * dt, k4v = lens_accel(P, x + k3x * dt, h2);
    const V3 sx = (k1x + (k2x + k3x) * 2.0f) + k4x;
    const V3 sv = (k1v + (k2v + k3v) * 2.0f) + k4v;
    x1 = x + sx * w;
    v1 = v + sv * w;
}
// State of a bent path segment that is marched in instalments (so that a wave is not held back by its
// longest segment): the photon itself lives in the caller's (x, v).
struct LensState {
    float remaining, travelled, h2;
    int steps_left;
    bool first, inside;        // inside: h2 is valid and the RK4 march is in progress
};
BT_DEV void lens_begin(const BtLaunch &P, LensState &st) {
    st.remaining = P.clip_max;
    st.travelled = 0.0f;
    st.h2 = 0.0f;
    st.steps_left = P.lens_max_steps;
    st.first = true;
    st.inside = false;
}
// Advances the bent segment from (x, v) by at most `budget` RK4 steps.  Returns 2 = not finished (call
// again), 1 = hit (h = hit on the chord (x, v), which are updated to that chord), 0 = miss ((x, v) = the ray
// that reaches the root), -1 = captured by the horizon.  st.travelled = path length before the returned
// chord.  The arithmetic and its order are those of lens_trace() in oracle/bt_oracle.c.
template <bool RECTS = true>
BT_DEV int lens_advance(const BtLaunch &P, V3 &x, V3 &v, LensState &st, HitRec &h, int budget, unsigned long long &steps) {
    const V3 c = mk(P.lens_c);
    const float R2 = P.lens_radius * P.lens_radius, rs2 = P.lens_rs * P.lens_rs;
    h.prim = -1;
    for (;;) {
        if (!st.inside) {
            const V3 rel = x - c;
            const float r2 = len2(rel);
            if (!(r2 <= R2)) {
                // straight flight to the sphere of influence (or to the end of the clip)
                const float hb = dot(rel, v), cc = r2 - R2, disc = hb * hb - cc;
                float t_enter = __builtin_inff();
                if (disc >= 0.0f) {
                    const float te = -hb - sqrt_bt(disc);
                    if (te > 0.0f) t_enter = te;
                }
                const float seg = fminf(t_enter, st.remaining);
                h = intersect<RECTS, false>(P, x, v, st.first ? P.clip_min : 0.0f, seg, -1);
                if (h.prim >= 0) return 1;
                if (!(t_enter < st.remaining)) return 0;
                x = x + v * t_enter;
                st.remaining -= t_enter;
                st.travelled += t_enter;
                st.first = false;
            }
            st.h2 = len2(cross(x - c, v));
            st.inside = true;
        }
        for (;;) {
            if (budget-- <= 0) return 2;
            if (st.steps_left-- <= 0) {        // step budget exhausted: the segment is abandoned as a miss
                v = normalize(v);
                return 0;
            }
            steps += 1;
            V3 x1, v1;
            lens_rk4(P, st.h2, x, v, x1, v1);
            const V3 chord = x1 - x;
            const float l2 = len2(chord), rl = lens_rsqrt(l2);
            const float len = l2 * rl;
            const V3 dirn = chord * rl;
            const float seg = fminf(len, st.remaining);
            // the chord starts within lens_radius of the centre: if it is no longer than lens_margin only the listed
            // rows can be touched (BtLaunch::lens_prims)
            if (len <= P.lens_margin)
                h = intersect_listed<RECTS>(P, x, dirn, st.first ? P.clip_min : 0.0f, seg);
            else
                h = intersect<RECTS, false>(P, x, dirn, st.first ? P.clip_min : 0.0f, seg, -1);
            if (h.prim >= 0) {
                v = dirn;
                return 1;
            }
            if (!(len < st.remaining)) {
                v = dirn;
                return 0;
            }
            st.remaining -= len;
            st.travelled += len;
            st.first = false;
            x = x1;
            v = v1;
            const V3 rel = x - c;
            const float r2 = len2(rel);
            if (r2 <= rs2) return -1;
            if (r2 > R2 && dot(rel, v) > 0.0f) {
                v = normalize(v);
                st.inside = false;
                break;
            }
        }
    }
}

} // namespace
