// bt_kernels_sorted.hip -- the render kernel with workgroup-level regrouping of paths.
//
// Same arithmetic, same per-path event sequence and the same per-pixel summation order as
// bt_render_kernel (bt_kernels.hip) -- the outputs are bit-identical -- but the 256 paths of a
// 16x16 pixel tile no longer belong to fixed lanes.  Each path's state lives in an LDS slot
// (six float4 planes, structure-of-arrays, 96 B per path); lanes are workers.  One iteration:
//
//   1. a lane picks the slot `order[tid]`, loads the path (6 x ds_read_b128),
//   2. runs the path's pending random EVENT (camera ray | Diffuse | Metallic | Glass | volume
//      step) and then TRACEs the new ray (one segment), classifies the hit -> next event kind,
//      stores the path back (6 x ds_write_b128),
//   3. the workgroup counting-sorts the slots by next event kind: per-wave ballots + popcounts,
//      per-wave totals through LDS, lane rank from mbcnt -> new `order`.
//
// After the sort consecutive lanes hold paths with the same pending event, so whole waves run
// only the camera code, or only the Diffuse code, ... instead of every wave running every
// branch at partial occupancy.  This is the north-star's "ballot / prefix-sum compaction of
// rays to tame divergence", done at LDS scope (the ray state never leaves the CU), and its
// "state staged in LDS".  Terminated paths regenerate in place (kind GEN), finished pixels sort
// to the end (kind DONE) and their lanes skip the iteration.
#include "bt_device.hpp"

namespace {

// sort order of the kinds: neighbours share code (the three surface kinds share the basis block)
enum { K_DIFFUSE = 0, K_METALLIC = 1, K_GLASS = 2, K_VOLUME = 3, K_GEN = 4, K_DONE = 5, K_COUNT = 6 };
enum { F_FRONT = 1 << 4, F_INSIDE = 1 << 5, F_VOLBACK = 1 << 6, F_HAVE_FIRST = 1 << 7 };   // packed word, above the kind

constexpr int SLOTS = 256;

BT_DEV float4 mk4(V3 v, float w) { return make_float4(v.x, v.y, v.z, w); }
BT_DEV V3 xyz(float4 v) { return mk(v.x, v.y, v.z); }
BT_DEV float as_f(int v) { return __int_as_float(v); }
BT_DEV float as_f(uint32_t v) { return __uint_as_float(v); }

} // namespace

template <int OUTPUT>
__global__ __launch_bounds__(256, 5) void bt_render_sorted_kernel(BtLaunch P) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int PLANES = OUTPUT == 0 ? 6 : 7;

    // ---- LDS carve: path planes | order[2] | per-wave kind counts | scene tables ----
    float4 *plane = (float4 *)smem;                                    // [PLANES][SLOTS]
    unsigned short *order = (unsigned short *)(plane + PLANES * SLOTS); // [2][SLOTS]
    uint32_t *counts = (uint32_t *)(order + 2 * SLOTS);                // [K_COUNT][4 waves]
    SceneLds S;
    {
        unsigned char *p = (unsigned char *)(counts + K_COUNT * 4);
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
            l.kind_object = (R.kind & BT_PRIM_SHAPE_MASK) | (R.object << 8);
            l.material = R.material;
            l.volume = R.volume;
            l.rcp_radius = 0.0f;
            lite[i] = l;
        }
        for (int i = threadIdx.x; i < P.n_materials; i += blockDim.x) mats[i] = P.materials[i];
        for (int i = threadIdx.x; i < P.n_volumes; i += blockDim.x) vols[i] = P.volumes[i];
        for (int i = threadIdx.x; i < P.n_lights; i += blockDim.x) lights[i] = P.lights[i];
        for (int i = threadIdx.x; i < P.n_light_faces; i += blockDim.x) faces[i] = P.light_faces[i];
        const bool dens_lds = P.n_density > 0 && P.n_density <= BT_DENSITY_LDS_MAX;
        if (dens_lds)
            for (int i = threadIdx.x; i < P.n_density; i += blockDim.x) dens[i] = P.density[i];
        S.lite = lite; S.materials = mats; S.volumes = vols; S.lights = lights; S.faces = faces;
        S.density = dens_lds ? dens : P.density;
    }

    // ---- tile mapping: slot s of the workgroup is pixel (s & 15, s >> 4) of the tile ----
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t tile = P.sharded ? (blockIdx.x * P.world + P.rank) : blockIdx.x;
    const uint32_t tx = tile % P.tiles_x, ty = tile / P.tiles_x;
    const uint32_t nn = (uint32_t)(P.subsample_n * P.subsample_n);
    const uint32_t total_in_frame = (uint32_t)P.samples * nn;
    const uint32_t sample0 = P.sample_base * nn;
    const V3 mcx = mk(P.cam_cx), mcy = mk(P.cam_cy), mcz = mk(P.cam_cz);
    auto slot_pixel = [&](uint32_t s, uint32_t &px, uint32_t &py) {
        // same pixel <-> (wave, lane) layout as bt_render_kernel: 8x8 quadrants
        const uint32_t w = s >> 6, l = s & 63;
        px = tx * BT_TILE_DIM + (((w & 1) << 3) | (l & 7));
        py = ty * BT_TILE_DIM + (((w >> 1) << 3) | (l >> 3));
    };
    auto slot_out = [&](uint32_t s, uint32_t px, uint32_t py) -> float * {
        const uint32_t w = s >> 6, l = s & 63;
        const uint32_t lx = ((w & 1) << 3) | (l & 7), ly = ((w >> 1) << 3) | (l >> 3);
        return P.sharded ? P.out + ((size_t)blockIdx.x * (BT_TILE_DIM * BT_TILE_DIM) + ly * BT_TILE_DIM + lx) * 4
                         : P.out + ((size_t)py * P.width + px) * 4;
    };

    // ---- initial state: every in-frame pixel waits for the camera ray of its sample 0 ----
    {
        uint32_t px, py;
        slot_pixel(tid, px, py);
        const bool in_frame = (ty < P.tiles_y) && (px < P.width) && (py < P.height) && total_in_frame > 0;
        V3 acc = mk(0, 0, 0);
        if (in_frame) {
            const float *o = slot_out(tid, px, py);
            acc = mk(o[0], o[1], o[2]);                        // `*r += pixel.r` (buffer.rs:159-164)
        }
        plane[0 * SLOTS + tid] = mk4(mk(0, 0, 0), as_f(0u));                     // ro, k
        plane[1 * SLOTS + tid] = mk4(mk(0, 0, -1), as_f(0u));                    // rd, event
        plane[2 * SLOTS + tid] = mk4(mk(1, 1, 1), as_f(in_frame ? K_GEN : K_DONE)); // beta, packed
        plane[3 * SLOTS + tid] = mk4(mk(0, 0, 0), as_f(-1));                     // L, last_object
        plane[4 * SLOTS + tid] = mk4(acc, 0.0f);                                 // acc, hit_t
        plane[5 * SLOTS + tid] = mk4(mk(0, 0, 0), as_f(-1));                     // normal, prim
        if (OUTPUT != 0) plane[6 * SLOTS + tid] = mk4(mk(0, 0, 0), __builtin_inff());   // first AOV, first_depth
        order[tid] = (unsigned short)tid;
    }
    __syncthreads();

    unsigned long long segments = 0;
    int cur = 0;
    for (;;) {
        const uint32_t slot = order[cur * SLOTS + tid];
        const float4 s0 = plane[0 * SLOTS + slot], s1 = plane[1 * SLOTS + slot], s2 = plane[2 * SLOTS + slot];
        int packed = __float_as_int(s2.w);
        int kind = packed & 15;

        if (kind != K_DONE) {
            const float4 s3 = plane[3 * SLOTS + slot], s4 = plane[4 * SLOTS + slot], s5 = plane[5 * SLOTS + slot];
            V3 ro = xyz(s0), rd = xyz(s1), beta = xyz(s2), L = xyz(s3), acc = xyz(s4), normal = xyz(s5);
            uint32_t k = __float_as_uint(s0.w), event = __float_as_uint(s1.w);
            int bounce = (packed >> 8) & 0xff, vbounce = (packed >> 16) & 0xff;
            int last_object = __float_as_int(s3.w);
            const float hit_t = s4.w;
            const int prim = __float_as_int(s5.w);
            const bool front = (packed & F_FRONT) != 0, inside = (packed & F_INSIDE) != 0, vol_back = (packed & F_VOLBACK) != 0;
            bool have_first = (packed & F_HAVE_FIRST) != 0;
            V3 first = mk(0, 0, 0);
            float first_depth = __builtin_inff();
            if (OUTPUT != 0) {
                const float4 s6 = plane[6 * SLOTS + slot];
                first = xyz(s6);
                first_depth = s6.w;
            }
            uint32_t px, py;
            slot_pixel(slot, px, py);
            const uint32_t pixel_index = py * P.width + px;

            // mod.rs:304-315 -> Chunk::write_* -> Buffer::write_* (buffer.rs:159-178)
            auto finish_sample = [&]() {
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
                kind = k < total_in_frame ? K_GEN : K_DONE;
            };

            // =================== EVENT: the path's pending random event ===================
            const V3 pos = ro + rd * hit_t;                    // manifold.position of the stored hit
            const BtPrimLite &pl = S.lite[prim < 0 ? 0 : prim];
            const uint32_t sample_index = sample0 + k;
            const U4 u = philox(pixel_index, sample_index, kind == K_GEN ? 0u : event, 0u, P.seed_lo, P.seed_hi);
            const uint32_t w1 = kind == K_METALLIC ? u.x : (kind == K_GLASS ? u.y : u.z);
            const uint32_t w2 = kind == K_METALLIC ? u.y : (kind == K_GLASS ? u.z : u.w);
            const float r1 = uniform_sample(w1, 0.0f, P.tau_scale), r2 = uniform_sample(w2, 0.0f, P.one_scale);
            float sn, cs;
            sincos_bt(r1, sn, cs);

            V3 new_o = pos, dir = rd;
            bool late_end = false;
            const int ev = kind;

            if (ev == K_GEN) {
                // ---- camera ray (mod.rs:271-302, ray.rs:103-113,126-137) ----
                float u_sub = 0.0f, v_sub = 0.0f;
                if (P.subsample_n > 1) {
                    const uint32_t n = (uint32_t)P.subsample_n;
                    const uint32_t sub = k % (n * n);
                    const float width_sub = 1.0f / (float)n;
                    u_sub = (float)(sub % n) * width_sub;
                    v_sub = (float)(sub / n) * width_sub;
                }
                const float v0 = (float)py * P.pixel_height - 1.0f;
                const float u0 = (float)px * P.pixel_width - 1.0f;
                const float u_offset = u_sub * P.pixel_width + uniform_sample(u.x, P.jitter_u_lo, P.jitter_u_scale);
                const float v_offset = v_sub * P.pixel_height + uniform_sample(u.y, P.jitter_v_lo, P.jitter_v_scale);
                const float uu = u0 + u_offset, vv = v0 + v_offset;
                const float yrot = P.xfov * 0.5f * -uu;
                const float xrot = P.yfov * 0.5f * -vv;
                float sy, cy, sx, cx;
                sincos_bt(yrot, sy, cy);
                sincos_bt(xrot, sx, cx);
                const V3 d_cam = mk(-(cx * sy), sx, -(cx * cy));
                new_o = mk(P.cam_t) + mk(0.0f, 0.0f, 0.0f);
                dir = normalize_or_zero(xf_vector(mcx, mcy, mcz, d_cam));
                if (P.has_focus) {                        // mod.rs:286-299; disk angle = r1, radius = r2
                    const V3 d1 = normalize(dir);
                    const V3 defocus = (mk(P.disk_x) * cs + mk(P.disk_y) * sn) * r2;
                    const V3 defocus_offset = xf_vector(mcx, mcy, mcz, defocus * P.aperture);
                    const float frac_f_z = P.focus / fabsf(d_cam.z);
                    new_o = new_o + defocus_offset;
                    dir = d1 * frac_f_z - defocus_offset;
                }
                beta = mk(1, 1, 1);
                L = mk(0, 0, 0);
                bounce = 0; vbounce = 0; last_object = -1;
                event = 1;
                have_first = false;
                first = mk(0, 0, 0);
                first_depth = __builtin_inff();
            } else {
                event += 1;
                const BtMaterial &M = S.materials[pl.material];
                int light_index = 0;
                bool to_light = false;
                if (ev == K_DIFFUSE) {
                    light_index = (int)__umulhi(u.x, (uint32_t)P.n_lights);   // material.rs:106-119
                    to_light = bernoulli(u.y, 0.5f);                          // Pdf::Mix (:269-275)
                }
                const bool is_cosine = ev == K_DIFFUSE && !to_light;
                const bool in_frame_of_normal = is_cosine || ev == K_METALLIC || ev == K_GLASS;
                // UnitSphere (distr.rs:10-21), UnitHemisphere (:48-59, z = 1 - r2), Cosine (:86-97)
                const float sq = sqrtf(is_cosine ? r2 : r2 * (1.0f - r2));
                const float lx_ = (is_cosine ? cs : cs * 2.0f) * sq;
                const float ly_ = (is_cosine ? sn : sn * 2.0f) * sq;
                float lz_ = in_frame_of_normal ? 1.0f - r2 : 1.0f - 2.0f * r2;
                if (is_cosine) lz_ = sqrtf(1.0f - r2);
                V3 v = mk(lx_, ly_, lz_);
                if (in_frame_of_normal) {
                    V3 z_axis = normalize(normal), x_axis, y_axis;
                    orthonormal_pair(z_axis, x_axis, y_axis);
                    v = (x_axis * lx_ + y_axis * ly_) + z_axis * lz_;
                }

                if (ev == K_DIFFUSE) {
                    if (to_light) {                                           // Pdf::Light (:262-268)
                        const BtLight &Lt = S.lights[light_index];
                        V3 point;
                        if (Lt.kind == BT_LIGHT_SPHERE) {                     // sphere.rs:40-42
                            point = mk(Lt.centre) + v * Lt.radius;
                        } else if (Lt.kind == BT_LIGHT_RECT) {
                            point = face_random_point(S.faces[Lt.face_first], u.z, u.w);
                        } else if (Lt.kind == BT_LIGHT_CUBOID) {              // cuboid.rs:47-54
                            const U4 e = philox(pixel_index, sample_index, event - 1u, 1u, P.seed_lo, P.seed_hi);
                            const float chosen = uniform_sample(e.x, 0.0f, Lt.total_scale);
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
                        dir = v;                                              // Pdf::Diffuse (:224-230)
                    }
                } else if (ev == K_METALLIC) {                                // :231-239
                    dir = reflect(rd, normal) + v * M.roughness;
                } else if (ev == K_GLASS) {                                   // :240-261
                    const float ior = front ? M.inv_ior : M.ior;
                    const float cos_theta = fminf(dot(-rd, normal), 1.0f);
                    const float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
                    const float fr = fresnel(rd, normal, ior);
                    V3 base;
                    if (ior * sin_theta > 1.0f || bernoulli(u.x, fr))
                        base = reflect(rd, normal);
                    else
                        base = refract(rd, normal, ior);
                    dir = base + v * M.roughness;
                } else {
                    // ---- Volume::shade (volume.rs:26-60) ----
                    const BtVolume &vol = S.volumes[pl.volume < 0 ? 0 : pl.volume];
                    const V3 prim_c = mk(pl.c);
                    const V3 hsz = mk(pl.radius, pl.radius, pl.radius);
                    const V3 bmin = prim_c - hsz, bmax = prim_c + hsz;        // sphere.rs:35-38
                    const V3 size = bmax - bmin;
                    const V3 rel = pos - bmin;
                    const V3 coord = mk(rel.x / size.x, rel.y / size.y, rel.z / size.z);
                    const float density = P.volume_step * density_sample(vol, S.density, coord);
                    if (density >= 1.0f || bernoulli(u.x, density)) {
                        if (inside) new_o = pos - (rd * P.volume_step) * u24(u.y);
                        dir = v;
                        beta = beta * mk(0.8f, 0.8f, 0.8f);
                        if (OUTPUT != 0 && !have_first) {
                            have_first = true;
                            if (OUTPUT == 1) first = mk(0.8f, 0.8f, 0.8f);
                            if (OUTPUT == 2) first = normal;
                            if (OUTPUT == 3) first_depth = hit_t;
                        }
                    }
                    if (vol_back) {                                           // mod.rs:504-505
                        bounce += 1;
                        last_object = -1;
                    } else {                                                  // mod.rs:507-513
                        last_object = pl.kind_object >> 8;
                        vbounce += 1;
                    }
                }
            }

            // Ray::new normalizes (ray.rs:96-101); for the camera this is the last normalize of mod.rs:296-301
            const V3 nd = normalize(dir);

            if (ev == K_DIFFUSE || ev == K_METALLIC || ev == K_GLASS) {
                const BtMaterial &M = S.materials[pl.material];
                bool scatter = true;
                float weight = 1.0f;                                          // material.pdf / shade.pdf
                if (ev == K_DIFFUSE) {
                    const BtLight &Lt = S.lights[(int)__umulhi(u.x, (uint32_t)P.n_lights)];
                    const float pd = dot(normal, nd) * 0.318309886183790671538f;   // diffuse_pdf (:301-303)
                    const float plight = light_pdf(P, Lt, S, pos, nd);
                    const float p = lerpf(pd, plight, 0.5f);                  // :294-296
                    scatter = !(fabsf(p) <= 1e-5f);                           // Pdf::pdf (:279-286)
                    weight = pd / p;                                          // Material::pdf (:204) / shade.pdf
                }
                if (OUTPUT != 0 && !have_first) {
                    have_first = true;
                    if (scatter) {      // data.albedo ColorData (material.rs:99-104,140-145,169-174)
                        if (OUTPUT == 1) first = mk(M.albedo);
                        if (OUTPUT == 2) first = normal;
                        if (OUTPUT == 3) first_depth = hit_t;
                    } else {            // ColorData::from_emitted(emitted) (mod.rs:483-485)
                        if (OUTPUT == 1) first = mk(M.emitted);
                    }
                }
                if (scatter) {
                    beta = beta * (mk(M.albedo) * weight);
                    bounce += 1;
                    last_object = -1;
                } else {
                    late_end = true;
                }
            }
            ro = new_o;
            rd = nd;

            // sample() / sample_volumetric() return black past the limits (mod.rs:323-325, 352-354)
            if (!late_end) late_end = last_object >= 0 ? (vbounce > P.max_volume_bounces) : (bounce > P.max_bounces);

            float new_t = 0.0f;
            int new_prim = -1, flags = 0;
            V3 new_normal = mk(0, 0, 0);
            if (late_end) {
                finish_sample();                               // next kind: K_GEN or K_DONE
            } else {
                // =================== TRACE: try_hit / try_hit_volume (mod.rs:389-427) ===================
                const bool marching = last_object >= 0;
                if (!marching) vbounce = 0;                    // sample() -> sample_volume(.., 0), mod.rs:335
                segments += 1;
                const float tmin = marching ? 0.0f : P.clip_min;
                const float tmax = marching ? P.volume_step : P.clip_max;
                const HitRec h = intersect(P, ro, rd, tmin, tmax, last_object);
                if (h.prim < 0) {
                    // sample_root (mod.rs:429-452)
                    L = L + beta * mk(P.root_color);
                    if (OUTPUT != 0 && !have_first) {
                        have_first = true;
                        if (OUTPUT == 1) first = mk(P.root_albedo);
                        if (OUTPUT == 2) first = P.root_has_albedo ? -rd : mk(0, 0, 0);
                        if (OUTPUT == 3) first_depth = P.root_has_albedo ? P.clip_max : __builtin_inff();
                    }
                    finish_sample();
                } else {
                    const BtPrimLite &hl = S.lite[h.prim];
                    const int pshape = hl.kind_object & 0xff;
                    new_prim = h.prim;
                    new_t = h.t;
                    const V3 hpos = ro + rd * h.t;
                    bool vol_face = false, hfront = false;
                    if (h.inside) {                       // generate_volume_manifold (sphere.rs:63-83)
                        flags |= F_INSIDE;
                        vol_face = true;
                    } else if (pshape == BT_PRIM_SPHERE) { // generate_surface_manifold (sphere.rs:85-119)
                        V3 nrm = hpos - mk(hl.c);
                        nrm = mk(nrm.x / hl.radius, nrm.y / hl.radius, nrm.z / hl.radius);
                        hfront = dot(rd, nrm) < 0.0f;
                        new_normal = hfront ? nrm : -nrm;
                        vol_face = hl.volume >= 0;
                        if (vol_face && !hfront) flags |= F_VOLBACK;
                    } else {                              // rect.rs:138-142
                        hfront = h.p_neg;
                        new_normal = hfront ? mk(hl.c) : -mk(hl.c);
                    }
                    if (hfront) flags |= F_FRONT;
                    if (vol_face) {
                        kind = K_VOLUME;                  // sample_volume (mod.rs:488-523)
                    } else {
                        // sample_surface (mod.rs:454-486): emitted, then Material::shade
                        const BtMaterial &HM = S.materials[hl.material];
                        L = L + beta * mk(HM.emitted);
                        if (HM.kind == BT_MAT_DIFFUSE) kind = K_DIFFUSE;
                        else if (HM.kind == BT_MAT_METALLIC) kind = K_METALLIC;
                        else if (HM.kind == BT_MAT_GLASS) kind = K_GLASS;
                        else {
                            // Flat / Emissive: no scatter -> ColorData::from_emitted (mod.rs:483-485)
                            if (OUTPUT != 0 && !have_first) {
                                have_first = true;
                                if (OUTPUT == 1) first = mk(HM.emitted);
                            }
                            finish_sample();
                        }
                    }
                }
            }

            // ---- store the path ----
            packed = kind | flags | (have_first ? F_HAVE_FIRST : 0) | (bounce << 8) | (vbounce << 16);
            plane[0 * SLOTS + slot] = mk4(ro, as_f(k));
            plane[1 * SLOTS + slot] = mk4(rd, as_f(event));
            plane[2 * SLOTS + slot] = mk4(beta, as_f(packed));
            plane[3 * SLOTS + slot] = mk4(L, as_f(last_object));
            plane[4 * SLOTS + slot] = mk4(acc, new_t);
            plane[5 * SLOTS + slot] = mk4(new_normal, as_f(new_prim));
            if (OUTPUT != 0) plane[6 * SLOTS + slot] = mk4(first, first_depth);
        }

        // =================== counting sort of the slots by next event kind ===================
        unsigned long long mine = 0;
        uint32_t wave_count[K_COUNT];
#pragma unroll
        for (int kk = 0; kk < K_COUNT; ++kk) {
            const unsigned long long m = __ballot(kind == kk);
            wave_count[kk] = (uint32_t)__popcll(m);
            if (kind == kk) mine = m;
        }
        const uint32_t rank_in_wave = __builtin_amdgcn_mbcnt_hi((uint32_t)(mine >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mine, 0u));
        if (lane == 0) {
#pragma unroll
            for (int kk = 0; kk < K_COUNT; ++kk) counts[kk * 4 + wave] = wave_count[kk];
        }
        __syncthreads();
        uint32_t base = 0, done_total = 0;
#pragma unroll
        for (int kk = 0; kk < K_COUNT; ++kk) {
            const uint4 c = *(const uint4 *)&counts[kk * 4];
            const uint32_t tot = c.x + c.y + c.z + c.w;
            const uint32_t before = (wave > 0 ? c.x : 0u) + (wave > 1 ? c.y : 0u) + (wave > 2 ? c.z : 0u);
            if (kk < kind) base += tot;
            if (kk == kind) base += before;
            if (kk == K_DONE) done_total = tot;
        }
        cur ^= 1;
        order[cur * SLOTS + base + rank_in_wave] = (unsigned short)slot;
        __syncthreads();
        if (done_total == SLOTS) break;
    }

    // ---- write the pixel sums back: lane tid owns slot tid again ----
    {
        uint32_t px, py;
        slot_pixel(tid, px, py);
        if ((ty < P.tiles_y) && (px < P.width) && (py < P.height)) {
            const float4 a = plane[4 * SLOTS + tid];
            float *o = slot_out(tid, px, py);
            o[0] = a.x; o[1] = a.y; o[2] = a.z;
        }
    }
    if (P.counters) {
        unsigned long long s = wave_sum(segments);
        if (lane == 0 && s) atomicAdd(&P.counters[0], s);
    }
}

extern "C" size_t bt_sorted_state_bytes(int output) {
    const size_t planes = output == 0 ? 6 : 7;
    return planes * SLOTS * sizeof(float4) + 2 * SLOTS * sizeof(unsigned short) + K_COUNT * 4 * sizeof(uint32_t);
}

extern "C" hipError_t bt_launch_render_sorted(const BtLaunch *P, int output, unsigned grid, size_t scene_lds_bytes,
                                              hipStream_t stream) {
    dim3 g(grid), b(256);
    const size_t lds = bt_sorted_state_bytes(output) + scene_lds_bytes;
    switch (output) {
    case 0: hipLaunchKernelGGL(bt_render_sorted_kernel<0>, g, b, lds, stream, *P); break;
    case 1: hipLaunchKernelGGL(bt_render_sorted_kernel<1>, g, b, lds, stream, *P); break;
    case 2: hipLaunchKernelGGL(bt_render_sorted_kernel<2>, g, b, lds, stream, *P); break;
    default: hipLaunchKernelGGL(bt_render_sorted_kernel<3>, g, b, lds, stream, *P); break;
    }
    return hipGetLastError();
}
