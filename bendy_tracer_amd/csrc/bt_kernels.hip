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
// Mapping to the hardware:
//   * a workgroup owns a block of 256/S pixels (S = 1..32) and all of their samples in this launch.  The block's
//     (pixel, sample) pairs are a work queue in LDS: a lane whose path has ended takes the next pair, every sample's
//     value is parked in HBM and the last wave of the workgroup adds the parked values to the frame in sample order --
//     the reference's per-pixel summation order, no atomics on the frame;
//   * a launch of only a few work items per lane of the GPU is PACKED instead: one workgroup per workgroup slot, each
//     owning every k-th pixel block behind one queue, so that the GPU drains its longest paths once, not once per
//     generation of workgroups (own builds, template flag PACKED);
//   * in the sphere-only builds a wave votes every iteration whether it runs the camera event or the scatter / volume
//     events; the lanes of the other kind keep their state for the next iteration (phase voting);
//   * every loop iteration is TRACE (one path segment, all lanes) followed by exactly ONE random event per lane --
//     a Diffuse / Metallic / Glass scatter, a volume step, or, for a lane whose path just ended, the camera ray of
//     its next sample.  The event's Philox block, its sin/cos, its basis construction and its final normalize are
//     shared by all event kinds, so those instructions run with every lane active; a wave never idles on its
//     longest path;
//   * the primitive table is read with wave-uniform indices through the constant address space -> scalar (SMEM)
//     loads that broadcast through SGPRs; per-lane lookups (hit primitive, material, light, density) go to tables
//     staged in LDS; the camera block of the launch parameters is read where it is used, not kept in SGPRs;
//   * template flags select a build without rect code / without the volume march for scenes that have neither;
//   * no MFMA: there is no dense contraction on this path.
//
// Round 3 removed what had lost every measurement of rounds 1 and 2 (the logs stay under profiles/): the lane-owns-pixel
// mapping, the streaming queue with its ring of parked units, the regrouping kernel (bt_kernels_sorted.hip) and the A/B
// knobs BT_VOTE3, BT_VOTE_SOFT_K, BT_XCD_ROTATE, BT_PK_PAIRS, BT_WG_THREADS, BT_VOTE_RECTS, BT_NUM_SGPR, BT_WAVES_EXACT.
// Round 3's own experiments are gone again too, each bit-exact and each measured slower or no faster: persistent workgroups
// that claim pixel blocks with the sums in a second kernel (profiles/r04b), a pool of path records in LDS through which paths
// change lanes -- a "march stack" for paths that enter a volume (profiles/r04d) and end-of-block compaction (profiles/r04c,
// r04e, r04f).
#include "bt_device.hpp"

#define BT_SUM_BATCH 8             // parked values a lane of the summing wave has in flight (16: no difference, profiles/r04k)

// Developer build (-DBT_PROFILE): s_memtime stamps around the sections of the render loop, summed per wave into
// counters[2..]; shares of wave cycles are printed by bt_scene_last_stats.  Not part of the product build.
// Developer build (-DBT_LANESTAT, implies the 12 counters of BT_PROFILE): per wave-iteration popcounts of what the lanes
// do, summed into counters[2..]: [2] wave iterations, [3] lanes that trace, [4..8] lanes per event kind (camera, Diffuse,
// Metallic, Glass, volume), [9] lanes waiting for their phase, [10] lanes that have left the loop (queue empty).
#ifdef BT_LANESTAT
#define BT_LS(i, mask) do { ls_acc[i] += (unsigned long long)__popcll(mask); } while (0)   // every lane still in the loop counts; max over lanes at the end
#else
#define BT_LS(i, mask)
#endif
#ifdef BT_PROFILE
#define BT_PROF_DECL unsigned long long prof_t = __builtin_readcyclecounter(), prof_acc[BT_N_COUNTERS - 2] = {}
#define BT_PROF(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); prof_acc[i] += now_ - prof_t; prof_t = now_; } while (0)
#else
#define BT_PROF_DECL
#define BT_PROF(i)
#endif

namespace {

// A parked sample value: 12 bytes (round 1 parked a float4 with an unused w -- a quarter of the block queue's HBM traffic).
struct Parked { float x, y, z; };

enum { EV_GEN = 0, EV_DIFFUSE = 1, EV_METALLIC = 2, EV_GLASS = 3, EV_VOLUME = 4 };

// A path in flight, as it changes lanes in the drain of a packed launch (80 bytes): ray, throughput, radiance; event counter,
// pixel, work item; bounce | volume bounce << 16; in-volume object + 1 | vote wait << 24 | held << 31; the held hit.
struct __attribute__((aligned(16))) PathRec { float f[12]; uint32_t w[8]; };

// lanes of a wave64 mask below this lane
BT_DEV uint32_t lanes_below(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
// ---- where a pixel block lies in the frame ---------------------------------------------------------------------------
// BtLaunch::slices = NS in {1,2,4,8,16,32}: a 16x16 tile is cut into NS blocks of pxb = 256/NS pixels -- whole 8x8 quadrants
// down to 64 pixels, then 8x4, 4x4, 4x2 pixels, numbered row-major inside the tile.  Block b of the launch is block
// b mod NS of the launch's tile b / NS.  The tile (a division by tiles_x: umulhi + one fix-up step) and the block's corner
// inside it depend on the block alone: computed once per block (wave-uniform: scalar), not once per work item.
struct BlockGeom { uint32_t NS, LOG_NS, pxb, LOG_PXB, LBW, WMASK; };
BT_DEV BlockGeom block_geom(const BtLaunch &P) {
    BlockGeom g;
    g.NS = (uint32_t)P.slices;
    g.LOG_NS = (uint32_t)__builtin_ctz(g.NS);
    g.pxb = 256u >> g.LOG_NS;                          // pixels per block
    g.LOG_PXB = 8u - g.LOG_NS;
    g.LBW = g.pxb >= 32 ? 3u : 2u;                     // log2 of a block row: whole quadrants and 8x4 blocks are 8 pixels wide
    g.WMASK = (1u << g.LBW) - 1u;
    return g;
}
struct BlockRef { uint32_t px0, py0, tile_ok, slot; };
BT_DEV BlockRef block_ref(const BtLaunch &P, const BlockGeom &g, uint32_t b) {      // b: block in launch order
    const uint32_t slot = b >> g.LOG_NS, sub = b & (g.NS - 1u);
    const uint32_t tile = P.sharded ? (slot * P.world + P.rank) : slot;
    uint32_t ty = __umulhi(tile, P.tiles_x_magic), tx = tile - ty * P.tiles_x;      // tile / tiles_x, exact after the fix-up
    if (tx >= P.tiles_x) { ty += 1u; tx -= P.tiles_x; }
    // corner of block `sub` inside the tile: 128 pixels = the quadrant row `sub`, 64 = quadrant `sub`, below that blocks
    // of 8x4 / 4x4 / 4x2 pixels numbered row-major
    uint32_t bx0 = 0, by0 = 0;
    if (g.pxb == 128) by0 = sub << 3;
    else if (g.pxb == 64) { bx0 = (sub & 1u) << 3; by0 = (sub >> 1) << 3; }
    else if (g.pxb < 64) {
        const uint32_t lbh = g.LOG_PXB - g.LBW, lnbx = 4u - g.LBW;
        bx0 = (sub & ((1u << lnbx) - 1u)) << g.LBW;
        by0 = (sub >> lnbx) << lbh;
    }
    BlockRef r;
    r.px0 = tx * BT_TILE_DIM + bx0;
    r.py0 = ty * BT_TILE_DIM + by0;
    r.tile_ok = ty < P.tiles_y ? 1u : 0u;
    r.slot = slot;
    return r;
}
// pixel q of a block: quadrants 1 .. 3 of a 128- or 256-pixel block sit to the right of / below quadrant 0
struct PixelRef { uint32_t px, py; bool in_frame; };
BT_DEV PixelRef pixel_of(const BtLaunch &P, const BlockGeom &g, const BlockRef &B, uint32_t q) {
    PixelRef r;
    r.px = B.px0 + (q & g.WMASK) + ((q >> 3) & 8u);
    r.py = B.py0 + ((q & 63u) >> g.LBW) + ((q >> 4) & 8u);
    r.in_frame = B.tile_ok && r.px < P.width && r.py < P.height;
    return r;
}
// the pixel's running sum: row-major frame, or this rank's tile-major shard
BT_DEV float *out_of(const BtLaunch &P, const BlockRef &B, const PixelRef &r) {
    return P.sharded ? P.out + ((size_t)B.slot * (BT_TILE_DIM * BT_TILE_DIM) + (r.py & 15u) * BT_TILE_DIM + (r.px & 15u)) * 4
                     : P.out + ((size_t)r.py * P.width + r.px) * 4;
}

// `*r += pixel.r` (buffer.rs:159-164) for every parked sample of block b's pixels, in sample order -- the additions a
// lane that owned the pixel would perform in a register, in the same order, hence the same bits.  Executed by ONE wave
// (`lane` = 0 .. 63); src = the block's parked values, src[k * pxb + pixel].
BT_DEV void sum_block(const BtLaunch &P, const BlockGeom &g, uint32_t b, uint32_t T, const Parked *src, uint32_t lane) {
    const BlockRef B = block_ref(P, g, b);
    const uint32_t pxb = g.pxb;
    if (pxb >= 64) {
        for (uint32_t q = lane; q < pxb; q += 64) {
            const PixelRef r = pixel_of(P, g, B, q);
            if (!r.in_frame) continue;
            float *o = out_of(P, B, r);
            const Parked *s = src + q;
            V3 sum = mk(o[0], o[1], o[2]);
            uint32_t kk = 0;
            for (; kk + BT_SUM_BATCH <= T; kk += BT_SUM_BATCH) {   // BT_SUM_BATCH loads in flight, additions strictly in order
                Parked v[BT_SUM_BATCH];
#pragma unroll
                for (int j = 0; j < BT_SUM_BATCH; ++j) v[j] = s[(size_t)(kk + j) * pxb];
#pragma unroll
                for (int j = 0; j < BT_SUM_BATCH; ++j) sum = sum + mk(v[j].x, v[j].y, v[j].z);
            }
            for (; kk < T; ++kk) {
                const Parked v = s[(size_t)kk * pxb];
                sum = sum + mk(v.x, v.y, v.z);
            }
            o[0] = sum.x;
            o[1] = sum.y;
            o[2] = sum.z;
        }
    } else {
        // 32, 16 or 8 pixels (deep launches, T in the hundreds): J = 64 / pxb lanes per pixel fetch interleaved
        // samples (8 J in flight per pixel), lane (q, 0) adds them in sample order out of the others' registers
        const uint32_t J = 64u >> g.LOG_PXB, q = lane & (pxb - 1u), jl = lane >> g.LOG_PXB;
        const PixelRef r = pixel_of(P, g, B, q);
        const bool owner = jl == 0 && r.in_frame;
        float *o = out_of(P, B, r);
        const Parked *s = src + q;
        V3 sum = mk(0.0f, 0.0f, 0.0f);
        if (owner) sum = mk(o[0], o[1], o[2]);
        for (uint32_t kk = 0; kk < T; kk += 8 * J) {
            Parked v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const uint32_t k2 = kk + (uint32_t)u * J + jl;
                v[u] = k2 < T ? s[(size_t)k2 * pxb] : Parked{0.0f, 0.0f, 0.0f};
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                for (uint32_t jj = 0; jj < J; ++jj) {
                    const int from = (int)(q + jj * pxb);
                    const V3 val = mk(__shfl(v[u].x, from, 64), __shfl(v[u].y, from, 64), __shfl(v[u].z, from, 64));
                    if (kk + (uint32_t)u * J + jj < T) sum = sum + val;
                }
        }
        if (owner) {
            o[0] = sum.x;
            o[1] = sum.y;
            o[2] = sum.z;
        }
    }
}


// A workgroup of a packed launch (BtLaunch::wg_blocks > 1) owns the blocks first, first + stride, ...: `n` of them, parked
// back to back.  All its threads sum, one pixel each at a time, BT_SUM_BATCH parked values in flight, additions strictly in
// sample order.
BT_DEV void sum_blocks(const BtLaunch &P, const BlockGeom &g, uint32_t first, uint32_t stride, uint32_t n, uint32_t T,
                       const Parked *src, uint32_t thread, uint32_t n_threads) {
    const uint32_t LOG_ROWS = P.log_rows;              // a block's samples are padded to 2^log_rows rows of pxb parked values
    for (uint32_t p = thread; p < (n << g.LOG_PXB); p += n_threads) {
        const uint32_t j = p >> g.LOG_PXB, q = p & (g.pxb - 1u);
        const BlockRef B = block_ref(P, g, first + j * stride);
        const PixelRef r = pixel_of(P, g, B, q);
        if (!r.in_frame) continue;
        float *o = out_of(P, B, r);
        const Parked *s = src + ((size_t)j << (LOG_ROWS + g.LOG_PXB)) + q;
        V3 sum = mk(o[0], o[1], o[2]);
        uint32_t kk = 0;
        for (; kk + BT_SUM_BATCH <= T; kk += BT_SUM_BATCH) {
            Parked v[BT_SUM_BATCH];
#pragma unroll
            for (int u = 0; u < BT_SUM_BATCH; ++u) v[u] = s[(size_t)(kk + u) << g.LOG_PXB];
#pragma unroll
            for (int u = 0; u < BT_SUM_BATCH; ++u) sum = sum + mk(v[u].x, v[u].y, v[u].z);
        }
        for (; kk < T; ++kk) {
            const Parked v = s[(size_t)kk << g.LOG_PXB];
            sum = sum + mk(v.x, v.y, v.z);
        }
        o[0] = sum.x;
        o[1] = sum.y;
        o[2] = sum.z;
    }
}

} // namespace

// --------------------------------------------------------------------------------------------
// The render kernel.  OUTPUT: 0 Full, 1 Albedo, 2 Normal, 3 Depth (tracer/mod.rs:108-115).
// Block = 256 threads; 7 waves per SIMD caps the allocation at 72 VGPRs (round 2, without the SLP vectorizer;
// profiles/r03c/ab_waves_noslp.log, ab_waves_per_class.log).
#ifndef BT_WAVES_PER_SIMD
#define BT_WAVES_PER_SIMD 7
#endif
#ifndef BT_WAVES_PER_SIMD_VOLS
#define BT_WAVES_PER_SIMD_VOLS BT_WAVES_PER_SIMD     // sphere scenes with volumes (own knob for A/B runs)
#endif
#ifndef BT_WAVES_PER_SIMD_RECTS
#define BT_WAVES_PER_SIMD_RECTS 7
#endif
#ifndef BT_WAVES_PER_SIMD_LENS
#define BT_WAVES_PER_SIMD_LENS 6       // lens builds: 80 VGPRs + ~100 B of scratch per lane still beat 4 waves without
#endif                                 // scratch (665 -> 719 Msamples/s, profiles/r01g/ab_lens_waves.log)
// LENS switches the (non-reference, default-off) gravitational-lens extension of bt_device.hpp in.
#ifndef BT_SKIP_DIR
#define BT_SKIP_DIR 1          // a wave of pass-through march steps skips the direction sampling
#endif
#ifndef BT_LENS_BATCH
#define BT_LENS_BATCH 8            // RK4 steps a lane marches per loop iteration before it yields
#endif
// RECTS = false: sphere-only scenes (scene.json, volume.json, cloud.json) run a build without any rect / cuboid code.
// VOLS = false: no sphere carries a volume (scene.json, the Cornell boxes): the march and Volume::shade drop out.
// PACKED = true: the builds for packed launches (BtLaunch::wg_blocks > 1; not with the lens extension), so that the
// other builds carry none of their code -- as a run-time switch it cost C3 4 % (profiles/r04u).
template <int OUTPUT, bool LENS, bool RECTS, bool VOLS, bool PACKED>
__global__ __launch_bounds__(256, LENS ? BT_WAVES_PER_SIMD_LENS : (RECTS ? BT_WAVES_PER_SIMD_RECTS : (VOLS ? BT_WAVES_PER_SIMD_VOLS : BT_WAVES_PER_SIMD))) void bt_render_kernel(BtLaunch P) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ uint32_t s_waves_done;      // block queue: waves of this workgroup that have parked all their samples
    __shared__ uint32_t s_next_item;       // the workgroup's work queue (next unclaimed (pixel, sample) pair)
    __shared__ uint32_t s_segments;        // path segments traced by this workgroup
    __shared__ uint32_t s_pool_paths[2], s_pool_waves[2];   // packed builds, drain rounds: live paths / waves that hold any (two sets, alternating)
    if (threadIdx.x == 0) {
        s_waves_done = 0;
        s_next_item = 0;
        s_segments = 0;
        if (PACKED && RECTS && !VOLS && OUTPUT == 0) s_pool_paths[0] = s_pool_paths[1] = s_pool_waves[0] = s_pool_waves[1] = 0;
    }

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
            l.kind_object = (R.kind & BT_PRIM_SHAPE_MASK) | (R.object << 8);
            l.material = R.material;
            l.volume = R.volume;
            l.rcp_radius = ((R.kind & BT_PRIM_SHAPE_MASK) == BT_PRIM_SPHERE && R.radius >= 0x1p-20f && R.radius <= 0x1p20f) ? refined_rcp(R.radius) : 0.0f;
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
        __syncthreads();
    }
    BtVolBox *const vbox = (BtVolBox *)(smem + P.table_lds_bytes);      // VOLS builds: bt_types.h BtVolBox, one per primitive
    if (VOLS && P.vbox_lds_bytes) {
        for (int i = threadIdx.x; i < P.n_prims; i += blockDim.x) {
            const BtPrim &R = P.prims[i];
            BtVolBox bx;
            const V3 c = mk(R.c), hsz = mk(R.radius, R.radius, R.radius);
            const V3 bmin = c - hsz, bmax = c + hsz, size = bmax - bmin;            // sphere.rs:35-38, volume.rs:29-31
            bx.bmin.x = bmin.x; bx.bmin.y = bmin.y; bx.bmin.z = bmin.z;
            bx.size.x = size.x; bx.size.y = size.y; bx.size.z = size.z;
            bx.rcp.x = refined_rcp(size.x); bx.rcp.y = refined_rcp(size.y); bx.rcp.z = refined_rcp(size.z);
            const bool ok = size.x >= 0x1p-20f && size.x <= 0x1p20f && size.y >= 0x1p-20f && size.y <= 0x1p20f &&
                            size.z >= 0x1p-20f && size.z <= 0x1p20f;
            bx.ok = ok ? 1.0f : 0.0f;
            bx.pad0 = bx.pad1 = 0.0f;
            vbox[i] = bx;
        }
        __syncthreads();
    }

    // ---- tile / pixel mapping ----
    // A workgroup owns one pixel block of pxb = 256 / slices pixels (block_ref() above): its pxb * T (pixel, sample) pairs are
    // work items i = k * pxb + pixel, handed out through an LDS counter (one atomic per wave and iteration, see the loop)
    // -- a lane whose path has ended takes the next item, so all 256 lanes stay busy until the block's samples run out, and
    // 64 consecutive items are the same sample of neighbouring pixels (coherent camera rays).  Every sample's value is
    // parked at scratch[block * pxb * T + i]; the last wave to finish adds them to the frame in sample order (end of the
    // kernel).
    const BlockGeom G = block_geom(P);
    const uint32_t pxb = G.pxb, LOG_PXB = G.LOG_PXB;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t nn = (uint32_t)(P.subsample_n * P.subsample_n);
    const uint32_t T = (uint32_t)P.samples * nn;       // samples per pixel in this launch
    const uint32_t sample0 = P.sample_base * nn;
    // block queue: this workgroup's one block (launch order = blockIdx.x) -- or, in a packed launch (wg_blocks > 1: gridDim.x
    // workgroups for the launch's blocks), the blocks blockIdx.x, blockIdx.x + gridDim.x, ... behind ONE queue: item
    // i = ((j << log_rows | k) << LOG_PXB) + pixel for sample k of the workgroup's j-th block (rows k >= T are holes: T is
    // padded to a power of two so that neither j nor k costs a division)
    constexpr bool packed = PACKED;
    const uint32_t my_blocks = packed ? P.wg_blocks - (blockIdx.x < P.wg_blocks_rem ? 0u : 1u) : 1u;
    const uint32_t n_items = packed ? (my_blocks << (P.log_rows + LOG_PXB)) : pxb * T;      // work items of this workgroup
    const BlockRef B_own = block_ref(P, G, blockIdx.x);
    // where this workgroup parks: every workgroup of a packed launch has room for wg_blocks blocks
    auto park = [&]() -> Parked * {
        return (Parked *)P.scratch + (size_t)blockIdx.x * (packed ? (size_t)P.wg_blocks << (P.log_rows + LOG_PXB) : (size_t)n_items);
    };

    // the lane's current work item: pixel_index keys the Philox counter; park_i = the item's number i = k * pxb + pixel in
    // the block, where its value is parked (+ the workgroup's base; the sample number k comes out of it, one register less
    // than keeping both)
    uint32_t px = 0, py = 0, pixel_index = 0, park_i = 0;

    // per-lane path state
    V3 ro = mk(0, 0, 0), rd = mk(0, 0, -1), beta = mk(1, 1, 1), L = mk(0, 0, 0);
    V3 first = mk(0, 0, 0);            // first non-pass-through albedo / normal (AOV outputs)
    float first_depth = __builtin_inff();
    bool have_first = false;
    int bounce = 0, vbounce = 0, last_object = -1;
    uint32_t event = 0;
    bool pending = true;               // the lane has no ray yet: its next event is the camera ray
    // phase voting (BtLaunch::phase_vote): a lane whose scatter event lost the vote keeps its hit for the next iteration
    constexpr bool VOTE = !RECTS && !LENS;    // pays where the events, not TRACE, are most of an iteration
    bool held = false;
    float held_t = 0.0f;
    int held_info = 0, waited = 0;     // held_info = prim | inside << 29 | p_neg << 30
    // path segments of this lane (bt_stats::segments): counted in 32 bits per lane and added up per workgroup in LDS --
    // one device atomic per workgroup, issued by the wave that sums the block.  (A lane sees at most
    // scratch cap / 12 B / 256 items per launch and bt_api.cpp keeps items x longest path below 2^32 per workgroup.)
    uint32_t segments = 0;
    unsigned long long lens_steps = 0;
    LensState lens;                    // lens extension: the bent segment in progress (LENS builds only)
    bool bent = false;
    lens_begin(P, lens);

    // mod.rs:304-315 -> Chunk::write_* -> Buffer::write_* (buffer.rs:159-178): one sample is done
    auto finish_sample = [&]() {
        V3 value;
        if (OUTPUT == 0) {
            value = L;
        } else if (OUTPUT == 3) {
            float depth = (first_depth - P.clip_min) / (P.clip_max - P.clip_min);
            depth = fminf(fmaxf(depth, 0.0f), 1.0f);
            value = mk(depth, depth, depth);
        } else {
            value = first;
        }
        park()[park_i] = Parked{value.x, value.y, value.z};
    };

    // Packed builds, the drain (BtLaunch::pool_records > 0): once the queue is empty the workgroup's waves meet at the end of every
    // iteration, put the paths still in flight into LDS records and take them back densely packed -- waves 1 .. 3 run out of
    // paths and stop issuing instructions for a handful of live lanes each.  Scheduling only: a path's state moves between lanes,
    // its operations and their order do not change.  Every wave takes part in every round (barriers pair up by count) until
    // a round finds no path left.
    PathRec *const pool = (PathRec *)(smem + P.pool_lds_offset);
    // Compiled into the rect build only: measured (profiles/r04y), it takes 4 - 14 % off packed Cornell-box launches and nothing off
    // sphere and volume launches (short drains; marches), whose builds its code made ~5 % slower.
    constexpr bool CAN_COMPACT = PACKED && RECTS && !VOLS && OUTPUT == 0;    // (a PathRec carries no first-hit AOV state)
    const bool compacting = CAN_COMPACT && P.pool_records > 0;
    bool dry_lane = false;             // this lane found the queue empty
    uint32_t drain_it = 0;             // iterations since the wave saw the queue empty (wave-uniform)

    BT_PROF_DECL;
#ifdef BT_LANESTAT
    unsigned long long ls_acc[9] = {};
    const unsigned long long ls_all = __ballot(true);
#endif
    for (;;) {
        do {                                              // (`continue` below = on to the latch at the end of the iteration)
        if (compacting && dry_lane && pending) continue;  // the queue is empty and this lane has no path: nothing to do
        BT_LS(0, 1ull);
        BT_LS(8, ls_all & ~__ballot(true));
        BT_PROF(0);                                       // loop overhead / previous iteration's tail
        int ev = EV_GEN;
        // manifold of this iteration's hit (shading events only)
        V3 pos = ro, normal = mk(0, 0, 0);
        float hit_depth = 0.0f;                         // Manifold.t of the hit, for the Depth output
        bool front = false, inside = false, vol_back = false;
        int pobject = -1, mat_index = 0, vol_index = 0;
        V3 prim_c = mk(0, 0, 0);
        float prim_radius = 0.0f;
        int hit_prim = 0;

        BT_LS(1, __ballot(!pending && !(VOTE && held)));
        if (!pending) {
            // ---- TRACE: try_hit (mod.rs:389-402) / try_hit_volume (mod.rs:404-427) ----
            const bool marching = VOLS && last_object >= 0;
            if (!marching) vbounce = 0;                               // sample() -> sample_volume(.., 0), mod.rs:335
            const float tmin = marching ? 0.0f : P.clip_min;
            const float tmax = marching ? P.volume_step : P.clip_max;
            HitRec h;
            bool ended = false, captured = false;
            float travelled = 0.0f;
            if (LENS && !marching) {
                // bent segment, marched BT_LENS_BATCH RK4 steps per iteration: (ro, rd) is the photon; at the
                // end they are the chord that hits, or the ray that reaches the root
                if (!bent) {
                    lens_begin(P, lens);
                    bent = true;
                    segments += 1;
                }
                const int r = lens_advance<RECTS>(P, ro, rd, lens, h, BT_LENS_BATCH, lens_steps);
                if (r == 2) continue;                     // still on its way: no event for this lane yet
                bent = false;
                captured = r < 0;
                travelled = lens.travelled;
            } else if (VOTE && held) {                // the hit found one iteration ago (phase voting; it travels in the path's record)
                h.t = held_t;
                h.prim = held_info & 0x1fffffff;
                h.inside = (held_info >> 29) & 1;
                h.p_neg = (held_info >> 30) & 1;
            } else {
                segments += 1;
                h = intersect<RECTS, VOLS, RECTS && !VOLS && !LENS>(P, ro, rd, tmin, tmax, last_object);
            }
            if (captured) {
                ended = true;                         // swallowed by the horizon: the path returns black
            } else if (h.prim < 0) {
                // sample_root (mod.rs:429-452)
                L = L + beta * mk(P.root_color);
                if (OUTPUT != 0 && !have_first) {
                    have_first = true;
                    if (OUTPUT == 1) first = mk(P.root_albedo);
                    if (OUTPUT == 2) first = P.root_has_albedo ? -rd : mk(0, 0, 0);
                    if (OUTPUT == 3) first_depth = P.root_has_albedo ? P.clip_max : __builtin_inff();
                }
                ended = true;
            } else {
                const BtPrimLite &pl = S.lite[h.prim];
                const int pshape = pl.kind_object & 0xff;
                pobject = pl.kind_object >> 8;
                prim_c = mk(pl.c);
                prim_radius = pl.radius;
                hit_prim = h.prim;
                hit_depth = LENS ? h.t + travelled : h.t;
                pos = ro + rd * h.t;
                bool vol_face = false;
                if (VOLS && h.inside) {               // generate_volume_manifold (sphere.rs:63-83)
                    inside = true;
                    vol_face = true;
                } else if (!RECTS || pshape == BT_PRIM_SPHERE) { // generate_surface_manifold (sphere.rs:85-119)
                    // normal = (position - centre) / radius (sphere.rs:95-99): div_refined() with the sphere's refined
                    // reciprocal -- the IEEE quotient's bits while every operand is 0 or within [2^-60, 2^60] (it is ~radius
                    // here); a wave with a lane outside that range divides exactly
                    V3 nrm = pos - prim_c;
                    const float nax = fabsf(nrm.x), nay = fabsf(nrm.y), naz = fabsf(nrm.z), rr = pl.rcp_radius;
                    const bool in_range = (rr != 0.0f) & (nax == 0.0f || (nax >= 0x1p-60f && nax <= 0x1p60f)) &
                                          (nay == 0.0f || (nay >= 0x1p-60f && nay <= 0x1p60f)) & (naz == 0.0f || (naz >= 0x1p-60f && naz <= 0x1p60f));
                    if (__ballot(!in_range) == 0ull)
                        nrm = mk(div_refined(nrm.x, pl.radius, rr), div_refined(nrm.y, pl.radius, rr), div_refined(nrm.z, pl.radius, rr));
                    else
                        nrm = mk(nrm.x / pl.radius, nrm.y / pl.radius, nrm.z / pl.radius);
                    front = dot(rd, nrm) < 0.0f;
                    normal = front ? nrm : -nrm;
                    vol_face = VOLS && pl.volume >= 0;
                    vol_back = vol_face && !front;
                } else {                              // rect.rs:138-142
                    front = h.p_neg;
                    normal = front ? prim_c : -prim_c;
                }
                if (vol_face) {
                    vol_index = pl.volume;
                    ev = EV_VOLUME;                   // sample_volume (mod.rs:488-523)
                } else {
                    // sample_surface (mod.rs:454-486): emitted, then Material::shade
                    mat_index = pl.material;
                    const BtMaterial &M = S.materials[mat_index];
                    if (!(VOTE && held)) L = L + beta * mk(M.emitted);
                    if (M.kind == BT_MAT_DIFFUSE) ev = EV_DIFFUSE;
                    else if (M.kind == BT_MAT_METALLIC) ev = EV_METALLIC;
                    else if (M.kind == BT_MAT_GLASS) ev = EV_GLASS;
                    else {
                        // Flat / Emissive: no scatter -> ColorData::from_emitted (mod.rs:483-485)
                        if (OUTPUT != 0 && !have_first) {
                            have_first = true;
                            if (OUTPUT == 1) first = mk(M.emitted);
                        }
                        ended = true;
                    }
                }
            }
            if (ended) finish_sample();
            if (VOTE && P.phase_vote && ev != EV_GEN) {       // in case this lane's event loses the vote below
                held_t = h.t;
                held_info = h.prim | ((int)h.inside << 29) | ((int)h.p_neg << 30);
            }
        }
        pending = false;
        if (compacting && dry_lane && ev == EV_GEN) {     // the queue is empty: a lane whose path has just ended is done (and has no vote)
            pending = true;
            continue;
        }

        if (VOTE && P.phase_vote) {
            // ---- which events run this iteration?  The kind more lanes want (camera | scatter / volume step); nobody waits
            // more than max_wait iterations.  Everything here is wave-uniform mask arithmetic on the scalar unit; the lane's
            // verdict is its bit of `served_m`.
            const bool want_gen = ev == EV_GEN;
            const unsigned long long m_gen = __ballot(want_gen), m_sc = __ballot(!want_gen);
            const uint32_t n_gen = popc64(m_gen), n_sc = popc64(m_sc);
            // a lane of the losing side that has waited long enough is served in THIS iteration together with the winners
            // (its whole kind runs, as without the vote) -- the majority does not lose an iteration to it
            const unsigned long long starving = __ballot(waited >= P.phase_vote);
            const bool run_gen = n_gen >= n_sc || (starving & m_gen) != 0;
            const bool run_sc = n_sc > n_gen || (starving & m_sc) != 0;
            const unsigned long long served_m = (run_gen ? m_gen : 0ull) | (run_sc ? m_sc : 0ull);
            const bool served = __builtin_amdgcn_inverse_ballot_w64(served_m);
            BT_LS(7, __ballot(!served));
            if (!served) {
                waited += 1;
                pending = want_gen;                   // no ray yet | the hit stays in held_t / held_info
                held = !want_gen;
                continue;
            }
            waited = 0;
            held = false;
        }

        // ---- a lane whose path has ended (or that has none yet) moves on to its next sample ----
        {
            const unsigned long long need = __ballot(ev == EV_GEN);
            if (need) {                                                   // one LDS atomic for the whole wave
                const int leader = __ffsll((long long)need) - 1;
                uint32_t base = 0;
                if ((int)lane == leader) base = atomicAdd(&s_next_item, popc64(need));
                base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
                if (ev == EV_GEN) {
                    const uint32_t i = base + lanes_below(need);
                    if (i >= n_items) {                                   // the block's samples are all taken
                        if (!compacting) goto queue_empty;                // this lane is done
                        dry_lane = true;                                  // the wave learns of it at the end of the iteration
                        pending = true;
                        continue;
                    }
                    park_i = i;                                           // (+ the workgroup's base, see finish_sample)
                    BlockRef B_i = B_own;
                    bool hole = false;
                    if (packed) {                                         // which of the workgroup's blocks, which row of it
                        const uint32_t row = i >> LOG_PXB;
                        B_i = block_ref(P, G, blockIdx.x + (row >> P.log_rows) * gridDim.x);
                        hole = (row & P.row_mask) >= T;
                    }
                    const PixelRef r = pixel_of(P, G, B_i, i & (pxb - 1u));
                    px = r.px;
                    py = r.py;
                    if (!r.in_frame || hole) {
                        pending = true;                                   // pixel outside the frame (edge tile): skip it
                        continue;
                    }
                    pixel_index = py * P.width + px;
                }
            }
        }
        BT_PROF(1);                                       // TRACE + hit classification

        BT_LS(2, __ballot(ev == EV_GEN)); BT_LS(3, __ballot(ev == EV_DIFFUSE)); BT_LS(4, __ballot(ev == EV_METALLIC));
        BT_LS(5, __ballot(ev == EV_GLASS)); BT_LS(6, __ballot(ev == EV_VOLUME));
        // ---- the lane's one random event of this iteration (numerics contract N6) ----
        // block queue: the item's sample number comes out of its item number (one register less than keeping both)
        const uint32_t k_now = packed ? (park_i >> LOG_PXB) & P.row_mask : park_i >> LOG_PXB;
        const uint32_t sample_index = sample0 + k_now;
        const U4 u = philox(pixel_index, sample_index, ev == EV_GEN ? 0u : event, 0u, P.seed_lo, P.seed_hi);
        // slots of the two angular draws: Metallic [0],[1]; Glass [1],[2]; everything else [2],[3]
        const uint32_t w1 = ev == EV_METALLIC ? u.x : (ev == EV_GLASS ? u.y : u.z);
        const uint32_t w2 = ev == EV_METALLIC ? u.y : (ev == EV_GLASS ? u.z : u.w);
        const float r1 = uniform_sample(w1, 0.0f, P.tau_scale), r2 = uniform_sample(w2, 0.0f, P.one_scale);
        // Volume::shade's scatter decision (volume.rs:26-35) comes first: a march step that passes through needs no
        // sampled direction, and a wave whose lanes all pass through skips the angular draws below altogether
        bool vol_scatter = false;
        if (VOLS && ev == EV_VOLUME) {
            const float density = P.vbox_lds_bytes ? march_density_box(P, S, vol_index, vbox[hit_prim], pos)
                                                   : march_density(P, S, vol_index, prim_c, prim_radius, pos);
            vol_scatter = density >= 1.0f || bernoulli(u.x, density);
        }
        const bool wave_needs_dir = !VOLS || !BT_SKIP_DIR || __ballot(ev != EV_VOLUME || vol_scatter) != 0ull;
        float sn = 0.0f, cs = 0.0f;
        if (wave_needs_dir) sincos_bt(r1, sn, cs);
        BT_PROF(2);                                       // Philox + shared sin/cos

        V3 new_o = pos, dir = rd;
        bool late_end = false;

        if (ev == EV_GEN) {
            // ---- camera ray (mod.rs:271-302, ray.rs:103-113,126-137) ----
            float u_sub = 0.0f, v_sub = 0.0f;
            if (P.subsample_n > 1) {
                const uint32_t n = (uint32_t)P.subsample_n;
                const uint32_t subpx = k_now % (n * n);
                const float width_sub = 1.0f / (float)n;
                u_sub = (float)(subpx % n) * width_sub;
                v_sub = (float)(subpx / n) * width_sub;
            }
            // The camera block of the launch parameters (~30 dwords) is read from the kernarg segment HERE, through
            // a pointer the compiler cannot see through: otherwise it hoists the loads into the prologue, where
            // they live in SGPRs across the whole loop and push other values out into spills.
            typedef const __attribute__((address_space(4))) BtLaunch BtLaunchK;
            BtLaunchK *C = (BtLaunchK *)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(C));
            const V3 mcx = mk(C->cam_cx.x, C->cam_cx.y, C->cam_cx.z), mcy = mk(C->cam_cy.x, C->cam_cy.y, C->cam_cy.z),
                     mcz = mk(C->cam_cz.x, C->cam_cz.y, C->cam_cz.z);
            const float v0 = (float)py * C->pixel_height - 1.0f;
            const float u0 = (float)px * C->pixel_width - 1.0f;
            const float u_offset = u_sub * C->pixel_width + uniform_sample(u.x, C->jitter_u_lo, C->jitter_u_scale);
            const float v_offset = v_sub * C->pixel_height + uniform_sample(u.y, C->jitter_v_lo, C->jitter_v_scale);
            const float uu = u0 + u_offset, vv = v0 + v_offset;
            const float yrot = C->xfov * 0.5f * -uu;
            const float xrot = C->yfov * 0.5f * -vv;
            float sy, cy, sx, cx;
            sincos_small_bt(yrot, sy, cy);              // |angle| <= fov / 2: k = 0 for every frustum below 90 degrees
            sincos_small_bt(xrot, sx, cx);
            const V3 d_cam = mk(-(cx * sy), sx, -(cx * cy));
            // Affine3A * Ray: origin = translation + 0; direction = normalize(normalize_or_zero(M*d)),
            // the outer normalize being the shared one below
            new_o = mk(C->cam_t.x, C->cam_t.y, C->cam_t.z) + mk(0.0f, 0.0f, 0.0f);
            dir = normalize_or_zero(xf_vector(mcx, mcy, mcz, d_cam));
            if (C->has_focus) {                       // mod.rs:286-299; disk angle = r1, radius = r2
                const V3 d1 = normalize(dir);
                const V3 defocus = (mk(C->disk_x.x, C->disk_x.y, C->disk_x.z) * cs + mk(C->disk_y.x, C->disk_y.y, C->disk_y.z) * sn) * r2;
                const V3 defocus_offset = xf_vector(mcx, mcy, mcz, defocus * C->aperture);
                const float frac_f_z = C->focus / fabsf(d_cam.z);
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
            BT_PROF(3);                                   // camera ray
        } else {
            event += 1;
            // ---- direction sample in the local frame (math/distr.rs) ----
            const BtMaterial &M = S.materials[mat_index];
            int light_index = 0;
            bool to_light = false;
            if (ev == EV_DIFFUSE) {
                light_index = (int)__umulhi(u.x, (uint32_t)P.n_lights);   // material.rs:106-119
                to_light = bernoulli(u.y, 0.5f);                          // Pdf::Mix (:269-275)
            }
            const bool is_cosine = ev == EV_DIFFUSE && !to_light;
            const bool in_frame_of_normal = is_cosine || ev == EV_METALLIC || ev == EV_GLASS;
            // UnitSphere (distr.rs:10-21), UnitHemisphere (:48-59, z = 1 - r2), Cosine (:86-97)
            V3 v = mk(0.0f, 0.0f, 0.0f);
            if (wave_needs_dir) {
                const float sq = sqrt_bt(is_cosine ? r2 : r2 * (1.0f - r2));
                const float lx_ = (is_cosine ? cs : cs * 2.0f) * sq;
                const float ly_ = (is_cosine ? sn : sn * 2.0f) * sq;
                float lz_ = in_frame_of_normal ? 1.0f - r2 : 1.0f - 2.0f * r2;
                if (is_cosine) lz_ = sqrt_bt(1.0f - r2);
                v = mk(lx_, ly_, lz_);
                if (in_frame_of_normal) {
                    V3 z_axis = normalize(normal), x_axis, y_axis;
                    orthonormal_pair(z_axis, x_axis, y_axis);
                    v = (x_axis * lx_ + y_axis * ly_) + z_axis * lz_;
                }
            }

            if (ev == EV_DIFFUSE) {
                if (to_light) {                                           // Pdf::Light (:262-268)
                    const BtLight &Lt = S.lights[light_index];
                    V3 point;
                    if (Lt.kind == BT_LIGHT_SPHERE) {                     // sphere.rs:40-42
                        point = mk(Lt.centre) + v * Lt.radius;
                    } else if (RECTS && Lt.kind == BT_LIGHT_RECT) {
                        point = face_random_point(S.faces[Lt.face_first], u.z, u.w);
                    } else if (RECTS && Lt.kind == BT_LIGHT_CUBOID) {     // cuboid.rs:47-54
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
            } else if (ev == EV_METALLIC) {                               // :231-239
                dir = reflect(rd, normal) + v * M.roughness;
            } else if (ev == EV_GLASS) {                                  // :240-261
                const float ior = front ? M.inv_ior : M.ior;
                const float cos_theta = fminf(dot(-rd, normal), 1.0f);
                const float sin_theta = sqrt_bt(1.0f - cos_theta * cos_theta);
                const float fr = fresnel(rd, normal, ior);
                V3 base;
                if (ior * sin_theta > 1.0f || bernoulli(u.x, fr))
                    base = reflect(rd, normal);
                else
                    base = refract(rd, normal, ior);
                dir = base + v * M.roughness;
            } else if (VOLS) {
                // ---- Volume::shade (volume.rs:26-60) ----
                if (vol_scatter) {
                    if (inside) new_o = pos - (rd * P.volume_step) * u24(u.y);
                    dir = v;
                    beta = beta * mk(0.8f, 0.8f, 0.8f);
                    if (OUTPUT != 0 && !have_first) {
                        have_first = true;
                        if (OUTPUT == 1) first = mk(0.8f, 0.8f, 0.8f);
                        if (OUTPUT == 2) first = normal;
                        if (OUTPUT == 3) first_depth = hit_depth;
                    }
                }                                                         // else pass through: Ray::new(pos, rd)
                if (vol_back) {                                           // mod.rs:504-505
                    bounce += 1;
                    last_object = -1;
                } else {                                                  // mod.rs:507-513
                    last_object = pobject;
                    vbounce += 1;
                }
            }
        }

        BT_PROF(4);                                       // scatter direction / volume step
        // Ray::new normalizes (ray.rs:96-101); for the camera this is the last normalize of mod.rs:296-301
        const V3 nd = normalize(dir);

        if (ev == EV_DIFFUSE || ev == EV_METALLIC || ev == EV_GLASS) {
            const BtMaterial &M = S.materials[mat_index];
            bool scatter = true;
            float weight = 1.0f;                                          // material.pdf / shade.pdf
            if (ev == EV_DIFFUSE) {
                const BtLight &Lt = S.lights[(int)__umulhi(u.x, (uint32_t)P.n_lights)];
                const float pd = dot(normal, nd) * 0.318309886183790671538f;   // diffuse_pdf (:301-303)
                const float plight = P.n_lights == 1 ? light_pdf_only_light<RECTS>(P, S, pos, nd) : light_pdf<RECTS>(P, Lt, S, pos, nd);
                const float p = lerpf(pd, plight, 0.5f);                  // :294-296
                scatter = !(fabsf(p) <= 1e-5f);                           // Pdf::pdf (:279-286)
                weight = pd / p;                                          // Material::pdf (:204) / shade.pdf
            }
            if (OUTPUT != 0 && !have_first) {
                have_first = true;
                if (scatter) {      // data.albedo ColorData (material.rs:99-104,140-145,169-174)
                    if (OUTPUT == 1) first = mk(M.albedo);
                    if (OUTPUT == 2) first = normal;
                    if (OUTPUT == 3) first_depth = hit_depth;
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

        // sample() / sample_volumetric() return black past the limits (mod.rs:323-325, 352-354):
        // the path ends before its next TRACE
        if (!late_end) late_end = (VOLS && last_object >= 0) ? (vbounce > P.max_volume_bounces) : (bounce > P.max_bounces);
        if (late_end) {
            finish_sample();
            pending = true;
        }
        BT_PROF(5);                                       // normalize, pdf weight (light_pdf), bookkeeping
        } while (0);
        // ---- the latch: every lane of the wave passes here once per iteration, together -- the drain rounds of the packed builds
        // contain barriers, so they must not sit where lanes that skip the body could run ahead of the others (at the top of the
        // loop the compiler split `round; if (no path) continue;` off as an inner loop of its own: a wave's idle lanes then met
        // the barrier alone, again and again)
        if (compacting && __ballot(dry_lane) != 0ull) {                   // wave-uniform: the queue is empty
            dry_lane = true;
            {                                                             // a round per iteration (every 2nd: +0.7 %, every 4th: +2 %; profiles/r04z, r05x)
                const uint32_t set = drain_it & 1u;
                const bool alive = !pending;                              // a path in flight (possibly holding its hit for the vote)
                const unsigned long long m = __ballot(alive);
                uint32_t base = 0;
                if (lane == 0) {
                    base = atomicAdd(&s_pool_paths[set], popc64(m));
                    if (m) atomicAdd(&s_pool_waves[set], 1u);
                }
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                const uint32_t slot = base + lanes_below(m);
                if (alive && slot < P.pool_records) {
                    PathRec r;
                    r.f[0] = ro.x; r.f[1] = ro.y; r.f[2] = ro.z; r.f[3] = rd.x; r.f[4] = rd.y; r.f[5] = rd.z;
                    r.f[6] = beta.x; r.f[7] = beta.y; r.f[8] = beta.z; r.f[9] = L.x; r.f[10] = L.y; r.f[11] = L.z;
                    r.w[0] = event; r.w[1] = pixel_index; r.w[2] = park_i;
                    r.w[3] = (uint32_t)bounce | ((uint32_t)vbounce << 16);
                    r.w[4] = ((uint32_t)(last_object + 1) & 0xffffffu) | ((uint32_t)waited << 24) | (held ? 0x80000000u : 0u);
                    r.w[5] = __float_as_uint(held_t); r.w[6] = (uint32_t)held_info; r.w[7] = 0u;
                    pool[slot] = r;
                }
                __syncthreads();
                const uint32_t total = *(volatile uint32_t *)&s_pool_paths[set], holders = *(volatile uint32_t *)&s_pool_waves[set];
                if (threadIdx.x == 0) { s_pool_paths[set ^ 1u] = 0; s_pool_waves[set ^ 1u] = 0; }   // next round's set: last read a round ago
                if (total == 0u) break;                                   // no path left in the workgroup: every wave leaves here
                if (total <= P.pool_records && holders > (total + 63u) / 64u) {   // the paths fit fewer waves than hold them now
                    pending = threadIdx.x >= total;
                    if (!pending) {
                        const PathRec r = pool[threadIdx.x];
                        ro = mk(r.f[0], r.f[1], r.f[2]); rd = mk(r.f[3], r.f[4], r.f[5]);
                        beta = mk(r.f[6], r.f[7], r.f[8]); L = mk(r.f[9], r.f[10], r.f[11]);
                        event = r.w[0]; pixel_index = r.w[1]; park_i = r.w[2];
                        bounce = (int)(r.w[3] & 0xffffu); vbounce = (int)(r.w[3] >> 16);
                        last_object = (int)(r.w[4] & 0xffffffu) - 1; waited = (int)((r.w[4] >> 24) & 0x7fu); held = (r.w[4] >> 31) != 0u;
                        held_t = __uint_as_float(r.w[5]); held_info = (int)r.w[6];
                    }
                }
                __syncthreads();                                          // records are read before the next round overwrites them
            }
            drain_it += 1u;
        }
    }
queue_empty:;

    // ---- the end of a workgroup --------------------------------------------------------------------------------------
    // Block queue: the last wave of the workgroup to get here performs `*r += pixel.r` (buffer.rs:159-164) for every parked
    // sample of the block's pixels, in sample order (sum_block).  The parked values were written by waves of this
    // workgroup (same CU, same L1 / L2), so workgroup-scope release / acquire is all the ordering that is needed.
    if (P.counters) {
        uint32_t sg = segments;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sg += __shfl_xor(sg, off, 64);
        if (lane == 0 && sg) atomicAdd(&s_segments, sg);              // LDS; ahead of this wave's s_waves_done below
    }
    if (packed) {
        // one generation of workgroups: nobody waits for this workgroup's wave slots, so its waves meet at a barrier and sum together
        __syncthreads();
        if (P.counters && threadIdx.x == 0) {
            const uint32_t total = *(volatile uint32_t *)&s_segments;
            if (total) atomicAdd(&P.counters[0], (unsigned long long)total);
        }
        sum_blocks(P, G, blockIdx.x, gridDim.x, my_blocks, T, park(), threadIdx.x, blockDim.x);
    } else {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        uint32_t arrived = 0;
        if (lane == 0) arrived = atomicAdd(&s_waves_done, 1u);
        arrived = (uint32_t)__builtin_amdgcn_readfirstlane((int)arrived);
        if (arrived == (blockDim.x >> 6) - 1u) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if (P.counters && lane == 0) {
                const uint32_t total = *(volatile uint32_t *)&s_segments;
                if (total) atomicAdd(&P.counters[0], (unsigned long long)total);
            }
            sum_block(P, G, blockIdx.x, T, park(), lane);
        }
    }
    if (P.counters) {
        if (LENS) {
            unsigned long long ls = wave_sum(lens_steps);
            if (lane == 0 && ls) atomicAdd(&P.counters[1], ls);
        }
#ifdef BT_LANESTAT
        for (int i = 0; i < 9; ++i) {
            unsigned long long v = ls_acc[i];
            for (int off = 32; off > 0; off >>= 1) {
                const unsigned long long o = __shfl_xor(v, off, 64);
                v = o > v ? o : v;
            }
            if (lane == 0) atomicAdd(&P.counters[2 + i], v);
        }
#elif defined(BT_PROFILE)
        if (lane == 0)
            for (int i = 0; i < BT_N_COUNTERS - 2; ++i) atomicAdd(&P.counters[2 + i], prof_acc[i]);
#endif
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
// numerics contract N9: x^(1/2.4) = exp2(log2(x) / 2.4), same polynomials as the oracle
BT_DEV float log2_bt(float x) {
    const uint32_t xi = __float_as_uint(x);
    int e = (int)((xi >> 23) & 0xffu) - 127;
    float m = __uint_as_float((xi & 0x7fffffu) | 0x3f800000u);
    if (m > 1.41421356f) { m = m * 0.5f; e += 1; }
    const float t = m - 1.0f;
    const float s = t / (2.0f + t), s2 = s * s;
    float p = __builtin_fmaf(s2, 0.0909090909f, 0.1111111111f);
    p = __builtin_fmaf(p, s2, 0.1428571429f);
    p = __builtin_fmaf(p, s2, 0.2f);
    p = __builtin_fmaf(p, s2, 0.3333333333f);
    const float ln = __builtin_fmaf(p * s2, s, s) * 2.0f;
    return __builtin_fmaf(ln, 1.4426950408889634f, (float)e);
}
BT_DEV float exp2_bt(float y) {
    const float k = __builtin_rintf(y), r = y - k;
    const float z = r * 0.6931471805599453f;
    float p = __builtin_fmaf(z, 1.984126984e-4f, 1.388888889e-3f);
    p = __builtin_fmaf(p, z, 8.333333333e-3f);
    p = __builtin_fmaf(p, z, 4.166666667e-2f);
    p = __builtin_fmaf(p, z, 1.666666667e-1f);
    p = __builtin_fmaf(p, z, 0.5f);
    p = __builtin_fmaf(p, z, 1.0f);
    p = __builtin_fmaf(p, z, 1.0f);
    const int ki = (int)k;
    if (ki < -126) return 0.0f;
    if (ki > 127) return __builtin_inff();
    return p * __uint_as_float((uint32_t)(ki + 127) << 23);
}
BT_DEV float linear_to_srgb(float x) {              // color.rs:14-20
    if (x <= 0.0031308f) return 12.92f * x;
    if (!(x < 3.0e38f)) return x;
    return 1.055f * exp2_bt(log2_bt(x) * (1.0f / 2.4f)) - 0.055f;
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
    // grid = tiles to render; a tile is P->slices workgroups (see the mapping in the kernel)
    const bool packed = P->wg_blocks > 1;          // bt_api.cpp packs launches without the lens only
    if (packed && P->lens_on) return hipErrorInvalidValue;
    dim3 g(packed ? P->n_workgroups : grid * (unsigned)P->slices), b(256);
    // scene classes: bit 0 = some sphere carries a volume (volume.json, cloud.json), bit 1 = rects / cuboids present
    // (the Cornell boxes); scene.json is class 0
    const int cls = (P->any_rects ? 2 : 0) | (P->any_volumes ? 1 : 0);
    // scene tables beyond the default 64 KB of dynamic LDS (hundreds of objects): gfx950 has 160 KB per CU, the limit
    // has to be raised per kernel; one workgroup per CU is then all that fits
#define BT_LAUNCH(O, L, R, V, K)                                                                                 \
    do {                                                                                                         \
        if (lds_bytes > 48 * 1024)                                                                               \
            (void)hipFuncSetAttribute((const void *)bt_render_kernel<O, L, R, V, K>,                             \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);               \
        hipLaunchKernelGGL((bt_render_kernel<O, L, R, V, K>), g, b, lds_bytes, stream, *P);                      \
    } while (0)
#define BT_LAUNCH_OUT(L, R, V, K)                                                                                \
    switch (output) {                                                                                            \
    case 0: BT_LAUNCH(0, L, R, V, K); break;                                                                     \
    case 1: BT_LAUNCH(1, L, R, V, K); break;                                                                     \
    case 2: BT_LAUNCH(2, L, R, V, K); break;                                                                     \
    default: BT_LAUNCH(3, L, R, V, K); break;                                                                    \
    }
#define BT_LAUNCH_CLASS(L, K)                                                                                    \
    if (cls == 3) { BT_LAUNCH_OUT(L, true, true, K) } else if (cls == 2) { BT_LAUNCH_OUT(L, true, false, K) }      \
    else if (cls == 1) { BT_LAUNCH_OUT(L, false, true, K) } else { BT_LAUNCH_OUT(L, false, false, K) }
    if (packed) {
        BT_LAUNCH_CLASS(false, true)
    } else if (P->lens_on) {
        BT_LAUNCH_CLASS(true, false)
    } else {
        BT_LAUNCH_CLASS(false, false)
    }
#undef BT_LAUNCH_CLASS
#undef BT_LAUNCH_OUT
#undef BT_LAUNCH
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
