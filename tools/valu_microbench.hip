// valu_microbench.hip -- issue cost (SIMD cycles per wave64 instruction) of the opcode classes the render kernel
// is made of, measured on the MI355X at 1 / 2 / 4 / 7 waves per SIMD.  Developer tool (not part of the product):
//   hipcc --offload-arch=gfx950 -O2 tools/valu_microbench.hip -o tools/valu_microbench && tools/valu_microbench out.json
// Every kernel runs ITERS iterations of a body of 64 instructions of ONE opcode on 8 independent registers (a single
// wave then shows the dependent-issue cost, several waves the issue-port cost), stamped with s_memtime.  Reported
// per opcode and occupancy W:  median over waves of (cycles of the loop / instructions) / W_here, W_here = waves that
// ran on that wave's SIMD (from HW_ID) -- the cycles of SIMD issue one instruction takes when W waves compete.
// tools/isa_histogram.py weights the static histogram with the W = 7 row.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <tuple>
#include <vector>

#define HIP_OK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_)); return 1; } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));
typedef int i4 __attribute__((ext_vector_type(4)));

struct Stamp { unsigned long long t0, t1; unsigned hw, xcc; };

#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BODY64(X) R8(X) R8(X) R8(X) R8(X) R8(X) R8(X) R8(X) R8(X)

#define KERNEL(NAME, ONE)                                                                                        \
    __global__ __launch_bounds__(256) void k_##NAME(Stamp *out, int iters, float fx, float fy, const i4 *mem) {  \
        float a[8];                                                                                              \
        unsigned long long b[8];                                                                                 \
        unsigned c[8];                                                                                           \
        f2 p[8];                                                                                                 \
        i4 q[8];                                                                                                 \
        unsigned s0 = (unsigned)iters, s1 = (unsigned)iters * 3u;                                                \
        unsigned long long m0 = ~0ull;                                                                           \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                          \
            a[i] = fx + (float)(threadIdx.x + i) * 1e-3f;                                                        \
            c[i] = threadIdx.x * 2654435761u + i;                                                                \
            b[i] = (unsigned long long)c[i] | 1ull;                                                              \
            p[i] = (f2){a[i], a[i] + 0.5f};                                                                      \
            q[i] = (i4){0, 0, 0, 0};                                                                             \
        }                                                                                                        \
        float x = fx, y = fy;                                                                                    \
        unsigned ux = (unsigned)(fx * 1000.0f) | 1u;                                                             \
        f2 px = (f2){fx, fy};                                                                                    \
        __shared__ int lds[256];                                                                                 \
        lds[threadIdx.x] = (int)((threadIdx.x * 4u) & 255u);                                                     \
        unsigned la = (threadIdx.x & 63u) * 4u;                                                                  \
        const char *gaddr = (const char *)mem + 4096 + ((size_t)blockIdx.x * 256 + threadIdx.x) * 16;            \
        (void)gaddr;                                                                                             \
        __syncthreads();                                                                                         \
        asm volatile("s_nop 0" : "+s"(s0), "+s"(s1), "+s"(m0));                                                  \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                              \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                       \
        for (int it = 0; it < iters; ++it) {                                                                     \
            BODY64(ONE)                                                                                          \
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                          \
        }                                                                                                        \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                              \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                       \
        float acc = (float)(s0 + s1) + (float)m0 + (float)la + x + px.x + (float)ux;                             \
        _Pragma("unroll") for (int i = 0; i < 8; ++i)                                                            \
            acc += a[i] + (float)b[i] + p[i].x + p[i].y + (float)c[i] + (float)(q[i].x + q[i].w);               \
        if (acc == 123.456f) out[0].t0 = (unsigned long long)acc + lds[la / 4];                                  \
        if ((threadIdx.x & 63) == 0) {                                                                           \
            unsigned hw, xcc;                                                                                    \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));                                     \
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));                                   \
            Stamp s; s.t0 = t0; s.t1 = t1; s.hw = hw; s.xcc = xcc;                                               \
            out[blockIdx.x * 4 + (threadIdx.x >> 6)] = s;                                                        \
        }                                                                                                        \
    }

// ---- one instruction per macro; i = the independent register it works on ----
#define OP_v_add_f32(i) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
#define OP_v_mul_f32(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
#define OP_v_fma_f32(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
#define OP_v_mul_f32_sgpr(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "s"(s0));
#define OP_v_pk_mul_f32(i) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(p[i]) : "v"(px));
#define OP_v_pk_add_f32(i) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(p[i]) : "v"(px));
#define OP_v_pk_fma_f32(i) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(p[i]) : "v"(px));
#define OP_v_mad_u64_u32(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(b[i]) : "v"(ux), "v"(c[i]) : "vcc");
#define OP_v_mul_hi_u32(i) asm volatile("v_mul_hi_u32 %0, %1, %0" : "+v"(c[i]) : "v"(ux));
#define OP_v_mul_lo_u32(i) asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(c[i]) : "v"(ux));
#define OP_v_mul_u32_u24(i) asm volatile("v_mul_u32_u24 %0, %1, %0" : "+v"(c[i]) : "v"(ux));
#define OP_v_rcp_f32(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
#define OP_v_sqrt_f32(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
#define OP_v_rsq_f32(i) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i]));
#define OP_v_div_scale_f32(i) asm volatile("v_div_scale_f32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y) : "vcc");
#define OP_v_div_fmas_f32(i) asm volatile("v_div_fmas_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y) : "vcc");
#define OP_v_div_fixup_f32(i) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));
#define OP_v_cndmask_b32(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(x) : );
#define OP_v_cmp_lt_f32(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(x) : "vcc");
#define OP_v_cmp_lt_f32_sdst(i) asm volatile("v_cmp_lt_f32 s[20:21], %0, %1" : : "v"(a[i]), "v"(x) : "s20", "s21");
#define OP_v_mov_b32(i) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(x));
#define OP_v_xor_b32(i) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(c[i]) : "v"(ux));
#define OP_v_add_u32(i) asm volatile("v_add_u32 %0, %1, %0" : "+v"(c[i]) : "v"(ux));
#define OP_v_lshl_add_u64(i) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(b[i]) : "v"(b[(i + 1) & 7]));
#define OP_v_cvt_f32_u32(i) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(a[i]) : "v"(c[i]));
#define OP_v_readlane_b32(i) asm volatile("v_readlane_b32 s20, %0, 3" : : "v"(a[i]) : "s20");
#define OP_v_writelane_b32(i) asm volatile("v_writelane_b32 %0, %1, 5" : "+v"(a[i]) : "s"(s0));
#define OP_ds_bpermute_b32(i) asm volatile("ds_bpermute_b32 %0, %1, %0" : "+v"(c[i]) : "v"(la) : "memory");
#define OP_ds_read_b32(i) asm volatile("ds_read_b32 %0, %1" : "=v"(c[i]) : "v"(la) : "memory");
#define OP_ds_read_b128(i) asm volatile("ds_read_b128 %0, %1" : "=v"(q[i]) : "v"(la) : "memory");
#define OP_s_add_u32(i) asm volatile("s_add_u32 %0, %0, %1" : "+s"(s0) : "s"(s1) : "scc");
#define OP_s_and_b64(i) asm volatile("s_and_b64 %0, %0, exec" : "+s"(m0) : : "scc");
#define OP_s_saveexec_pair(i) asm volatile("s_and_saveexec_b64 s[20:21], %0\n s_or_b64 exec, exec, s[20:21]" : : "s"(m0) : "s20", "s21", "scc");
#define OP_saveexec_valu(i) asm volatile("s_and_saveexec_b64 s[20:21], %1\n v_add_f32 %0, %2, %0\n s_or_b64 exec, exec, s[20:21]" : "+v"(a[i]) : "s"(m0), "v"(x) : "s20", "s21", "scc");
#define OP_branch_not_taken(i) asm volatile("s_cmp_eq_u32 %0, 0\n s_cbranch_scc1 1f\n1:" : : "s"(s1) : "scc");
#define OP_branch_execz_not_taken(i) asm volatile("s_cbranch_execz 1f\n v_add_f32 %0, %1, %0\n1:" : "+v"(a[i]) : "v"(x));
#define OP_branch_taken(i) asm volatile("s_branch 1f\n s_nop 0\n1:" :::);
#define OP_s_load_dwordx4(i) asm volatile("s_load_dwordx4 %0, %1, 0x0" : "=s"(q[i]) : "s"(mem) : "memory");
#define OP_valu_salu_pair(i) asm volatile("v_add_f32 %0, %2, %0\n s_add_u32 %1, %1, 1" : "+v"(a[i]), "+s"(s0) : "v"(x) : "scc");
#define OP_valu_2salu(i) asm volatile("v_add_f32 %0, %2, %0\n s_add_u32 %1, %1, 1\n s_xor_b32 %1, %1, 5" : "+v"(a[i]), "+s"(s0) : "v"(x) : "scc");
#define OP_valu_nop(i) asm volatile("v_add_f32 %0, %1, %0\n s_nop 0" : "+v"(a[i]) : "v"(x));
#define OP_valu_dep_chain(i) asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[0]) : "v"(x));
#define OP_valu_then_readlane(i) asm volatile("v_add_f32 %0, %1, %0\n v_readlane_b32 s20, %0, 3" : "+v"(a[i]) : "v"(x) : "s20");

#define OP_v_cndmask_e64_sgpr(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "s"(m0));
#define OP_cmp_nop_cndmask(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n s_nop 1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(x) : "vcc");
#define OP_cmp_cndmask_other(i) asm volatile("v_cmp_lt_f32 vcc, %0, %2\n v_cndmask_b32 %1, %1, %2, vcc" : "+v"(a[i]), "+v"(a[(i + 4) & 7]) : "v"(x) : "vcc");
#define OP_cmp64_cndmask64(i) asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %1\n s_nop 1\n v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a[i]) : "v"(x) : "s20", "s21");
#define OP_v_add_f32_inline(i) asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(a[i]));
#define OP_v_add_f32_literal(i) asm volatile("v_add_f32 %0, 0x3f8ccccd, %0" : "+v"(a[i]));
#define OP_v_fma_f32_sgpr(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "s"(s0));
#define OP_v_pk_mul_f32_sgpr(i) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(p[i]) : "s"(m0));
#define OP_v_pk_fma_f32_sgpr(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "s"(m0), "v"(px));
#define OP_v_max_f32(i) asm volatile("v_max_f32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
#define OP_v_mul_f32_e64_mod(i) asm volatile("v_mul_f32_e64 %0, -%1, |%0|" : "+v"(a[i]) : "v"(x));
#define OP_v_fmac_f32(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));
#define OP_v_fmamk_f32(i) asm volatile("v_fmamk_f32 %0, %1, 0x3f8ccccd, %0" : "+v"(a[i]) : "v"(x));
#define OP_v_and_b32_sgpr(i) asm volatile("v_and_b32 %0, %1, %0" : "+v"(c[i]) : "s"(s0));
#define OP_v_bfe_u32(i) asm volatile("v_bfe_u32 %0, %0, 3, 7" : "+v"(c[i]));
#define OP_v_lshrrev_b32(i) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(c[i]));
#define OP_v_and_or_b32(i) asm volatile("v_and_or_b32 %0, %0, %1, %1" : "+v"(c[i]) : "v"(ux));
#define OP_v_add3_u32(i) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(c[i]) : "v"(ux));
#define OP_v_cvt_u32_f32(i) asm volatile("v_cvt_u32_f32 %0, %1" : "=v"(c[i]) : "v"(a[i]));
#define OP_v_rndne_f32(i) asm volatile("v_rndne_f32 %0, %0" : "+v"(a[i]));
#define OP_v_floor_f32(i) asm volatile("v_floor_f32 %0, %0" : "+v"(a[i]));
#define OP_v_mov_b32_sgpr(i) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "s"(s0));
#define OP_v_mov_b64(i) asm volatile("v_mov_b64 %0, %1" : "=v"(b[i]) : "v"(b[(i + 1) & 7]));
#define OP_s_mul_i32(i) asm volatile("s_mul_i32 %0, %0, %1" : "+s"(s0) : "s"(s1));
#define OP_s_cselect_b64(i) asm volatile("s_cselect_b64 %0, %0, exec" : "+s"(m0));
#define OP_s_mov_b32(i) asm volatile("s_mov_b32 %0, %1" : "=s"(s0) : "s"(s1));
#define OP_ds_write2_b32(i) asm volatile("ds_write2_b32 %0, %1, %1 offset1:1" : : "v"(la), "v"(c[i]) : "memory");
#define OP_ds_read2_b32(i) asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(b[i]) : "v"(la) : "memory");
#define OP_valu4_salu1(i) asm volatile("v_add_f32 %0, %2, %0\n v_mul_f32 %0, %2, %0\n v_add_f32 %0, %2, %0\n v_mul_f32 %0, %2, %0\n s_add_u32 %1, %1, 1" : "+v"(a[i]), "+s"(s0) : "v"(x) : "scc");
#define OP_valu2_sgprvalu2(i) asm volatile("v_add_f32 %0, %2, %0\n v_mul_f32 %0, %1, %0\n v_add_f32 %0, %2, %0\n v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "s"(s0), "v"(x));

#define OP_mix_max_add(i) asm volatile("v_max_f32 %0, %2, %0\n v_add_f32 %1, %2, %1" : "+v"(a[i]), "+v"(a[(i + 4) & 7]) : "v"(x));
#define OP_mix_rcp_add3(i) asm volatile("v_rcp_f32 %0, %0\n v_add_f32 %1, %2, %1\n v_mul_f32 %1, %2, %1\n v_add_f32 %1, %2, %1" : "+v"(a[i]), "+v"(a[(i + 4) & 7]) : "v"(x));
#define OP_mix_rcp_add1(i) asm volatile("v_rcp_f32 %0, %0\n v_add_f32 %1, %2, %1" : "+v"(a[i]), "+v"(a[(i + 4) & 7]) : "v"(x));
#define OP_cmp_nop_cndmask3(i) asm volatile("v_cmp_lt_f32 vcc, %0, %3\n s_nop 1\n v_cndmask_b32 %0, %0, %3, vcc\n v_cndmask_b32 %1, %1, %3, vcc\n v_cndmask_b32 %2, %2, %3, vcc" : "+v"(a[i]), "+v"(a[(i + 3) & 7]), "+v"(a[(i + 5) & 7]) : "v"(x) : "vcc");
#define OP_cmp64_cndmask64x3(i) asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %3\n s_nop 1\n v_cndmask_b32_e64 %0, %0, %3, s[20:21]\n v_cndmask_b32_e64 %1, %1, %3, s[20:21]\n v_cndmask_b32_e64 %2, %2, %3, s[20:21]" : "+v"(a[i]), "+v"(a[(i + 3) & 7]), "+v"(a[(i + 5) & 7]) : "v"(x) : "s20", "s21");
#define OP_mix_pk_add2(i) asm volatile("v_pk_mul_f32 %0, %3, %0\n v_add_f32 %1, %2, %1\n v_mul_f32 %1, %2, %1" : "+v"(p[i]), "+v"(a[i]) : "v"(x), "v"(px));
#define OP_mix_mad64_xor2(i) asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_xor_b32 %1, %2, %1\n v_xor_b32 %1, %3, %1" : "+v"(b[i]), "+v"(c[i]) : "v"(ux), "v"(c[(i + 1) & 7]) : "vcc");
#define OP_mix_max_sgprmul(i) asm volatile("v_max_f32 %0, %2, %0\n v_mul_f32 %1, %3, %1" : "+v"(a[i]), "+v"(a[(i + 4) & 7]) : "v"(x), "s"(s0));
#define OP_v_min_f32(i) asm volatile("v_min_f32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
#define OP_v_sub_f32(i) asm volatile("v_sub_f32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
#define OP_v_and_b32(i) asm volatile("v_and_b32 %0, %1, %0" : "+v"(c[i]) : "v"(ux));
#define OP_v_or_b32(i) asm volatile("v_or_b32 %0, %1, %0" : "+v"(c[i]) : "v"(ux));
#define OP_v_lshlrev_b32(i) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(c[i]));
#define OP_v_sub_u32(i) asm volatile("v_sub_u32 %0, %1, %0" : "+v"(c[i]) : "v"(ux));
#define OP_v_add_co_u32(i) asm volatile("v_add_co_u32 %0, vcc, %1, %0" : "+v"(c[i]) : "v"(ux) : "vcc");
#define OP_v_cmp_class(i) asm volatile("v_cmp_class_f32 vcc, %0, %1" : : "v"(a[i]), "v"(ux) : "vcc");
#define OP_v_ldexp_f32(i) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(a[i]) : "v"(ux));
#define OP_v_mul_legacy(i) asm volatile("v_mul_legacy_f32 %0, %1, %0" : "+v"(a[i]) : "v"(x));
#define OP_v_fma_f32_neg(i) asm volatile("v_fma_f32 %0, -%1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
#define OP_v_add_f32_dpp(i) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
#define OP_global_store_x4(i) asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(gaddr), "v"(q[i]) : "memory");

#define OP_cnd_vcc_e64(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(x));
#define OP_cnd_vcc_interleaved(i) asm volatile("v_cndmask_b32 %0, %0, %2, vcc\n v_add_f32 %1, %2, %1" : "+v"(a[i]), "+v"(a[(i + 4) & 7]) : "v"(x));
#define OP_cmp_nop_cnd3_e64vcc(i) asm volatile("v_cmp_lt_f32 vcc, %0, %3\n s_nop 1\n v_cndmask_b32_e64 %0, %0, %3, vcc\n v_cndmask_b32_e64 %1, %1, %3, vcc\n v_cndmask_b32_e64 %2, %2, %3, vcc" : "+v"(a[i]), "+v"(a[(i + 3) & 7]), "+v"(a[(i + 5) & 7]) : "v"(x) : "vcc");
#define OP_cmp_nop_cnd3_spaced(i) asm volatile("v_cmp_lt_f32 vcc, %0, %3\n s_nop 1\n v_cndmask_b32 %0, %0, %3, vcc\n v_add_f32 %4, %3, %4\n v_cndmask_b32 %1, %1, %3, vcc\n v_add_f32 %4, %3, %4\n v_cndmask_b32 %2, %2, %3, vcc\n v_add_f32 %4, %3, %4" : "+v"(a[i]), "+v"(a[(i + 3) & 7]), "+v"(a[(i + 5) & 7]) : "v"(x), "v"(y) : "vcc");
#define OP_salu_vcc_cnd3(i) asm volatile("s_and_b64 vcc, exec, %3\n v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc" : "+v"(a[i]), "+v"(a[(i + 3) & 7]), "+v"(a[(i + 5) & 7]) : "s"(m0), "v"(x) : "vcc", "scc");
#define OP_cnd_const_vcc(i) asm volatile("v_cndmask_b32 %0, 0, %1, vcc" : "=v"(a[i]) : "v"(x));
#define OP_addc_vcc(i) asm volatile("v_addc_co_u32 %0, vcc, %0, %1, vcc" : "+v"(c[i]) : "v"(ux) : "vcc");

#define ALL_OPS(X)                                                                                               \
    X(v_add_f32, 1) X(v_mul_f32, 1) X(v_fma_f32, 1) X(v_mul_f32_sgpr, 1) X(v_pk_mul_f32, 1) X(v_pk_add_f32, 1)   \
    X(v_pk_fma_f32, 1) X(v_mad_u64_u32, 1) X(v_mul_hi_u32, 1) X(v_mul_lo_u32, 1) X(v_mul_u32_u24, 1)             \
    X(v_rcp_f32, 1) X(v_sqrt_f32, 1) X(v_rsq_f32, 1) X(v_div_scale_f32, 1) X(v_div_fmas_f32, 1)                  \
    X(v_div_fixup_f32, 1) X(v_cndmask_b32, 1) X(v_cmp_lt_f32, 1) X(v_cmp_lt_f32_sdst, 1) X(v_mov_b32, 1)         \
    X(v_xor_b32, 1) X(v_add_u32, 1) X(v_lshl_add_u64, 1) X(v_cvt_f32_u32, 1) X(v_readlane_b32, 1)                \
    X(v_writelane_b32, 1) X(ds_bpermute_b32, 1) X(ds_read_b32, 1) X(ds_read_b128, 1) X(s_add_u32, 1)             \
    X(s_and_b64, 1) X(s_saveexec_pair, 2) X(saveexec_valu, 3) X(branch_not_taken, 2) X(branch_execz_not_taken, 2) \
    X(branch_taken, 1) X(s_load_dwordx4, 1) X(valu_salu_pair, 2) X(valu_2salu, 3) X(valu_nop, 2)                 \
    X(valu_dep_chain, 1) X(valu_then_readlane, 2) X(v_cndmask_e64_sgpr, 1) X(cmp_nop_cndmask, 2)                 \
    X(cmp_cndmask_other, 2) X(cmp64_cndmask64, 2) X(v_add_f32_inline, 1) X(v_add_f32_literal, 1)                 \
    X(v_fma_f32_sgpr, 1) X(v_pk_mul_f32_sgpr, 1) X(v_pk_fma_f32_sgpr, 1) X(v_max_f32, 1) X(v_mul_f32_e64_mod, 1) \
    X(v_fmac_f32, 1) X(v_fmamk_f32, 1) X(v_and_b32_sgpr, 1) X(v_bfe_u32, 1) X(v_lshrrev_b32, 1)                  \
    X(v_and_or_b32, 1) X(v_add3_u32, 1) X(v_cvt_u32_f32, 1) X(v_rndne_f32, 1) X(v_floor_f32, 1)                  \
    X(v_mov_b32_sgpr, 1) X(v_mov_b64, 1) X(s_mul_i32, 1) X(s_cselect_b64, 1) X(s_mov_b32, 1)                     \
    X(ds_write2_b32, 1) X(ds_read2_b32, 1) X(valu4_salu1, 5) X(valu2_sgprvalu2, 4) X(mix_max_add, 2)             \
    X(mix_rcp_add3, 4) X(mix_rcp_add1, 2) X(cmp_nop_cndmask3, 5) X(cmp64_cndmask64x3, 5) X(mix_pk_add2, 3)       \
    X(mix_mad64_xor2, 3) X(mix_max_sgprmul, 2) X(v_min_f32, 1) X(v_sub_f32, 1) X(v_and_b32, 1) X(v_or_b32, 1)    \
    X(v_lshlrev_b32, 1) X(v_sub_u32, 1) X(v_add_co_u32, 1) X(v_cmp_class, 1) X(v_ldexp_f32, 1)                   \
    X(v_mul_legacy, 1) X(v_fma_f32_neg, 1) X(v_add_f32_dpp, 1) X(global_store_x4, 1) X(cnd_vcc_e64, 1)           \
    X(cnd_vcc_interleaved, 2) X(cmp_nop_cnd3_e64vcc, 5) X(cmp_nop_cnd3_spaced, 8) X(salu_vcc_cnd3, 4)            \
    X(cnd_const_vcc, 1) X(addc_vcc, 1)

#define DEFINE(NAME, N) KERNEL(NAME, OP_##NAME)
ALL_OPS(DEFINE)

typedef void (*kern_t)(Stamp *, int, float, float, const i4 *);
struct Op { const char *name; kern_t fn; int insts_per_slot; };
static int iters_for(const char *n) {
    const std::string s(n);
    if (s == "ds_read_b128") return 64;
    if (s == "global_store_x4") return 256;
    if (s == "ds_bpermute_b32" || s == "s_load_dwordx4" || s == "ds_read_b32" || s == "ds_write2_b32" || s == "ds_read2_b32") return 512;
    return 2048;
}
#define ENTRY(NAME, N) {#NAME, k_##NAME, N},
static const Op OPS[] = {ALL_OPS(ENTRY)};

int main(int argc, char **argv) {
    const char *out_path = argc > 1 ? argv[1] : "valu_issue_costs.json";
    hipDeviceProp_t prop;
    HIP_OK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    Stamp *d = nullptr;
    i4 *mem = nullptr;
    const int max_blocks = cus * 8;
    HIP_OK(hipMalloc((void **)&d, sizeof(Stamp) * max_blocks * 4));
    HIP_OK(hipMalloc((void **)&mem, 4096 + (size_t)max_blocks * 256 * 16));
    HIP_OK(hipMemset(mem, 0, 4096));
    std::vector<Stamp> h(max_blocks * 4);
    FILE *f = fopen(out_path, "w");
    if (!f) { perror(out_path); return 1; }
    fprintf(f, "{\n \"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d,\n \"unit\": \"SIMD cycles per wave64 instruction = median over SIMDs of (first stamp .. last stamp on the SIMD) / instructions issued on it; s_memtime ticks\",\n \"ops\": {\n",
            prop.name, cus, prop.clockRate / 1000);
    const int Ws[] = {1, 2, 4, 7};
    bool first = true;
    const char *filter = argc > 2 ? argv[2] : nullptr;
    for (const Op &op : OPS) {
        if (filter && !strstr(op.name, filter)) continue;
        fprintf(f, "%s  \"%s\": {", first ? "" : ",\n", op.name);
        first = false;
        printf("%-26s", op.name);
        const int iters = iters_for(op.name);
        for (int wi = 0; wi < 4; ++wi) {
            const int W = Ws[wi], blocks = cus * W;
            HIP_OK(hipMemset(d, 0, sizeof(Stamp) * blocks * 4));
            hipEvent_t e0, e1;
            HIP_OK(hipEventCreate(&e0));
            HIP_OK(hipEventCreate(&e1));
            hipLaunchKernelGGL(op.fn, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 0.9999f, mem);   // warms the instruction cache
            HIP_OK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(op.fn, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 0.9999f, mem);
            HIP_OK(hipEventRecord(e1, 0));
            HIP_OK(hipDeviceSynchronize());
            float ms = 0.0f;
            HIP_OK(hipEventElapsedTime(&ms, e0, e1));
            HIP_OK(hipEventDestroy(e0));
            HIP_OK(hipEventDestroy(e1));
            HIP_OK(hipMemcpy(h.data(), d, sizeof(Stamp) * blocks * 4, hipMemcpyDeviceToHost));
            // per SIMD (xcc, se, sh, cu, simd -- HW_ID bits simd 5:4, cu 11:8, sh 12, se 15:13): the waves that ran there, how
            // much of the time they overlapped (sum of wave lifetimes / span), and the SIMD cycles per instruction
            struct Acc { double sum = 0; unsigned long long lo = ~0ull, hi = 0; int n = 0; };
            std::map<std::tuple<unsigned, unsigned>, Acc> per_simd;
            for (int i = 0; i < blocks * 4; ++i) {
                Acc &a = per_simd[{h[i].xcc & 0xf, h[i].hw & 0xfff0u & ~0xc0u}];
                a.sum += (double)(h[i].t1 - h[i].t0);
                a.lo = std::min(a.lo, h[i].t0);
                a.hi = std::max(a.hi, h[i].t1);
                a.n += 1;
            }
            const double n_inst = (double)iters * 64.0 * op.insts_per_slot;
            std::vector<double> cost, per_wave, conc;
            for (auto &kv : per_simd) {
                const Acc &a = kv.second;
                cost.push_back((double)(a.hi - a.lo) / (n_inst * a.n));      // span / instructions issued on this SIMD
                conc.push_back(a.sum / (double)(a.hi - a.lo));
            }
            for (int i = 0; i < blocks * 4; ++i) per_wave.push_back((double)(h[i].t1 - h[i].t0) / n_inst);
            std::sort(cost.begin(), cost.end());
            std::sort(per_wave.begin(), per_wave.end());
            std::sort(conc.begin(), conc.end());
            const double med = cost[cost.size() / 2], medw = per_wave[per_wave.size() / 2], medc = conc[conc.size() / 2];
            const double ns_per_inst = (double)ms * 1e6 / (n_inst * (double)blocks * 4.0 / 1024.0);   // wall clock, per SIMD
            fprintf(f, "%s\"W%d\": {\"simd_cycles_per_inst\": %.3f, \"wave_cycles_per_inst\": %.3f, \"waves_overlapping\": %.2f, "
                       "\"simd_ns_per_inst_wallclock\": %.4f, \"simds_used\": %zu}",
                    wi ? ", " : "", W, med, medw, medc, ns_per_inst, per_simd.size());
            printf("  W%d %5.2f (wave %6.2f x%.1f, %.3f ns)", W, med, medw, medc, ns_per_inst);
        }
        fprintf(f, "}");
        printf("\n");
    }
    fprintf(f, "\n }\n}\n");
    fclose(f);
    return 0;
}
