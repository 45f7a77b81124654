#!/usr/bin/env python3
"""profiles/valu_issue_costs.json from the microbenchmark and the kernels' static opcode mix.

    python3 tools/make_issue_costs.py profiles/r02a/valu_microbench.json [--asm FILE.s] > profiles/valu_issue_costs.json

For every PMC opcode class (SQ_INSTS_VALU_ADD_F32, ..., "OTHER" = SQ_INSTS_VALU minus the classes) the cost is the mean,
over the class's instructions in the three default render-kernel instantiations (static count, tools/isa_histogram.py's
classification), of the issue cycles the microbenchmark measured for that opcode at 7 waves per SIMD -- the pure-stream
figure, or the mixed-stream figure where a pure stream of the opcode is pathological (v_cndmask_b32 on VCC).
tools/pmc_collect.py multiplies the dynamic class counts of a --pmc pass with these costs: the issue-cost-weighted share of
VALU issue slots in use (`valu_issue_weighted_frac`).
"""
import collections
import json
import os
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa_histogram as ih

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    mb = json.load(open(sys.argv[1]))["ops"]
    asm = sys.argv[sys.argv.index("--asm") + 1] if "--asm" in sys.argv else None
    if not asm:
        asm = os.path.join(tempfile.mkdtemp(prefix="bt_isa_"), "bt_kernels.s")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-slp-vectorize", "-std=c++17", "-S", "--cuda-device-only",
                               "-o", asm, os.path.join(ROOT, "bendy_tracer_amd", "csrc", "bt_kernels.hip")], stderr=subprocess.DEVNULL)

    def c(name, w="W7"):
        return mb[name][w]["simd_cycles_per_inst"]

    plain = (c("v_add_f32") + c("v_mul_f32") + c("v_fma_f32") + c("v_xor_b32") + c("v_add_u32")) / 5
    sgpr_or_other = (c("v_mul_f32_sgpr") + c("v_max_f32") + c("v_bfe_u32") + c("v_add3_u32")) / 4   # SGPR operand / "second-class" opcode
    issue = {
        "plain (v_add/mul/fma/sub_f32, logic, add/sub_u32, lshrrev; VGPR / literal / inline operands)": plain,
        "SGPR operand, v_max/min, v_floor/rndne, v_bfe, 3-operand integer, v_lshlrev, 64-bit moves": sgpr_or_other,
        "v_pk_*_f32 (two f32 operations per lane)": c("v_pk_mul_f32"),
        "v_mul_lo/hi_u32, v_mad_u64_u32": (c("v_mul_lo_u32") + c("v_mul_hi_u32") + c("v_mad_u64_u32")) / 3,
        "v_rcp/rsq/sqrt_f32": c("v_rcp_f32"),
        "v_cvt_*": c("v_cvt_f32_u32"),
        "v_cmp_* (writes VCC / SGPR pair), v_div_scale, v_readlane": c("v_cmp_lt_f32"),
        "v_cndmask_b32 among other VALU instructions": c("cmp_nop_cnd3_spaced") if "cmp_nop_cnd3_spaced" in mb else plain,
        "v_cndmask_b32_e32 back to back on VCC (pathological, whole SIMD stalls; not seen to matter in the render kernels)": c("v_cndmask_b32"),
    }

    def op_cost(op):
        if op.startswith(("v_rcp_", "v_rsq_", "v_sqrt_")):
            return c("v_rcp_f32")
        if op.startswith("v_pk_"):
            return c("v_pk_mul_f32")
        if op.startswith(("v_mad_u64_u32", "v_mad_i64_i32")):
            return c("v_mad_u64_u32")
        if op.startswith(("v_mul_lo_", "v_mul_hi_")):
            return c("v_mul_lo_u32")
        if op.startswith("v_cvt_"):
            return c("v_cvt_f32_u32")
        if op.startswith(("v_cmp", "v_div_scale", "v_readlane", "v_readfirstlane", "v_add_co", "v_addc_co", "v_subb_co", "v_sub_co")):
            return c("v_cmp_lt_f32")
        if op.startswith(("v_div_fmas", "v_div_fixup", "v_writelane", "v_lshl_add_u64", "v_mov_b64", "v_max", "v_min", "v_med3",
                          "v_floor", "v_rndne", "v_trunc", "v_ceil", "v_fract", "v_bfe", "v_and_or", "v_add3", "v_lshlrev", "v_ldexp",
                          "v_lshl_or", "v_or3", "v_add_lshl", "v_lshl_add", "v_perm", "v_alignbit", "v_bfi")):
            return sgpr_or_other
        if op.startswith("v_cndmask"):
            return issue["v_cndmask_b32 among other VALU instructions"] if op.endswith("e32") else c("v_cndmask_e64_sgpr")
        return plain

    # the same opcode in a MIXED stream (second-class opcodes alternating with plain ones issue at the plain rate:
    # mix_max_add, mix_max_sgprmul; a compare and its selects among other VALU work: cmp_nop_cnd3_spaced; a quarter-rate
    # multiply between xors: mix_mad64_xor2) -- what a real kernel's stream looks like
    second_mixed = 2.0 * c("mix_max_add") - c("v_add_f32")
    cmp_mixed = c("cmp_nop_cnd3_spaced")
    mad_mixed = 3.0 * c("mix_mad64_xor2") - 2.0 * c("v_xor_b32")

    def op_cost_mixed(op):
        pure = op_cost(op)
        if op.startswith(("v_rcp_", "v_rsq_", "v_sqrt_", "v_pk_", "v_cvt_", "v_readlane", "v_readfirstlane", "v_writelane",
                          "v_div_scale", "v_div_fmas", "v_div_fixup")):
            return pure
        if op.startswith(("v_mad_u64_u32", "v_mad_i64_i32", "v_mul_lo_", "v_mul_hi_")):
            return min(pure, mad_mixed)
        if op.startswith(("v_cmp", "v_cndmask", "v_add_co", "v_addc_co", "v_subb_co", "v_sub_co")):
            return min(pure, cmp_mixed)
        return min(pure, max(plain, second_mixed))

    pmc_class = {"valu_add_f32 (ADD_F32)": "ADD_F32", "valu_mul_f32 (MUL_F32)": "MUL_F32", "valu_fma_f32 (FMA_F32)": "FMA_F32",
                 "valu_trans (TRANS_F32)": "TRANS_F32", "valu_cvt (CVT)": "CVT"}
    sums, sums_mixed, counts = collections.Counter(), collections.Counter(), collections.Counter()
    for name, ops in ih.kernels_in(asm).items():
        if "bt_render_kernel" not in name or not any(s in name for s in ih.DEFAULT_KERNELS.values()):
            continue
        for op in ops:
            cls, _ = ih.classify(op)
            if not cls.startswith("valu"):
                continue
            if cls in pmc_class:
                k = pmc_class[cls]
            elif op.startswith("v_pk_add"):
                k = "ADD_F32"
            elif op.startswith("v_pk_mul"):
                k = "MUL_F32"
            elif op.startswith("v_pk_fma"):
                k = "FMA_F32"
            elif op.startswith(("v_mad_u64", "v_mad_i64", "v_lshl_add_u64", "v_lshlrev_b64", "v_lshrrev_b64")):
                k = "INT64"
            elif cls.startswith(("valu_int", "valu_int_mul32")):
                k = "INT32"
            else:
                k = "OTHER"
            sums[k] += op_cost(op)
            sums_mixed[k] += op_cost_mixed(op)
            counts[k] += 1
    class_cost = {k: round(sums[k] / counts[k], 3) for k in sorted(counts)}
    class_cost_mixed = {k: round(sums_mixed[k] / counts[k], 3) for k in sorted(counts)}
    json.dump({"source": os.path.relpath(os.path.abspath(sys.argv[1]), ROOT),
               "unit": "SIMD cycles of VALU issue per wave64 instruction (s_memtime ticks = shader cycles), 7 waves per SIMD",
               "issue_cycles": {k: round(v, 3) for k, v in issue.items()},
               "class_cost": class_cost, "class_cost_mixed": class_cost_mixed, "static_instructions_per_class": dict(counts),
               "note": "class_cost = static-mix mean over the C3 / C4 / Cornell work-queue instantiations, every opcode at the cost of a "
                       "stream of its own (an upper bound: the weighted sum can exceed the cycles there are); class_cost_mixed = the same "
                       "with the mixed-stream cost where the microbenchmark measured one (second-class opcodes, compares and selects, "
                       "quarter-rate multiplies between other instructions); packed f32 instructions are filed under the ADD / MUL / FMA "
                       "class of their operation"}, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
