#!/usr/bin/env python3
"""Static instruction histogram of the render-kernel instantiations, weighted by issue cost.

    python3 tools/isa_histogram.py [--asm FILE.s] [--costs profiles/<tag>/valu_issue_costs.json] [KERNEL_SUBSTR ...]

Compiles bendy_tracer_amd/csrc/bt_kernels.hip to gfx950 assembly (hipcc -S, the product's flags) unless --asm names
an existing listing, and prints for every kernel whose mangled name contains one of the substrings (default: the
C3 / Cornell / C4 work-queue instantiations) the number of instructions per opcode class and the classes' share
of the issue-cost-weighted total.  Classes follow the SQ_INSTS_VALU_* PMC counters where one exists, so that the
static mix inside a class (e.g. how many INT32 instructions are quarter-rate multiplies) can be combined with the
dynamic class counts of a `rocprofv3 --pmc` pass (tools/profile.sh -> profiles/<tag>/pmc_summary.json).

Issue costs (cycles a wave64 instruction occupies its SIMD's VALU issue port with several waves resident) come
from tools/valu_microbench.hip, measured on the MI355X; the defaults below are MI355X_MICROARCH.md's constants.
"""
import argparse
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_KERNELS = {
    "C3 (spheres, no volumes)": "ILi0ELb0ELb0ELb0E",
    "C4 (spheres + volumes)": "ILi0ELb0ELb0ELb1E",
    "Cornell / C2 (rects)": "ILi0ELb0ELb1ELb0E",
}
# cycles of VALU issue per wave64 instruction at >= 2 waves / SIMD (MI355X_MICROARCH.md: plain 2 on a SIMD-32,
# transcendentals twice a plain op's cost; quarter-rate 32x32 integer multiplies); overridden by --costs
DEFAULT_COST = {"plain": 2.0, "pk_f32": 4.0, "trans": 4.0, "quarter": 8.0, "f64": 4.0, "lane": 2.0}

QUARTER = ("v_mad_u64_u32", "v_mad_i64_i32", "v_mul_hi_u32", "v_mul_lo_u32", "v_mul_hi_i32", "v_mul_lo_i32")
TRANS = ("v_rcp_", "v_rsq_", "v_sqrt_", "v_exp_", "v_log_", "v_sin_", "v_cos_")
LANE = ("v_readlane", "v_writelane", "v_readfirstlane", "v_permlane", "v_mov_b32_dpp", "v_bpermute")


def classify(op):
    """-> (class, cost key or None).  Classes mirror the SQ PMC counters."""
    if op.startswith("v_"):
        if op.startswith(QUARTER):
            return "valu_int_mul32 (INT32/INT64, quarter rate)", "quarter"
        if op.startswith(TRANS):
            return "valu_trans (TRANS_F32)", "trans"
        if op.startswith("v_pk_"):
            return "valu_pk_f32 (ADD/MUL/FMA_F32, packed)", "pk_f32"
        if op.startswith(LANE):
            return "valu_lane (readlane / writelane)", "lane"
        if op.startswith(("v_div_scale", "v_div_fmas", "v_div_fixup")):
            return "valu_div_helpers", "plain"
        if op.startswith(("v_cvt_", "v_rndne", "v_trunc", "v_floor", "v_ceil", "v_fract")):
            return "valu_cvt (CVT)", "plain"
        if op.startswith(("v_fma_f32", "v_fmac_f32", "v_mad_f32", "v_mac_f32", "v_fmaak", "v_fmamk")):
            return "valu_fma_f32 (FMA_F32)", "plain"
        if op.startswith(("v_mul_f32", "v_mul_legacy")):
            return "valu_mul_f32 (MUL_F32)", "plain"
        if op.startswith(("v_add_f32", "v_sub_f32", "v_subrev_f32")):
            return "valu_add_f32 (ADD_F32)", "plain"
        if op.startswith(("v_cmp", "v_cndmask", "v_max", "v_min", "v_med3")):
            return "valu_cmp_select", "plain"
        if op.startswith("v_mov"):
            return "valu_mov", "plain"
        if re.match(r"v_.*_f64", op):
            return "valu_f64", "f64"
        return "valu_int_logic (INT32: add / shift / logic / bfe / perm)", "plain"
    if op.startswith("s_"):
        if op.startswith(("s_load", "s_buffer_load", "s_scratch_load")):
            return "smem", None
        if op.startswith("s_waitcnt"):
            return "s_waitcnt", None
        if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_endpgm", "s_barrier")):
            return "branch", None
        if op.startswith("s_nop"):
            return "s_nop", None
        return "salu", None
    if op.startswith("ds_"):
        return "lds", None
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem", None
    return "other", None


def kernels_in(asm_path):
    """-> {mangled name: [opcodes]} for every .amdhsa kernel in the listing."""
    out, cur, name = {}, None, None
    label = re.compile(r"^(_Z\w+):")
    inst = re.compile(r"^\t([a-z][a-z0-9_]+)\b")
    for line in open(asm_path):
        m = label.match(line)
        if m:
            name, cur = m.group(1), []
            out[name] = cur
            continue
        if cur is None:
            continue
        if line.startswith(".Lfunc_end"):
            cur = None
            continue
        m = inst.match(line)
        if m and not line.startswith("\t."):
            cur.append(m.group(1))
    return out


def meta_of(asm_path, name):
    meta = {}
    pat = re.compile(r"\.set %s\.(num_vgpr|numbered_sgpr|private_seg_size), (\d+)" % re.escape(name))
    for line in open(asm_path):
        m = pat.search(line)
        if m:
            meta[m.group(1)] = int(m.group(2))
    return meta


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--asm")
    ap.add_argument("--costs")
    ap.add_argument("--json", help="write the histograms here")
    ap.add_argument("kernels", nargs="*")
    args = ap.parse_args()
    asm = args.asm
    if not asm:
        asm = os.path.join(tempfile.mkdtemp(prefix="bt_isa_"), "bt_kernels.s")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-slp-vectorize", "-std=c++17", "-S",
                               "--cuda-device-only", "-o", asm,
                               os.path.join(ROOT, "bendy_tracer_amd", "csrc", "bt_kernels.hip")],
                              stderr=subprocess.DEVNULL)
    cost = dict(DEFAULT_COST)
    if args.costs:
        cost.update(json.load(open(args.costs)).get("issue_cycles", {}))
    wanted = {k: k for k in args.kernels} if args.kernels else DEFAULT_KERNELS
    ks = kernels_in(asm)
    result = {}
    for title, sub in wanted.items():
        for name, ops in ks.items():
            if "bt_render_kernel" not in name or sub not in name:
                continue
            hist, weighted = collections.Counter(), collections.Counter()
            opcodes = collections.defaultdict(collections.Counter)
            for op in ops:
                c, ck = classify(op)
                hist[c] += 1
                opcodes[c][op] += 1
                if ck:
                    weighted[c] += cost[ck]
            total_w = sum(weighted.values())
            n_valu = sum(v for c, v in hist.items() if c.startswith("valu"))
            print(f"\n== {title}: {name}  {meta_of(asm, name)}")
            print(f"   {len(ops)} instructions, {n_valu} VALU, issue-cost-weighted VALU cycles {total_w:.0f} "
                  f"(mean {total_w / max(n_valu, 1):.2f} cycles per VALU instruction; 2.00 = all plain)")
            for c, n in hist.most_common():
                w = weighted.get(c)
                top = ", ".join(f"{o} {k}" for o, k in opcodes[c].most_common(4))
                print(f"   {c:58s} {n:6d}  " + (f"{100 * w / total_w:5.1f}% of VALU issue  " if w else " " * 25) + top)
            result[name] = {"title": title, "instructions": len(ops), "valu": n_valu, "weighted_valu_cycles": total_w,
                            "classes": dict(hist), "weighted": dict(weighted),
                            "opcodes": {c: dict(v) for c, v in opcodes.items()}}
    if args.json:
        json.dump({"issue_cycles": cost, "kernels": result}, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    sys.exit(main())
