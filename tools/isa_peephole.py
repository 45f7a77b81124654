#!/usr/bin/env python3
"""Peephole pass over the gfx950 assembly of the render kernels (part of the product build, see csrc/Makefile).

    isa_peephole.py IN.s OUT.s [--mode runs|all|off]

One rewrite, found with tools/valu_microbench.hip on the MI355X (profiles/r02a/valu_microbench.log):

  two VOP2 `v_cndmask_b32_e32 vD, vA, vB, vcc` in a row stall the whole SIMD for ~19 cycles per instruction (every wave on
  it, not just the issuing one; the cost does not shrink with occupancy), while the VOP3 encoding of the same
  instruction, `v_cndmask_b32_e64 vD, vA, vB, vcc`, issues in ~4 cycles and a VOP2 one that follows any other vector
  instruction in ~2.  hipcc always shrinks a select whose mask is VCC to VOP2, and three-component selects
  (`normal = front ? n : -n`) come out as runs of them.

mode `runs` (default) re-encodes every v_cndmask_b32_e32 that directly follows another one as _e64; `all` re-encodes
every one.  Same operation, same operands, same result bits -- only the encoding (8 instead of 4 bytes) changes.
VOP3 on gfx9 cannot carry a 32-bit literal, so an instruction with a literal operand is left alone.
"""
import re
import sys

INLINE_INT = {str(i) for i in range(-16, 65)}
INLINE_FLT = {"0.5", "-0.5", "1.0", "-1.0", "2.0", "-2.0", "4.0", "-4.0", "0", "0.15915494", "0.15915494309189532"}
CND = re.compile(r"^(\s*)v_cndmask_b32_e32\s+(v\d+),\s*([^,]+),\s*(v\d+),\s*vcc\s*(;.*)?$")
INST = re.compile(r"^\s+[a-z][a-z0-9_]+\b")


def convertible(src0):
    src0 = src0.strip()
    return bool(re.fullmatch(r"v\d+", src0)) or src0 in INLINE_INT or src0 in INLINE_FLT


def run(lines, mode):
    out, prev_was_cnd, changed = [], False, 0
    for line in lines:
        m = CND.match(line.rstrip("\n"))
        if m and mode != "off" and convertible(m.group(3)) and (mode == "all" or prev_was_cnd):
            out.append(f"{m.group(1)}v_cndmask_b32_e64 {m.group(2)}, {m.group(3).strip()}, {m.group(4)}, vcc\n")
            changed += 1
            prev_was_cnd = True
            continue
        if INST.match(line) and not line.lstrip().startswith("."):
            op = line.split()[0]
            if m:
                prev_was_cnd = True
            elif op.startswith(("v_", "ds_", "global_", "buffer_", "flat_", "scratch_")):
                prev_was_cnd = False      # another vector / memory instruction in between: no stall (measured)
            # scalar instructions, s_nop and branches in between are not known to help: the run continues
        out.append(line)
    return out, changed


def main():
    if len(sys.argv) < 3:
        sys.exit(__doc__)
    mode = "runs"
    if "--mode" in sys.argv:
        mode = sys.argv[sys.argv.index("--mode") + 1]
    lines = open(sys.argv[1]).readlines()
    out, changed = run(lines, mode)
    open(sys.argv[2], "w").writelines(out)
    print(f"isa_peephole: {changed} v_cndmask_b32_e32 -> _e64 ({mode})", file=sys.stderr)


if __name__ == "__main__":
    main()
