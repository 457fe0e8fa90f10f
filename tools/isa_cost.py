#!/usr/bin/env python3
"""Vector-ALU issue cost of a stretch of a gfx950 ISA listing, priced with the per-opcode rates tools/op_rate.hip measured
(profiles/r03_op_rates.txt): 2 clocks per wave64 instruction for fp32 fma / mul / add / sub, u32 add / sub, and / or / xor /
not, mov and right shifts (an fma with an SGPR source: 4), 8 for the transcendentals, 4 for every other vector instruction.
usage: isa_cost.py listing.s first_line last_line [--top N]"""
import re
import sys
from collections import Counter

FULL = {"v_fma_f32", "v_fmac_f32", "v_fmamk_f32", "v_fmaak_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32",
        "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_mov_b32",
        "v_lshrrev_b32", "v_ashrrev_i32", "v_mul_lo_u16", "v_mul_legacy_f32", "v_nop"}
QUARTER = {"v_log_f32", "v_exp_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag_f32",
           "v_exp_legacy_f32", "v_log_legacy_f32"}


def price(line):
    m = re.match(r"\s*(v_[a-z0-9_]+)\s*(.*)", line)
    if not m:
        return None
    op, args = m.group(1), m.group(2).split(";")[0]
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    if op.endswith("_dpp") or op.endswith("_sdwa"):
        return base + " (dpp/sdwa)", 4
    if base in QUARTER:
        return base, 8
    if base in FULL:
        if base in ("v_fma_f32", "v_fmac_f32") and re.search(r"(^|[ ,\-|])s\d+|s\[", args):
            return base + " (sgpr)", 4
        return base, 2
    return base, 4


def march_loop_counts(path, kernel="_ZN2vx14render_dvr_ldsILi16ELb0ELb0ELb0E"):
    """the all-lanes march loop of the headline kernel in the listing: (vector instructions every wave step executes,
    vector instructions of the block only wave steps with a sample inside the sample range execute, priced clocks of the
    two).  The loop is the first block after the kernel's main loop header that holds the four ds_read2_b32 of a sample's
    eight taps; the in-range block is the stretch its one forward s_cbranch_scc1 jumps over.  bench.py's NECESSARY_VALU /
    NECESSARY_CLK are the hand-derived minima these counts are held against (tests/test_isa_lint.py)."""
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith(kernel) and re.match(r"\w+:", l))   # the function's label
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    first = next(i for i in range(start, end) if "ds_read2_b32" in lines[i])
    top = max(i for i in range(start, first) if re.match(r"\.LBB\d+_\d+:", lines[i]))
    label = lines[top].split(":")[0]
    back = next(i for i in range(first, end) if re.match(r"\s*s_c?branch(_\w+)?\s+" + re.escape(label) + r"\b", lines[i]))
    skip = next(i for i in range(first, back) if re.match(r"\s*s_cbranch_scc1\s+\.LBB", lines[i]))
    target = lines[skip].split()[-1]
    join = next(i for i in range(skip, back) if lines[i].startswith(target + ":"))

    def count(a, b):
        n = clk = 0
        for l in lines[a:b]:
            if l.lstrip().startswith(";"):
                continue
            pr = price(l)
            if pr:
                n += 1; clk += pr[1]
        return n, clk
    always = [count(top, skip), count(join, back)]
    tf = count(skip, join)
    reads = sum(1 for l in lines[top:back] if "ds_read2_b32" in l)
    return {"per_step": always[0][0] + always[1][0], "per_step_clk": always[0][1] + always[1][1],
            "per_tf_step": tf[0], "per_tf_step_clk": tf[1], "tap_reads": reads, "lines": (top + 1, back + 1)}


def main():
    path, a, b = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 12
    lines = open(path).read().splitlines()[a - 1:b]
    cost, count = Counter(), Counter()
    other = Counter()
    for l in lines:
        if l.lstrip().startswith(";"):
            continue
        pr = price(l)
        if pr:
            cost[pr[0]] += pr[1]; count[pr[0]] += 1
        else:
            m = re.match(r"\s*((s|ds|global|flat|scratch|buffer)_[a-z0-9_]+)", l)
            if m:
                other[m.group(2)] += 1
    n, c = sum(count.values()), sum(cost.values())
    print(f"{n} vector instructions, {c} clocks at the measured rates ({c / max(n, 1):.2f} per instruction); other: {dict(other)}")
    for op, cl in cost.most_common(top):
        print(f"  {op:28s} x{count[op]:3d}  {cl:4d} clk")


if __name__ == "__main__":
    main()
