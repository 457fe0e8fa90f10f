#!/usr/bin/env python3
"""Lint of the gfx950 ISA listing (make asm -> volxel_amd/csrc/vx_api.s) for one miscompile of this toolchain's
register allocator (root cause of the wrong pixels of render_generic<DVR_PHONG, LAYOUT_REF> at 8 waves per SIMD,
DESIGN.md section 5.3):

    .LBB53_108:                                   ; join block of `if (tap in bounds) {...}`
        scratch_store_dword off, v58, off offset:40   ; 4-byte Folded Spill      <-- under the if's narrowed EXEC
        scratch_store_dword off, v57, off offset:36   ; 4-byte Folded Spill
        s_or_b64 exec, exec, s[6:7]                   ; the exec restore that should open the block

The greedy allocator placed VGPR spill code at the top of a join block AHEAD of the `s_or_b64 exec, exec, <saved>`
that re-enables the lanes the `if` had masked off.  Those lanes miss the store (or the reload) and later read a
stale slot.  Nothing in the source is undefined; `-mllvm -vgpr-regalloc=basic` compiles the same source correctly.

The lint flags a block targeted by `s_cbranch_execz` when the only vector instructions between its label and its
first `s_or_b64 exec, exec, ...` are of the kinds the register allocator inserts (scratch spills / reloads, plain
VGPR-to-VGPR or AGPR copies): in the code the allocator was given the exec restore opens such a block, so nothing
else can have got ahead of it.  (A block that starts with arithmetic and meets an `s_or_b64 exec` further down is
the else-arm of an if / else whose restore closes the arm: not this pattern.)  SGPR reloads (v_readlane), scalar
instructions and waits may legitimately stand ahead of the restore.

Second form of the same fault: when the if-body is short the skip branch is dropped and the join block is printed
without a label, so the misplaced code shows as a run of spill instructions directly ahead of a mid-block
`s_or_b64 exec, exec, s[a:b]`.  A spill store there is legitimate when the body defined the stored register (the
allocator stores after the definition); it is flagged when nothing between the matching `s_and_saveexec_b64 s[a:b]`
and the store writes that register, and a reload directly ahead of the restore (its use can only come after it) is
flagged always.

  tools/check_exec_prologue.py volxel_amd/csrc/vx_api.s [--kernels REGEX] [-v]
exit code 1 if any site is flagged."""
import re
import sys

VECTOR = re.compile(r"^\s*(v_(?!readlane|readfirstlane)|scratch_|global_|flat_|buffer_|ds_)")
LABEL = re.compile(r"^(\.LBB\d+_\d+):")
FUNC = re.compile(r"^(_Z[A-Za-z0-9_]+):")
EXEC_OR = re.compile(r"^\s*s_or_b64 exec, exec, ")
EXECZ = re.compile(r"^\s*s_cbranch_execz (\.LBB\d+_\d+)")
RA_INSERTED = re.compile(r"^\s*(scratch_(load|store)_\w+ .*Folded (Spill|Reload)|v_mov_b32_e32 v\d+, v\d+\s*$|"
                         r"v_accvgpr_(read|write)_b32 [av]\d+, [av]\d+\s*$|v_mov_b64_e32 v\[\d+:\d+\], v\[\d+:\d+\]\s*$)")
END_OF_BLOCK = re.compile(r"^\s*(s_cbranch|s_branch|s_endpgm|s_setpc)")


def scan(path, kernel_re=None, verbose=False):
    lines = open(path).read().split("\n")
    # function extents
    funcs = []
    for i, l in enumerate(lines):
        m = FUNC.match(l)
        if m:
            funcs.append((i, m.group(1)))
    funcs.append((len(lines), None))
    findings = []
    stats = {}
    for (start, name), (end, _) in zip(funcs, funcs[1:]):
        if kernel_re and not re.search(kernel_re, name):
            continue
        targets = set()
        for l in lines[start:end]:
            m = EXECZ.match(l)
            if m:
                targets.add(m.group(1))
        n_blocks = n_spills = 0
        for l in lines[start:end]:
            if "Folded Spill" in l or "Folded Reload" in l:
                n_spills += 1
        i = start
        while i < end:
            m = LABEL.match(lines[i])
            if m and m.group(1) in targets:
                n_blocks += 1
                j = i + 1
                before = []
                found = False
                while j < end and not LABEL.match(lines[j]) and not END_OF_BLOCK.match(lines[j]):
                    if EXEC_OR.match(lines[j]):
                        found = True
                        break
                    if VECTOR.match(lines[j]):
                        before.append((j + 1, lines[j].rstrip()))
                    j += 1
                if found and before and all(RA_INSERTED.match(t) for _, t in before):
                    findings.append((name, m.group(1), before))
            i += 1
        # label-less join blocks: spill code directly ahead of a mid-block exec restore
        SKIPPABLE = re.compile(r"^\s*(s_waitcnt|s_nop|v_readlane_b32|;|$)")
        for j in range(start, end):
            mo = re.match(r"^\s*s_or_b64 exec, exec, (s\[\d+:\d+\])", lines[j])
            if not mo or LABEL.match(lines[j - 1]):
                continue
            k = j - 1
            run = []
            while k > start and (SKIPPABLE.match(lines[k]) or RA_INSERTED.match(lines[k])):
                if RA_INSERTED.match(lines[k]) and "Folded" in lines[k]:
                    run.append(k)
                k -= 1
            if not run or LABEL.match(lines[k]):
                continue   # nothing there, or the run starts a labelled block (handled above)
            # the body: back to the instruction that saved exec into this SGPR pair, without crossing a label
            b = k
            opened = None
            while b > start and not LABEL.match(lines[b]):
                if re.match(r"^\s*(s_and_saveexec_b64|s_or_saveexec_b64) %s," % re.escape(mo.group(1)), lines[b]) or \
                   re.match(r"^\s*s_mov_b64 %s, exec" % re.escape(mo.group(1)), lines[b]):
                    opened = b
                    break
                b -= 1
            if opened is None:
                continue
            bad = []
            for r in run:
                mm = re.match(r"^\s*scratch_(load|store)_dword(?:x\d)? (?:off, )?(v\d+|v\[\d+:\d+\])", lines[r])
                if not mm:
                    continue
                if mm.group(1) == "load":
                    bad.append((r + 1, lines[r].rstrip()))
                    continue
                reg = mm.group(2)
                nums = [int(x) for x in re.findall(r"\d+", reg)]
                regs = set(range(nums[0], nums[-1] + 1))
                written = False
                for q in range(opened + 1, r):
                    w = re.match(r"^\s*(?:v_\w+|ds_read\w*|global_load\w*|flat_load\w*|scratch_load\w*|buffer_load\w*) (v\d+|v\[\d+:\d+\])", lines[q])
                    if w:
                        wn = [int(x) for x in re.findall(r"\d+", w.group(1))]
                        if regs & set(range(wn[0], wn[-1] + 1)):
                            written = True
                            break
                    if ";;#ASMSTART" in lines[q]:   # inline asm: look at its text
                        w2 = re.match(r"^\s*\w+ (v\d+)", lines[q + 1])
                        if w2 and int(w2.group(1)[1:]) in regs:
                            written = True
                            break
                if not written:
                    bad.append((r + 1, lines[r].rstrip()))
            if bad:
                findings.append((name, "(no label) restore at line %d" % (j + 1), bad))
        stats[name] = (n_blocks, n_spills)
    return findings, stats


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("-")]
    kernel_re = None
    if "--kernels" in sys.argv:
        kernel_re = sys.argv[sys.argv.index("--kernels") + 1]
        args = [a for a in args if a != kernel_re]
    verbose = "-v" in sys.argv
    path = args[0]
    findings, stats = scan(path, kernel_re, verbose)
    if verbose:
        for name, (nb, ns) in stats.items():
            print(f"{name[:90]:90s} execz join blocks {nb:4d}  spill instructions {ns:4d}")
    for name, label, before in findings:
        print(f"FLAGGED {name}\n   {label}: vector instructions ahead of the exec restore:")
        for ln, text in before:
            print(f"      line {ln}: {text}")
    print(f"{len(findings)} flagged join block(s) in {len(stats)} kernel(s)")
    return 1 if findings else 0


if __name__ == "__main__":
    sys.exit(main())
