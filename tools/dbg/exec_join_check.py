#!/usr/bin/env python3
"""Static check of the compiler's gfx950 assembly for one miscompile seen in round 5 ("the fault that was not an LDS overrun"):
vector instructions placed at the head of the block where a divergent region ends, AHEAD of the s_or_b64 that restores EXEC -- so a
register copy every lane needs afterwards (there: the lane index handed to a non-inlined rg_export) is made in the lanes of the region only.

What is flagged: a label that some `s_cbranch_execz` jumps to, followed -- before the `s_or_b64 exec, exec, s[..]` that closes the region --
by register-to-register copies (v_mov vA, vB) into vector registers which are read again after EXEC has been restored (before being written).
Usage: exec_join_check.py file.s [...]; exit code 1 when something is flagged."""
import re
import sys

VDST = re.compile(r"^\s*((?:v_|ds_read|ds_bpermute|ds_swizzle|global_load|flat_load|buffer_load|scratch_load)\w*)\s+(v\[?\d+(?::\d+)?\]?)")
VCOPY = re.compile(r"^\s*v_mov_b(?:32|64)_e32\s+(v\[?\d+(?::\d+)?\]?),\s*(v\[?\d+(?::\d+)?\]?)\s*$")


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def vregs_in(text):
    out = set()
    for t in re.findall(r"v\[\d+:\d+\]|v\d+", text):
        out |= regs(t)
    return out


def check(path):
    lines = open(path).read().split("\n")
    code = [(i, l.split(";")[0].rstrip()) for i, l in enumerate(lines)]
    code = [(i, l) for i, l in code if l.strip() and not l.strip().startswith(".") or re.match(r"^\.LBB\w+:", l.strip() if l else "")]
    targets = set(re.findall(r"s_cbranch_execz\s+(\.LBB\w+)", "\n".join(l for _, l in code)))
    label_at = {}
    for k, (i, l) in enumerate(code):
        m = re.match(r"^(\.LBB\w+):", l.strip())
        if m:
            label_at[m.group(1)] = k
    func = None
    func_at = []
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            func_at.append((i, m.group(1)))
    flagged = []
    for lab in sorted(targets):
        if lab not in label_at:
            continue
        k = label_at[lab] + 1
        pre = []          # (line, dst regs) of VALU writes ahead of the exec restore
        while k < len(code):
            i, l = code[k]
            s = l.strip()
            if re.match(r"^\.LBB\w+:", s) or s.startswith("s_branch") or s.startswith("s_cbranch") or s.startswith("s_endpgm") or s.startswith("s_setpc") or s.startswith("s_swappc"):
                pre = []
                break
            if re.match(r"s_or_b64\s+exec,\s*exec,", s) or re.match(r"s_mov_b64\s+exec,", s):
                break
            if "saveexec" in s or re.match(r"s_\w+\s+exec,", s):   # the else side of an if/else (s_andn2_saveexec): other lanes, on purpose
                pre = []
                break
            m = VCOPY.match(l)   # a register-to-register copy: the lanes outside the region keep what they had
            if m:
                pre.append((i, regs(m.group(1)), s))
            k += 1
        else:
            pre = []
        if not pre:
            continue
        # after the restore: is a register written under the narrow EXEC read before it is written again (straight-line scan, a few blocks)?
        live = set()
        for _, r, _ in pre:
            live |= r
        k += 1
        steps = 0
        hit = None
        while k < len(code) and steps < 400 and live:
            i, l = code[k]
            s = l.strip()
            k += 1
            if re.match(r"^\.LBB\w+:", s):
                continue
            steps += 1
            if s.startswith("s_endpgm"):
                break
            mb = re.match(r"s_branch\s+(\.LBB\w+)", s)
            if mb and mb.group(1) in label_at:
                k = label_at[mb.group(1)] + 1
                continue
            m = VDST.match(l)
            srcs = vregs_in(l)
            if m and not m.group(1).startswith(("v_cmp", "v_cmpx")):
                dst = regs(m.group(2))
                srcs_only = vregs_in(l[l.index(m.group(2)) + len(m.group(2)):])
                if srcs_only & live:
                    hit = (i, s, sorted(srcs_only & live))
                    break
                live -= dst
            elif srcs & live:
                hit = (i, s, sorted(srcs & live))
                break
        if hit:
            fn = [n for (a, n) in func_at if a <= pre[0][0]]
            flagged.append((path, fn[-1] if fn else "?", lab, pre, hit))
    return flagged


def main():
    bad = 0
    for p in sys.argv[1:]:
        for path, fn, lab, pre, hit in check(p):
            bad += 1
            print("%s: %s\n  at %s, ahead of the EXEC restore:" % (path, fn[:110], lab))
            for i, r, s in pre:
                print("    line %d: %s" % (i + 1, s))
            print("  read with EXEC restored at line %d: %s (v%s)" % (hit[0] + 1, hit[1], ",v".join(map(str, hit[2]))))
    print("%d place(s) flagged" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
