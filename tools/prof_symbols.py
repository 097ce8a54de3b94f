#!/usr/bin/env python3
"""tools/prof_symbols.py <samples written under $BSX_PROF_SAMPLE> : CPU time by function.  Offsets inside this repository's shared objects are
named with `nm` on the local build (the same binaries travel to the GPU box); other objects are reported as a whole."""
import bisect
import collections
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def symtab(path):
    out = subprocess.run(["nm", "-n", "--defined-only", path], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode()
    addr, name = [], []
    for l in out.splitlines():
        p = l.split()
        if len(p) == 3 and p[1] in "tTwW":
            addr.append(int(p[0], 16)); name.append(p[2])
    return addr, name


def main():
    by = collections.Counter(); tot = 0; tabs = {}
    for l in open(sys.argv[1]):
        if l.startswith("#"):
            continue
        n, off, obj = l.split(None, 2)
        n, off, obj = int(n), int(off, 16), obj.strip()
        tot += n
        via = ""
        if obj.startswith("libc<-"):   # a sample in a system library whose return address points into this repository's code
            obj = obj[6:]; via = "libc called from "
        base = os.path.basename(obj)
        local = {"libbiscuit_amd.so": "biscuit_amd/libbiscuit_amd.so", "liboracle_port.so": "oracle/liboracle_port.so"}.get(base)
        if not local and os.path.exists(obj) and len(sys.argv) > 3 and sys.argv[3] in base:
            local = obj   # (the same image here and on the GPU box: system libraries resolve too -- dynamic symbols only)
        if local:
            if base not in tabs:
                tabs[base] = symtab(os.path.join(ROOT, local))
                if not tabs[base][0]:
                    out = subprocess.run(["nm", "-n", "-D", "--defined-only", os.path.join(ROOT, local)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode()
                    a, nmn = [], []
                    for l2 in out.splitlines():
                        q = l2.split()
                        if len(q) == 3 and q[1] in "tTwWiI":
                            a.append(int(q[0], 16)); nmn.append(q[2])
                    tabs[base] = (a, nmn)
            a, nm = tabs[base]
            i = bisect.bisect_right(a, off) - 1
            by[via + (nm[i] if i >= 0 else "?") + " [" + base + "]"] += n
        else:
            by["[" + base + "]"] += n
    print("%d samples (ms of CPU)" % tot)
    for k, v in by.most_common(int(sys.argv[2]) if len(sys.argv) > 2 else 45):
        print("%8d %5.1f%%  %s" % (v, 100.0 * v / tot, k))


if __name__ == "__main__":
    main()
