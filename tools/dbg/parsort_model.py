#!/usr/bin/env python3
"""Model of the wave-parallel form of klib's introsort (ksort.h:184-234) used by the chain filter's sort (k_regions.hip, rg_introsort_par):
every Hoare partition is computed at once from the two stop masks, the final insertion pass is a stable rank by counting.  Checked here
against the sequential algorithm on random keys with many ties (the order of equal weights is part of the result)."""
import random
import sys


def lt(a, b):          # flt_lt: heavier first
    return a[0] > b[0]


def klib_combsort(a, s, t):
    """ks_combsort (ksort.h:162-183) over a[s..t]"""
    m = t - s + 1
    gap = m
    while True:
        if gap > 2:
            gap = int(gap / 1.2473309501039786540366528676643)
            if gap == 9 or gap == 10:
                gap = 11
        swapped = False
        for i in range(0, m - gap):
            if lt(a[s + i + gap], a[s + i]):
                a[s + i], a[s + i + gap] = a[s + i + gap], a[s + i]
                swapped = True
        if not (swapped or gap > 2):
            break
    if gap != 1:
        for i in range(s + 1, t + 1):
            j = i
            while j > s and lt(a[j], a[j - 1]):
                a[j], a[j - 1] = a[j - 1], a[j]
                j -= 1


def klib_introsort(a, comb=False):
    n = len(a)
    if n < 1:
        return True
    if n == 2:
        if lt(a[1], a[0]):
            a[0], a[1] = a[1], a[0]
        return True
    d = 2
    while (1 << d) < n:
        d += 1
    stack = []
    s, t = 0, n - 1
    d <<= 1
    while True:
        if s < t:
            d -= 1
            if d == 0:
                if not comb:
                    return False      # (the depth limit: klib goes into its comb sort)
                klib_combsort(a, s, t)
                t = s
                continue
            i, j = s, t
            k = i + ((j - i) >> 1) + 1
            if lt(a[k], a[i]):
                if lt(a[k], a[j]):
                    k = j
            else:
                k = i if lt(a[j], a[i]) else j
            rp = a[k]
            if k != t:
                a[k], a[t] = a[t], a[k]
            while True:
                i += 1
                while lt(a[i], rp):
                    i += 1
                j -= 1
                while i <= j and lt(rp, a[j]):
                    j -= 1
                if j <= i:
                    break
                a[i], a[j] = a[j], a[i]
            a[i], a[t] = a[t], a[i]
            if i - s > t - i:
                if i - s > 16:
                    stack.append((s, i - 1, d))
                s = i + 1 if t - i > 16 else t
            else:
                if t - i > 16:
                    stack.append((i + 1, t, d))
                t = i - 1 if i - s > 16 else s
        else:
            if not stack:
                for i in range(1, n):
                    j = i
                    while j > 0 and lt(a[j], a[j - 1]):
                        a[j], a[j - 1] = a[j - 1], a[j]
                        j -= 1
                return True
            s, t, d = stack.pop()


def par_partition(a, s, t):
    """positions s..t, pivot already at a[t]; returns the partition point"""
    wp = a[t][0]
    L = [p for p in range(s + 1, t + 1) if a[p][0] <= wp]          # where `do ++i while (a[i] < rp)` can stop
    R = [p for p in range(t - 1, s, -1) if a[p][0] >= wp]          # where `do --j while (rp < a[j])` can stop, from the right
    setL, setR = set(L), set(R)
    partL, partR = {}, {}
    for p in range(s + 1, t + 1):
        if p in setL:
            kL = sum(1 for q in L if q <= p)
            above = sum(1 for q in R if q > p)
            if above >= kL:
                partL[kL] = p
        if p in setR:
            kR = sum(1 for q in R if q >= p)
            below = sum(1 for q in L if q < p)
            if below >= kR:
                partR[kR] = p
    assert len(partL) == len(partR)
    K = len(partL)
    vals = list(a)
    for k in range(1, K + 1):
        a[partL[k]], a[partR[k]] = vals[partR[k]], vals[partL[k]]
    first_free = min(p for p in L if p not in partL.values())
    i = min(first_free, partR[K]) if K else first_free
    a[i], a[t] = a[t], a[i]
    return i


def par_introsort(a, comb=False):
    n = len(a)
    if n < 2:
        return True
    if n == 2:
        if lt(a[1], a[0]):
            a[0], a[1] = a[1], a[0]
        return True
    d = 2
    while (1 << d) < n:
        d += 1
    stack = []
    s, t = 0, n - 1
    d <<= 1
    while True:
        if s < t:
            d -= 1
            if d == 0:
                if not comb:
                    return False
                klib_combsort(a, s, t)    # (one lane, on the array as the partitions so far left it)
                t = s
                continue
            i, j = s, t
            k = i + ((j - i) >> 1) + 1
            if lt(a[k], a[i]):
                if lt(a[k], a[j]):
                    k = j
            else:
                k = i if lt(a[j], a[i]) else j
            if k != t:
                a[k], a[t] = a[t], a[k]
            i = par_partition(a, s, t)
            if i - s > t - i:
                if i - s > 16:
                    stack.append((s, i - 1, d))
                s = i + 1 if t - i > 16 else t
            else:
                if t - i > 16:
                    stack.append((i + 1, t, d))
                t = i - 1 if i - s > 16 else s
        else:
            if not stack:
                break
            s, t, d = stack.pop()
    # insertion pass = stable order by weight
    rank = [sum(1 for q in range(n) if a[q][0] > a[p][0] or (a[q][0] == a[p][0] and q < p)) for p in range(n)]
    out = [None] * n
    for p in range(n):
        out[rank[p]] = a[p]
    a[:] = out
    return True


def main():
    rnd = random.Random(7)
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    n_fallback = 0
    for case in range(n_cases):
        n = rnd.choice([3, 5, 16, 17, 18, 33, 40, 64, 65, 100, 128, 129, 200, 256]) if case % 3 else rnd.randint(1, 256)
        kind = rnd.randint(0, 4)
        if kind == 0:
            w = [rnd.randint(19, 22) for _ in range(n)]
        elif kind == 1:
            w = [rnd.randint(19, 22) if rnd.random() < 0.95 else rnd.randint(30, 150) for _ in range(n)]
        elif kind == 2:
            w = [rnd.randint(1, 300) for _ in range(n)]
        elif kind == 3:
            w = [20] * n
        else:
            w = sorted((rnd.randint(19, 40) for _ in range(n)), reverse=rnd.random() < 0.5)
        a = [(w[i], i) for i in range(n)]
        b = list(a)
        ra, rb = klib_introsort(a), par_introsort(b)
        assert ra == rb, (case, n, kind)
        if not ra:
            n_fallback += 1
            continue
        assert a == b, (case, n, kind, a, b)
    print("ok: %d cases, %d hit the depth limit in both" % (n_cases, n_fallback))


if __name__ == "__main__":
    main()
