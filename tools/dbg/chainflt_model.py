#!/usr/bin/env python3
"""Model of the chain filter as the LDS tiers run it (k_regions.hip, stage D of rg_task): mem_chain_flt's overlap loop (memchain.c:426-482) with a lane
per candidate chain, the test in integers (thresholds T and D per chain), the chains nobody can drop entered up front and nothing tested at all
when that is every chain.  Input: the chains of a strand search in the order klib's sort left them (begin, end, weight, alt).  Output: which
survive.  `sequential` is the reference's loop as written, in float where the reference computes in float; tests/test_chainflt_model.py holds
the two against each other."""
import math
import numpy as np

f32 = np.float32


def sequential(opt, ch):
    """memchain.c:426-482 on chains already sorted by weight; returns the kept flag of every chain"""
    n = len(ch)
    kept = [0] * n
    first = [-1] * n
    if n == 0:
        return kept
    kept[0] = 3
    to_keep = [0]
    for i in range(1, n):
        bi, ei, wi, ai = ch[i]
        large = 0
        broke = False
        for k in to_keep:
            bk, ek, wk, ak = ch[k]
            b_max, e_min = max(bk, bi), min(ek, ei)
            if e_min > b_max and (not ak or ai):
                li, lj = ei - bi, ek - bk
                min_l = min(li, lj)
                if f32(e_min - b_max) >= f32(min_l) * f32(opt["mask_level"]) and min_l < opt["max_chain_gap"]:
                    large = 1
                    if first[k] < 0:
                        first[k] = i
                    if f32(wi) < f32(wk) * f32(opt["drop_ratio"]) and wk - wi >= opt["min_seed_len"] << 1:
                        broke = True
                        break
        if not broke:
            to_keep.append(i)
            kept[i] = 2 if large else 3
    for k in to_keep:
        if first[k] >= 0:
            kept[first[k]] = 1
    i = k = 0
    while i < n:
        if not (kept[i] == 0 or kept[i] == 3):
            k += 1
            if k >= opt["max_chain_extend"]:
                break
        i += 1
    while i < n:
        if kept[i] < 3:
            kept[i] = 0
        i += 1
    return kept


def flt_T(opt, l):      # smallest overlap that is "significant" for a chain of query length l; 0xffff: never
    if not l < opt["max_chain_gap"]:
        return 0xffff
    t = int(math.ceil(float(f32(l) * f32(opt["mask_level"]))))
    return max(t, 1)


def flt_D(opt, w):      # a chain is dropped by a kept chain of weight w (that it overlaps significantly) iff its own weight is below this
    d = int(math.ceil(float(f32(w) * f32(opt["drop_ratio"]))))
    e = w - (opt["min_seed_len"] << 1) + 1
    d = min(d, e)
    return 0 if d < 0 else min(d, 0x7fff)


def wave(opt, ch):
    """the device's order of work; returns the kept flags (1, 2, 3 as the reference has them except where the kernel says they cannot matter)"""
    n = len(ch)
    if n == 0:
        return []
    T = [flt_T(opt, e - b) for b, e, w, a in ch]
    D = [flt_D(opt, w) for b, e, w, a in ch]
    # the prefix nobody can drop
    d0 = D[0]
    U = n
    for i in range(n):
        if ch[i][2] < d0:
            U = i
            break
    kept_list = list(range(U))          # entries of the kept list = sorted positions
    state = {i: 3 for i in range(U)}    # kept list entry -> 2 / 3
    first = {i: -1 for i in range(U)}
    all_kept = U == n and opt["max_chain_extend"] >= n

    def hit(i, k):
        bi, ei, wi, ai = ch[i]
        bk, ek, wk, ak = ch[k]
        ov = min(ek, ei) - max(bk, bi)
        h = ov >= min(T[i], T[k])
        if ak and not ai:
            h = False
        return h

    if not all_kept:
        for base in range(0, n, 64):
            lanes = [i for i in range(base, min(base + 64, n)) if i >= 1]
            live = {i: True for i in lanes}
            large = {i: False for i in lanes}
            pre = {i: i < U for i in lanes}
            k_end = base + 64 if base + 64 <= U else len(kept_list)
            for kk in range(k_end):                      # the list as it stands
                k = kept_list[kk]
                hits = [i for i in lanes if live[i] and (not pre[i] or kk < i) and hit(i, k)]
                for i in hits:
                    large[i] = True
                if hits:
                    if first[k] < 0:
                        first[k] = min(hits)
                    for i in hits:
                        if ch[i][2] < D[k]:
                            live[i] = False
            for i in lanes:
                if pre[i] and live[i] and large[i]:
                    state[i] = 2
            um = [i for i in lanes if live[i] and not pre[i]]
            while um:
                l = um[0]                                  # the next kept chain
                hits = [i for i in lanes if live[i] and i > l and hit(i, l)]
                for i in hits:
                    large[i] = True
                dropped = [i for i in hits if ch[i][2] < D[l]]
                for i in dropped:
                    live[i] = False
                kept_list.append(l)
                state[l] = 2 if large[l] else 3
                first[l] = min(hits) if hits else -1
                um = [i for i in um if i != l and i not in dropped]
    kept = [0] * n
    for k in kept_list:
        kept[k] = state[k]
    for k in kept_list:
        if first[k] >= 0:
            kept[first[k]] = 1
    if opt["max_chain_extend"] < n:
        i = k = 0
        while i < n:
            if not (kept[i] == 0 or kept[i] == 3):
                k += 1
                if k >= opt["max_chain_extend"]:
                    break
            i += 1
        while i < n:
            if kept[i] < 3:
                kept[i] = 0
            i += 1
    return kept
