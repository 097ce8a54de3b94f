"""CPU: the chain filter in the form the device runs it (a lane per candidate chain, integer thresholds, the undroppable prefix, nothing tested when
every chain is kept: tools/dbg/chainflt_model.py, which mirrors k_regions.hip stage D step by step) keeps exactly the chains mem_chain_flt keeps
(memchain.c:426-482 as written, float where the reference is float), and the same kept values wherever they can matter (max_chain_extend set)."""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools", "dbg"))
import chainflt_model as M   # noqa: E402


def _opt(rnd, kind):
    o = {"mask_level": 0.5, "drop_ratio": 0.5, "min_seed_len": 19, "max_chain_gap": 10000, "max_chain_extend": 1 << 30}
    if kind == 1:
        o.update(mask_level=rnd.choice([0.3, 0.5, 0.65, 0.8, 0.95, 0.0]), drop_ratio=rnd.choice([0.3, 0.5, 0.7, 0.9, 1.2]), min_seed_len=rnd.choice([10, 19, 25]))
    elif kind == 2:
        o.update(max_chain_gap=rnd.choice([30, 60, 100, 200]))
    elif kind == 3:
        o.update(max_chain_extend=rnd.choice([1, 2, 3, 5, 20]))
    return o


def _chains(rnd, l_query, shape):
    n = rnd.choice([1, 2, 3, 10, 63, 64, 65, 100, 128, 129, 200, 256])
    ch = []
    for _ in range(n):
        if shape == 0 or (shape == 1 and rnd.random() < 0.93):      # chance matches of a 3-letter 19-mer
            ln = rnd.randint(19, 22)
            b = rnd.randint(0, l_query - ln)
            ch.append((b, b + ln, ln, 0))
        elif shape == 2:                                            # a repeat family: many chains of similar, large weight
            ln = rnd.randint(60, l_query)
            b = rnd.randint(0, l_query - ln)
            ch.append((b, b + ln, rnd.randint(ln // 2, ln), 1 if rnd.random() < 0.1 else 0))
        else:                                                       # the read's own chain and relatives
            ln = rnd.randint(40, l_query)
            b = rnd.randint(0, l_query - ln)
            ch.append((b, b + ln, rnd.randint(30, ln), 1 if rnd.random() < 0.2 else 0))
    ch.sort(key=lambda c: -c[2])      # any order among equal weights is a possible outcome of the sort: the filter takes what it gets
    return ch


def test_wave_form_keeps_what_the_reference_keeps():
    rnd = random.Random(5)
    n_all_kept = n_some = 0
    for case in range(2500):
        opt = _opt(rnd, case % 4)
        ch = _chains(rnd, rnd.choice([100, 150, 250]), rnd.randint(0, 3))
        a, b = M.sequential(opt, ch), M.wave(opt, ch)
        assert [x != 0 for x in a] == [x != 0 for x in b], (case, opt, ch)
        if opt["max_chain_extend"] < len(ch):
            assert a == b, (case, opt, ch)            # the kept values decide who survives there: they have to be the reference's
        if all(a):
            n_all_kept += 1
        else:
            n_some += 1
    assert n_all_kept > 100 and n_some > 100


def test_integer_thresholds_are_the_float_tests():
    """overlap >= T(min_l) <=> (float)overlap >= (float)min_l * mask_level, and w_i < D(w_k) <=> the drop rule, over every value the kernel can see"""
    import numpy as np
    f32 = np.float32
    for ml in (0.0, 0.1, 0.3, 0.5, 0.65, 0.8, 0.95, 1.0):
        for gap in (50, 10000):
            opt = {"mask_level": ml, "drop_ratio": 0.5, "min_seed_len": 19, "max_chain_gap": gap}
            for l in range(1, 300):
                t = M.flt_T(opt, l)
                for ov in range(-3, l + 3):
                    want = ov > 0 and f32(ov) >= f32(l) * f32(ml) and l < gap
                    assert (ov >= t) == bool(want), (ml, gap, l, ov, t)
            # T of the shorter chain is the smaller T
            ts = [M.flt_T(opt, l) for l in range(1, 300)]
            assert ts == sorted(ts)
    for dr in (0.3, 0.5, 0.7, 0.9, 1.0, 1.2):
        for msl in (10, 19, 25):
            opt = {"mask_level": 0.5, "drop_ratio": dr, "min_seed_len": msl, "max_chain_gap": 10000}
            ds = []
            for wk in range(1, 400):
                d = M.flt_D(opt, wk)
                ds.append(d)
                for wi in range(1, wk + 1):
                    want = f32(wi) < f32(wk) * f32(dr) and wk - wi >= msl << 1
                    assert (wi < d) == bool(want), (dr, msl, wk, wi, d)
            assert ds == sorted(ds)     # a lighter kept chain drops no more than a heavier one: the undroppable chains are a prefix
