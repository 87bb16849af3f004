"""The share rule of the LSH pipeline (csrc/fs_lsh.hip) as arithmetic, on the
CPU: the inequality its two skips rest on, and the two properties of the gate's subsets --
on random tables with norms spread by a factor of thirty.  (That the kernels apply it to the
reference's records without changing them is tests/test_gpu_realistic_table.py and
tools/stress_share.py, on the GPU.)"""

import itertools

import numpy as np
import pytest


def _table(rng, rows, dim, groups):
    base = rng.standard_normal((groups, dim))
    emb = base[rng.integers(0, groups, size=rows)] + 0.45 * rng.standard_normal((rows, dim))
    emb /= np.linalg.norm(emb, axis=1, keepdims=True)
    emb *= rng.lognormal(np.log(6.0), 0.8, size=(rows, 1))
    emb[rng.integers(0, rows, size=3)] = 0.0
    return emb


def _components(emb, gamma):
    n = np.linalg.norm(emb, axis=1)
    unit = np.divide(emb, n[:, None], out=np.zeros_like(emb), where=n[:, None] > 0)
    near = (unit @ unit.T > gamma) & (n[:, None] > 0) & (n[None, :] > 0)
    comp = np.arange(len(emb))
    changed = True
    while changed:                                   # label propagation to the smallest member
        new = np.where(near, comp[None, :], len(emb)).min(axis=1)
        new = np.minimum(new, comp)
        changed = bool((new != comp).any())
        comp = new
    return comp


@pytest.mark.parametrize("gamma", [0.5, 0.7, 0.85])
def test_cosine_of_two_windows_is_bounded_by_the_shares_of_their_far_slots(gamma):
    rng = np.random.default_rng(3)
    emb = _table(rng, 300, 24, 40)
    comp = _components(emb, gamma)
    q = (emb ** 2).sum(axis=1)
    n = 6
    worst = -1.0
    for _ in range(4000):
        f = rng.integers(0, len(emb), size=n)
        s = f.copy()
        # a script window related to the fan window: some slots the same word, some a word of the
        # same component, some anything
        for k in range(n):
            r = rng.random()
            if r < 0.35:
                mates = np.nonzero(comp == comp[f[k]])[0]
                s[k] = mates[rng.integers(0, len(mates))]
            elif r < 0.6:
                s[k] = rng.integers(0, len(emb))
        ff, ss = q[f].sum(), q[s].sum()
        if ff == 0 or ss == 0:
            continue
        cos = (emb[f] * emb[s]).sum() / np.sqrt(ff * ss)
        far = comp[f] != comp[s]
        A, B = q[f][far].sum() / ff, q[s][far].sum() / ss
        bound = np.sqrt((1 - A) * (1 - B)) + gamma * np.sqrt(A * B)
        assert cos <= bound + 1e-9
        assert bound <= np.sqrt(1 - A * (1 - gamma ** 2)) + 1e-12      # ... and that by the one-sided form
        worst = max(worst, cos - bound)
    assert worst > -0.5                               # (the bound is met from close by: identical windows)


def test_every_heavy_set_holds_a_minimal_heavy_subset_the_gate_asks_for():
    """share_asks: with squared norms as integers qi = floor(q * scale), a subset M is heavy when
    sum_M qi >= thr, thr = floor(lim * sum qi) - N - 2; it is asked for when no slot can go
    (sum_M qi - min_M qi < thr).  (1) Every subset that holds the share `lim` of the real squared
    norms is heavy in integers; (2) every heavy set contains a subset that is asked for."""
    rng = np.random.default_rng(5)
    n, lim = 6, np.float32((1 - 0.37255) * (1 - 1e-6))
    subsets = [m for m in range(1, 1 << n)]
    for _ in range(3000):
        q = rng.lognormal(np.log(30.0), 1.0, size=n) * (rng.random(n) > 0.05)
        scale = 2.0 ** 20 / max(q.max() * 1.7, 3.0)
        qi = np.floor(q * scale).astype(np.int64)
        thr = int(np.floor(np.float32(lim) * np.float32(qi.sum()))) - n - 2
        if thr <= 0:
            continue
        bits = lambda m: [k for k in range(n) if m >> k & 1]
        heavy = {m for m in subsets if qi[bits(m)].sum() >= thr}
        asked = {m for m in heavy if qi[bits(m)].sum() - qi[bits(m)].min() < thr}
        for m in subsets:
            if q[bits(m)].sum() >= float(lim) * q.sum():
                assert m in heavy                                                   # (1)
        for m in heavy:
            assert any(a & m == a for a in asked)                                   # (2)
        assert len(asked) <= 20                       # an antichain of subsets of six slots (kEnumCap)


def test_signatures_never_make_a_slot_far_that_is_not():
    """The pairs' test compares a few bits of a hash of the component ids: equal components have
    equal signatures (a far slot is never invented); restated from csrc/fs_hash.h."""
    def mul24(a, m):
        return ((a & 0xFFFFFF) * m) & 0xFFFFFFFF

    def rotl(x, r):
        return ((x << r) | (x >> (32 - r))) & 0xFFFFFFFF

    def sig(comp, n):
        x = comp ^ (comp >> 16)
        mixed = mul24(x, 0x9E3779) ^ rotl(mul24(x >> 8, 0x85EBCB), 16)
        b = min(10, 64 // n)
        return (mixed >> 7) & ((1 << b) - 1)

    seen = {}
    for n in (3, 6, 8, 12):
        values = {sig(c, n) for c in range(20000)}
        assert len(values) > (1 << min(10, 64 // n)) * 0.9      # ... and the bits are used
        seen[n] = values
    assert sig(123, 6) == sig(123, 6)
