#!/usr/bin/env python3
"""What bounds the LDS conflicts of the HALF-slab sweep (32-byte slab rows: eight rows per service group, eight bank windows): a
simulation of random rows of L entries over 4000 columns - LDS cycles per group of eight rows when the slice is swept for T steps
with the optimal (edge-coloured) order: T + the entries a window class holds beyond T; `fixed`: the slice's rows in their slots,
`regrouped`: the best of 64 random deals of the sixteen rows to the two groups.  (CPU only.)"""
import numpy as np

rng = np.random.default_rng(0)


def sim(length, ts, trials=200, cands=64):
    res = {t: [0, 0] for t in ts}
    for _ in range(trials):
        cols = np.stack([rng.choice(4000, length, replace=False) for _ in range(16)])
        cnt = np.stack([np.bincount(c & 7, minlength=8) for c in cols])
        parts = [np.arange(16)] + [rng.permutation(16) for _ in range(cands)]
        sums = [(cnt[p[:8]].sum(0), cnt[p[8:]].sum(0)) for p in parts]
        for t in ts:
            ov = [np.maximum(a - t, 0).sum() + np.maximum(b - t, 0).sum() for a, b in sums]
            res[t][0] += ov[0]
            res[t][1] += min(ov)
    return {t: (v[0] / trials, v[1] / trials) for t, v in res.items()}


for length in (12, 16, 21, 26, 34, 41, 51, 67):
    pieces = -(-length // 32)
    t0 = 32 * (pieces - 1) + ((length - 32 * (pieces - 1) + 3) & ~3)
    ts = [t0 + 4 * k for k in range(0, 4)]
    r = sim(length, ts)
    print(f"L={length:2d} cycles per group, fixed / regrouped: " + " | ".join(f"T={t}: {(2 * t + r[t][0]) / 2:.1f} / {(2 * t + r[t][1]) / 2:.1f}" for t in ts))
