"""Helpers for reading tests/golden/*.npz (data only; produced by make_golden.py)."""
import glob
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REAL = ["cora", "citeseer", "film", "texas"]
SYN = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "syn_*.npz")))


def load(name):
    return dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))


def dense_features(g, key="feat_data"):
    n, f = int(g["n_nodes"]), int(g["n_feat"])
    x = np.zeros((n, f), np.float32)
    rr = np.repeat(np.arange(n), np.diff(g["feat_indptr"]))
    x[rr, g["feat_indices"]] = g[key]
    return x


KR_FIXTURES = SYN + ["real_texas", "real_cora", "real_cora_s200"]


def load_kr(name):
    """per-epoch kernel-regression golden of fixture `name` (make_golden_kr.py): dict(epochs, seed, sample_max and, per
    classifier, train / val node-id lists per epoch, g_results / x_results per epoch, p)"""
    z = np.load(os.path.join(GOLDEN_DIR, "kr_epochs.npz"))
    out = {"epochs": int(z["epochs"]), "seed": int(z[f"{name}/seed"]), "sample_max": float(z[f"{name}/sample_max"])}
    for clf in ("kernel_reg0", "kernel_reg1"):
        tr, va = z[f"{name}/train_{clf}"], z[f"{name}/val_{clf}"]
        out[clf] = dict(node_sets=[(t[t >= 0].astype(np.int64), v[v >= 0].astype(np.int64)) for t, v in zip(tr, va)],
                        g_results=z[f"{name}/g_results_{clf}"], x_results=z[f"{name}/x_results_{clf}"], p=float(z[f"{name}/p_{clf}"]))
    return out


def welch_p(g, x):
    """the metric's p-value from the epochs' accuracies (utils/homophily_metrics.py:335-347)"""
    from scipy.stats import ttest_ind
    g, x = np.asarray(g, np.float32), np.asarray(x, np.float32)
    _, p = ttest_ind(x, g, axis=0, equal_var=False, nan_policy="propagate")
    return float(p / 2 if np.mean((g > x).astype(np.float32)) <= 0.5 else 1 - p / 2)


def p_tolerance(g_ref, x_ref, n_val, rows, trials=400, seed=0):
    """How far the Welch p-value can move when every epoch's accuracy is off by at most `rows` validation rows: the largest
    deviation over `trials` random perturbations of that size (both result vectors) - the p-value bound that a per-epoch
    accuracy bound of rows / n_val implies for THESE accuracies (a p-value near 0 or 1 barely moves, one near 0.5 does)."""
    rng = np.random.default_rng(seed)
    g_ref, x_ref = np.asarray(g_ref, np.float64), np.asarray(x_ref, np.float64)
    p0 = welch_p(g_ref, x_ref)
    worst = 0.0
    for _ in range(trials):
        dg = rng.integers(-rows, rows + 1, g_ref.shape) / n_val
        dx = rng.integers(-rows, rows + 1, x_ref.shape) / n_val
        worst = max(worst, abs(welch_p(np.clip(g_ref + dg, 0, 1), np.clip(x_ref + dx, 0, 1)) - p0))
    return worst + 1e-9


def assert_gntk_close(got, want, want_linear, n_layers):
    """A GNTK kernel block against the reference's (utils/homophily_metrics.py:232-257), entry by entry, at the tolerance the
    reference's own formula allows:
      linear kernel G / 2 (and every arc-cosine entry whose cosine is below 0.999): rtol 2e-5 - products and sums of fp32;
      arc-cosine entries with cos = G / nu >= 0.999 (the diagonal, near-duplicate rows): acos(c) and sqrt(nu^2 - G^2) are evaluated
      where d acos / dc = 1 / sqrt(1 - c^2) blows up - a rounding of c by eps moves acos by sqrt(2 eps) = 4.9e-4 and the entry by
      G sqrt(2 eps) / (2 pi), 1.6e-4 of its value; the reference's own entries carry that error (against fp64), so nothing
      tighter than 2e-4 can be asked THERE - and is asked nowhere else (round 5 applied 2e-4 to every entry of both kernels).
    want_linear: the golden linear block of the same matrix (its cosines locate the ill-conditioned entries)."""
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    scale = np.abs(want).max()
    tol = np.full(want.shape, 2e-5)
    if n_layers == 1:
        d = np.sqrt(np.maximum(np.diag(np.asarray(want_linear, np.float64)), 1e-300))
        cos = np.asarray(want_linear, np.float64) / (d[:, None] * d[None, :])
        tol[cos >= 0.999] = 2e-4
        assert (cos >= 0.999).mean() < 0.05  # (the loose bound applies to a sliver of the block: the diagonal and near-duplicates)
    err = np.abs(got - want)
    bad = err > tol * np.abs(want) + 2e-6 * scale
    assert not bad.any(), (int(bad.sum()), float((err / np.maximum(np.abs(want), 1e-300))[bad].max()))
