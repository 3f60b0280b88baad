"""Synthetic inputs shaped like the reference's `data_synthesis/` files.

The reference ships pre-generated graphs (`data_synthesis/{800,4000}/{h}/adj_{h}_{s}.pt`); the generating rule,
verified on those files, is restated here so that benchmarks and tests need no data files:

  * N nodes, C equal contiguous classes (label = node // (N/C));
  * every node has exactly `k` distinct same-class out-neighbours (never itself) and `int(k/h) - k` distinct
    out-neighbours drawn uniformly from the other classes; directed, no duplicates, no self loops;
  * `data_synthesis/800` is k=2, `data_synthesis/4000` is k=10 (the directory name is k * 400, synthetic_plot.py:22);
  * features: dense fp32 [N,F], ~10 % non-zeros, rows L1-normalised (pubmed-sample statistics).
Host-side numpy (numpy PCG64 seeded by (seed, h)); the graphs are inputs, not part of the timed hot path.
"""
import numpy as np

# the 30 levels of synthetic_plot.py:19-20 and the 10-level subset used by BASELINE configs[1]/[2]
H_LEVELS_30 = [0.05, 0.1, 0.15, 0.16, 0.165, 0.17, 0.175, 0.18, 0.185, 0.19, 0.195, 0.2, 0.21, 0.22, 0.23, 0.24,
               0.25, 0.3, 0.35, 0.4, 0.45, 0.5, 0.55, 0.6, 0.65, 0.7, 0.75, 0.8, 0.85, 0.9]
H_LEVELS_10 = [0.05, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9]
H_LEVELS_10_K10 = [0.15, 0.2, 0.25, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9]  # `4000` set lacks 0.05 / 0.1 (Appendix B.2)


def out_degree(k, h):
    return int(k / h)


def regular_graph(n, n_classes, k, h, seed):
    """-> (src int64[E], dst int64[E], labels int64[n]) with E = n * int(k/h), rows sorted by column."""
    assert n % n_classes == 0
    m = n // n_classes
    d = out_degree(k, h)
    n_other = d - k
    assert 0 < k <= m - 1 and 0 <= n_other <= n - m
    rng = np.random.default_rng([int(seed), int(round(h * 1000)), n, k])
    labels = np.arange(n) // m
    node = np.arange(n)
    # k same-class neighbours: smallest random keys among the m-1 candidates
    keys = rng.random((n, m))
    keys[node, node % m] = np.inf
    same = np.argpartition(keys, k - 1, axis=1)[:, :k] + (labels * m)[:, None]
    if n_other > 0:
        keys = rng.random((n, n - m))
        pick = np.argpartition(keys, n_other - 1, axis=1)[:, :n_other] if n_other < n - m else np.tile(np.arange(n - m), (n, 1))
        other = pick + (pick >= (labels * m)[:, None]) * m  # skip the node's own class block
        dst = np.concatenate([same, other], axis=1)
    else:
        dst = same
    dst = np.sort(dst, axis=1)
    src = np.repeat(node, d)
    return src.astype(np.int64), dst.reshape(-1).astype(np.int64), labels.astype(np.int64)


# Share of the rows that are exact copies of another row of the same class (0: none).  The reference's synthetic features are real
# nodes' rows sampled per class WITH replacement: its pubmed sample holds 58 - 67 duplicate pairs per 2000 rows (3.3 %; tests/golden/
# syn_*.npz), enough to put a duplicate pair into 39 % of the 300-row train blocks of the kernel-regression metric.  bench.py sets
# it to that share (--dup-frac), so that the deflation those blocks need is inside every nine-scalar figure it prints.
DUPLICATE_FRACTION = 0.0


def features(n, f, seed, density=0.1, labels=None, signal=1.0, duplicates=None, n_classes=5):
    """Row-L1-normalised sparse-ish dense features, fp32 [n, f].

    duplicates (default DUPLICATE_FRACTION): that share of the rows copy another row of their class (contiguous classes of n /
    n_classes nodes, the generator's labels) - the reference's sampling with replacement.

    With `labels`, the features carry class information the way the reference's do (its synthetic features are
    sampled from real nodes of the same class): class c switches features of "its" block of columns on with
    twice the base density, the others with less, keeping the mean density."""
    rng = np.random.default_rng([int(seed), n, f, 77])
    x = rng.random((n, f), dtype=np.float32) * np.float32(0.33)
    dens = np.full((n, f), density, np.float32)
    if labels is not None:
        c = int(np.max(labels)) + 1
        block = (np.arange(f) * c // f)[None, :] == np.asarray(labels)[:, None]
        dens = np.where(block, density * (1 + signal), density * (1 - signal / max(c - 1, 1))).astype(np.float32)
    x *= rng.random((n, f), dtype=np.float32) < dens
    x[np.arange(n), rng.integers(0, f, n)] += np.float32(0.01)  # no empty rows
    x = (x / x.sum(1, keepdims=True)).astype(np.float32)
    dup = DUPLICATE_FRACTION if duplicates is None else duplicates
    if dup > 0 and n >= 2 * n_classes:
        rng2 = np.random.default_rng([int(seed), n, f, 78])  # (its own stream: the rows above do not depend on the share)
        m = n // n_classes
        rows = rng2.choice(n, int(round(dup * n)), replace=False)
        block = np.minimum(rows // m, n_classes - 1)
        source = block * m + rng2.integers(0, m, rows.shape[0])
        keep = (source != rows) & ~np.isin(source, rows)  # (copies of originals only, every original copied once: classes of two,
        keep[np.setdiff1d(np.arange(rows.shape[0]), np.unique(source, return_index=True)[1])] = False  # as in the pubmed sample)
        x[rows[keep]] = x[source[keep]]
    return x


def random_graph(n, n_edges, seed, power_law=False):
    """Undirected-pair generator for the twitch-scale config C5 (csv absent from the reference checkout)."""
    rng = np.random.default_rng([int(seed), n, n_edges])
    if power_law:
        w = 1.0 / np.arange(1, n + 1) ** 0.8
        w /= w.sum()
        src = rng.choice(n, n_edges, p=w)
    else:
        src = rng.integers(0, n, n_edges)
    dst = rng.integers(0, n, n_edges)
    return src.astype(np.int64), dst.astype(np.int64)
