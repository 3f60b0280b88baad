#!/usr/bin/env python3
"""Per-epoch golden vectors of the classifier-based performance metric (kernel regression), from the REAL reference.

Companion of make_golden.py (same stubs, same container-only rule: /root/reference is imported here and nowhere else).
The reference's `classifier_based_performance_metric` (utils/homophily_plot.py:271-368, utils/homophily_metrics.py:260-349)
returns one p-value; what pins a solver is what happens inside each epoch.  The call is run unchanged with two names in the
reference MODULE's namespace wrapped by recorders:

  * `random_disassortative_splits` - every mask it returns is kept: per epoch the class-balanced sample of the nodes (when
    nnodes > sample_max) and the train / val / test masks over that sample;
  * `ttest_ind` - its two arguments are the epochs' accuracies `X_results`, `G_results`.

Stored per (fixture, classifier): the train / validation NODE ids of every epoch (ascending, -1 padded), the accuracies of
the graph-aware and the feature-only regression per epoch, the p-value.  Data only - no reference source text.

    python tests/golden/make_golden_kr.py     # -> tests/golden/kr_epochs.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (stubs + reference import)

EPOCHS = 8
SYN = (("800", 0.5, 0), ("800", 0.05, 0), ("800", 0.9, 1), ("4000", 0.2, 0), ("4000", 0.15, 2))
# (tag, dataset, sample_max, seed): texas <= sample_max -> every node in every epoch; cora at the CLI test's 200 and at the
# sweep's 500 (300 train rows)
REAL = (("real_texas", "texas", 200.0, 11), ("real_cora", "cora", 500.0, 11), ("real_cora_s200", "cora", 200.0, 11))


class Recorder:
    """wraps the two names inside a reference module for the duration of one call"""

    def __init__(self, module):
        self.module = module
        self.masks, self.results = [], None

    def __enter__(self):
        self._splits, self._ttest = self.module.random_disassortative_splits, self.module.ttest_ind

        def splits(*a, **kw):
            out = self._splits(*a, **kw)
            self.masks.append(tuple(m.cpu().numpy().copy() for m in out))
            return out

        def ttest(x, g, *a, **kw):
            self.results = (np.asarray(x, np.float64).copy(), np.asarray(g, np.float64).copy())
            return self._ttest(x, g, *a, **kw)

        self.module.random_disassortative_splits, self.module.ttest_ind = splits, ttest
        return self

    def __exit__(self, *exc):
        self.module.random_disassortative_splits, self.module.ttest_ind = self._splits, self._ttest


def node_sets(masks, n_nodes, sampled, epochs):
    """the recorded masks -> per epoch (train node ids, validation node ids), ascending"""
    per_epoch = 2 if sampled else 1
    assert len(masks) == per_epoch * epochs, (len(masks), per_epoch, epochs)
    out = []
    for e in range(epochs):
        if sampled:
            sample = np.flatnonzero(masks[2 * e][0])
            tr, va, te = masks[2 * e + 1]
        else:
            sample = np.arange(n_nodes)
            tr, va, te = masks[e]
        out.append((sample[np.flatnonzero(tr)], sample[np.flatnonzero(va | te)]))
    return out


def pad(rows):
    w = max(len(r) for r in rows)
    return np.stack([np.concatenate([r, np.full(w - len(r), -1)]) for r in rows]).astype(np.int32)


def record(module, call, n_nodes, sample_max, seed, epochs):
    out = {}
    for ci, clf in enumerate(("kernel_reg0", "kernel_reg1")):
        torch.manual_seed(seed)
        with Recorder(module) as rec:
            p = call(clf)
        p = float(p[0] if isinstance(p, tuple) else p)
        sets = node_sets(rec.masks, n_nodes, n_nodes > sample_max, epochs)
        out[f"train_{clf}"] = pad([s[0] for s in sets])
        out[f"val_{clf}"] = pad([s[1] for s in sets])
        out[f"x_results_{clf}"], out[f"g_results_{clf}"] = rec.results
        out[f"p_{clf}"] = np.float64(p)
        print(f"    {clf}: p = {p:.6f}  G = {np.round(rec.results[1], 4)}  X = {np.round(rec.results[0], 4)}")
    return out


def main():
    uf, hm, hp = mg._import_reference()
    torch.set_num_threads(8)
    out = {"epochs": np.int64(EPOCHS)}
    for k_dir, h, seed in SYN:  # synthetic_plot.py:81-101 (dense flavour, sample_max 500); seed 5 as in make_golden.py
        tag = f"syn_{k_dir}_{h}_{seed}"
        print(f"[golden-kr] {tag}")
        feats = torch.load(f"./data_synthesis/features/pubmed/pubmed_{seed}.pt").clone().detach().float()
        features = torch.tensor(uf.preprocess_features(feats)).clone().detach()
        adj_raw = torch.load(f"./data_synthesis/{k_dir}/{h}/adj_{h}_{seed}.pt").coalesce().to_dense().clone().detach().float()
        label = torch.load(f"./data_synthesis/{k_dir}/{h}/label_{h}_{seed}.pt").to_dense().clone().detach().float()
        n = adj_raw.shape[0]
        adj = torch.tensor(uf.normalize(adj_raw + torch.eye(n)))
        lab = torch.argmax(label, 1)
        rec = record(hp, lambda clf: hp.classifier_based_performance_metric(features, adj, lab, sample_max=500, base_classifier=clf,
                                                                            epochs=EPOCHS), n, 500, 5, EPOCHS)
        out.update({f"{tag}/{k}": v for k, v in rec.items()})
        out[f"{tag}/seed"], out[f"{tag}/sample_max"] = np.int64(5), np.float64(500)
    for tag, name, sample_max, seed in REAL:  # homophily_tests.py:133-137 (sparse flavour, raw adjacency, raw features)
        print(f"[golden-kr] {tag}")
        adj_raw, features, labels = uf.full_load_data_large(name)
        adj_raw = adj_raw.coalesce()
        n = labels.shape[0]
        rec = record(hm, lambda clf: hm.classifier_based_performance_metric(features, adj_raw, labels, sample_max, base_classifier=clf,
                                                                            epochs=EPOCHS), n, sample_max, seed, EPOCHS)
        out.update({f"{tag}/{k}": v for k, v in rec.items()})
        out[f"{tag}/seed"], out[f"{tag}/sample_max"] = np.int64(seed), np.float64(sample_max)
    np.savez_compressed(os.path.join(HERE, "kr_epochs.npz"), **out)
    print("wrote kr_epochs.npz:", os.path.getsize(os.path.join(HERE, "kr_epochs.npz")), "bytes")


if __name__ == "__main__":
    main()
