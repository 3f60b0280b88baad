"""ORACLE side (test / measurement infrastructure, never imported by the product): the six step scalars of a synthetic sweep job
computed THE WAY THE REFERENCE COMPUTES THEM - dense n x n torch tensors on the host, `nonzero()` edge lists, scatter_add, Python
loops over the classes - restated from utils/homophily_plot.py (the "dense flavour" synthetic_plot.py:101-108 calls).  This is
the call pattern bench.py's `cpu_baseline` times ("Python-loop metrics"); oracle.py / wdg_oracle.c compute the
same numbers from one CSR pass and are what the GPU path is checked against.

Pinned: tests/test_oracle_golden.py compares every scalar with tests/golden/syn_*.npz, which the real reference produced
(tests/golden/make_golden.py).  Each block cites the reference lines it restates."""
import numpy as np
import torch


def normalised_dense_adjacency(src, dst, n):
    """synthetic_plot.py:87-92 + utils/util_funcs.py:29-36 (`normalize`): dense A + I, rows divided by their sums (inf -> 0)"""
    a = torch.zeros((n, n), dtype=torch.float32)
    a[torch.as_tensor(src), torch.as_tensor(dst)] = 1.0
    a = a + torch.eye(n)
    inv = 1.0 / a.sum(1)
    inv[torch.isinf(inv)] = 0.0
    return inv[:, None] * a


def _edge_homophily(adj, onehot):
    """utils/homophily_plot.py:43-54: binarise, drop the diagonal, fraction of the remaining entries whose endpoints share a class
    (through the dense n x n same-class indicator onehot onehot^T)"""
    pattern = (adj > 0).float()
    pattern = pattern - torch.diag(torch.diag(pattern))
    same = onehot @ onehot.t()
    return (same * pattern).sum() / pattern.sum()


def _node_homophily(adj, labels):
    """utils/homophily_plot.py:81-100: edge list from two nonzero() calls, per-node fraction of same-class entries (self loop
    included), mean over the nodes that have entries"""
    rows, cols = adj.nonzero()[:, 0], adj.nonzero()[:, 1]
    edges = torch.tensor(np.vstack((rows, cols)), dtype=torch.long).contiguous()
    deg = torch.bincount(edges[0]).float()
    hit = (labels[edges[0]] == labels[edges[1]]).float()
    per_node = torch.zeros(adj.shape[0]).scatter_add(0, edges[0], hit) / deg
    return per_node[deg != 0].mean()


def _compatibility(edge_pairs, labels):
    """utils/homophily_plot.py:103-127: class-by-class compatibility matrix, one scatter_add per class, rows normalised"""
    s, t = edge_pairs[:, 0], edge_pairs[:, 1]
    keep = (labels[s] >= 0) * (labels[t] >= 0)
    c = int(labels.max()) + 1
    h = torch.zeros((c, c))
    ls, lt = labels[s[keep]], labels[t[keep]]
    for k in range(c):
        tgt = lt[torch.where(ls == k)[0]]
        h[k].scatter_add_(src=torch.ones_like(tgt).to(h.dtype), dim=-1, index=tgt)
    return h / h.sum(1, keepdim=True)


def _class_homophily(adj, labels):
    """utils/homophily_plot.py:130-151 (`our_measure`): diagonal removed (isolated rows get a loop back), compatibility matrix of
    the nonzero() pairs, clamp(H_kk - p_k, 0) summed over the classes in a Python loop (NaN terms skipped) / (C - 1)"""
    a = adj - torch.diag(torch.diag(adj))
    a = a + torch.diag((a.sum(1) == 0).float())
    h = _compatibility(a.nonzero(), labels)
    known = labels[labels >= 0]
    prop = known.unique(return_counts=True)[1].float() / known.shape[0]
    c = int(labels.max()) + 1
    total = 0
    for k in range(c):
        term = torch.clamp(h[k, k] - prop[k], min=0)
        if not torch.isnan(term):
            total += term
    return total / (c - 1)


def _class_distribution(adj, labels):
    """utils/homophily_plot.py:154-186: degrees from the coalesced sparse copy minus the loop, the C x C class-pair counts in a
    DOUBLE Python loop of masked sums over the loop-free edge list, zeros replaced by 1e-8"""
    idx = adj.to_sparse().coalesce().indices()
    deg = idx[0].unique(return_counts=True)[1] - 1
    idx = adj.to_sparse().coalesce().indices()
    idx = idx[:, idx[0] != idx[1]]  # (remove_self_loops, :24-40)
    s, t = idx[0], idx[1]
    c = int(labels.max()) + 1
    p = labels.unique(return_counts=True)[1] / labels.shape[0]
    p_bar, pc = torch.zeros(c), torch.zeros((c, c))
    for i in range(c):
        p_bar[i] = deg[torch.where(labels == i)].sum()
        for j in range(c):
            pc[i, j] = (labels[t[torch.where(labels[s] == i)]] == j).sum()
    p_bar, pc = p_bar / deg.sum(), pc / deg.sum()
    p_bar[torch.where(p_bar == 0)], pc[torch.where(pc == 0)] = 1e-8, 1e-8
    return p, p_bar, pc


def _adjusted_homophily(adj, onehot):
    """utils/homophily_plot.py:172-177: its own class_distribution() and edge_homophily() calls"""
    _p, p_bar, _pc = _class_distribution(adj, onehot.argmax(1))
    e = _edge_homophily(adj, onehot)
    return (e - (p_bar ** 2).sum()) / (1 - (p_bar ** 2).sum())


def _label_informativeness(adj, onehot):
    """utils/homophily_plot.py:180-186: class_distribution() again"""
    _p, p_bar, pc = _class_distribution(adj, onehot.argmax(1))
    return 2 - (pc * torch.log(pc)).sum() / (p_bar * torch.log(p_bar)).sum()


def _soft_las(adj, onehot):
    """utils/homophily_plot.py:189-231 as synthetic_plot.py:106 calls it (features = label one-hot, NTK None, hard None, LP 1,
    ifsum 1): the dense n x n post-aggregation similarity (A Z)(A Z)^T - both products formed twice -, its per-class column sums
    in a Python loop, the intra / inter ratio per node, the share of nodes at >= 1 (NaN -> 0)"""
    sim = (adj @ onehot) @ (adj @ onehot).t()
    labels = onehot.argmax(1)
    c = int(labels.max()) + 1
    w = torch.zeros(adj.shape[0], c)
    for i in range(c):
        w[:, i] = sim[:, labels == i].sum(1)
    n = labels.shape[0]
    same = (onehot @ onehot.t()).sum(1)
    own = w[np.arange(n), labels]
    ratio = (own / same) / ((w.sum(1) - own) / (n - same))
    ratio[torch.isnan(ratio)] = 0
    return (ratio >= 1).float().mean()


def six_scalars(adj, onehot):
    """the step's six scalars in synthetic_plot.py:101-107's order of METRIC_NAMES: edge, node, class, adjusted homophily, label
    informativeness, soft LAS - each through its own reference-pattern routine (nothing shared between them, as in the reference)"""
    labels = onehot.argmax(1)
    return (float(_edge_homophily(adj, onehot)), float(_node_homophily(adj, labels)), float(_class_homophily(adj, labels)),
            float(_adjusted_homophily(adj, onehot)), float(_label_informativeness(adj, onehot)), float(_soft_las(adj, onehot)))


def job(src, dst, labels, x, n, n_classes):
    """one sweep job the reference's way: dense normalised A + I, the dense aggregation torch.spmm(adj, X) (utils/homophily_plot.py:246,
    what every kernel-regression call starts with), the six scalars -> (Y, scalars, stored entries of A + I)"""
    adj = normalised_dense_adjacency(src, dst, n)
    onehot = torch.eye(n_classes)[torch.as_tensor(labels)]
    y = torch.spmm(adj, x)
    return y, six_scalars(adj, onehot), int((adj > 0).sum())
