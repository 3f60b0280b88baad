"""MI355X-backed twin of the reference's `utils/homophily_plot.py` (dense-adjacency flavour used by the
synthetic sweep, synthetic_plot.py:94-109).

Signatures follow the reference: `adj` / `A` are DENSE [N,N] tensors (here: anything `CsrGraph.from_any`
accepts - a dense tensor is turned into the CSR of its non-zeros by csrc/graph_build.hip), `label` is one-hot
[N,C] where the reference wants one-hot, an integer vector where it wants integers.  Differences from the
sparse flavour are kept: the diagonal is removed for edge homophily, kept for node
homophily, isolated rows get a self loop in `our_measure`, `idx_train` is an INDEX vector (Q6), and the
classifier metric returns the p-value only.
"""
import math
import random

import numpy as np
import torch

from .. import ops
from ..ops import CsrGraph
from . import _tails
from . import homophily_metrics as _hm
from .util_funcs import random_disassortative_splits  # noqa: F401

pi = math.pi
device = torch.device("cuda:0" if torch.cuda.is_available() else "cpu")  # reference: utils/homophily_plot.py:15-19

remove_self_loops = _hm.remove_self_loops


def _graph(a):
    return CsrGraph.from_any(a)


def _int_labels(label, dev):
    label = label.to(dev)
    return torch.argmax(label, 1) if label.dim() == 2 else label


def edge_homophily(adj, label):
    """sum(label label^T * (adj>0 minus diag)) / sum(adj>0 minus diag).  reference: utils/homophily_plot.py:43-53."""
    g = _graph(adj)
    st = ops.edge_label_stats(g, _int_labels(label, g.device), per_row=False)
    return _tails.edge_homophily_noself(st)


def generalized_edge_homophily(adj, features, label, sample_max=20000, iteration=100):
    """reference: utils/homophily_plot.py:56-78.  Below `sample_max` NODES: mean cosine over non-loop entries.

    The reference's sampling branch re-slices `adj` in place every iteration (so it only works for one
    iteration); here every iteration samples nodes from the original graph."""
    g = _graph(adj)
    nnodes = label.shape[0]
    if nnodes < sample_max:
        total, cnt, _ = _hm._edge_cosine_sum(g, features)
        return total / cnt.float()
    vals = np.zeros(iteration)
    rows, cols = g.row_indices(), g.col.to(torch.int64)
    for i in range(iteration):
        smp = torch.tensor(random.sample(list(np.arange(nnodes)), int(sample_max)), device=g.device)
        inside = torch.zeros(nnodes, dtype=torch.bool, device=g.device)
        inside[smp] = True
        ent = torch.nonzero(inside[rows] & inside[cols] & (rows != cols)).view(-1)
        total, _, _ = _hm._edge_cosine_sum(g, features, ent)
        vals[i] = float(total) / max(int(ent.shape[0]), 1)
    return np.mean(vals)


def node_homophily(A, labels):
    """Self loops are KEPT here (A.nonzero()).  reference: utils/homophily_plot.py:81-99."""
    g = _graph(A)
    st = ops.edge_label_stats(g, _int_labels(labels, g.device))
    return _tails.node_homophily_withself(st)


def node_homophily_edge_idx(edge_index, labels, num_nodes):
    """reference: utils/homophily_plot.py:92-99 (no loop removal in this flavour)."""
    g = CsrGraph.from_coo(edge_index[0], edge_index[1], num_nodes, None, ops.COO_KEEP_DUPLICATES)
    st = ops.edge_label_stats(g, _int_labels(labels, g.device))
    return _tails.node_homophily_withself(st)


def _compat_with_isolated(g, lab):
    st = ops.edge_label_stats(g, lab)
    compat = st["compat"].clone()
    iso = (st["row_nnz_noself"] == 0) & (lab >= 0)  # A + diag(rowsum(A - diag) == 0): utils/homophily_plot.py:131-132
    if bool(iso.any()):
        c = st["n_classes"]
        li = lab[iso].to(torch.int64)
        compat.view(-1).index_add_(0, li * c + li, torch.ones_like(li))
    return compat


def compact_matrix_edge_idx(edge_index, labels):
    """reference: utils/homophily_plot.py:102-123; `edge_index` is [E,2] (rows of A.nonzero()), loops included."""
    labels = labels.squeeze()
    g = CsrGraph.from_coo(edge_index[:, 0], edge_index[:, 1], labels.shape[0], None, ops.COO_KEEP_DUPLICATES)
    lab = _int_labels(labels, g.device)
    st = ops.edge_label_stats(g, lab, per_row=True)
    compat = st["compat"].clone()
    # loops are NOT removed by this function in the plot flavour: add them back to the diagonal cells
    loops = (st["row_nnz"] - st["row_nnz_noself"]).to(torch.int64)
    ok = lab >= 0
    c = st["n_classes"]
    li = lab[ok].to(torch.int64)
    compat.view(-1).index_add_(0, li * c + li, loops[ok])
    return _tails.compat_matrix(compat)


def our_measure(A, label):
    """Class homophily on a dense adjacency: diagonal removed, isolated rows get a self loop.
    reference: utils/homophily_plot.py:126-147."""
    g = _graph(A)
    lab = _int_labels(label.squeeze(), g.device)
    return _tails.class_homophily(_compat_with_isolated(g, lab), lab)


def class_distribution(A, labels):
    """reference: utils/homophily_plot.py:150-172."""
    g = _graph(A)
    lab = _int_labels(labels, g.device)
    st = ops.edge_label_stats(g, lab, per_row=False)
    return _tails.class_distribution(st, lab)


def adjusted_homo(A, label):
    """reference: utils/homophily_plot.py:175-180 (edge term WITHOUT loops in this flavour)."""
    g = _graph(A)
    lab = _int_labels(label, g.device)
    st = ops.edge_label_stats(g, lab, per_row=False)
    _, p_bar, _ = _tails.class_distribution(st, lab)
    return _tails.adjusted(_tails.edge_homophily_noself(st), p_bar)


def label_informativeness(A, label):
    """reference: utils/homophily_plot.py:183-186."""
    g = _graph(A)
    lab = _int_labels(label, g.device)
    st = ops.edge_label_stats(g, lab, per_row=False)
    _, p_bar, pc = _tails.class_distribution(st, lab)
    return _tails.label_informativeness(p_bar, pc)


def similarity(features, adj, label, NTK=None, hard=None, LP=1, ifsum=1, idx_train=None):
    """reference: utils/homophily_plot.py:189-235.  `idx_train` is an index vector in this flavour (Q6)."""
    if NTK:
        g = _graph(adj)
        dev = g.device
        x = features.to(dev, torch.float32)
        ip = torch.clamp(ops.gemm(x, x, transb=True), 0, 1)
        ip = (ip * (torch.pi - torch.acos(ip))) / (2 * torch.pi)
        left = ops.spmm(g, ip)                      # adj @ K
        inner = ops.spmm(g, left.t().contiguous())  # adj @ (adj K)^T = (adj K adj^T)^T, symmetric
        label = label.to(dev)
        labels = torch.argmax(label, 1)
        if idx_train is not None:
            ix = torch.as_tensor(idx_train, device=dev).long()
            labels, label, inner = labels[ix], label[ix], inner[ix][:, ix]
        c = int(labels.max().item()) + 1
        w = torch.stack([inner[:, labels == i].sum(1) for i in range(c)], 1).double()
        return _tails.las_from_weights(w, labels, label[:, :c], hard, LP, ifsum)
    mask = None
    if idx_train is not None:
        mask = torch.as_tensor(idx_train).to(torch.int32) if torch.as_tensor(idx_train).dtype != torch.bool else idx_train
    return _hm.similarity(features, adj, label, hard=hard, LP=LP, ifsum=ifsum, idx_train=mask)


def gntk_homophily_(features, adj, sample, n_layers):
    """reference: utils/homophily_plot.py:238-268."""
    return _hm.gntk_homophily_(features, adj, sample, n_layers)


def classifier_based_performance_metric(features, adj, labels, sample_max, rcond=1e-15, base_classifier='kernel_reg1',
                                        epochs=100):
    """reference: utils/homophily_plot.py:271-368 - same loop as the sparse flavour, p-value only
    (`rcond` is accepted and, as in the reference, not forwarded to pinv)."""
    p, _ = _hm.classifier_based_performance_metric(features, adj, labels, sample_max, base_classifier=base_classifier,
                                                   epochs=epochs)
    return p
