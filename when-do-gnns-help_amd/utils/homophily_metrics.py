"""MI355X-backed twin of the reference's `utils/homophily_metrics.py` (sparse-adjacency flavour).

Same function names and signatures; `A` / `adj` are torch sparse COO tensors (or a
`wdg_amd.ops.CsrGraph`), labels are integer vectors, results are 0-dim tensors.  The work is done by
  - csrc/graph_build.hip  : `.coalesce()` -> int32 CSR                       (rows A1, A7)
  - csrc/edge_stats.hip   : one integer pass for rows A8-A11
  - csrc/spmm.hip         : `torch.spmm(adj, features)`                      (row A6)
  - csrc/las.hip          : aggregation similarity without the n x n Gram    (row A12)
  - csrc/gemm.hip         : sampled Gram products on the fp32 MFMA pipe      (row A13)
Quirks of the reference that callers may rely on are kept and cited inline.
"""
import math
import os
import random
import time

import numpy as np
import torch
from scipy.stats import ttest_ind

from .. import ops
from ..ops import CsrGraph
from . import _tails
from .util_funcs import accuracy, random_disassortative_splits  # noqa: F401  (re-exported like the reference)

pi = math.pi
device = torch.device("cuda:0" if torch.cuda.is_available() else "cpu")  # reference: utils/homophily_metrics.py:17-21


def _graph(a):
    return CsrGraph.from_any(a)


def _labels(labels, dev):
    if not isinstance(labels, torch.Tensor):
        labels = torch.as_tensor(np.asarray(labels))
    return labels.to(dev)


def remove_self_loops(edge_index, edge_attr=None):
    """reference: utils/homophily_metrics.py:24-40 (boolean mask on a [2,E] index tensor)."""
    mask = edge_index[0] != edge_index[1]
    return edge_index[:, mask], (edge_attr if edge_attr is None else edge_attr[mask])


# ------------------------------------------------------------------------------------------- A8
def edge_homophily(A, labels, ignore_negative=False):
    """Fraction of stored entries (self loops included) whose endpoints share a label.
    reference: utils/homophily_metrics.py:43-57.

    2-D one-hot `labels` reproduce the element-wise quirk of homophily_tests.py:114-116.
    `ignore_negative=True` raises in the reference (np.mean of a torch bool mask); here it returns the mean
    over entries whose two labels are non-negative, which is what its docstring promises."""
    g = _graph(A)
    labels = _labels(labels, g.device)
    if labels.dim() == 2:
        c = labels.shape[1]
        st = ops.edge_label_stats(g, torch.argmax(labels, 1), c, per_row=False)
        return _tails.edge_homophily_onehot_quirk(st, c)
    st = ops.edge_label_stats(g, labels, per_row=False)
    return _tails.edge_homophily_labeled(st) if ignore_negative else _tails.edge_homophily_all(st)


# ------------------------------------------------------------------------------------------- A9
def node_homophily(A, labels):
    """reference: utils/homophily_metrics.py:60-68."""
    g = _graph(A)
    st = ops.edge_label_stats(g, _labels(labels, g.device))
    return _tails.node_homophily_noself(st)


def node_homophily_edge_idx(edge_idx, labels, num_nodes):
    """reference: utils/homophily_metrics.py:71-78; duplicates in `edge_idx` count with multiplicity.

    (The reference divides a length-`num_nodes` vector by `bincount(src)`, which fails when the highest-numbered
    node has no non-loop edge; that case simply works here.)"""
    g = CsrGraph.from_coo(edge_idx[0], edge_idx[1], num_nodes, None, ops.COO_KEEP_DUPLICATES)
    st = ops.edge_label_stats(g, _labels(labels, g.device))
    return _tails.node_homophily_noself(st)


# ------------------------------------------------------------------------------------------- A10
def _edge_index_graph(edge_index, label):
    n = int(label.shape[0])
    return CsrGraph.from_coo(edge_index[0], edge_index[1], n, None, ops.COO_KEEP_DUPLICATES)


def compact_matrix_edge_idx(edge_idx, labels):
    """c x c compatibility matrix, row-normalised; negative labels = unlabelled; self loops removed.
    reference: utils/homophily_metrics.py:81-102."""
    labels = labels.squeeze()
    g = _edge_index_graph(edge_idx, labels)
    st = ops.edge_label_stats(g, _labels(labels, g.device), per_row=False)
    return _tails.compat_matrix(st["compat"])


def our_measure(edge_index, label):
    """Class homophily h_hat.  reference: utils/homophily_metrics.py:105-123.

    Takes an EDGE INDEX [2,E] like the reference (which fails on a sparse adjacency); a sparse
    adjacency / CsrGraph is accepted too and read as its stored pattern."""
    label = label.squeeze()
    if isinstance(edge_index, CsrGraph) or (isinstance(edge_index, torch.Tensor) and edge_index.layout != torch.strided):
        g = _graph(edge_index)
    else:
        g = _edge_index_graph(edge_index, label)
    lab = _labels(label, g.device)
    st = ops.edge_label_stats(g, lab, per_row=False)
    return _tails.class_homophily(st["compat"], lab)


# ------------------------------------------------------------------------------------------- A11
def class_distribution(A, labels):
    """(p, p_bar, pc); assumes every node carries one self loop (deg = row count - 1).
    reference: utils/homophily_metrics.py:126-147."""
    g = _graph(A)
    lab = _labels(labels, g.device)
    st = ops.edge_label_stats(g, lab, per_row=False)
    return _tails.class_distribution(st, lab)


def adjusted_homo(A, label):
    """reference: utils/homophily_metrics.py:150-155 (edge term includes loops, p_bar excludes them)."""
    g = _graph(A)
    lab = _labels(label, g.device)
    st = ops.edge_label_stats(g, lab, per_row=False)
    _, p_bar, _ = _tails.class_distribution(st, lab)
    return _tails.adjusted(_tails.edge_homophily_all(st), p_bar)


def label_informativeness(A, label):
    """reference: utils/homophily_metrics.py:158-161."""
    g = _graph(A)
    lab = _labels(label, g.device)
    st = ops.edge_label_stats(g, lab, per_row=False)
    _, p_bar, pc = _tails.class_distribution(st, lab)
    return _tails.label_informativeness(p_bar, pc)


# ------------------------------------------------------------------------------------------- generalised edge homophily
def _edge_cosine_sum(g, features, entries=None):
    """sum over stored non-loop entries (or the listed entry ids) of cos(x_u, x_v); NaN -> 0."""
    x = features.to(g.device, torch.float32)
    if entries is not None:  # sampled entries: loops are kept, as the reference's sampling branch does
        sim = ops.edge_cosine(g, x, entries=entries, skip_self=False)
        return sim.double().sum().float(), torch.tensor(sim.shape[0], device=g.device), sim
    sim = ops.edge_cosine(g, x, skip_self=True)  # csrc/edge_cosine.hip: one wave per stored entry
    n_noself = (g.row_indices() != g.col.to(torch.int64)).sum()
    return sim.double().sum().float(), n_noself, sim


def generalized_edge_homophily(adj, features, label, sample_max=75000, iteration=10):
    """Mean cosine similarity over edges.  reference: utils/homophily_metrics.py:164-187.

    Below `sample_max` entries the reference forms the dense N x N cosine matrix and masks it with (adj>0) minus
    the diagonal; the same sum is taken here edge by edge.  Above it, `iteration` draws of `sample_max` entries
    from Python's global `random` (loops included), as in the reference."""
    g = _graph(adj)
    nedges = g.nnz
    if nedges < sample_max:
        total, cnt, _ = _edge_cosine_sum(g, features)
        return total / cnt.float()
    vals = np.zeros(iteration)
    for i in range(iteration):
        smp = torch.tensor(random.sample(list(np.arange(nedges)), int(sample_max)), device=g.device)
        total, _, _ = _edge_cosine_sum(g, features, smp)
        vals[i] = float(total) / int(sample_max)
    return np.mean(vals)


# ------------------------------------------------------------------------------------------- A12
def similarity(features, adj, label, hard=None, LP=1, ifsum=1, idx_train=None):
    """Label-aggregation similarity (aggregation homophily before the `2s-1` map).
    reference: utils/homophily_metrics.py:190-229.

    H = adj @ features on the GPU, then W = (H_s H_s^T) Y_s evaluated as H_s (H_s^T Y_s) in fp64 by csrc/las.hip
    (the n x n Gram never exists).  `idx_train` is a bool mask or None."""
    g = _graph(adj)
    dev = g.device
    label = label.to(dev)
    h = ops.spmm(g, features.to(dev))
    labels_all = torch.argmax(label, 1)
    rows = None
    if idx_train is not None:
        idx_train = idx_train.to(dev)
        rows = (torch.nonzero(idx_train).view(-1) if idx_train.dtype == torch.bool else idx_train).to(torch.int32)
    labels_sel = labels_all if rows is None else labels_all[rows.long()]
    c = int(labels_sel.max().item()) + 1
    default_path = (LP == 1 and ifsum == 1)
    cnt, n, w = ops.las(h, labels_all, c, rows=rows, want_weights=not default_path)
    if default_path:
        # kernel counts assume one-hot `label` rows (degs_label == class size), which is what every call site passes
        return (cnt[0] if hard is None else cnt[1]).to(torch.float32) / n
    lab_sel = label if rows is None else label[rows.long()]
    return _tails.las_from_weights(w, labels_sel, lab_sel[:, :c], hard, LP, ifsum)


# ------------------------------------------------------------------------------------------- A13
def _arccos_kernel(gram, n_layers):
    """GNTK-style map of a Gram matrix.  reference: utils/homophily_metrics.py:236-244."""
    eps = 1e-8
    d = torch.sqrt(torch.diag(gram))
    nrm = d.reshape(-1, 1) * d.reshape(1, -1)
    nrm = (nrm > eps) * nrm + eps * (nrm <= eps)
    if n_layers != 1:
        return gram
    arccos = torch.acos(torch.div(gram, nrm))
    sqrt = torch.sqrt(torch.square(nrm) - torch.square(gram))
    arccos = torch.where(torch.isnan(arccos), torch.zeros_like(arccos), arccos)
    sqrt = torch.where(torch.isnan(sqrt), torch.zeros_like(sqrt), sqrt)
    return 1 / pi * (gram * (pi - arccos) + sqrt)


def _as_index(idx, dev):
    """bool mask or index vector (both occur) -> int64 index vector on `dev`"""
    idx = torch.as_tensor(idx)
    if idx.dtype == torch.bool:
        idx = torch.nonzero(idx).view(-1)
    return idx.to(dev)


def _gntk_from_aggregate(h, features, sample, n_layers):
    smp = torch.as_tensor(np.asarray(sample.cpu() if isinstance(sample, torch.Tensor) else sample), device=h.device)
    if smp.dtype == torch.bool:
        smp = torch.nonzero(smp).view(-1)
    hs, xs = h[smp].contiguous(), features[smp].contiguous()
    g_gram = ops.gemm(hs, hs, transb=True)
    x_gram = ops.gemm(xs, xs, transb=True)
    return _arccos_kernel(g_gram, n_layers) / 2, _arccos_kernel(x_gram, n_layers) / 2


def gntk_homophily_(features, adj, sample, n_layers):
    """(K_G / 2, K_X / 2): sampled Gram of the aggregated and of the raw features, arc-cosine kernel when
    n_layers == 1.  reference: utils/homophily_metrics.py:232-257."""
    g = _graph(adj)
    features = features.to(g.device, torch.float32)
    h = ops.spmm(g, features)  # computed once (the reference computes it twice per call)
    return _gntk_from_aggregate(h, features, sample, n_layers)


def classifier_based_performance_metric(features, adj, labels, sample_max, base_classifier='kernel_reg1', epochs=100,
                                        solver=None):
    """Classifier-based performance metric -> (p_value, seconds).  reference: utils/homophily_metrics.py:260-349.

    GPU: the aggregation A X (hoisted out of the epoch loop - it is loop invariant), the sampled
    Gram products and the arc-cosine map (hoisted too when nnodes <= sample_max: the sample is then every node).
    Host, exactly as in the reference: split sampling from torch's CPU generator, `np.linalg.pinv` (the reference
    moves the kernels to the CPU for it, :286-290), sklearn GNB/SVM, scipy's Welch t-test.

    solver="device" (or WDG_KR_SOLVER=device) keeps the kernel regression on the GPU: the kernels
    never leave the device, the two train blocks of an epoch are pseudo-inverted by one batched symmetric
    eigendecomposition (`torch.linalg.pinv(hermitian=True)`, rocSOLVER underneath) and the predictions are two MFMA
    products.  For a symmetric matrix V diag(1/lambda) V^T IS the SVD pseudo-inverse.  Cut-off: |lambda| <=
    n eps |lambda|_max (torch's default) instead of numpy's 1e-15 sigma_max - the null space of a kernel with zero or
    duplicate rows comes out of LAPACK's SVD as exact zeros (cut either way) but out of an fp32 eigensolver as
    +-1e-6 lambda_max noise, which numpy's cut-off would invert.  Per-epoch accuracies then match the host path to a
    few validation nodes on well-conditioned kernels (tests/test_gpu_api.py); on rank-deficient linear kernels
    (kernel_reg0, F < n_train) the host path inverts ITS rounding noise and the two agree only statistically."""
    solver = solver or os.environ.get("WDG_KR_SOLVER", "host")
    if solver not in ("host", "device"):
        raise ValueError(f"unknown solver {solver!r}")
    from sklearn import svm
    from sklearn.naive_bayes import GaussianNB

    g = _graph(adj)
    dev = g.device
    features = features.to(dev, torch.float32)
    labels = labels.to(dev)
    nnodes = labels.shape[0]
    if labels.dim() > 1:
        labels = labels.flatten()
    G_results, X_results, diff_results = torch.zeros(epochs), torch.zeros(epochs), torch.zeros(epochs)
    t_time = time.time()
    h_agg = ops.spmm(g, features)
    n_cls = int(labels.max().item()) + 1
    labels_cpu = labels.cpu()
    fixed_kernels = None  # nnodes <= sample_max: every epoch uses all nodes -> the kernels are loop invariant as well
    for j in range(epochs):
        if nnodes <= sample_max:
            sample = np.arange(nnodes)
            label_onehot = torch.eye(n_cls)[labels_cpu]
            labels_sample = labels_cpu
        else:
            sample, _, _ = random_disassortative_splits(labels, labels.max() + 1, sample_max / nnodes)
            sample = sample.cpu()
            label_onehot = torch.eye(n_cls)[labels_cpu][sample, :]
            labels_sample = labels_cpu[sample]
        idx_train, idx_val, idx_test = random_disassortative_splits(labels_sample, labels_sample.max() + 1)
        idx_train, idx_val = idx_train.cpu(), (idx_val + idx_test).cpu()
        if base_classifier in {'kernel_reg0', 'kernel_reg1'}:
            nlayers = 0 if base_classifier == 'kernel_reg0' else 1
            if nnodes <= sample_max and fixed_kernels is not None:
                K_graph, K = fixed_kernels
            else:
                K_graph, K = _gntk_from_aggregate(h_agg, features, sample, nlayers)
                if solver == "host":
                    K_graph, K = K_graph.cpu(), K.cpu()
                if nnodes <= sample_max:
                    fixed_kernels = (K_graph, K)
            preds = []
            if solver == "device":
                tr, va = _as_index(idx_train, dev), _as_index(idx_val, dev)
                k_tt = torch.stack([kern[tr][:, tr] for kern in (K_graph, K)])
                coef = torch.linalg.pinv(k_tt, hermitian=True) @ label_onehot[idx_train].to(dev)
                for kern, cf in zip((K_graph, K), coef):
                    preds.append(ops.gemm(kern[va][:, tr].contiguous(), cf.contiguous()).cpu())
            else:
                for kern in (K_graph, K):
                    k_tt = kern[idx_train, :][:, idx_train]
                    k_vt = kern[idx_val, :][:, idx_train]
                    preds.append(k_vt @ (torch.tensor(np.linalg.pinv(k_tt.numpy())) @ label_onehot[idx_train]))
            acc_g = accuracy(labels_sample[idx_val], preds[0])
            acc_x = accuracy(labels_sample[idx_val], preds[1])
        else:
            smp = torch.as_tensor(np.asarray(sample))
            if smp.dtype == torch.bool:
                smp = torch.nonzero(smp).view(-1)
            X = features[smp.to(dev)].cpu()
            X_agg = h_agg[smp.to(dev)].cpu()
            if base_classifier == 'gnb':
                mk = lambda: GaussianNB()  # noqa: E731
            elif base_classifier == 'svm_rbf':
                mk = lambda: svm.SVC(kernel='rbf', gamma=0.5, C=0.1)  # noqa: E731
            elif base_classifier == 'svm_poly':
                mk = lambda: svm.SVC(kernel='poly', degree=3, C=1)  # noqa: E731
            elif base_classifier == 'svm_linear':
                mk = lambda: svm.SVC(kernel='linear')  # noqa: E731
            else:
                raise ValueError(f"unknown base_classifier {base_classifier!r}")
            x_model, g_model = mk().fit(X[idx_train], labels_sample[idx_train]), mk().fit(X_agg[idx_train], labels_sample[idx_train])
            X_pred, G_pred = torch.tensor(x_model.predict(X[idx_val])), torch.tensor(g_model.predict(X_agg[idx_val]))
            acc_g = torch.mean(G_pred.eq(labels_sample[idx_val]).float())
            acc_x = torch.mean(X_pred.eq(labels_sample[idx_val]).float())
        diff_results[j] = (acc_g > acc_x)
        G_results[j], X_results[j] = acc_g, acc_x

    _, p = ttest_ind(X_results.detach().cpu(), G_results.detach().cpu(), axis=0, equal_var=False, nan_policy='propagate')
    p = p / 2 if torch.mean(diff_results) <= 0.5 else 1 - p / 2
    return p, time.time() - t_time
