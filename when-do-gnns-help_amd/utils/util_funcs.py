"""MI355X-backed twin of the reference's `utils/util_funcs.py` (graph preprocessing).

Same function names, argument meaning and guards as the reference (`inf -> 0`, `rowsum==0 -> 1`); the
arithmetic runs in the HIP kernels of csrc/graph_build.hip.  Return kinds follow the reference (scipy in ->
scipy out, torch in -> torch out) so call sites such as `homophily_tests.py:80-104` and
`synthetic_plot.py:81-92` work unchanged; the `*_csr` variants return the device CSR directly and are what
the rest of this package uses.
"""
import numpy as np
import scipy.sparse as sp
import torch

from .. import ops
from ..ops import CsrGraph

device = torch.device("cuda:0" if torch.cuda.is_available() else "cpu")  # reference: utils/util_funcs.py:23-24


def _is_scipy(mx):
    return sp.issparse(mx)


def _csr_to_scipy(g, dtype=np.float64):
    rowptr, col, val = g.rowptr.cpu().numpy(), g.col.cpu().numpy(), g.val.cpu().numpy()
    return sp.csr_matrix((val.astype(dtype), col, rowptr), shape=(g.n_rows, g.n_cols))


# ------------------------------------------------------------------------------------------- row scaling
def normalize(mx):
    """Row-normalise: diag(1/rowsum) mx, inf -> 0.  reference: utils/util_funcs.py:29-36."""
    if _is_scipy(mx):
        # any shape (feature matrices are N x F): the CSR arrays are uploaded as they are - the COO builder is for square
        # adjacencies; coefficients in fp64 like scipy's, stored values are the fp32 roundings (device values are fp32)
        g = ops.normalise_values(CsrGraph.from_scipy_csr(mx), ops.NORM_RW, ops.PREC_F64)
        return _csr_to_scipy(g, mx.dtype)
    if isinstance(mx, torch.Tensor):
        return ops.row_l1_normalise(mx)
    return ops.row_l1_normalise(torch.as_tensor(np.asarray(mx, dtype=np.float32))).cpu().numpy()


def preprocess_features(features):
    """Row-normalise a feature matrix.  reference: utils/util_funcs.py:39-46 (same arithmetic as `normalize`)."""
    return normalize(features)


def normalize_tensor(mx, symmetric=0):
    """Dense torch matrix: D^-1 M (symmetric=0) or D^-1/2 M D^-1/2.  reference: utils/util_funcs.py:365-380.

    The reference builds an N x N `torch.diag` and multiplies (O(N^2 F)); here the rw case is one row-scaling
    kernel, and the symmetric case goes dense -> CSR -> coefficient kernels -> dense."""
    if symmetric == 0:
        return ops.row_l1_normalise(mx)
    g = ops.normalise_values(CsrGraph.from_dense(mx), ops.NORM_SYM, ops.PREC_F32)
    return g.to_torch_sparse().to_dense()


def normalized_adjacency_csr(adj, symmetric=0, add_self_loops=True, prec=ops.PREC_F32):
    """A_hat = D^-1 (A+I) or D^-1/2 (A+I) D^-1/2 as a device CSR (no dense detour).

    Equivalent of homophily_tests.py:83-85 (`normalize_tensor(eye + adj.to_dense(), symmetric).to_sparse()`) and
    of :98-104 (`sparse_mx_to_torch_sparse_tensor(sys_/row_normalized_adjacency(adj))`, prec=PREC_F64)."""
    flags = ops.COO_ADD_SELF_LOOPS if add_self_loops else 0
    g = CsrGraph.from_any(adj, flags)
    return ops.normalise_values(g, ops.NORM_SYM if symmetric else ops.NORM_RW, prec)


def row_normalized_adjacency(adj):
    """(A + I) with every row divided by its L1 norm.  reference: utils/util_funcs.py:383-390.

    Returns a scipy COO like the reference; values are the fp32-rounded coefficients (the reference's next step,
    `sparse_mx_to_torch_sparse_tensor`, casts to fp32 anyway - utils/util_funcs.py:402)."""
    g = normalized_adjacency_csr(sp.coo_matrix(adj), symmetric=0, prec=ops.PREC_F64)
    return _csr_to_scipy(g).tocoo()


def sys_normalized_adjacency(adj):
    """D^-1/2 (A + I) D^-1/2 with rowsum==0 -> 1 and inf -> 0.  reference: utils/util_funcs.py:418-426."""
    g = normalized_adjacency_csr(sp.coo_matrix(adj), symmetric=1, prec=ops.PREC_F64)
    return _csr_to_scipy(g).tocoo()


def sparse_mx_to_torch_sparse_tensor(sparse_mx):
    """scipy sparse -> torch sparse COO fp32 (int64 indices), built directly on the GPU.

    reference: utils/util_funcs.py:400-407 (numpy vstack + astype on the host: 6.7 s at 13.8 M nnz)."""
    coo = sparse_mx.tocoo()
    dev = device
    idx = torch.stack([torch.as_tensor(coo.row.astype(np.int64)).to(dev), torch.as_tensor(coo.col.astype(np.int64)).to(dev)])
    val = torch.as_tensor(coo.data).to(dev).to(torch.float32)
    return torch.sparse_coo_tensor(idx, val, torch.Size(coo.shape))


# ------------------------------------------------------------------------------------------- support
def accuracy(labels, output):
    """reference: utils/util_funcs.py:393-397."""
    preds = output.max(1)[1].type_as(labels)
    correct = preds.eq(labels).double().sum()
    return correct / len(labels)


def index_to_mask(index, size):
    """reference: utils/util_funcs.py:478-481."""
    mask = torch.zeros(size, dtype=torch.bool, device=index.device)
    mask[index] = 1
    return mask


def _disassortative_split_indices(labels, c, training_percentage=0.6):
    """(train, val, test) index tensors on the host, drawn from torch's global CPU generator exactly like the reference's
    routine (utils/util_funcs.py:454-475): one `randperm` per class, then one over the remainder."""
    n = labels.shape[0]
    per_class = []
    for k in range(c):
        members = torch.nonzero(labels == k).view(-1)
        per_class.append(members[torch.randperm(members.shape[0])])
    n_train = int(round(training_percentage * (n / c)))
    n_val = int(round(0.2 * n))
    train = torch.cat([p[:n_train] for p in per_class])
    rest = torch.cat([p[n_train:] for p in per_class])
    rest = rest[torch.randperm(rest.shape[0])]
    return train, rest[:n_val], rest[n_val:]


def random_disassortative_splits(labels, num_classes, training_percentage=0.6):
    """Class-balanced train / 20 % val / rest test boolean masks.  reference: utils/util_funcs.py:454-475.

    Host logic (no kernel): it draws from torch's global CPU generator in the same order as the reference
    (one `randperm` per class, then one over the remainder), so a given `torch.manual_seed` yields the
    reference's masks.  `num_classes` is a tensor or int; Python `round` (banker's) as in the reference."""
    labels = labels.cpu()
    n = labels.shape[0]
    c = int(num_classes.item()) if isinstance(num_classes, torch.Tensor) else int(num_classes)
    masks = [index_to_mask(ix, n) for ix in _disassortative_split_indices(labels, c, training_percentage)]
    return tuple(m.to(device) for m in masks)


def kernel_regression_epoch_indices(labels, sample_max, epochs):
    """The node sets of every epoch of classifier_based_performance_metric (utils/homophily_metrics.py:268-281): per epoch the
    class-balanced sample (all nodes when nnodes <= sample_max), its train rows and its validation (val + test) rows, as
    ascending NODE ids - drawn from torch's global CPU generator in the reference's order (seed it before the call).
    -> list of (train int64 [n_train], val int64 [n_val]) host tensors."""
    labels = labels.cpu().flatten()
    n = labels.shape[0]
    c = int(labels.max().item()) + 1
    out = []
    for _ in range(epochs):
        if n <= sample_max:
            sample = torch.arange(n)
        else:
            smp, _, _ = _disassortative_split_indices(labels, c, sample_max / n)
            sample = torch.sort(smp).values  # (a boolean mask in the reference: ascending node order)
        lab_s = labels[sample]
        tr, va, te = _disassortative_split_indices(lab_s, int(lab_s.max().item()) + 1)
        train = sample[torch.sort(tr).values]
        val = sample[torch.sort(torch.cat([va, te])).values]
        out.append((train, val))
    return out


def rand_train_test_idx(label, train_prop=.6, valid_prop=.2, ignore_negative=True):
    """reference: utils/util_funcs.py:484-508 (numpy global RNG)."""
    labeled = torch.where(label != -1)[0] if ignore_negative else label
    n = labeled.shape[0]
    n_tr, n_va = int(n * train_prop), int(n * valid_prop)
    perm = torch.as_tensor(np.random.permutation(n))
    parts = perm[:n_tr], perm[n_tr:n_tr + n_va], perm[n_tr + n_va:]
    if not ignore_negative:
        return parts
    return tuple(labeled[p] for p in parts)
