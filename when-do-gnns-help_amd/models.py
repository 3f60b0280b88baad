"""SGC-1 / GCN-2 on the HIP kernels (build-defined model API).

The reference repo ships no model code - `gnns_on_syn.py:1-249` is a table of accuracies obtained with upstream
ACM-GNN code (`README.md:79-87`) - so the architecture is defined here, following what that table names
("GCN" with random-walk A_hat, "SGC" one step):
    SGC-1 : logits = (A_hat X) W                     (aggregation computed once and cached)
    GCN-2 : logits = A_hat relu(A_hat (X W0)) W1     (transform-then-aggregate: hidden << F), bias-free layers
Forward AND backward run on csrc/spmm*.hip (A_hat^T for the backward pass is the transposed CSR: the synthetic
graphs are directed) and csrc/gemm.hip (exact-fp32 MFMA).  PyTorch supplies autograd bookkeeping, dropout,
log-softmax / NLL on [N, C] logits and Adam.
"""
import torch

from . import ops
from .ops import CsrGraph
from .utils.util_funcs import accuracy, random_disassortative_splits


class NormAdj:
    """A_hat = diag(r) (A [+ I]) diag(c) kept factored: CSR pattern + its transpose + the two scale vectors."""

    def __init__(self, adj, symmetric=0, add_self_loops=True):
        g = CsrGraph.from_any(adj, ops.COO_ADD_SELF_LOOPS if add_self_loops else 0)
        d = ops.degree_norm(g, ops.NORM_SYM if symmetric else ops.NORM_RW, ops.PREC_F32)["dinv"]
        self.graph, self.graph_t = g, g.transpose()
        self.row_scale, self.col_scale = d, (d if symmetric else None)
        self.n = g.n_rows

    def matmul(self, x):
        return _Aggregate.apply(x, self)


class _Aggregate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, adj):
        ctx.adj = adj
        return ops.spmm(adj.graph, x, row_scale=adj.row_scale, col_scale=adj.col_scale)

    @staticmethod
    def backward(ctx, gy):
        a = ctx.adj  # (R A C)^T = C A^T R
        return ops.spmm(a.graph_t, gy.contiguous(), row_scale=a.col_scale, col_scale=a.row_scale), None


class _Linear(torch.autograd.Function):
    """y = act(x @ w): forward and both gradients on wdg_gemm_f32."""

    @staticmethod
    def forward(ctx, x, w, relu):
        y = ops.gemm(x, w, relu=relu)
        ctx.save_for_backward(x, w, y if relu else torch.empty(0))
        ctx.relu = relu
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, y = ctx.saved_tensors
        g = gy.contiguous()
        if ctx.relu:
            g = g * (y > 0)
        gx = ops.gemm(g, w, transb=True) if ctx.needs_input_grad[0] else None  # g @ w^T
        gw = ops.gemm(x.t().contiguous(), g)                                   # x^T @ g
        return gx, gw, None


class SGC1(torch.nn.Module):
    def __init__(self, nfeat, nclass):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.empty(nfeat, nclass))
        torch.nn.init.xavier_uniform_(self.weight)
        self._cache = None

    def forward(self, adj, x):
        if self._cache is None or self._cache[0] is not x:
            with torch.no_grad():
                self._cache = (x, ops.spmm(adj.graph, x, row_scale=adj.row_scale, col_scale=adj.col_scale))
        return _Linear.apply(self._cache[1], self.weight, False)


class GCN2(torch.nn.Module):
    def __init__(self, nfeat, nclass, nhid=64, dropout=0.5):
        super().__init__()
        self.w0 = torch.nn.Parameter(torch.empty(nfeat, nhid))
        self.w1 = torch.nn.Parameter(torch.empty(nhid, nclass))
        torch.nn.init.xavier_uniform_(self.w0)
        torch.nn.init.xavier_uniform_(self.w1)
        self.dropout = dropout

    def forward(self, adj, x):
        h = torch.relu(adj.matmul(_Linear.apply(x, self.w0, False)))
        h = torch.nn.functional.dropout(h, self.dropout, self.training)
        return adj.matmul(_Linear.apply(h, self.w1, False))


def train_eval(model, adj, x, labels, masks=None, epochs=200, lr=0.01, weight_decay=5e-4):
    """Full-batch training with Adam + cross-entropy; model selection on validation accuracy.
    masks = (train, val, test) boolean tensors; default: `random_disassortative_splits` (60/20/20).
    Returns dict(val_acc, test_acc, epochs)."""
    dev = adj.graph.device
    x, labels = x.to(dev, torch.float32), labels.to(dev)
    model = model.to(dev)
    if masks is None:
        masks = random_disassortative_splits(labels, labels.max() + 1)
    tr, va, te = (m.to(dev) for m in masks)
    opt = torch.optim.Adam(model.parameters(), lr=lr, weight_decay=weight_decay)
    best = dict(val_acc=-1.0, test_acc=0.0, epoch=0)
    for ep in range(epochs):
        model.train()
        opt.zero_grad()
        out = torch.log_softmax(model(adj, x), 1)
        loss = torch.nn.functional.nll_loss(out[tr], labels[tr])
        loss.backward()
        opt.step()
        model.eval()
        with torch.no_grad():
            out = model(adj, x)
            v, t = float(accuracy(labels[va], out[va])), float(accuracy(labels[te], out[te]))
        if v > best["val_acc"]:
            best = dict(val_acc=v, test_acc=t, epoch=ep)
    best["epochs"] = epochs
    return best
