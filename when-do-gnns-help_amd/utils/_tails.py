"""Scalar "tails" of the homophily metrics: a handful of fp32 operations on the exact integer statistics that
csrc/edge_stats.hip produces.  Shared by the sparse flavour (homophily_metrics.py) and
the dense flavour (homophily_plot.py).  Everything stays on the GPU; results are 0-dim fp32 tensors like the
reference's."""
import torch


def f32(t):
    return t.to(torch.float32)


def edge_homophily_all(st):
    """matches / stored entries, self loops included.  reference: utils/homophily_metrics.py:50-56."""
    return f32(st["totals"][1]) / f32(st["totals"][0])


def edge_homophily_labeled(st):
    """ignore_negative=True branch (:52-54): only entries whose two labels are >= 0."""
    return f32(st["totals"][3]) / f32(st["totals"][2])


def edge_homophily_onehot_quirk(st, c):
    """homophily_tests.py:114-116 passes one-hot labels, so `labels[src]==labels[dst]` is an [E,C] boolean and
    the mean runs over E*C cells: C cells agree on a matching edge, C-2 on a non-matching one."""
    nnz, m = f32(st["totals"][0]), f32(st["totals"][1])
    return (m * c + (nnz - m) * (c - 2)) / (nnz * c)


def edge_homophily_noself(st):
    """dense flavour: diagonal removed first.  reference: utils/homophily_plot.py:48-51."""
    return f32(st["totals"][5]) / f32(st["totals"][4])


def node_homophily_noself(st):
    """mean over nodes with a non-loop entry of matching/non-loop degree.  reference: utils/homophily_metrics.py:73-78."""
    deg = f32(st["row_nnz_noself"])
    hs = f32(st["row_match_noself"]) / deg
    return hs[deg != 0].mean()


def node_homophily_withself(st):
    """dense flavour keeps self loops (A.nonzero()).  reference: utils/homophily_plot.py:85-99."""
    deg = f32(st["row_nnz"])
    loops = f32(st["row_nnz"] - st["row_nnz_noself"])
    hs = (f32(st["row_match_noself"]) + loops) / deg
    return hs[deg != 0].mean()


def compat_matrix(compat):
    """H = K / rowsum(K) (NaN rows when a class has no labelled edge).  reference: utils/homophily_metrics.py:101."""
    k = f32(compat)
    return k / k.sum(1, keepdim=True)


def class_homophily(compat, labels):
    """sum_k max(H_kk - p_k, 0) / (C-1), NaN terms skipped.  reference: utils/homophily_metrics.py:110-123."""
    labels = labels.squeeze()
    c = int(labels.max().item()) + 1
    h = compat_matrix(compat)
    nz = labels[labels >= 0]
    counts = nz.unique(return_counts=True)[1]
    prop = counts.float() / nz.shape[0]
    terms = torch.clamp(torch.diagonal(h)[:c] - prop[:c], min=0)
    val = torch.where(torch.isnan(terms), torch.zeros_like(terms), terms).sum()
    return val / (c - 1)


def class_distribution(st, labels):
    """(p, p_bar, pc) of utils/homophily_metrics.py:126-147: deg = |P_u| - 1, zeros -> 1e-8."""
    labels = labels.squeeze()
    p = labels.unique(return_counts=True)[1] / labels.shape[0]
    tot = f32(st["classdeg"].sum())
    p_bar = f32(st["classdeg"]) / tot
    pc = f32(st["compat"]) / tot
    p_bar = torch.where(p_bar == 0, torch.full_like(p_bar, 1e-8), p_bar)
    pc = torch.where(pc == 0, torch.full_like(pc, 1e-8), pc)
    return p, p_bar, pc


def adjusted(edge_homo, p_bar):
    """reference: utils/homophily_metrics.py:153."""
    s = torch.sum(p_bar ** 2)
    return (edge_homo - s) / (1 - s)


def label_informativeness(p_bar, pc):
    """reference: utils/homophily_metrics.py:160."""
    return 2 - torch.sum(pc * torch.log(pc)) / torch.sum(p_bar * torch.log(p_bar))


def las_from_weights(w, labels, label, hard, LP, ifsum):
    """Every branch of utils/homophily_metrics.py:201-229 given W (n x C, fp64 on device)."""
    n, c = w.shape
    if ifsum != 1:
        cnt = torch.bincount(labels, minlength=c).to(w.dtype)
        w = w / cnt[None, :]
    idx = torch.arange(n, device=w.device)
    own = w[idx, labels]
    label = label.to(w.dtype)
    if hard is None:
        if ifsum == 1:
            nnodes, degs = n, label @ label.sum(0)
        else:
            nnodes, degs = c, torch.ones(n, dtype=w.dtype, device=w.device)
        if LP == 1:
            ratio = (own / degs) / ((w.sum(1) - own) / (nnodes - degs))
            ratio = torch.where(torch.isnan(ratio), torch.zeros_like(ratio), ratio)
            return torch.mean((ratio >= 1).float())
        return torch.mean((((w - w * label).sum(1) <= 0) & ((w * label).sum(1) >= 0)).float())
    if LP == 1:
        return torch.mean(torch.argmax(w, 1).eq(labels).float())
    return torch.mean((((w - w * label).max(1)[0] <= 0.) & ((w * label).sum(1) >= 0)).float())
