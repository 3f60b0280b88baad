"""CPU ORACLE - test infrastructure only (NOT product code).

A CPU restatement of the reference's aggregation / homophily-metric hot path
(SitaoLuan/When-Do-GNNs-Help).  Heavy loops live in plain C
(`wdg_oracle.c`, compiled with gcc into `oracle/_build/libwdg_oracle.so`); the
scalar "tails" of each metric are numpy.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import this
module; the shipped package (`when-do-gnns-help_amd/`) never does.

Parity status: PINNED.  `tests/test_oracle_golden.py` checks every function here
against `tests/golden/*.npz`, which were produced by running the real reference
in the build container (`tests/golden/make_golden.py`).

Citations `file:line` are into the reference checkout.
"""
import ctypes
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")
_SO = os.path.join(_BUILD, "libwdg_oracle.so")
_SRC = os.path.join(_HERE, "wdg_oracle.c")

SYMMETRISE, BINARISE, ADD_SELF_LOOPS, DROP_SELF_LOOPS, KEEP_DUPLICATES = 1, 2, 4, 8, 16
NORM_RW, NORM_SYM = 0, 1
PREC_F32, PREC_F64 = 0, 1


def build(force=False):
    """Compile the C restatement (gcc, a second or two)."""
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(_SRC):
        os.makedirs(_BUILD, exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-o", _SO, _SRC, "-lm"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.orc_coo_to_csr.restype = ctypes.c_int64
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dt):
    return None if a is None else np.ascontiguousarray(a, dtype=dt)


# ----------------------------------------------------------------------------- C wrappers
def coo_to_csr(src, dst, n, val=None, flags=0):
    """(rowptr int32[N+1], col int32[nnz], val fp32[nnz]); see orc_coo_to_csr."""
    src, dst, val = _c(src, np.int64), _c(dst, np.int64), _c(val, np.float32)
    e = src.shape[0]
    cap = (2 if flags & SYMMETRISE else 1) * e + n + 1
    rowptr = np.zeros(n + 1, np.int32)
    col = np.zeros(cap, np.int32)
    out = np.zeros(cap, np.float32)
    nnz = lib().orc_coo_to_csr(_p(src), _p(dst), _p(val), ctypes.c_int64(e), ctypes.c_int32(n),
                               ctypes.c_int(flags), _p(rowptr), _p(col), _p(out))
    if nnz < 0:
        raise IndexError("edge index out of range")
    return rowptr, col[:nnz].copy(), out[:nnz].copy()


def degree_norm(rowptr, val, mode, prec):
    n = rowptr.shape[0] - 1
    rowsum, cnt, dinv = np.zeros(n, np.float32), np.zeros(n, np.int32), np.zeros(n, np.float64)
    val = _c(val, np.float32)
    lib().orc_degree_norm(_p(rowptr), _p(val), ctypes.c_int32(n), ctypes.c_int(mode), ctypes.c_int(prec),
                          _p(rowsum), _p(cnt), _p(dinv))
    return rowsum, cnt, dinv


def normalise_values(rowptr, col, val, mode, prec, dinv):
    n = rowptr.shape[0] - 1
    out = np.zeros(col.shape[0], np.float32)
    val = _c(val, np.float32)
    lib().orc_normalise_values(_p(rowptr), _p(col), _p(val), ctypes.c_int32(n), ctypes.c_int(mode),
                               ctypes.c_int(prec), _p(dinv), _p(out))
    return out


def normalised_csr(rowptr, col, val, mode, prec):
    """A_hat values for an already self-looped CSR: D^-1 A or D^-1/2 A D^-1/2."""
    _, _, dinv = degree_norm(rowptr, val, mode, prec)
    return normalise_values(rowptr, col, val, mode, prec, dinv)


def spmm_csr(rowptr, col, val, x, f64acc=False):
    x = _c(x, np.float32)
    n, f = rowptr.shape[0] - 1, x.shape[1]
    val = _c(val, np.float32)
    y = np.zeros((n, f), np.float64 if f64acc else np.float32)
    fn = lib().orc_spmm_csr_f64acc if f64acc else lib().orc_spmm_csr_f32
    fn(_p(rowptr), _p(col), _p(val), _p(x), ctypes.c_int64(x.shape[1]), _p(y), ctypes.c_int64(f),
       ctypes.c_int32(n), ctypes.c_int32(f))
    return y


def spmm_csr_t(rowptr, col, val, x, ncols):
    x = _c(x, np.float32)
    n, f = rowptr.shape[0] - 1, x.shape[1]
    val = _c(val, np.float32)
    y = np.zeros((ncols, f), np.float32)
    lib().orc_spmm_csr_t_f32(_p(rowptr), _p(col), _p(val), _p(x), ctypes.c_int64(f), _p(y), ctypes.c_int64(f),
                             ctypes.c_int32(n), ctypes.c_int32(ncols), ctypes.c_int32(f))
    return y


def edge_label_stats(rowptr, col, labels, c):
    n = rowptr.shape[0] - 1
    labels = _c(labels, np.int32)
    st = dict(totals=np.zeros(6, np.int64), row_nnz=np.zeros(n, np.int32), row_nnz_noself=np.zeros(n, np.int32),
              row_match_noself=np.zeros(n, np.int32), compat=np.zeros((c, c), np.int64),
              classdeg=np.zeros(c, np.int64))
    lib().orc_edge_label_stats(_p(rowptr), _p(col), _p(labels), ctypes.c_int32(n), ctypes.c_int32(c),
                               _p(st["totals"]), _p(st["row_nnz"]), _p(st["row_nnz_noself"]),
                               _p(st["row_match_noself"]), _p(st["compat"]), _p(st["classdeg"]))
    return st


def las_weights(h, labels, c, f64=False):
    h = _c(h, np.float32)
    labels = _c(labels, np.int32)
    n, f = h.shape
    w = np.zeros((n, c), np.float64 if f64 else np.float32)
    fn = lib().orc_las_weights_f64 if f64 else lib().orc_las_weights_f32
    fn(_p(h), ctypes.c_int64(f), _p(labels), ctypes.c_int32(n), ctypes.c_int32(f), ctypes.c_int32(c), _p(w))
    return w


def gemm(a, b, bias=None, relu=False):
    a, b, bias = _c(a, np.float32), _c(b, np.float32), _c(bias, np.float32)
    m, k = a.shape
    n = b.shape[1]
    out = np.zeros((m, n), np.float32)
    lib().orc_gemm_f32(_p(a), ctypes.c_int64(k), _p(b), ctypes.c_int64(n), _p(bias), ctypes.c_int(int(relu)),
                       _p(out), ctypes.c_int64(n), ctypes.c_int32(m), ctypes.c_int32(n), ctypes.c_int32(k))
    return out


def gram(h, sample):
    h, sample = _c(h, np.float32), _c(sample, np.int64)
    ns = sample.shape[0]
    g = np.zeros((ns, ns), np.float32)
    lib().orc_gram_f32(_p(h), ctypes.c_int64(h.shape[1]), _p(sample), ctypes.c_int32(ns),
                       ctypes.c_int32(h.shape[1]), _p(g))
    return g


def row_l1_normalise(x, use_abs=False):
    x = _c(x, np.float32)
    y = np.zeros_like(x)
    lib().orc_row_l1_normalise(_p(x), ctypes.c_int64(x.shape[1]), _p(y), ctypes.c_int64(x.shape[1]),
                               ctypes.c_int32(x.shape[0]), ctypes.c_int32(x.shape[1]), ctypes.c_int(int(use_abs)))
    return y


# ----------------------------------------------------------------------------- metric tails (numpy)
def _f32(x):
    return np.float32(x)


def edge_homophily_sparse(st, labels_2d_classes=None):
    """utils/homophily_metrics.py:50-56: mean over ALL stored entries (loops included).

    labels_2d_classes=C reproduces the one-hot quirk of homophily_tests.py:114-116
    (element-wise compare of one-hot rows, mean over E*C booleans)."""
    nnz, m = int(st["totals"][0]), int(st["totals"][1])
    if labels_2d_classes is None:
        return m / nnz
    c = labels_2d_classes
    return (m * c + (nnz - m) * (c - 2)) / (nnz * c)


def node_homophily_sparse(st):
    """utils/homophily_metrics.py:73-78: loops removed, nodes without a non-loop entry skipped."""
    deg = st["row_nnz_noself"].astype(np.float32)
    sel = deg != 0
    hs = st["row_match_noself"].astype(np.float32)[sel] / deg[sel]
    return float(hs.astype(np.float64).mean())


def compat_matrix(st):
    """utils/homophily_metrics.py:97-101: H = K / rowsum(K) in fp32 (NaN rows when a class has no edge)."""
    k = st["compat"].astype(np.float32)
    with np.errstate(invalid="ignore", divide="ignore"):
        return k / k.sum(1, keepdims=True)


def class_homophily(st, labels):
    """utils/homophily_metrics.py:105-123 ("our_measure")."""
    labels = np.asarray(labels)
    c = int(labels.max()) + 1
    h = compat_matrix(st)
    nz = labels[labels >= 0]
    counts = np.unique(nz, return_counts=True)[1]
    prop = counts.astype(np.float32) / np.float32(nz.shape[0])
    val = np.float32(0)
    for k in range(c):
        t = np.float32(max(h[k, k] - prop[k], 0)) if not np.isnan(h[k, k]) else np.float32("nan")
        if not np.isnan(t):
            val += t
    return float(val / np.float32(c - 1))


def class_distribution(st, labels):
    """utils/homophily_metrics.py:126-147 -> (p, p_bar, pc)."""
    labels = np.asarray(labels)
    n = labels.shape[0]
    p = np.unique(labels, return_counts=True)[1] / n
    tot = np.float32(st["classdeg"].sum())  # == sum(deg) with deg = |P_u| - 1
    p_bar = st["classdeg"].astype(np.float32) / tot
    pc = st["compat"].astype(np.float32) / tot
    p_bar[p_bar == 0] = 1e-8
    pc[pc == 0] = 1e-8
    return p, p_bar.astype(np.float32), pc.astype(np.float32)


def adjusted_homophily(st, labels):
    """utils/homophily_metrics.py:150-155."""
    _, p_bar, _ = class_distribution(st, labels)
    eh = np.float32(edge_homophily_sparse(st))
    s = np.float32((p_bar.astype(np.float32) ** 2).sum())
    return float((eh - s) / (np.float32(1) - s))


def label_informativeness(st, labels):
    """utils/homophily_metrics.py:158-161."""
    _, p_bar, pc = class_distribution(st, labels)
    return float(np.float32(2) - np.float32((pc * np.log(pc)).sum()) / np.float32((p_bar * np.log(p_bar)).sum()))


# dense ("plot") flavour: same integers, different bookkeeping -------------------------------
def edge_homophily_dense(st):
    """utils/homophily_plot.py:43-53: diagonal removed first."""
    return int(st["totals"][5]) / int(st["totals"][4])


def node_homophily_dense(st):
    """utils/homophily_plot.py:81-99: A.nonzero() keeps self loops."""
    deg = st["row_nnz"].astype(np.float32)
    loops = (st["row_nnz"] - st["row_nnz_noself"]).astype(np.float32)
    sel = deg != 0
    hs = (st["row_match_noself"].astype(np.float32) + loops)[sel] / deg[sel]
    return float(hs.astype(np.float64).mean())


def class_homophily_dense(st, labels):
    """utils/homophily_plot.py:126-147: diagonal removed, isolated rows get a self loop."""
    labels = np.asarray(labels)
    st2 = dict(st)
    k = st["compat"].copy()
    iso = np.nonzero(st["row_nnz_noself"] == 0)[0]
    for u in iso:
        if labels[u] >= 0:
            k[labels[u], labels[u]] += 1
    st2["compat"] = k
    return class_homophily(st2, labels)


def adjusted_homophily_dense(st, labels):
    """utils/homophily_plot.py:175-180: edge term WITHOUT loops, p_bar as in the sparse flavour."""
    _, p_bar, _ = class_distribution(st, labels)
    eh = np.float32(edge_homophily_dense(st))
    s = np.float32((p_bar ** 2).sum())
    return float((eh - s) / (np.float32(1) - s))


# ----------------------------------------------------------------------------- LAS tail
def las_from_weights(w, labels, label_onehot=None, hard=None, LP=1, ifsum=1):
    """utils/homophily_metrics.py:201-229 given W (n x C).  labels: int vector of the n rows."""
    w = np.asarray(w)
    labels = np.asarray(labels)
    n, c = w.shape
    if label_onehot is None:
        label_onehot = np.eye(c, dtype=w.dtype)[labels]
    if ifsum != 1:
        cnt = np.bincount(labels, minlength=c).astype(w.dtype)
        w = w / cnt[None, :]
    own = w[np.arange(n), labels]
    if hard is None:
        if ifsum == 1:
            nnodes = n
            degs = (label_onehot @ label_onehot.sum(0)).astype(w.dtype)
        else:
            nnodes, degs = c, np.ones(n, w.dtype)
        if LP == 1:
            with np.errstate(invalid="ignore", divide="ignore"):
                ratio = (own / degs) / ((w.sum(1) - own) / (nnodes - degs))
            ratio[np.isnan(ratio)] = 0
            return float((ratio >= 1).mean())
        return float((((w - w * label_onehot).sum(1) <= 0) & ((w * label_onehot).sum(1) >= 0)).mean())
    if LP == 1:
        return float((np.argmax(w, 1) == labels).mean())
    return float((((w - w * label_onehot).max(1) <= 0) & ((w * label_onehot).sum(1) >= 0)).mean())


def similarity(features, rowptr, col, val, label_onehot, hard=None, LP=1, ifsum=1, idx_train=None, f64=False):
    """utils/homophily_metrics.py:190-229 (and the dense twin utils/homophily_plot.py:189-235 with NTK=None).

    idx_train: bool mask (metrics flavour, :198) or None."""
    h = spmm_csr(rowptr, col, val, features)
    labels = np.argmax(label_onehot, 1)
    onehot = np.asarray(label_onehot, np.float32)
    if idx_train is not None:
        idx = np.asarray(idx_train)
        idx = np.nonzero(idx)[0] if idx.dtype == np.bool_ else idx
        h, labels, onehot = h[idx], labels[idx], onehot[idx]
    c = int(labels.max()) + 1
    w = las_weights(h, labels, c, f64=f64)
    return las_from_weights(w, labels, onehot[:, :c], hard=hard, LP=LP, ifsum=ifsum)


# ----------------------------------------------------------------------------- GNTK / kernel regression
def gntk_kernels(features, rowptr, col, val, sample, n_layers):
    """utils/homophily_metrics.py:232-257 -> (K_G/2, K_X/2) fp32."""
    eps = np.float32(1e-8)
    x = _c(features, np.float32)
    hagg = spmm_csr(rowptr, col, val, x)

    def arc(g):
        d = np.sqrt(np.diag(g))
        nrm = d[:, None] * d[None, :]
        nrm = np.where(nrm > eps, nrm, eps).astype(np.float32)
        if n_layers != 1:
            return g
        with np.errstate(invalid="ignore"):
            ac = np.arccos(g / nrm)
            sq = np.sqrt(np.square(nrm) - np.square(g))
        ac[np.isnan(ac)] = 0
        sq[np.isnan(sq)] = 0
        return (np.float32(1 / math.pi) * (g * (np.float32(math.pi) - ac) + sq)).astype(np.float32)

    return arc(gram(hagg, sample)) / np.float32(2), arc(gram(x, sample)) / np.float32(2)


def kernel_regression_accuracies(features, rowptr, col, val, labels, node_sets, n_layers):
    """The kernel-regression branch of one call of classifier_based_performance_metric, epoch by epoch
    (utils/homophily_metrics.py:283-297, utils/homophily_plot.py:301-316): per epoch the kernels of the epoch's sample
    (gntk_homophily_, :232-257), `K_val_train @ (np.linalg.pinv(K_train_train) @ onehot[train])` in fp32 with numpy's default
    rcond (1e-15: every singular value of an fp32 block is kept), arg-max, mean hit rate on the validation rows.
    node_sets: list of (train node ids, validation node ids) - the reference's boolean masks as ascending ids.
    -> (G_results [epochs], X_results [epochs]) float64, like the reference's per-epoch result vectors.  Pinned per epoch
    by tests/test_oracle_golden.py against tests/golden/kr_epochs.npz."""
    labels = np.asarray(labels, np.int64)
    c = int(labels.max()) + 1
    onehot = np.eye(c, dtype=np.float32)[labels]
    g_res, x_res = [], []
    for train, valid in node_sets:
        train, valid = np.asarray(train, np.int64), np.asarray(valid, np.int64)
        sample = np.sort(np.concatenate([train, valid]))
        pos = {int(v): i for i, v in enumerate(sample)}
        tr = np.array([pos[int(v)] for v in train])
        va = np.array([pos[int(v)] for v in valid])
        accs = []
        for kern in gntk_kernels(features, rowptr, col, val, sample, n_layers):
            alpha = np.linalg.pinv(kern[np.ix_(tr, tr)]) @ onehot[train]
            pred = (kern[np.ix_(va, tr)] @ alpha).argmax(1)
            accs.append(float(np.mean((pred == labels[valid]).astype(np.float32))))
        g_res.append(accs[0])
        x_res.append(accs[1])
    return np.array(g_res), np.array(x_res)


def welch_p_value(g_results, x_results):
    """utils/homophily_metrics.py:335-347: Welch t-test of the epochs' accuracies, folded by which side won more epochs"""
    from scipy.stats import ttest_ind
    g, x = np.asarray(g_results, np.float32), np.asarray(x_results, np.float32)
    _, p = ttest_ind(x, g, axis=0, equal_var=False, nan_policy="propagate")
    return float(p / 2 if np.mean((g > x).astype(np.float32)) <= 0.5 else 1 - p / 2)


def _seq_sum_f32(rows):
    """fp32 sum of the rows of a [n, F] array, row after row (what `np.add.reduce(axis=0)` does on a C-contiguous float32 matrix)"""
    acc = np.zeros(rows.shape[1], np.float32)
    for r in rows:
        acc = (acc + r).astype(np.float32)
    return acc


def gnb_fit(x_train, y_train):
    """Gaussian naive Bayes, fit: the algorithm of `sklearn.naive_bayes.GaussianNB().fit(X, y)` as the reference calls it
    (utils/homophily_metrics.py:296-303, utils/homophily_plot.py:317-324: default priors, var_smoothing = 1e-9) - scikit-learn is a
    third-party dependency that is not part of the reference checkout; the build container holds 1.7.2 on numpy 2.2, which
    produced the golden p-values (tests/golden/make_golden.py).  Restated from its published source (`_partial_fit`,
    `_update_mean_variance`) with the arithmetic made explicit:
      * float32 inputs stay float32: per present class c (ascending) theta_c = fp32 mean, var_c = fp32 population variance of the
        class's rows in their order, both by SEQUENTIAL fp32 sums over the rows (np.mean / np.var, axis 0);
      * epsilon = float32(1e-9) * max_f fp32-var(all rows)[f] in fp32 (numpy 2: a Python float is weak beside a float32 scalar);
      * theta, var are stored as float64; var += epsilon in float64; prior_c = count_c / n.
    -> dict(classes [k], theta [k, F] f64, var [k, F] f64, prior [k] f64, epsilon f32).  Pinned against sklearn itself by
    tests/test_oracle_golden.py (attributes bit for bit)."""
    x = np.ascontiguousarray(x_train, np.float32)
    y = np.asarray(y_train)
    n = x.shape[0]

    def mean_var(rows):
        cnt = np.float32(rows.shape[0])
        mean = (_seq_sum_f32(rows) / cnt).astype(np.float32)
        d = (rows - mean).astype(np.float32)
        return mean, (_seq_sum_f32((d * d).astype(np.float32)) / cnt).astype(np.float32)

    eps = np.float32(np.float32(1e-9) * mean_var(x)[1].max())
    classes = np.unique(y)
    theta = np.zeros((len(classes), x.shape[1]))
    var = np.zeros((len(classes), x.shape[1]))
    count = np.zeros(len(classes))
    for i, c in enumerate(classes):
        rows = x[y == c]
        theta[i], var[i] = mean_var(rows)
        count[i] = rows.shape[0]
    var += eps
    return dict(classes=classes, theta=theta, var=var, prior=count / n, epsilon=eps)


def gnb_joint_log_likelihood(model, x):
    """`GaussianNB._joint_log_likelihood` in float64: log prior_c - 1/2 sum_f log(2 pi var_cf) - 1/2 sum_f (x_f - theta_cf)^2 / var_cf"""
    x = np.asarray(x, np.float32)
    out = []
    for i in range(len(model["classes"])):
        n_ij = -0.5 * np.sum(np.log(2.0 * np.pi * model["var"][i]))
        n_ij = n_ij - 0.5 * np.sum(((x - model["theta"][i]) ** 2) / model["var"][i], 1)
        out.append(np.log(model["prior"][i]) + n_ij)
    return np.array(out).T


def gnb_predict(model, x):
    """`GaussianNB.predict`: classes[arg-max of the joint log likelihood] (the first maximum)"""
    return model["classes"][np.argmax(gnb_joint_log_likelihood(model, x), axis=1)]


def gnb_accuracies(x_feat, x_agg, labels, node_sets):
    """The GNB branch of one call of classifier_based_performance_metric, epoch by epoch (utils/homophily_metrics.py:296-312):
    fit on the raw and on the aggregated features of the epoch's train rows (ascending ids: the reference indexes with boolean
    masks), predict its validation rows, mean hit rate.  -> (G_results, X_results) float64 [epochs]"""
    labels = np.asarray(labels, np.int64)
    g_res, x_res = [], []
    for train, valid in node_sets:
        train, valid = np.sort(np.asarray(train, np.int64)), np.sort(np.asarray(valid, np.int64))
        accs = []
        for m in (x_agg, x_feat):
            pred = gnb_predict(gnb_fit(m[train], labels[train]), m[valid])
            accs.append(float(np.mean((pred == labels[valid]).astype(np.float32))))
        g_res.append(accs[0])
        x_res.append(accs[1])
    return np.array(g_res), np.array(x_res)
