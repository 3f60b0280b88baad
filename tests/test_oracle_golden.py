"""Pins the CPU oracle (oracle/) to golden vectors captured from the real reference.

CPU-only.  Every oracle function used as a checker in the GPU parity tests is
first proven here against `tests/golden/*.npz` (see tests/golden/make_golden.py).
"""
import numpy as np
import pytest

from _golden import REAL, SYN, dense_features, load


def _close(a, b, rtol=1e-6, atol=1e-7):
    np.testing.assert_allclose(np.asarray(a, np.float64), np.asarray(b, np.float64), rtol=rtol, atol=atol)


# --------------------------------------------------------------------------- indexing (bit exact)
@pytest.mark.parametrize("name", REAL)
def test_csr_build_selfloops_bitexact(oracle, name):
    g = load("real_" + name)
    n = int(g["n_nodes"])
    # reference: eye + A, then coalesce (homophily_tests.py:83-85)
    rowptr, col, val = oracle.coo_to_csr(g["adj_row"], g["adj_col"], n, g["adj_val"], oracle.ADD_SELF_LOOPS)
    rows = np.repeat(np.arange(n, dtype=np.int32), np.diff(rowptr))
    np.testing.assert_array_equal(rows, g["small_rw_row"])
    np.testing.assert_array_equal(col, g["small_rw_col"])
    # a pre-existing self loop must have weight 2
    loops_in = int((g["adj_row"] == g["adj_col"]).sum())
    assert int(((rows == col) & (val == 2)).sum()) == loops_in
    # large path (scipy COO order is row-major sorted too)
    np.testing.assert_array_equal(rows, g["large_sym_row"])
    np.testing.assert_array_equal(col, g["large_sym_col"])


@pytest.mark.parametrize("name", REAL)
@pytest.mark.parametrize("tag,mode", [("rw", 0), ("sym", 1)])
def test_normalised_values(oracle, name, tag, mode):
    g = load("real_" + name)
    n = int(g["n_nodes"])
    rowptr, col, val = oracle.coo_to_csr(g["adj_row"], g["adj_col"], n, g["adj_val"], oracle.ADD_SELF_LOOPS)
    v32 = oracle.normalised_csr(rowptr, col, val, mode, oracle.PREC_F32)
    v64 = oracle.normalised_csr(rowptr, col, val, mode, oracle.PREC_F64)
    _close(v32, g[f"small_{tag}_val"], rtol=5e-7, atol=0)   # torch fp32 path, a few ulp (powf vs torch.pow)
    _close(v64, g[f"large_{tag}_val"], rtol=1.2e-7, atol=0)  # scipy fp64 path cast to fp32, <= 1 ulp
    # degree is an exact integer count
    _, cnt, _ = oracle.degree_norm(rowptr, val, mode, oracle.PREC_F64)
    np.testing.assert_array_equal(cnt, np.bincount(g["small_rw_row"], minlength=n))


@pytest.mark.parametrize("name", REAL)
def test_feature_row_normalise(oracle, name):
    g = load("real_" + name)
    x = dense_features(g)
    rr = np.repeat(np.arange(x.shape[0]), np.diff(g["feat_indptr"]))
    xn = oracle.row_l1_normalise(x)
    _close(xn[rr, g["feat_indices"]], g["featn_data"], rtol=2e-7, atol=0)
    xl = oracle.row_l1_normalise(x, use_abs=True)
    _close(xl[rr, g["feat_indices"]], g["featl1_data"], rtol=2e-7, atol=0)


# --------------------------------------------------------------------------- aggregation
@pytest.mark.parametrize("name", REAL)
@pytest.mark.parametrize("path,tag", [("small", "rw"), ("small", "sym"), ("large", "rw"), ("large", "sym")])
def test_spmm_real(oracle, name, path, tag):
    g = load("real_" + name)
    n = int(g["n_nodes"])
    x = dense_features(g, "featn_data" if path == "small" else "featl1_data")
    # use the reference's own normalised values: isolates the SpMM itself
    rowptr, col, val = oracle.coo_to_csr(g[f"{path}_{tag}_row"], g[f"{path}_{tag}_col"], n, g[f"{path}_{tag}_val"])
    y = oracle.spmm_csr(rowptr, col, val, x)
    rows = g["sample_rows"]
    # <= 1e-5 relative to the row scale (north-star tolerance), observed ~1e-7
    scale = np.abs(g[f"{path}_{tag}_y_rows"]).max()
    np.testing.assert_allclose(y[rows], g[f"{path}_{tag}_y_rows"], rtol=1e-5, atol=1e-6 * scale)
    _close(y.sum(1, dtype=np.float64), g[f"{path}_{tag}_y_rowsum"], rtol=1e-5, atol=1e-7)
    _close(y.sum(0, dtype=np.float64), g[f"{path}_{tag}_y_colsum"], rtol=1e-5, atol=1e-7)
    _close(np.linalg.norm(y.astype(np.float64)), g[f"{path}_{tag}_y_fro"], rtol=1e-6)
    # fp64-accumulated yardstick agrees as well
    y64 = oracle.spmm_csr(rowptr, col, val, x, f64acc=True)
    np.testing.assert_allclose(y, y64, rtol=1e-5, atol=1e-6 * scale)


@pytest.mark.parametrize("name", SYN)
def test_spmm_synthetic_end_to_end(oracle, name):
    """synthetic_plot.py:81-92 pipeline: A + I -> D^-1 -> dense spmm, from raw inputs."""
    g = load(name)
    n = int(g["n_nodes"])
    rowptr, col, val = oracle.coo_to_csr(g["adj_row"], g["adj_col"], n, None, oracle.ADD_SELF_LOOPS)
    np.testing.assert_array_equal(np.repeat(np.arange(n, dtype=np.int32), np.diff(rowptr)), g["norm_row"])
    np.testing.assert_array_equal(col, g["norm_col"])
    vhat = oracle.normalised_csr(rowptr, col, val, oracle.NORM_RW, oracle.PREC_F32)
    _close(vhat, g["norm_val"], rtol=1.2e-7, atol=0)
    x = oracle.row_l1_normalise(dense_features(g))
    y = oracle.spmm_csr(rowptr, col, vhat, x)
    scale = np.abs(g["y_rows"]).max()
    np.testing.assert_allclose(y[g["sample_rows"]], g["y_rows"], rtol=1e-5, atol=1e-6 * scale)
    _close(np.linalg.norm(y.astype(np.float64)), g["y_fro"], rtol=1e-6)
    # known answers: row-regular out-degree, rows of A_hat sum to one
    d = np.diff(rowptr)
    assert d.min() == d.max()
    _close(np.add.reduceat(vhat, rowptr[:-1]), np.ones(n), rtol=1e-6)


# --------------------------------------------------------------------------- edge / label metrics
@pytest.mark.parametrize("name", REAL)
def test_sparse_flavour_metrics(oracle, name):
    g = load("real_" + name)
    n, labels = int(g["n_nodes"]), g["labels"]
    c = int(labels.max()) + 1
    rowptr, col, _ = oracle.coo_to_csr(g["small_rw_row"], g["small_rw_col"], n, g["small_rw_val"])
    st = oracle.edge_label_stats(rowptr, col, labels, c)
    _close(oracle.edge_homophily_sparse(st), g["m_edge_homo"], rtol=1e-6)
    _close(oracle.edge_homophily_sparse(st, labels_2d_classes=c), g["m_edge_homo_onehot_quirk"], rtol=1e-6)
    _close(oracle.node_homophily_sparse(st), g["m_node_homo"], rtol=1e-6)
    _close(oracle.class_homophily(st, labels), g["m_class_homo"], rtol=1e-5, atol=1e-7)
    _close(oracle.adjusted_homophily(st, labels), g["m_adj_homo"], rtol=1e-5, atol=1e-7)
    _close(oracle.label_informativeness(st, labels), g["m_label_info"], rtol=2e-4, atol=2e-7)
    p, p_bar, pc = oracle.class_distribution(st, labels)
    _close(p, g["cd_p"], rtol=1e-6)
    _close(p_bar, g["cd_p_bar"], rtol=1e-6)
    _close(pc, g["cd_pc"], rtol=1e-6)
    h = oracle.compat_matrix(st)
    np.testing.assert_allclose(h, g["compat_H"], rtol=1e-6, equal_nan=True)


@pytest.mark.parametrize("name", SYN)
def test_dense_flavour_metrics(oracle, name):
    g = load(name)
    n, labels = int(g["n_nodes"]), g["labels"]
    c = int(labels.max()) + 1
    rowptr, col, _ = oracle.coo_to_csr(g["norm_row"], g["norm_col"], n, g["norm_val"])
    st = oracle.edge_label_stats(rowptr, col, labels, c)
    _close(oracle.edge_homophily_dense(st), g["m_edge_homo"], rtol=1e-6)
    _close(oracle.node_homophily_dense(st), g["m_node_homo"], rtol=1e-6)
    _close(oracle.class_homophily_dense(st, labels), g["m_class_homo"], rtol=1e-5, atol=1e-7)
    _close(oracle.adjusted_homophily_dense(st, labels), g["m_adj_homo"], rtol=1e-4, atol=2e-7)
    _close(oracle.label_informativeness(st, labels), g["m_label_info"], rtol=2e-3, atol=2e-6)
    # known answer: edge homophily of the generator is k / int(k/h) exactly
    k = int(st["row_match_noself"][0])
    assert (st["row_match_noself"] == k).all()
    assert oracle.edge_homophily_dense(st) == k / (np.diff(rowptr)[0] - 1)


# --------------------------------------------------------------------------- LAS / aggregation homophily
@pytest.mark.parametrize("name", REAL)
def test_aggregation_homophily_real(oracle, name):
    g = load("real_" + name)
    n, labels = int(g["n_nodes"]), g["labels"]
    c = int(labels.max()) + 1
    onehot = np.eye(c, dtype=np.float32)[labels]
    rowptr, col, val = oracle.coo_to_csr(g["adj_row"], g["adj_col"], n, g["adj_val"])
    tol = 1.01 / n  # at most one borderline node (SURVEY 7.2 "threshold metrics")
    soft = oracle.similarity(onehot, rowptr, col, val, onehot, hard=None)
    hard = oracle.similarity(onehot, rowptr, col, val, onehot, hard=1)
    assert abs((2 * soft - 1) - g["m_agg_soft"]) <= 2 * tol
    assert abs((2 * hard - 1) - g["m_agg_hard"]) <= 2 * tol
    mask = g["las_mask"]
    tolm = 1.01 / mask.sum()
    assert abs(oracle.similarity(onehot, rowptr, col, val, onehot, idx_train=mask) - g["m_agg_soft_masked"]) <= tolm
    assert abs(oracle.similarity(onehot, rowptr, col, val, onehot, hard=1, idx_train=mask)
               - g["m_agg_hard_masked"]) <= tolm
    # fp64 middle-product form agrees (integer-valued H: exact)
    assert abs(oracle.similarity(onehot, rowptr, col, val, onehot, f64=True) - soft) <= tol


@pytest.mark.parametrize("name", ["cora", "texas"])
def test_similarity_real_valued_features(oracle, name):
    g = load("real_" + name)
    n, labels = int(g["n_nodes"]), g["labels"]
    c = int(labels.max()) + 1
    onehot = np.eye(c, dtype=np.float32)[labels]
    x = dense_features(g, "featn_data")
    rowptr, col, val = oracle.coo_to_csr(g["small_rw_row"], g["small_rw_col"], n, g["small_rw_val"])
    tol = 3.01 / n
    assert abs(oracle.similarity(x, rowptr, col, val, onehot) - g["m_sim_feat_soft"]) <= tol
    assert abs(oracle.similarity(x, rowptr, col, val, onehot, hard=1) - g["m_sim_feat_hard"]) <= tol


@pytest.mark.parametrize("name", SYN)
def test_las_synthetic(oracle, name):
    g = load(name)
    n, labels = int(g["n_nodes"]), g["labels"]
    onehot = np.eye(int(labels.max()) + 1, dtype=np.float32)[labels]
    rowptr, col, val = oracle.coo_to_csr(g["norm_row"], g["norm_col"], n, g["norm_val"])
    tol = 2.01 / n
    assert abs(oracle.similarity(onehot, rowptr, col, val, onehot) - g["m_soft_las"]) <= tol
    assert abs(oracle.similarity(onehot, rowptr, col, val, onehot, hard=1) - g["m_hard_las"]) <= tol


# --------------------------------------------------------------------------- GNTK kernels
@pytest.mark.parametrize("name", ["cora", "film"])
@pytest.mark.parametrize("nl", [0, 1])
def test_gntk_kernels(oracle, name, nl):
    g = load("real_" + name)
    n = int(g["n_nodes"])
    x = dense_features(g)
    rowptr, col, val = oracle.coo_to_csr(g["adj_row"], g["adj_col"], n, g["adj_val"])
    kg, kx = oracle.gntk_kernels(x, rowptr, col, val, g["gntk_sample"], nl)
    from _golden import assert_gntk_close
    assert_gntk_close(kg, g[f"gntk_KG_l{nl}"], g["gntk_KG_l0"], nl)
    assert_gntk_close(kx, g[f"gntk_KX_l{nl}"], g["gntk_KX_l0"], nl)


# --------------------------------------------------------------------------- plain helpers
def test_gemm_and_transpose_spmm(oracle):
    rng = np.random.default_rng(0)
    a, b = rng.standard_normal((37, 53), np.float32), rng.standard_normal((53, 11), np.float32)
    np.testing.assert_allclose(oracle.gemm(a, b), a @ b, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(oracle.gemm(a, b, bias=b[0], relu=True), np.maximum(a @ b + b[0], 0), rtol=1e-5,
                               atol=1e-5)
    src, dst = rng.integers(0, 40, 300), rng.integers(0, 40, 300)
    rowptr, col, val = oracle.coo_to_csr(src, dst, 40, rng.random(300, np.float32))
    x = rng.standard_normal((40, 9), np.float32)
    dense = np.zeros((40, 40), np.float64)
    np.add.at(dense, (np.repeat(np.arange(40), np.diff(rowptr)), col), val)
    np.testing.assert_allclose(oracle.spmm_csr(rowptr, col, val, x), dense @ x, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(oracle.spmm_csr_t(rowptr, col, val, x, 40), dense.T @ x, rtol=1e-5, atol=1e-5)


def test_coo_to_csr_flag_semantics(oracle):
    src = np.array([0, 0, 1, 2, 2, 2, 3])
    dst = np.array([1, 1, 0, 2, 0, 3, 3])
    # coalesce: duplicates summed
    rp, col, val = oracle.coo_to_csr(src, dst, 4)
    assert rp.tolist() == [0, 1, 2, 5, 6] and col.tolist() == [1, 0, 0, 2, 3, 3] and val[0] == 2
    # to_undirected: both directions, unique pattern, loops kept once
    rp, col, val = oracle.coo_to_csr(src, dst, 4, None, oracle.SYMMETRISE | oracle.BINARISE)
    assert col.tolist() == [1, 2, 0, 0, 2, 3, 2, 3] and (val == 1).all()
    # + I: existing loops become 2, missing ones 1
    rp, col, val = oracle.coo_to_csr(src, dst, 4, None, oracle.SYMMETRISE | oracle.BINARISE | oracle.ADD_SELF_LOOPS)
    diag = val[np.repeat(np.arange(4), np.diff(rp)) == col]
    assert diag.tolist() == [1, 1, 2, 2]
    with pytest.raises(IndexError):
        oracle.coo_to_csr(np.array([5]), np.array([0]), 4)
    # empty graph
    rp, col, val = oracle.coo_to_csr(np.zeros(0, np.int64), np.zeros(0, np.int64), 3, None, oracle.ADD_SELF_LOOPS)
    assert rp.tolist() == [0, 1, 2, 3] and col.tolist() == [0, 1, 2]


# --------------------------------------------------------------------------- kernel regression, per epoch
def _kr_inputs(oracle, name):
    """what the reference call of make_golden_kr.py was handed: (features, rowptr, col, val, labels)"""
    g = load(name.replace("_s200", ""))
    n = int(g["n_nodes"])
    if name.startswith("syn_"):  # synthetic_plot.py:81-92: row-L1 features, D^-1 (A + I)
        rowptr, col, val = oracle.coo_to_csr(g["adj_row"], g["adj_col"], n, None, oracle.ADD_SELF_LOOPS)
        val = oracle.normalised_csr(rowptr, col, val, oracle.NORM_RW, oracle.PREC_F32)
        x = oracle.row_l1_normalise(dense_features(g))
    else:                        # homophily_tests.py:133-137: raw adjacency, raw features
        rowptr, col, val = oracle.coo_to_csr(g["adj_row"], g["adj_col"], n, g["adj_val"])
        x = dense_features(g)
    return x, rowptr, col, val, g["labels"]


@pytest.mark.parametrize("name", ["syn_800_0.5_0", "syn_4000_0.2_0", "real_texas", "real_cora", "real_cora_s200"])
def test_kernel_regression_epochs(oracle, name):
    """The oracle's per-epoch restatement of the kernel-regression metric against what the reference computed in each epoch
    (tests/golden/kr_epochs.npz: node sets and accuracies recorded inside the reference's own call).  Exact or one validation
    row: the Gram's summation order differs from torch.mm's, and arg-max ties sit on rounding."""
    from _golden import load_kr
    kr = load_kr(name)
    x, rowptr, col, val, labels = _kr_inputs(oracle, name)
    for nl, clf in enumerate(("kernel_reg0", "kernel_reg1")):
        rec = kr[clf]
        sets = rec["node_sets"][:3]
        g_res, x_res = oracle.kernel_regression_accuracies(x, rowptr, col, val, labels, sets, nl)
        n_val = np.array([len(v) for _, v in sets], np.float64)
        assert (np.abs(g_res - rec["g_results"][:3]) * n_val <= 1.01).all(), (clf, g_res, rec["g_results"][:3])
        assert (np.abs(x_res - rec["x_results"][:3]) * n_val <= 1.01).all(), (clf, x_res, rec["x_results"][:3])
        assert abs(oracle.welch_p_value(rec["g_results"], rec["x_results"]) - rec["p"]) <= 1e-9


@pytest.mark.parametrize("name", ["syn_800_0.05_0", "syn_4000_0.15_2", "real_texas", "real_cora", "real_cora_s200"])
def test_epoch_node_sets_reproduce_the_reference_stream(name):
    """wdg_amd.utils.util_funcs.kernel_regression_epoch_indices (host logic of the product) draws, under the same
    torch.manual_seed, exactly the node sets the reference drew in every epoch (its masks as ascending node ids)."""
    import torch
    from _golden import load_kr
    from wdg_amd.utils.util_funcs import kernel_regression_epoch_indices
    kr = load_kr(name)
    labels = torch.from_numpy(load(name.replace("_s200", ""))["labels"])
    for clf in ("kernel_reg0", "kernel_reg1"):
        torch.manual_seed(kr["seed"])
        got = kernel_regression_epoch_indices(labels, kr["sample_max"], kr["epochs"])
        for (tr, va), (tr0, va0) in zip(got, kr[clf]["node_sets"]):
            assert np.array_equal(tr.numpy(), tr0) and np.array_equal(va.numpy(), va0)


@pytest.mark.parametrize("name", SYN)
def test_reference_pattern_restatement_reproduces_the_references_scalars(name):
    """oracle/ref_pattern.py - the dense-tensor / Python-loop call pattern of utils/homophily_plot.py that bench.py's cpu_baseline
    times - against the scalars the real reference computed for the same graph: exact to fp32 printing (same operations, same
    order)."""
    import torch
    from oracle import ref_pattern as rp
    d = load(name)
    n, lab = int(d["n_nodes"]), np.asarray(d["labels"])
    adj = rp.normalised_dense_adjacency(d["adj_row"], d["adj_col"], n)
    got = rp.six_scalars(adj, torch.eye(int(lab.max()) + 1)[torch.as_tensor(lab)])
    want = [float(d["m_" + k]) for k in ("edge_homo", "node_homo", "class_homo", "adj_homo", "label_info", "soft_las")]
    # (label informativeness - index 4 - is a difference of fp32 logarithm sums of ~0.005: torch's CPU reductions order their sums by the
    # host's vector width and threads, one fp32 rounding of the sums - 2.4e-7 absolute - is 5e-5 relative there: seen on a Zen 5 host)
    np.testing.assert_allclose(np.delete(got, 4), np.delete(want, 4), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(got[4], want[4], rtol=1e-6, atol=1e-6)
