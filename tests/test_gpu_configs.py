"""GPU parity at the sizes of BASELINE configs C4 (squirrel + chameleon real topology, heterophilous, skewed degrees)
and C5 (twitch-gamers scale: 168 114 nodes, ~13.8 M stored entries, F = 7 bf16 features, C = 2).
Real node features / the twitch csv are absent from the reference checkout, so features and labels are
synthetic of the right shape; the topology of C4 is the real one (tests/golden/topo_*.npz)."""
import time

import numpy as np
import pytest
import torch

from _golden import load

pytestmark = pytest.mark.gpu


def _np(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("name,f", [("squirrel", 2089), ("chameleon", 2325)])
def test_c4_real_topology(oracle, name, f):
    from wdg_amd import models, ops
    g0 = load("topo_" + name)
    n = int(g0["n_nodes"])
    rng = np.random.default_rng(17)
    x = (rng.random((n, f), dtype=np.float32) < 0.02) * rng.random((n, f), dtype=np.float32)
    x[np.arange(n), rng.integers(0, f, n)] = 1.0
    labels = rng.integers(0, 5, n)
    g = ops.CsrGraph.from_coo(g0["adj_row"], g0["adj_col"], n, None, ops.COO_ADD_SELF_LOOPS)
    rowptr, col, val = oracle.coo_to_csr(g0["adj_row"], g0["adj_col"], n, None, oracle.ADD_SELF_LOOPS)
    np.testing.assert_array_equal(_np(g.rowptr), rowptr)
    np.testing.assert_array_equal(_np(g.col), col)
    np.testing.assert_array_equal(_np(g.val), val)  # pre-existing loops doubled by A + I
    assert int((val == 2).sum()) == int((g0["adj_row"] == g0["adj_col"]).sum())
    st = ops.edge_label_stats(g, torch.from_numpy(labels))
    ref = oracle.edge_label_stats(rowptr, col, labels, 5)
    for k in ("totals", "row_nnz", "row_nnz_noself", "row_match_noself", "compat", "classdeg"):
        np.testing.assert_array_equal(_np(st[k]), ref[k], err_msg=k)
    xt = torch.from_numpy(x).cuda()
    for mode in (0, 1):
        d = ops.degree_norm(g, mode, ops.PREC_F32)["dinv"]
        y = ops.spmm(g, xt, row_scale=d, col_scale=d if mode else None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            ops.spmm(g, xt, row_scale=d, col_scale=d if mode else None, out=y)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 5
        vhat = oracle.normalised_csr(rowptr, col, val, mode, oracle.PREC_F32)
        want = oracle.spmm_csr(rowptr, col, vhat, x)
        np.testing.assert_allclose(_np(y), want, rtol=1e-5, atol=1e-6 * np.abs(want).max())
        print(f"[C4 {name} mode={mode}] {col.shape[0] / dt / 1e6:.0f} M edges/s, {dt * 1e6:.0f} us "
              f"(plan {ops.spmm_plan(n, n, f, 1, 0)})")
    # GCN-2 vs SGC-1 forward on this topology (config C4): finite, and SGC equals the oracle chain
    adj = models.NormAdj(g, symmetric=1, add_self_loops=False)  # g already holds A + I
    torch.manual_seed(0)
    sgc, gcn = models.SGC1(f, 5).cuda(), models.GCN2(f, 5, nhid=64, dropout=0.0).cuda().eval()
    with torch.no_grad():
        a, b = sgc(adj, xt), gcn(adj, xt)
    assert torch.isfinite(a).all() and torch.isfinite(b).all()
    vhat = oracle.normalised_csr(rowptr, col, val, 1, oracle.PREC_F32)
    want = oracle.gemm(oracle.spmm_csr(rowptr, col, vhat, x), _np(sgc.weight))
    np.testing.assert_allclose(_np(a), want, rtol=1e-5, atol=1e-5 * np.abs(want).max())


def test_c5_twitch_scale(oracle):
    from wdg_amd import ops, synth
    from wdg_amd.utils import homophily_metrics as hm
    from wdg_amd.utils.util_funcs import random_disassortative_splits
    n, e_und, f = 168114, 6797557, 7
    src, dst = synth.random_graph(n, e_und, seed=5)
    flags = ops.COO_SYMMETRISE | ops.COO_BINARISE | ops.COO_ADD_SELF_LOOPS  # to_undirected, then + I
    g = ops.CsrGraph.from_coo(src, dst, n, None, flags)
    rowptr, col, val = oracle.coo_to_csr(src, dst, n, None, oracle.SYMMETRISE | oracle.BINARISE | oracle.ADD_SELF_LOOPS)
    np.testing.assert_array_equal(_np(g.rowptr), rowptr)
    np.testing.assert_array_equal(_np(g.col), col)
    np.testing.assert_array_equal(_np(g.val), val)
    rng = np.random.default_rng(6)
    labels = rng.integers(0, 2, n)
    st = ops.edge_label_stats(g, torch.from_numpy(labels), per_row=False)
    ref = oracle.edge_label_stats(rowptr, col, labels, 2)
    for k in ("totals", "compat", "classdeg"):
        np.testing.assert_array_equal(_np(st[k]), ref[k], err_msg=k)
    # SGC-1 aggregation with bf16 features, fp32 accumulation (config C5)
    x = rng.standard_normal((n, f)).astype(np.float32)
    xb = torch.from_numpy(x).cuda().to(torch.bfloat16)
    d = ops.degree_norm(g, ops.NORM_SYM, ops.PREC_F32)["dinv"]
    y = ops.spmm(g, xb, row_scale=d, col_scale=d)
    assert g.narrow_ws is not None and list(g.narrow_parts) == [2]  # narrow kernel, 32-byte fp32 table in two column ranges
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        ops.spmm(g, xb, row_scale=d, col_scale=d, out=y)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"[C5 twitch-scale] nnz={col.shape[0]} bf16 F=7: {col.shape[0] / dt / 1e9:.2f} G edges/s, {dt * 1e6:.0f} us")
    vhat = oracle.normalised_csr(rowptr, col, val, 1, oracle.PREC_F32)
    want = oracle.spmm_csr(rowptr, col, vhat, _np(xb.float()))
    np.testing.assert_allclose(_np(y), want, rtol=1e-5, atol=1e-6 * np.abs(want).max())
    # random-walk A_hat = D^-1 (A+I): no column scale, so the bf16 rows are gathered as they are (16 bytes per column: the
    # whole table in one L2, one column range); and a two-column product with a column scale (four fp32 per column)
    drw = ops.degree_norm(g, ops.NORM_RW, ops.PREC_F32)["dinv"]
    y_rw = ops.spmm(g, xb, row_scale=drw)
    assert list(g.narrow_parts) == [2]  # (no second plan: one range)
    vrw = oracle.normalised_csr(rowptr, col, val, 0, oracle.PREC_F32)
    want = oracle.spmm_csr(rowptr, col, vrw, _np(xb.float()))
    np.testing.assert_allclose(_np(y_rw), want, rtol=1e-5, atol=1e-6 * np.abs(want).max())
    assert not g.unit_values  # (the random pairs hold a few self loops: A + I stores 2 there, the value stream is read)
    g1 = ops.CsrGraph.from_coo(src, dst, n, None, flags | ops.COO_DROP_SELF_LOOPS)  # binary A + I: the value stream is skipped
    assert g1.unit_values
    r1, c1, v1 = oracle.coo_to_csr(src, dst, n, None, oracle.SYMMETRISE | oracle.BINARISE | oracle.ADD_SELF_LOOPS | oracle.DROP_SELF_LOOPS)
    d1 = ops.degree_norm(g1, ops.NORM_RW, ops.PREC_F32)["dinv"]
    want = oracle.spmm_csr(r1, c1, oracle.normalised_csr(r1, c1, v1, 0, oracle.PREC_F32), _np(xb.float()))
    np.testing.assert_allclose(_np(ops.spmm(g1, xb, row_scale=d1)), want, rtol=1e-5, atol=1e-6 * np.abs(want).max())
    x2 = torch.from_numpy(x[:, :2].copy()).cuda()
    y2 = ops.spmm(g, x2, row_scale=d, col_scale=d)
    want = oracle.spmm_csr(rowptr, col, vhat, x[:, :2].copy())
    np.testing.assert_allclose(_np(y2), want, rtol=1e-5, atol=1e-6 * np.abs(want).max())
    # aggregation homophily with the 10 000-node class-balanced sample (homophily_tests.py:124-131)
    lab_t = torch.from_numpy(labels)
    onehot = torch.eye(2)[lab_t]
    torch.manual_seed(3)
    mask, _, _ = random_disassortative_splits(lab_t, lab_t.max() + 1, 10000 / n)
    adj_raw = ops.CsrGraph.from_coo(src, dst, n, None, ops.COO_SYMMETRISE | ops.COO_BINARISE)
    r2, c2, v2 = oracle.coo_to_csr(src, dst, n, None, oracle.SYMMETRISE | oracle.BINARISE)
    for hard in (None, 1):
        got = float(hm.similarity(onehot, adj_raw, onehot, hard=hard, LP=1, idx_train=mask))
        want = oracle.similarity(_np(onehot), r2, c2, v2, _np(onehot), hard=hard, idx_train=_np(mask), f64=True)
        assert abs(got - want) <= 1.01 / int(mask.sum())


@pytest.mark.parametrize("m,n,k,pad", [(2708, 7, 1433, 0), (5201, 5, 2089, 3), (1, 1, 1, 0), (9, 8, 4100, 0), (1000, 2, 7, 1), (17, 3, 2048, 0)])
def test_skinny_gemm_against_the_oracle(oracle, m, n, k, pad):
    """wdg_gemm_skinny_f32 (N <= 8: the SGC-1 head): any K (chunks of 2048 through LDS), unaligned rows (lda = K + pad), bias +
    relu, ragged last row group; fp32 within 1e-5 of the oracle's k-ordered product, bitwise reproducible"""
    from wdg_amd import ops
    rng = np.random.default_rng(m + 3 * n + k)
    store = rng.standard_normal((m, k + pad)).astype(np.float32)
    a, b, bias = store[:, :k], rng.standard_normal((k, n)).astype(np.float32), rng.standard_normal(n).astype(np.float32)
    at = torch.from_numpy(store).cuda()[:, :k]
    bt, biast = torch.from_numpy(b).cuda(), torch.from_numpy(bias).cuda()
    got = ops.gemm_skinny(at, bt)
    want = oracle.gemm(np.ascontiguousarray(a), b)
    np.testing.assert_allclose(_np(got), want, rtol=1e-5, atol=1e-5 * max(np.abs(want).max(), 1e-30))
    want = oracle.gemm(np.ascontiguousarray(a), b, bias, relu=True)
    np.testing.assert_allclose(_np(ops.gemm_skinny(at, bt, bias=biast, relu=True)), want, rtol=1e-5, atol=1e-5 * max(np.abs(want).max(), 1e-30))
    assert torch.equal(ops.gemm_skinny(at, bt), got)
    with pytest.raises(ValueError):
        from wdg_amd._lib import check, lib
        from wdg_amd.ops import _ptr, stream_handle
        c = torch.empty((m, 9), device="cuda")
        check(lib.wdg_gemm_skinny_f32(_ptr(at), k + pad, _ptr(torch.zeros((k, 9), device="cuda")), 9, None, 0, _ptr(c), 9, m, 9, k, stream_handle()), "x")


@pytest.mark.parametrize("name", ["cora", "squirrel"])
def test_sgc1_forward_in_both_orders(oracle, name):
    """SGC-1 inference: (A_hat X) W and A_hat (X W) (head first: skinny product, then an aggregation of C columns instead of F)
    give the oracle's logits within 1e-5 - BASELINE configs[0] (Cora) and [3] (squirrel topology)"""
    from wdg_amd import models, ops
    rng = np.random.default_rng(5)
    if name == "cora":
        g0 = load("real_cora")
        n, f, c = int(g0["n_nodes"]), int(g0["n_feat"]), 7
        x = np.zeros((n, f), np.float32)
        x[np.repeat(np.arange(n), np.diff(g0["feat_indptr"])), g0["feat_indices"]] = g0["featn_data"]
        src, dst = g0["adj_row"], g0["adj_col"]
    else:
        g0 = load("topo_squirrel")
        n, f, c = int(g0["n_nodes"]), 2089, 5
        x = ((rng.random((n, f), dtype=np.float32) < 0.02) * rng.random((n, f), dtype=np.float32)).astype(np.float32)
        src, dst = g0["adj_row"], g0["adj_col"]
    g = ops.CsrGraph.from_coo(src, dst, n, None, ops.COO_ADD_SELF_LOOPS)
    rowptr, col, val = oracle.coo_to_csr(src, dst, n, None, oracle.ADD_SELF_LOOPS)
    xt = torch.from_numpy(x).cuda()
    for symmetric in (0, 1):
        adj = models.NormAdj(g, symmetric=symmetric, add_self_loops=False)
        torch.manual_seed(1)
        sgc = models.SGC1(f, c).cuda().eval()
        with torch.no_grad():
            head_first = sgc(adj, xt)                      # eval mode, nothing cached: A_hat (X W)
            agg_first = sgc(adj, xt, order="agg_first")    # (A_hat X) W
            cached = sgc(adj, xt)                          # A_hat X is cached now: the default takes it
        vhat = oracle.normalised_csr(rowptr, col, val, symmetric, oracle.PREC_F32)
        w = _np(sgc.weight)
        want = oracle.gemm(oracle.spmm_csr(rowptr, col, vhat, x), w)
        tol = dict(rtol=1e-5, atol=1e-5 * np.abs(want).max())
        np.testing.assert_allclose(_np(agg_first), want, **tol)
        np.testing.assert_allclose(_np(head_first), want, **tol)
        np.testing.assert_allclose(_np(head_first), oracle.spmm_csr(rowptr, col, vhat, oracle.gemm(x, w)), **tol)
        assert torch.equal(cached, agg_first)


def test_graphed_inference_replays_the_eager_forward_bitwise():
    """models.graphed_inference: GCN-2 and SGC-1 (inference order) on the squirrel topology captured as one hipGraph each - the replay
    recomputes the logits in place, bit for bit the eager forward's, also after the features changed (same tensor, new values)"""
    from wdg_amd import models, ops
    rng = np.random.default_rng(9)
    g0 = load("topo_squirrel")
    n, f, c = int(g0["n_nodes"]), 2089, 5
    g = ops.CsrGraph.from_coo(g0["adj_row"], g0["adj_col"], n, None, ops.COO_ADD_SELF_LOOPS)
    adj = models.NormAdj(g, symmetric=1, add_self_loops=False)
    x = torch.from_numpy(((rng.random((n, f), dtype=np.float32) < 0.02) * rng.random((n, f), dtype=np.float32)).astype(np.float32)).cuda()
    x2 = torch.from_numpy(rng.random((n, f), dtype=np.float32)).cuda()
    torch.manual_seed(2)
    for model, kw in ((models.GCN2(f, c, nhid=64, dropout=0.5).cuda(), {}), (models.SGC1(f, c).cuda(), {"order": "head_first"})):
        replay, logits = models.graphed_inference(model, adj, x, **kw)
        for values in (x.clone(), x2):
            x.copy_(values)
            with torch.no_grad():
                want = model(adj, x, **kw).clone()
            logits.fill_(float("nan"))
            replay()
            torch.cuda.synchronize()
            assert torch.equal(logits, want), type(model).__name__


@pytest.mark.parametrize("m,n,k,splits", [(5201, 64, 2089, 0), (2277, 64, 2325, 0), (300, 16, 1500, 4), (129, 33, 1024, 16), (5, 3, 4096, 7)])
def test_split_k_gemm_against_the_oracle(oracle, monkeypatch, m, n, k, splits):
    """wdg_gemm_splitk_f32 (few output tiles, K >= 1024: the first layer of a GCN on squirrel / chameleon): partial products over
    ranges of K added in order - within 1e-5 of the oracle's k-ordered product, bitwise reproducible, bias + relu applied after
    the reduction, ragged last range; splits = 0: what ops.gemm picks by itself"""
    from wdg_amd import ops
    from wdg_amd._lib import lib
    rng = np.random.default_rng(m + n + k)
    a, b, bias = (rng.standard_normal(s_).astype(np.float32) for s_ in ((m, k), (k, n), (n,)))
    at, bt, biast = (torch.from_numpy(v).cuda() for v in (a, b, bias))
    if splits:
        monkeypatch.setenv("WDG_GEMM_SPLITK", str(splits))
    assert int(lib.wdg_gemm_splitk_plan(m, n, k)) == (splits if splits else int(lib.wdg_gemm_splitk_plan(m, n, k)))
    assert int(lib.wdg_gemm_splitk_plan(m, n, k)) > 1
    got = ops.gemm(at, bt, bias=biast, relu=True)
    want = np.maximum(oracle.gemm(a, b) + bias, 0)
    np.testing.assert_allclose(_np(got), want, rtol=1e-5, atol=1e-5 * np.abs(want).max())
    assert torch.equal(got, ops.gemm(at, bt, bias=biast, relu=True))
    plain = ops.gemm(at, bt)  # no bias, no activation; into a view with a leading dimension
    np.testing.assert_allclose(_np(plain), oracle.gemm(a, b), rtol=1e-5, atol=1e-5 * np.abs(want).max())
    store = torch.full((m, n + 3), -7.0, device="cuda")
    ops.gemm(at, bt, bias=biast, out=store[:, :n])
    assert torch.equal(store[:, :n], ops.gemm(at, bt, bias=biast)) and float(store[:, n:].min()) == -7.0 == float(store[:, n:].max())
    monkeypatch.setenv("WDG_GEMM_SPLITK", "1")  # never: the single chain
    one = ops.gemm(at, bt, bias=biast, relu=True)
    np.testing.assert_allclose(_np(got), _np(one), rtol=1e-5, atol=1e-5 * np.abs(want).max())


@pytest.mark.parametrize("f,dtype,scaled,ld_pad", [(1, "f32", True, 0), (3, "f32", False, 5), (4, "bf16", True, 0), (5, "bf16", False, 3),
                                                    (8, "bf16", False, 0), (8, "bf16", True, 0), (7, "f32", True, 1)])
def test_narrow_kernel_source_tables(oracle, f, dtype, scaled, ld_pad):
    """every source table of the narrow kernel (csrc/spmm_narrow.hip): four fp32 per column (F <= 4, any dtype, with or without
    a column scale), the bf16 row as it is (bf16 sources, no column scale), eight fp32 (the rest) - rows of X that are not whole
    16-byte units (ld = F + pad), explicit values beside unit ones, an empty row, a 3 000-entry hub row"""
    from wdg_amd import ops
    from wdg_amd._lib import lib
    rng = np.random.default_rng(100 * f + ld_pad)
    n = 40000
    src = np.concatenate([rng.integers(0, n, 400000), np.full(3000, 7), np.arange(n - 5)])
    dst = np.concatenate([rng.integers(0, n, 400000), rng.integers(0, n, 3000), np.arange(5, n)])
    keep = src != 11  # row 11 stays empty
    src, dst = src[keep], dst[keep]
    val = rng.random(src.shape[0]).astype(np.float32) + 0.5
    store = rng.standard_normal((n, f + ld_pad)).astype(np.float32)
    for use_val in (False, True):
        g = ops.CsrGraph.from_coo(src, dst, n, val if use_val else None, ops.COO_BINARISE if not use_val else 0)
        assert g.unit_values == (not use_val)
        rowptr, col, v = oracle.coo_to_csr(src, dst, n, val if use_val else None, oracle.BINARISE if not use_val else 0)
        xt = torch.from_numpy(store).cuda()
        x_np = store[:, :f]
        if dtype == "bf16":
            xt = xt.to(torch.bfloat16)
            x_np = _np(xt.float())[:, :f]
        xt = xt[:, :f]
        rs = torch.from_numpy(rng.random(n).astype(np.float32) + 0.5).cuda()
        cs = torch.from_numpy(rng.random(n).astype(np.float32) + 0.5).cuda() if scaled else None
        want_bytes = 16 if (f <= 4 or (dtype == "bf16" and not scaled)) else 32
        assert int(lib.wdg_spmm_narrow_col_bytes(f, int(dtype == "bf16"), int(scaled))) == want_bytes
        y = ops.spmm(g, xt, row_scale=rs, col_scale=cs)
        assert g.narrow_ws is not None
        vv = v * (_np(cs)[col] if scaled else 1.0)
        want = oracle.spmm_csr(rowptr, col, vv.astype(np.float32), np.ascontiguousarray(x_np)) * _np(rs)[:, None]
        np.testing.assert_allclose(_np(y), want, rtol=2e-5, atol=2e-5 * np.abs(want).max())
        assert torch.equal(y, ops.spmm(g, xt, row_scale=rs, col_scale=cs))
        assert float(y[11].abs().max()) == 0.0
