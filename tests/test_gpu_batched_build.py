"""The cold (one-pass) sweep path: a shard's graphs built together (ops.GraphBatch = one block-diagonal COO -> CSR build + the
batched SELL-16 build, one host read-back) against the per-graph builds, and the epochs' node sets drawn on the device
(ops.KrSets) against a host restatement of the documented generator and against the reference routine's statistics."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _coos(rng, shapes):
    out = []
    for n, e in shapes:
        src, dst = rng.integers(0, max(n, 1), e), rng.integers(0, max(n, 1), e)
        out.append((src.astype(np.int64), dst.astype(np.int64), n))
    return out


@pytest.mark.parametrize("flags", ["ADD_SELF_LOOPS", "SYMMETRISE|BINARISE", "0"])
def test_graph_batch_equals_per_graph_builds(flags):
    """rowptr / col / val, the degrees and every array of the SELL-16 copy: bit for bit what CsrGraph.from_coo + ensure_quad
    build for each graph on its own - graphs of different sizes, an empty one, duplicates, self loops, a two-block graph"""
    from wdg_amd import ops, synth
    rng = np.random.default_rng(7)
    fl = {"ADD_SELF_LOOPS": ops.COO_ADD_SELF_LOOPS, "SYMMETRISE|BINARISE": ops.COO_SYMMETRISE | ops.COO_BINARISE, "0": 0}[flags]
    coos = _coos(rng, [(300, 2000), (1, 3), (2000, 30000), (17, 0), (3000, 9000), (64, 4000)])
    for h, k in ((0.2, 10), (0.5, 2)):
        src, dst, _ = synth.regular_graph(2000, 5, k, h, 3)
        coos.append((src, dst, 2000))
    gb = ops.GraphBatch(coos, fl, quad=True)
    dn = gb.degree_norm(ops.NORM_SYM, ops.PREC_F32)
    assert len(gb.graphs) == len(coos)
    for (src, dst, n), g, d in zip(coos, gb.graphs, dn):
        ref = ops.CsrGraph.from_coo(src, dst, n, None, fl)
        assert torch.equal(g.rowptr, ref.rowptr) and torch.equal(g.col, ref.col) and torch.equal(g.val, ref.val)
        dr = ops.degree_norm(ref, ops.NORM_SYM, ops.PREC_F32)
        for key in ("rowsum", "cnt", "dinv", "dinv64"):
            assert torch.equal(d[key], dr[key]), key
        has = ref.ensure_quad()
        assert bool(g.quad) == has
        if has:
            q, r = g.quad, ref.quad
            for key in ("ext", "perm", "rows"):
                assert torch.equal(q[key], r[key]), key
            assert torch.equal(q["col"][:q["chunks"] * 256], r["col"][:r["chunks"] * 256])
            for key in ("block_cols", "n_blocks", "n_entries", "n_su", "split", "chunks", "n_slices"):
                assert q[key] == r[key], key
            assert np.array_equal(q["widths"], r["widths"])


def test_graph_batch_rejects_an_out_of_range_index_of_any_graph():
    from wdg_amd import ops
    rng = np.random.default_rng(1)
    coos = _coos(rng, [(50, 100), (40, 100), (60, 100)])
    coos[1][1][17] = 40  # one past the end of graph 1: inside the union, outside its graph
    with pytest.raises(IndexError):
        ops.GraphBatch(coos, ops.COO_ADD_SELF_LOOPS)
    coos[1][1][17] = -1
    with pytest.raises(IndexError):
        ops.GraphBatch(coos, 0)


def test_sweep_batch_builds_agree_bitwise():
    """SweepBatch(build="batched") == SweepBatch(build="per_graph"): aggregation output and the six step scalars"""
    from wdg_amd import sweep
    jobs = sweep.make_jobs([0.15, 0.3, 0.9], [0, 1], k=10, n_nodes=2000) + sweep.make_jobs([0.05, 0.5], [2], k=2, n_nodes=1000)
    a = sweep.SweepBatch(jobs, n_feat=64, gcn_hidden=16, build="batched")
    b = sweep.SweepBatch(jobs, n_feat=64, gcn_hidden=16, build="per_graph")
    assert hasattr(a, "graph_batch") and not hasattr(b, "graph_batch")
    for sb in (a, b):
        sb.step()
    torch.cuda.synchronize()
    for ya, yb in zip(a.y_agg, b.y_agg):  # (tiled or row-major alike in both)
        assert torch.equal(ya.t if hasattr(ya, "rowmajor") else ya, yb.t if hasattr(yb, "rowmajor") else yb)
    assert torch.equal(a.results(), b.results())
    for la, lb in zip(a.gcn["logits"], b.gcn["logits"]):
        assert torch.equal(la, lb)


# ---------------------------------------------------------------------------------------------- the device sampler
def _philox4x32_10(c0, c1, k0, k1):
    """numpy restatement of the documented generator (include/wdg.h): first output word of Philox4x32-10, counter {c0, c1, 0, 0}"""
    c0, c1 = np.asarray(c0, np.uint64), np.asarray(c1, np.uint64) + np.zeros_like(np.asarray(c0, np.uint64))
    c2, c3 = np.zeros_like(c0), np.zeros_like(c0)
    k0, k1 = np.uint64(k0), np.uint64(k1)
    m32 = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = np.uint64(0xD2511F53) * c0, np.uint64(0xCD9E8D57) * c2
        n0, n2 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & m32, ((p0 >> np.uint64(32)) ^ c3 ^ k1) & m32
        c1, c3, c0, c2 = p1 & m32, p0 & m32, n0, n2
        k0, k1 = (k0 + np.uint64(0x9E3779B9)) & m32, (k1 + np.uint64(0xBB67AE85)) & m32
    return c0


def _host_sets(labels, s_c, t_c, seed, epochs):
    n = len(labels)
    out = []
    for e in range(epochs):
        key = _philox4x32_10(np.arange(n), e, seed & 0xFFFFFFFF, seed >> 32)
        order = np.lexsort((np.arange(n), key, labels))  # by (class, key, node)
        train, val = [], []
        for c in range(len(s_c)):
            members = order[labels[order] == c]
            train += list(members[:t_c[c]])
            val += list(members[t_c[c]:s_c[c]])
        out.append((np.sort(train), np.sort(val)))
    return out


def test_device_node_sets_follow_the_documented_generator():
    from wdg_amd import ops
    rng = np.random.default_rng(5)
    cases = [(np.arange(2000) // 400, 500), (rng.integers(0, 3, 777), 200), (np.arange(150) % 5, 500), (rng.integers(0, 7, 2708), 500)]
    entries, want = [], []
    for i, (lab, sample_max) in enumerate(cases):
        lab = lab.astype(np.int64)
        s_c, t_c = ops.kr_split_sizes(lab, sample_max)
        seed = 0x1234567800000000 + 977 * i
        entries.append((torch.from_numpy(lab.astype(np.int32)).cuda(), s_c, t_c, seed))
        want.append(_host_sets(lab, s_c, t_c, seed, 3))
    ks = ops.KrSets(entries, 3)
    ks.launch()
    torch.cuda.synchronize()
    tr, va = ks.train.cpu().numpy(), ks.val.cpu().numpy()
    for i, sets in enumerate(want):
        for e, (t, v) in enumerate(sets):
            assert ks.n_train[i] == len(t) and ks.n_val[i] == len(v)
            assert np.array_equal(tr[i, e, :len(t)], t) and np.array_equal(va[i, e, :len(v)], v), (i, e)
    ks.launch()  # counter-based: the same draw again
    torch.cuda.synchronize()
    assert np.array_equal(ks.train.cpu().numpy(), tr)


def test_device_node_sets_have_the_reference_routines_statistics():
    """sizes per class exactly the reference's; every node equally likely to train / validate (chi-square against the uniform
    rates over 400 sets, next to the same statistic of the reference's host routine on torch's generator)"""
    from wdg_amd import ops
    from wdg_amd.utils.util_funcs import kernel_regression_epoch_indices
    n, c, epochs = 2000, 5, 400
    lab = np.arange(n) // (n // c)
    s_c, t_c = ops.kr_split_sizes(lab, 500)
    assert list(s_c) == [100] * 5 and list(t_c) == [60] * 5
    ks = ops.KrSets([(torch.from_numpy(lab.astype(np.int32)).cuda(), s_c, t_c, 42)], epochs)
    ks.launch()
    torch.cuda.synchronize()
    tr, va = ks.train[0].cpu().numpy(), ks.val[0].cpu().numpy()
    for e in range(epochs):
        assert (np.diff(tr[e]) > 0).all() and (np.diff(va[e]) > 0).all() and not np.intersect1d(tr[e], va[e]).size
        assert list(np.bincount(lab[tr[e]], minlength=c)) == [60] * 5 and list(np.bincount(lab[va[e]], minlength=c)) == [40] * 5
    torch.manual_seed(0)
    host = kernel_regression_epoch_indices(torch.from_numpy(lab), 500, epochs)
    htr = np.stack([t.numpy() for t, _ in host])

    def chi2(sets, rate):
        cnt = np.bincount(sets.reshape(-1), minlength=n).astype(np.float64)
        return float(((cnt - epochs * rate) ** 2 / (epochs * rate * (1 - rate))).sum())

    # chi-square with ~n degrees of freedom: mean n, sd sqrt(2 n) = 63; both generators must sit inside 5 sd
    for stat in (chi2(tr, 0.15), chi2(va, 0.10), chi2(htr, 0.15)):
        assert abs(stat - n) < 5 * np.sqrt(2 * n), stat
    # no correlation between consecutive sets: the overlap of two train sets averages 300 * 0.15 = 45 nodes
    overlap = np.mean([np.intersect1d(tr[e], tr[e + 1]).size for e in range(epochs - 1)])
    assert abs(overlap - 45) < 2.0


def test_nine_scalars_with_device_sampled_sets_agree_with_host_sampled_ones():
    """the same shard with the node sets drawn by the device sampler and by the reference's host routine: the six counter
    scalars and ge_homo are identical and the regressions' mean accuracies agree within sampling noise"""
    from wdg_amd import sweep
    jobs = sweep.make_jobs([0.2, 0.9], [0], k=10, n_nodes=2000)
    rows, accs = {}, {}
    for sampler in ("device", "host"):
        sb = sweep.SweepBatch(jobs, n_feat=128, gcn_hidden=0)
        sb.prepare_full(epochs=24, sample_max=500, sampler=sampler)
        sb.step()
        sb.launch_full()
        torch.cuda.synchronize()
        rows[sampler] = sb.full_metrics().numpy()
        accs[sampler] = sb.kr_accuracy().cpu().numpy()  # [job, classifier, epoch, kernel]
        assert accs[sampler].shape == (len(jobs), 2, 24, 2)
        assert (sb.kr_sets is not None) == (sampler == "device")
    assert np.array_equal(rows["device"][:, :7], rows["host"][:, :7])
    d, h = accs["device"].mean(2), accs["host"].mean(2)  # [job, classifier, kernel]
    assert np.abs(d - h).max() < 0.04, (d, h)               # 24 epochs of ~200 validation rows: sd of a mean ~ 0.007
    for r in (rows["device"], rows["host"]):
        assert ((r[:, 7:] >= 0) & (r[:, 7:] <= 1)).all()
