"""GPU parity of the batched sweep step (wdg_amd.sweep.SweepBatch = loop body of synthetic_plot.py:81-109 without
the kernel-regression metric, plus the GCN-2 forward): every stage of one step against the CPU oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rowmajor(y):
    """the aggregation's output as a row-major tensor (a tiled Y - ops.Tiled, 16-feature groups - copied out)"""
    return y.rowmajor() if hasattr(y, "rowmajor") else y



@pytest.mark.parametrize("fused_mlp,ride_labels", [("1", "1"), ("0", "0"), ("1", "0")])
@pytest.mark.parametrize("symmetric", [0, 1])
def test_sweep_step_against_oracle(oracle, symmetric, fused_mlp, ride_labels, monkeypatch):
    from wdg_amd import sweep, synth
    monkeypatch.setenv("WDG_SWEEP_FUSED_MLP", fused_mlp)
    monkeypatch.setenv("WDG_SWEEP_RIDE_LABELS", ride_labels)
    jobs = sweep.make_jobs([0.1, 0.5, 0.9], [0, 1], k=2, n_nodes=1000) + sweep.make_jobs([0.2, 0.4], [2], k=10, n_nodes=1000)
    batch = sweep.SweepBatch(jobs, n_feat=96, symmetric=symmetric, gcn_hidden=32)
    batch.step()
    batch.step()  # relaunch must give the same answers (counters re-zeroed)
    torch.cuda.synchronize()
    rows = batch.results().cpu().numpy()
    assert rows.shape == (len(jobs), sweep.STEP_METRICS) and len(sweep.METRIC_NAMES) == 9
    assert (batch.spmm_las is None) == (ride_labels == "1") and batch.agg_feat == (112 if ride_labels == "1" else 96)
    for i, j in enumerate(jobs):
        src, dst, lab = synth.regular_graph(j.n_nodes, j.n_classes, j.k, j.h, j.seed)
        x = synth.features(j.n_nodes, 96, j.seed)
        rowptr, col, val = oracle.coo_to_csr(src, dst, j.n_nodes, None, oracle.ADD_SELF_LOOPS)
        vhat = oracle.normalised_csr(rowptr, col, val, symmetric, oracle.PREC_F32)
        y = oracle.spmm_csr(rowptr, col, vhat, x)
        np.testing.assert_allclose(batch.y[i].cpu().numpy(), y, rtol=1e-5, atol=1e-6 * np.abs(y).max())
        onehot32 = np.eye(j.n_classes, dtype=np.float32)[lab]
        h_ref = oracle.spmm_csr(rowptr, col, vhat, onehot32)  # the label aggregation (riding along or its own launch)
        np.testing.assert_allclose(batch.h_las[i].cpu().numpy(), h_ref, rtol=1e-5, atol=1e-6)
        st = oracle.edge_label_stats(rowptr, col, lab, j.n_classes)
        want = [oracle.edge_homophily_dense(st), oracle.node_homophily_dense(st), oracle.class_homophily_dense(st, lab),
                oracle.adjusted_homophily_dense(st, lab), oracle.label_informativeness(st, lab)]
        np.testing.assert_allclose(rows[i, :5], want, rtol=2e-4, atol=3e-6)
        onehot = np.eye(j.n_classes, dtype=np.float32)[lab]
        soft = oracle.similarity(onehot, rowptr, col, vhat, onehot, f64=True)
        assert abs(rows[i, 5] - soft) <= 2.01 / j.n_nodes
        g = batch.gcn
        hid = oracle.gemm(y, g["w0"][i].cpu().numpy(), relu=True)
        assert (g["hid"] is None) == (fused_mlp == "1")  # the fused transform keeps the hidden layer in registers
        if g["hid"] is not None:
            np.testing.assert_allclose(g["hid"][i].cpu().numpy(), hid, rtol=1e-5, atol=1e-5 * np.abs(hid).max())
        logits = oracle.spmm_csr(rowptr, col, vhat, oracle.gemm(hid, g["w1"][i].cpu().numpy()))
        np.testing.assert_allclose(g["logits"][i].cpu().numpy(), logits, rtol=1e-5, atol=1e-5 * np.abs(logits).max())


def test_step_rest_graph_replay_equals_plain_launches():
    """SweepBatch.capture_rest(): the post-aggregation launches of a step (both streams, fork / join included) replayed
    from one hipGraph give bit for bit the outputs of the plain launches."""
    from wdg_amd import sweep
    jobs = sweep.make_jobs([0.2, 0.6], [0, 1], k=2, n_nodes=1000)
    batch = sweep.SweepBatch(jobs, n_feat=64, gcn_hidden=32)
    batch.step()
    torch.cuda.synchronize()
    want = ([l.clone() for l in batch.gcn["logits"]], batch.results().clone(), batch.las.counts.clone())
    replay = batch.capture_rest()
    for l in batch.gcn["logits"]:
        l.fill_(float("nan"))
    batch.spmm.launch()
    replay()
    torch.cuda.synchronize()
    for a, b in zip(batch.gcn["logits"], want[0]):
        assert torch.equal(a, b)
    assert torch.equal(batch.results(), want[1]) and torch.equal(batch.las.counts, want[2])


def test_whole_step_graph_replay_of_a_small_shard_equals_plain_launches():
    """SweepBatch.capture_step(): a 6-graph shard (what a rank holds when configs[2]'s 50 jobs are spread over 8 GPUs) steps
    from ONE hipGraph - the feature aggregation included - with bit for bit the outputs of the plain launches."""
    from wdg_amd import sweep, synth
    jobs = sweep.shard_jobs(sweep.make_jobs(synth.H_LEVELS_10_K10, range(5), k=10, n_nodes=2000), 8, 0)
    assert 6 <= len(jobs) <= 7
    batch = sweep.SweepBatch(jobs, n_feat=500, gcn_hidden=64)
    batch.step()
    torch.cuda.synchronize()
    want = ([y.clone() for y in batch.y], [l.clone() for l in batch.gcn["logits"]], batch.results().clone(), batch.las.counts.clone())
    replay = batch.capture_step()
    for t in list(batch.y) + list(batch.gcn["logits"]):
        t.fill_(float("nan"))
    replay()
    torch.cuda.synchronize()
    for a, b in zip(list(batch.y) + list(batch.gcn["logits"]), want[0] + want[1]):
        assert torch.equal(a, b)
    assert torch.equal(batch.results(), want[2]) and torch.equal(batch.las.counts, want[3])


def test_whole_sweep_over_feature_bases_equals_stand_alone_batches():
    """sweep.run_bases (synthetic_plot.py:64-109: every adjacency paired with features of every base; graphs built once per
    shard and shared by the bases, base-shards pipelined over two streams): every (shard, base) gives the rows of a stand-alone
    SweepBatch over the same inputs with the same node-set keys - uneven shards, widths on both sides of the fused-transform /
    one-slab limits, two sample sizes."""
    from wdg_amd import sweep, synth
    levels, samples = [0.2, 0.5, 0.8], [0, 1]
    graphs = {(h, s_): synth.regular_graph(600, 5, 4, h, s_) for h in levels for s_ in samples}
    bases = [("a", {s_: synth.features(600, 40, 10 + s_) for s_ in samples}, 500),
             ("b", {s_: synth.features(600, 530, 20 + s_) for s_ in samples}, 300),
             ("c", {s_: synth.features(600, 97, 30 + s_) for s_ in samples}, 500),
             # (a width the sweep driver propagates: label columns only in the aggregation, K(A_hat X) = A_hat K(X) A_hat^T - the
             # stand-alone batch below aggregates the features as well and takes the same route for the kernels)
             ("d", {s_: synth.features(600, 656, 40 + s_) for s_ in samples}, 500)]
    shards = []
    for lv in (levels[:2], levels[2:]):
        jobs = sweep.make_jobs(lv, samples, k=4, n_nodes=600)
        shards.append((jobs, [graphs[(j.h, j.seed)] for j in jobs]))
    got = {(si, bi): rows for si, bi, rows in sweep.run_bases(shards, bases, epochs=6, depth=2, first_seed=5)}
    assert sorted(got) == [(si, bi) for si in range(2) for bi in range(4)]
    for (si, bi), rows in got.items():
        jobs, gi = shards[si]
        _name, feats, sample_max = bases[bi]
        sb = sweep.SweepBatch(jobs, n_feat=next(iter(feats.values())).shape[1], gcn_hidden=0,
                              inputs=[(src, dst, lab, feats[j.seed]) for j, (src, dst, lab) in zip(jobs, gi)])
        sb.prepare_full(epochs=6, sample_max=sample_max, base_seed=5 + 1000 * bi)
        sb.step()
        sb.launch_full()
        want = sb.full_metrics()
        assert rows.shape == (len(jobs), 9)
        assert torch.equal(torch.nan_to_num(rows, nan=-7.0), torch.nan_to_num(want, nan=-7.0)), (si, bi)


def test_table_reuse_across_wide_feature_bases_equals_stand_alone_batches(monkeypatch):
    """run_bases over several WIDE feature bases (the propagated route: label columns only in the aggregation): the bases of a shard
    share one step (step twins), a base takes over the prepared batch of an earlier base of equal sample_max (rebind_features: same
    buffers, new features, new node-set keys) and the bases are visited out of order - every (shard, base) still gives the rows
    of a stand-alone SweepBatch over the same inputs, bit for bit, and the same rows as the driver without any reuse
    (WDG_SWEEP_REBIND=0 WDG_SWEEP_STEP_TWINS=0 WDG_SWEEP_PREFETCH_BUILD=0), unpipelined or with three base-shards in flight."""
    from wdg_amd import sweep, synth
    for name in ("WDG_SWEEP_REBIND", "WDG_SWEEP_STEP_TWINS", "WDG_SWEEP_PREFETCH_BUILD", "WDG_GRAM_ROUTE", "WDG_SWEEP_PROLOGUE"):
        monkeypatch.delenv(name, raising=False)  # (this test asserts WHICH bases were rebound / twinned under the defaults)
    levels, samples = [0.2, 0.5, 0.8], [0, 1]
    graphs = {(h, s_): synth.regular_graph(600, 5, 4, h, s_) for h in levels for s_ in samples}
    widths = [("p1", 656, 500), ("p2", 700, 500), ("n", 97, 500), ("p3", 720, 300), ("p4", 800, 500), ("p5", 650, 300)]
    bases = [(name, {s_: synth.features(600, w, 10 * (i + 1) + s_) for s_ in samples}, sm) for i, (name, w, sm) in enumerate(widths)]
    assert sweep._visit_order([(True, sm) if w >= 640 else ("own", i) for i, (_n, w, sm) in enumerate(widths)]) == [0, 3, 1, 2, 4, 5]
    shards = []
    for lv in (levels[:2], levels[2:]):
        jobs = sweep.make_jobs(lv, samples, k=4, n_nodes=600)
        shards.append((jobs, [graphs[(j.h, j.seed)] for j in jobs]))
    rebinds, twins = [], []
    orig_rebind, orig_twin = sweep.SweepBatch.rebind_features, sweep.SweepBatch._init_step_twin
    monkeypatch.setattr(sweep.SweepBatch, "rebind_features", lambda self, *a, **k: (rebinds.append(a[1]), orig_rebind(self, *a, **k))[1])
    monkeypatch.setattr(sweep.SweepBatch, "_init_step_twin", lambda self, *a, **k: (twins.append(a[2]), orig_twin(self, *a, **k))[1])
    got = {(si, bi): rows for si, bi, rows in sweep.run_bases(shards, bases, epochs=6, depth=2, first_seed=5)}
    assert sorted(got) == [(si, bi) for si in range(2) for bi in range(6)]
    assert sorted(rebinds) == [650, 650, 700, 700, 800, 800] and sorted(twins) == [720, 720]  # per shard: p2, p4, p5 rebound; p3 a twin of p1
    monkeypatch.setattr(sweep.SweepBatch, "rebind_features", orig_rebind)
    monkeypatch.setattr(sweep.SweepBatch, "_init_step_twin", orig_twin)
    for (si, bi), rows in got.items():
        jobs, gi = shards[si]
        _name, feats, sample_max = bases[bi]
        sb = sweep.SweepBatch(jobs, n_feat=next(iter(feats.values())).shape[1], gcn_hidden=0,
                              inputs=[(src, dst, lab, feats[j.seed]) for j, (src, dst, lab) in zip(jobs, gi)])
        sb.prepare_full(epochs=6, sample_max=sample_max, base_seed=5 + 1000 * bi)
        sb.step()
        sb.launch_full()
        want = sb.full_metrics()
        assert torch.equal(torch.nan_to_num(rows, nan=-7.0), torch.nan_to_num(want, nan=-7.0)), (si, bi)
    for depth in (1, 3):  # (unpipelined: a batch is free again at once; three in flight: a base may find no free batch and builds its own)
        for si, bi, rows in sweep.run_bases(shards, bases, epochs=6, depth=depth, first_seed=5):
            assert torch.equal(torch.nan_to_num(rows, nan=-7.0), torch.nan_to_num(got[(si, bi)], nan=-7.0)), (depth, si, bi)
    monkeypatch.setenv("WDG_SWEEP_PROLOGUE", "1")  # (round 6's feature prologue - a helper thread uploads the wide bases' features and
    for si, bi, rows in sweep.run_bases(shards, bases, epochs=6, depth=2, first_seed=5):  # launches their Grams ahead - : the same rows)
        assert torch.equal(torch.nan_to_num(rows, nan=-7.0), torch.nan_to_num(got[(si, bi)], nan=-7.0)), ("prologue", si, bi)
    monkeypatch.delenv("WDG_SWEEP_PROLOGUE")
    monkeypatch.setenv("WDG_SWEEP_REBIND", "0")
    monkeypatch.setenv("WDG_SWEEP_STEP_TWINS", "0")
    monkeypatch.setenv("WDG_SWEEP_PREFETCH_BUILD", "0")
    order = []
    for si, bi, rows in sweep.run_bases(shards, bases, epochs=6, depth=2, first_seed=5):
        order.append((si, bi))
        assert torch.equal(torch.nan_to_num(rows, nan=-7.0), torch.nan_to_num(got[(si, bi)], nan=-7.0)), (si, bi)
    assert order == [(si, bi) for si in range(2) for bi in range(6)]  # (the caller's order without reuse)


def test_a_graph_without_sell16_copy_in_a_wide_base_shard_falls_back_instead_of_raising(monkeypatch):
    """run_bases decided labels_only from the widths alone; one skewed graph in the shard (SELL-16 copy refused) made
    prepare_full raise and aborted the sweep.  Now the route is decided after the build (graphs built ahead) and a labels-only
    batch that cannot propagate falls back to the direct route over the on-demand aggregation (graphs built in place)."""
    from wdg_amd import ops, sweep, synth
    for name in ("WDG_SWEEP_REBIND", "WDG_SWEEP_STEP_TWINS", "WDG_SWEEP_PREFETCH_BUILD", "WDG_GRAM_ROUTE"):
        monkeypatch.delenv(name, raising=False)
    levels, samples = [0.2, 0.5, 0.8], [0]
    jobs = sweep.make_jobs(levels, samples, k=4, n_nodes=600)
    gi = [synth.regular_graph(600, 5, 4, j.h, j.seed) for j in jobs]
    bases = [(f"w{w}", {0: synth.features(600, w, 40 + w)}, 500) for w in (656, 700)]
    want = {bi: rows for _si, bi, rows in sweep.run_bases([(jobs, gi)], bases, epochs=4, depth=2, first_seed=3)}
    orig = ops.GraphBatch.finish

    def finish_and_refuse_one(self):
        orig(self)
        self.graphs[1].quad = False  # (what ensure_quad / the batched build decide for rows that would pad beyond 4 x)
    monkeypatch.setattr(ops.GraphBatch, "finish", finish_and_refuse_one)
    for prefetch in ("1", "0"):
        monkeypatch.setenv("WDG_SWEEP_PREFETCH_BUILD", prefetch)
        got = {bi: rows for _si, bi, rows in sweep.run_bases([(jobs, gi)], bases, epochs=4, depth=2, first_seed=3)}
        assert sorted(got) == [0, 1]
        for bi in got:
            assert got[bi].shape == want[bi].shape and bool(torch.isfinite(got[bi][:, :7]).all())
            # the scalars that do not depend on the kernels' route: equal to the all-quad run's to fp32 rounding
            assert torch.allclose(got[bi][:, :7], want[bi][:, :7], rtol=2e-5, atol=1e-6), (prefetch, bi)
            assert bool(((got[bi][:, 7:] >= 0) & (got[bi][:, 7:] <= 1)).all())


def test_nine_scalars_with_propagated_grams_match_the_direct_route(monkeypatch):
    """SweepBatch.prepare_full with WDG_GRAM_ROUTE=propagate (the aggregated features' kernels as A_hat K(X) A_hat^T: what the
    sweep takes for the reference's wide feature bases) against the direct route (the Gram of Y) on the golden synthetic fixtures,
    host-drawn node sets: the seven scalars that do not depend on the kernels identical, every epoch's accuracy within 2
    validation rows, the p-values within what that implies; and against the reference's own recorded epochs (kr_epochs.npz)
    within the same 2 rows as the direct route is held to."""
    from _golden import SYN, dense_features, load, load_kr, p_tolerance
    from wdg_amd import sweep
    from wdg_amd.utils import util_funcs as uf
    jobs, inputs = [], []
    for i, name in enumerate(SYN):
        g0 = load(name)
        n = int(g0["n_nodes"])
        x = uf.preprocess_features(torch.from_numpy(dense_features(g0))).cpu().numpy()
        jobs.append(sweep.Job(float(name.split("_")[2]), 100 + i, 10 if "_4000_" in name else 2, n, int(g0["labels"].max()) + 1))
        inputs.append((g0["adj_row"].astype(np.int64), g0["adj_col"].astype(np.int64), g0["labels"], x))
    rows, accs = {}, {}
    for route in ("direct", "propagate"):
        monkeypatch.setenv("WDG_GRAM_ROUTE", route)
        sb = sweep.SweepBatch(jobs, n_feat=inputs[0][3].shape[1], gcn_hidden=0, inputs=inputs)
        sb.prepare_full(epochs=8, sample_max=500, seed_of=lambda ji, clf: 5)
        assert sb.gram_route == route
        sb.step()
        sb.launch_full()
        torch.cuda.synchronize()
        rows[route] = sb.full_metrics().numpy()
        accs[route] = sb.kr_acc.copy()  # [job, clf, epoch, (graph, features)]
    assert np.array_equal(rows["direct"][:, :7], rows["propagate"][:, :7])
    assert np.array_equal(accs["direct"][..., 1], accs["propagate"][..., 1])  # the raw features' kernels are the same launch
    d_rows = np.abs(accs["direct"][..., 0].astype(np.float64) - accs["propagate"][..., 0]) * 200
    assert d_rows.max() <= 2.01, d_rows.max()
    for ji, name in enumerate(SYN):
        kr = load_kr(name)
        for ci, clf in enumerate(("kernel_reg0", "kernel_reg1")):
            g_ref, x_ref = kr[clf]["g_results"][:8], kr[clf]["x_results"][:8]
            assert np.abs(accs["propagate"][ji, ci, :, 0] - g_ref).max() <= 2.01 / 200, (name, clf)
            tol = p_tolerance(g_ref, x_ref, 200.0, 2)
            assert abs(rows["propagate"][ji, 7 + ci] - rows["direct"][ji, 7 + ci]) <= 2 * tol + 1e-12, (name, clf)


def test_common_sets_per_sample_share_the_raw_feature_regressions(monkeypatch):
    """prepare_full(sets="sample") (opt-in; the default draws per job like the reference): the homophily levels of one sample draw ONE sequence of node sets, keyed by the
    sample's identity, and the raw features' regression of an (epoch, classifier) is solved once per sample.
      * the table holds J * 2 E regressions on the aggregated features' kernels + G * 2 E on the raw features' (G samples);
      * the jobs of a sample read the SAME raw-feature accuracies; their graph-aware ones differ;
      * a job computes the same nine scalars bit for bit in whichever batch it sits (alone, or with the other levels);
      * sets="job" solves 4 E per job; both modes' mean accuracies agree within sampling noise (same distribution per job)."""
    from wdg_amd import sweep
    epochs = 12
    jobs = sweep.make_jobs([0.2, 0.5, 0.9], [0, 1], k=10, n_nodes=2000)  # 2 samples x 3 levels, seed-major
    rows, accs, totals = {}, {}, {}
    for mode in ("sample", "job"):
        sb = sweep.SweepBatch(jobs, n_feat=128, gcn_hidden=0)
        sb.prepare_full(epochs=epochs, sample_max=500, base_seed=3, sets=mode)
        sb.step()
        sb.launch_full()
        torch.cuda.synchronize()
        rows[mode] = sb.full_metrics().numpy()
        accs[mode] = sb.kr_acc.copy()  # [job, classifier, epoch, (graph-aware, features only)]
        totals[mode] = (sb.kr.n_jobs, sb.kr_total, sb.kr_set_mode)
        if mode == "sample":
            assert sb.kr_group.tolist() == [0, 0, 0, 1, 1, 1] and sb.kr_rep.tolist() == [0, 3]
            assert np.array_equal(sb.kr_accuracy().cpu().numpy(), accs[mode])
            tr = sb.kr_train.cpu().numpy()
            assert tr.shape[:2] == (2 * 2, epochs) and not np.array_equal(tr[0], tr[2])  # one sequence per (sample, classifier)
    J = len(jobs)
    assert totals["sample"] == (J * 2 * epochs + 2 * 2 * epochs, J * 2 * epochs + 2 * 2 * epochs, "sample")
    assert totals["job"] == (J * 4 * epochs, J * 4 * epochs, "job")
    a = accs["sample"]
    for g in (0, 1):
        assert np.array_equal(a[3 * g, ..., 1], a[3 * g + 1, ..., 1]) and np.array_equal(a[3 * g, ..., 1], a[3 * g + 2, ..., 1])
        assert not np.array_equal(a[3 * g, ..., 0], a[3 * g + 2, ..., 0])
    assert not np.array_equal(a[0, ..., 1], a[3, ..., 1])
    assert np.array_equal(rows["sample"][:, :7], rows["job"][:, :7])
    # same distribution per job: ~200 validation rows x 12 epochs -> sd of a mean accuracy ~ 0.01
    assert np.abs(accs["sample"].mean(2) - accs["job"].mean(2)).max() < 0.05
    # identity keys: a job alone in its batch draws the sets - and computes the rows - it has among the other levels
    for ji in (1, 5):
        sb = sweep.SweepBatch([jobs[ji]], n_feat=128, gcn_hidden=0)
        sb.prepare_full(epochs=epochs, sample_max=500, base_seed=3, sets="sample")
        sb.step()
        sb.launch_full()
        torch.cuda.synchronize()
        alone = sb.full_metrics().numpy()
        assert np.array_equal(alone[0], rows["sample"][ji], equal_nan=True), (ji, alone[0], rows["sample"][ji])
        assert np.array_equal(sb.kr_acc[0], accs["sample"][ji])
    # jobs of one seed whose LABEL vectors differ are different samples
    monkeypatch.setenv("WDG_SWEEP_KR_SETS", "sample")
    monkeypatch.setenv("WDG_SWEEP_RIDE_LABELS", "0")  # (label columns riding with the features want one label vector per seed)
    from wdg_amd import synth
    inputs = []
    for j in jobs[:2]:
        src, dst, lab = synth.regular_graph(j.n_nodes, j.n_classes, j.k, j.h, j.seed)
        inputs.append((src, dst, lab, synth.features(j.n_nodes, 64, j.seed)))
    flipped = inputs[1][2].copy()
    flipped[:2] = flipped[:2][::-1].copy() if flipped[0] != flipped[1] else (flipped[:2] + 1) % jobs[0].n_classes
    inputs[1] = (inputs[1][0], inputs[1][1], flipped, inputs[1][3])
    sb = sweep.SweepBatch(jobs[:2], n_feat=64, gcn_hidden=0, inputs=inputs)
    sb.prepare_full(epochs=2, sample_max=500)
    assert sb.kr_group.tolist() == [0, 1] and sb.kr.n_jobs == 2 * 4 * 2


def test_whole_sweep_rows_do_not_depend_on_the_world_size():
    """sweep.whole_sweep_rank + exchange_rows (bench.py's N-rank `sweep_whole`): the adjacencies dealt to 1, 2 and 3 "ranks" (run
    one after the other on this GPU; a rank's rows keyed by the pair's position in the job list) assemble to the SAME [pairs x
    bases, 9] table bit for bit - node sets are keyed by job identity, a graph's build and sum orders do not depend on its shard"""
    from wdg_amd import sweep, synth
    levels, samples = [0.2, 0.3, 0.5, 0.8], [0, 1]
    pairs = sweep.make_jobs(levels, samples, k=4, n_nodes=400)
    graphs = {(j.h, j.seed): synth.regular_graph(400, 5, 4, j.h, j.seed) for j in pairs}
    bases = [("a", {s_: synth.features(400, 48, 10 + s_) for s_ in samples}, 500),
             ("b", {s_: synth.features(400, 130, 20 + s_) for s_ in samples}, 300)]
    tables = {}
    for world in (1, 2, 3):
        table = torch.full((len(pairs) * len(bases), 9), float("nan"), dtype=torch.float64)
        seen = 0
        for rank in range(world):
            keys, rows = sweep.whole_sweep_rank(pairs, lambda j: graphs[(j.h, j.seed)], bases, world, rank, epochs=5,
                                                max_pairs_per_shard=3 if world == 1 else 80)
            assert rows.shape == (keys.shape[0], 9)
            table[keys] = rows
            seen += keys.shape[0]
        assert seen == len(pairs) * len(bases)
        assert not torch.isnan(table[:, :7]).any()
        tables[world] = table
    for world in (2, 3):
        assert torch.equal(torch.nan_to_num(tables[world], nan=-7.0), torch.nan_to_num(tables[1], nan=-7.0)), world


def test_batched_gemm_and_las_mixed_shapes(oracle):
    from wdg_amd import ops
    rng = np.random.default_rng(3)
    ent, refs = [], []
    for m, k, n in ((300, 40, 64), (2000, 500, 64), (17, 3, 5), (129, 64, 33)):
        a, b = rng.standard_normal((m, k)).astype(np.float32), rng.standard_normal((k, n)).astype(np.float32)
        bias = rng.standard_normal(n).astype(np.float32)
        ent.append((torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), torch.empty((m, n), device="cuda"),
                    torch.from_numpy(bias).cuda()))
        refs.append(oracle.gemm(a, b, bias, relu=True))
    gb = ops.GemmBatch(ent, relu=True)
    gb.launch()
    torch.cuda.synchronize()
    for (_, _, c, _), ref in zip(ent, refs):
        np.testing.assert_allclose(c.cpu().numpy(), ref, rtol=1e-5, atol=1e-5 * np.abs(ref).max())
    las_ent, want = [], []
    for n, f, c in ((500, 5, 5), (1300, 17, 3), (64, 5, 5)):
        h = rng.random((n, f), dtype=np.float32)
        lab = rng.integers(0, c, n)
        lab[:c] = np.arange(c)
        las_ent.append((torch.from_numpy(h).cuda(), torch.from_numpy(lab).cuda().to(torch.int32)))
        w = oracle.las_weights(h, lab, 5, f64=True)
        want.append((round(oracle.las_from_weights(w, lab) * n), round(oracle.las_from_weights(w, lab, hard=1) * n)))
    lb = ops.LasBatch(las_ent, 5)
    lb.launch()
    lb.launch()
    torch.cuda.synchronize()
    got = lb.counts.cpu().numpy()
    for i, (s, hd) in enumerate(want):
        assert abs(int(got[i, 0]) - s) <= 1 and abs(int(got[i, 1]) - hd) <= 1


@pytest.mark.parametrize("variant,env,family", [("column slab", {"WDG_SPMM_NO_QUAD": "1"}, 0), ("quad-row", {}, 5)])
def test_full_sweep_batch_every_item_exactly_once(monkeypatch, variant, env, family):
    """The bench-size batch of the `800` set (100 graphs) on the quad-row kernel (dynamic dealing of super-units inside a
    workgroup) and, with WDG_SPMM_NO_QUAD=1, on the CSR column-slab kernel: every (graph, feature group) piece must be produced
    by every launch - outputs are pre-filled with NaN - and repeated launches must be bitwise equal to the per-graph calls."""
    from wdg_amd import ops, sweep, synth
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    sb = sweep.SweepBatch(sweep.make_jobs(synth.H_LEVELS_10, range(10), k=2), n_feat=500, gcn_hidden=0)
    assert sb.spmm.plan()[0] == family
    want = {}
    for i in (0, 9, 37, 55, 89, 99):
        g, x, _y, d, _cs, _uv = sb.spmm.keep[i]
        want[i] = ops.spmm(g, x, row_scale=d, use_values=False).clone()
    for launch in range(6):
        for y in sb.y_agg:
            y.fill_(float("nan"))
        sb.spmm.launch()
        torch.cuda.synchronize()
        for i, (_g, _x, y, _d, _cs, _uv) in enumerate(sb.spmm.keep):
            y = _rowmajor(y)
            assert not bool(torch.isnan(y).any()), (variant, launch, i)
            if i in want:
                assert torch.equal(y, want[i]), (variant, launch, i)


@pytest.mark.parametrize("seeds,n_nodes", [(5, 2000), (10, 2000), (5, 4000), (2, 800)])
def test_full_c3_sweep_shard_on_the_quad_kernel(oracle, seeds, n_nodes):
    """BASELINE config C3 at full size - the `data_synthesis/4000`-equivalent shard the bench times: 10 h-levels (k = 10: 12 .. 67
    stored entries per row, the wide graphs split into CONT entries) x 5 (10) seeds, N = 2000, F = 500 (+ the label columns):
    every (graph, feature group) item is produced by every launch (outputs pre-filled with NaN), repeated launches are
    bitwise equal to the single-graph calls, and sampled graphs match the oracle within 1e-5.  n_nodes = 4000 / 800: the
    LITERAL reading of configs[2] / configs[1] (4000 nodes: one column block of 32-byte slab rows - feature groups of 8 - on
    the same pipelined loop; 800 nodes: one small block)."""
    from wdg_amd import ops, sweep, synth
    jobs = sweep.make_jobs(synth.H_LEVELS_10_K10, range(seeds), k=10, n_nodes=n_nodes)
    sb = sweep.SweepBatch(jobs, n_feat=500, gcn_hidden=0)
    assert sb.spmm.plan()[0] == 5 and sb.spmm.quad
    assert sb.spmm.flags & ops.SPMM_SMALL_OFFSETS and sb.spmm.flags & ops.SPMM_DMA_OK  # (split-form entries: the pipelined loop)
    assert sb.graphs[0].quad["n_blocks"] == 1 and sb.graphs[0].quad["half"] == (n_nodes == 4000)
    assert bool(sb.spmm.flags & ops.SPMM_HALF_SLAB) == (n_nodes == 4000)
    assert sb.edges == seeds * n_nodes * sum(int(10 / h) + 1 for h in synth.H_LEVELS_10_K10)
    sample = sorted({0, 3, 9, len(jobs) // 2, len(jobs) - 7, len(jobs) - 1})
    want = {}
    for i in sample:
        g, x, _y, d, _cs, _uv = sb.spmm.keep[i]
        want[i] = ops.spmm(g, x, row_scale=d, use_values=False).clone()
    for launch in range(4):
        for y in sb.y_agg:
            y.fill_(float("nan"))
        sb.spmm.launch()
        torch.cuda.synchronize()
        for i, (_g, _x, y, _d, _cs, _uv) in enumerate(sb.spmm.keep):
            y = _rowmajor(y)
            assert not bool(torch.isnan(y).any()), (launch, i)
            if i in want and n_nodes <= 2528:
                assert torch.equal(y, want[i]), (launch, i)
            elif i in want:  # the single-graph call of a wide 4000-node graph takes the band kernel (CSR order): same sums, other order
                torch.testing.assert_close(y, want[i], rtol=1e-5, atol=1e-6 * float(want[i].abs().max()))
    for i in sample:  # against the oracle: D^-1 (A + I) [X | onehot | 0]
        j = jobs[i]
        src, dst, lab = synth.regular_graph(j.n_nodes, j.n_classes, j.k, j.h, j.seed)
        rowptr, col, val = oracle.coo_to_csr(src, dst, j.n_nodes, None, oracle.ADD_SELF_LOOPS)
        vhat = oracle.normalised_csr(rowptr, col, val, 0, oracle.PREC_F32)
        y_ref = oracle.spmm_csr(rowptr, col, vhat, synth.features(j.n_nodes, 500, j.seed))
        np.testing.assert_allclose(sb.y[i].cpu().numpy(), y_ref, rtol=1e-5, atol=1e-6 * np.abs(y_ref).max())
        h_ref = oracle.spmm_csr(rowptr, col, vhat, np.eye(j.n_classes, dtype=np.float32)[lab])
        np.testing.assert_allclose(sb.h_las[i].cpu().numpy(), h_ref, rtol=1e-5, atol=1e-6)


def test_sweep_all_nine_scalars_against_golden():
    """The batched sweep job - all nine scalars of synthetic_plot.py:94-109 - on the reference's own `data_synthesis` fixtures
    (five graphs, one batch): the six counter-based metrics, generalized edge homophily (gathered from the features' Gram) and
    the kernel-regression p-values KR_L / KR_NL (kernels of all nodes in one fused MFMA launch, every (graph, classifier,
    epoch, kernel) regression in one launch of the register-resident Cholesky solver) against the golden scalars the real
    reference produced (seed 5, 4 epochs, sample_max 500; node sets drawn with the reference's RNG routine)."""
    from _golden import SYN, dense_features, load
    from wdg_amd import sweep
    from wdg_amd.utils import util_funcs as uf
    names = SYN
    jobs, inputs, gold = [], [], []
    for i, name in enumerate(names):
        g0 = load(name)
        n = int(g0["n_nodes"])
        x = uf.preprocess_features(torch.from_numpy(dense_features(g0))).cpu().numpy()  # synthetic_plot.py:81-83
        k = 10 if "_4000_" in name else 2
        jobs.append(sweep.Job(float(name.split("_")[2]), 100 + i, k, n, int(g0["labels"].max()) + 1))
        inputs.append((g0["adj_row"].astype(np.int64), g0["adj_col"].astype(np.int64), g0["labels"], x))
        gold.append(g0)
    sb = sweep.SweepBatch(jobs, n_feat=inputs[0][3].shape[1], gcn_hidden=0, inputs=inputs)
    sb.prepare_full(epochs=4, sample_max=500, seed_of=lambda ji, clf: 5)
    sb.step()
    sb.launch_full()
    torch.cuda.synchronize()
    rows = sb.full_metrics().numpy()
    assert rows.shape == (len(jobs), 9)
    for r, g0, j in zip(rows, gold, jobs):
        assert r[0] == pytest.approx(float(g0["m_edge_homo"]), rel=1e-6)
        assert r[1] == pytest.approx(float(g0["m_node_homo"]), rel=1e-6)
        assert r[2] == pytest.approx(float(g0["m_class_homo"]), rel=1e-5, abs=1e-7)
        assert r[3] == pytest.approx(float(g0["m_adj_homo"]), rel=1e-4, abs=2e-7)
        # (label informativeness = 2 - sum pc ln pc / sum p ln p, formed near 2 in fp32 by the reference and here: each side carries
        # ~2e-6 of absolute rounding whatever the value - 3.6e-5 on syn_4000_0.2 - so the bound is absolute; round 5 allowed rel 2e-3)
        assert r[4] == pytest.approx(float(g0["m_label_info"]), rel=1e-5, abs=4e-6)
        assert abs(r[5] - float(g0["m_soft_las"])) <= 2.01 / j.n_nodes
        assert r[6] == pytest.approx(float(g0["m_ge_homo"]), rel=2e-5)
    # the accuracies behind the p-values, per (job, classifier, epoch, kernel), against what the reference computed in each of
    # these epochs (tests/golden/kr_epochs.npz holds 8 epochs of the same seed: the first 4 are this run's): within 2 of the
    # 200 validation rows; the p-values within what that implies for these accuracies (_golden.p_tolerance)
    from _golden import load_kr, p_tolerance
    acc = sb.kr_accuracy().cpu().numpy().astype(np.float64)
    assert acc.shape == (len(jobs), 2, 4, 2)
    for ji, (name, r, g0) in enumerate(zip(names, rows, gold)):
        kr = load_kr(name)
        for ci, clf in enumerate(("kernel_reg0", "kernel_reg1")):
            g_ref, x_ref = kr[clf]["g_results"][:4], kr[clf]["x_results"][:4]
            assert np.abs(acc[ji, ci, :, 0] - g_ref).max() <= 2.01 / 200, (name, clf, acc[ji, ci, :, 0], g_ref)
            assert np.abs(acc[ji, ci, :, 1] - x_ref).max() <= 2.01 / 200, (name, clf, acc[ji, ci, :, 1], x_ref)
            assert abs(r[7 + ci] - float(g0[f"m_cpm_{clf}_seed5_e4_s500"])) <= p_tolerance(g_ref, x_ref, 200.0, 2)


def test_sweep_exchange_under_an_initialised_process_group(tmp_path):
    """SweepBatch + broadcast_jobs + gather_results together under an initialised `nccl` (= RCCL) process group - world size 1,
    what a one-GPU box can run; the world-size-2 logic is covered with gloo in tests/test_sweep_dist.py."""
    import os
    import subprocess
    import sys
    script = tmp_path / "one_rank.py"
    script.write_text("""
import os, sys, json
sys.path.insert(0, os.environ["WDG_ROOT"])
import torch, torch.distributed as dist
from wdg_amd import sweep, synth
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)
jobs = sweep.make_jobs(synth.H_LEVELS_10_K10[:4], range(2), k=10, n_nodes=1000)
got = sweep.broadcast_jobs(jobs if dist.get_rank() == 0 else [], dev)
mine = sweep.shard_jobs(got, dist.get_world_size(), dist.get_rank())
batch = sweep.SweepBatch(mine, n_feat=64, gcn_hidden=16)
batch.step()
rows = batch.results()
gathered = sweep.gather_results(rows, dev)
dist.barrier()
torch.cuda.synchronize()
assert len(gathered) == 1 and torch.equal(gathered[0].cpu(), rows.cpu())
print(json.dumps({"jobs": len(mine), "rows": list(rows.shape), "edge_homo": rows[:, 0].cpu().tolist()}))
dist.destroy_process_group()
""")
    env = dict(os.environ, WDG_ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), RANK="0", LOCAL_RANK="0",
               WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    import json
    out = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])  # (RCCL prints its library path, too)
    from wdg_amd import sweep, synth
    assert out["jobs"] == 8 and out["rows"] == [8, sweep.STEP_METRICS]
    want = [10 / int(10 / h) for h in synth.H_LEVELS_10_K10[:4]] * 2  # edge homophily of the generator: k / int(k / h)
    np.testing.assert_allclose(out["edge_homo"], want, rtol=1e-6)


@pytest.mark.parametrize("kind", ["sgc", "gcn"])
def test_batched_training_matches_per_graph_training(kind):
    """SURVEY 8(f) N4, batched: every graph's model trained inside one batch of launches (manual gradients through the
    transposed graphs, one Adam over the stacked parameters, epoch replayed from a hipGraph) ends where the per-graph
    autograd training of models.py ends when started from the same weights and splits."""
    from wdg_amd import models, sweep, synth
    jobs = sweep.make_jobs([0.2, 0.5, 0.8], range(2), k=2, n_nodes=600)
    sb = sweep.SweepBatch(jobs, n_feat=64, gcn_hidden=0)
    for s in sb.x:  # features with class signal, so that there is something to learn
        lab = synth.regular_graph(600, 5, 2, 0.5, s)[2]
        sb.x[s].copy_(torch.from_numpy(synth.features(600, 64, s, labels=lab)))
    epochs = 12
    res = {}
    for capture in (False, True):
        tb = sweep.TrainBatch(sb, kind=kind, hidden=16, seed=3)
        init = [p.detach().clone() for p in tb.params]
        res[capture] = (tb.run(epochs=epochs, capture=capture), [p.detach().clone() for p in tb.params], tb)
    for a, b in zip(res[False][1], res[True][1]):  # graph replay == eager launches, bitwise
        assert torch.equal(a, b)
    assert torch.equal(res[False][0]["val_acc"], res[True][0]["val_acc"])
    out, weights, tb = res[True]
    for j in (0, 3, 5):  # per-graph autograd training from the same start
        adj = models.NormAdj(sb.graphs[j], add_self_loops=False)
        masks = []
        for idx in (tb.tr[j], tb.va[j], tb.te[j]):
            m = torch.zeros(600, dtype=torch.bool, device="cuda")
            m[idx] = True
            masks.append(m)
        if kind == "sgc":
            model = models.SGC1(64, 5)
            with torch.no_grad():
                model.weight.copy_(init[0][j])
        else:
            model = models.GCN2(64, 5, nhid=16, dropout=0.0)
            with torch.no_grad():
                model.w0.copy_(init[0][j]); model.w1.copy_(init[1][j])
        ref = models.train_eval_graphed(model.cuda(), adj, sb.x[jobs[j].seed], tb.labels[j], masks=masks, epochs=epochs, capture=False)
        got = [w[j] for w in weights]
        for g, p in zip(got, model.parameters()):
            torch.testing.assert_close(g, p.detach(), rtol=2e-3, atol=2e-4)
        assert abs(float(out["val_acc"][j]) - ref["val_acc"]) <= 2.5 / tb.va.shape[1]
    assert float(out["val_acc"].mean()) > 0.3


def test_base_sweep_shares_graphs_and_matches_the_oracle_per_base(oracle):
    """synthetic_plot.py:64-65: the same adjacencies under several feature bases of different widths (here 3 of the 6
    reference widths at N = 1000): graphs are built once, every base's aggregation and step scalars match the oracle"""
    from wdg_amd import sweep, synth
    jobs = sweep.make_jobs([0.2, 0.6], [0, 1], k=2, n_nodes=1000)
    bases = (("pubmed", 500), ("film", 932), ("cora", 1433))
    bs = sweep.BaseSweep(jobs, bases)
    assert all(b.graphs[i] is bs.batches[0].graphs[i] for b in bs.batches for i in range(len(jobs)))
    assert bs.batches[0].graphs[0].quad is not None
    bs.step()
    torch.cuda.synchronize()
    res = bs.results()
    assert tuple(res.shape) == (3, len(jobs), sweep.STEP_METRICS)
    for bi, (name, width) in enumerate(bases):
        b = bs.batches[bi]
        for ji in (0, 3):
            j = jobs[ji]
            src, dst, lab = synth.regular_graph(j.n_nodes, j.n_classes, j.k, j.h, j.seed)
            rowptr, col, val = oracle.coo_to_csr(src, dst, j.n_nodes, None, oracle.ADD_SELF_LOOPS)
            vhat = oracle.normalised_csr(rowptr, col, val, 0, oracle.PREC_F32)
            x = synth.features(j.n_nodes, width, j.seed + 1000 * bi)
            want = oracle.spmm_csr(rowptr, col, vhat, x)
            np.testing.assert_allclose(b.y[ji].cpu().numpy(), want, rtol=1e-5, atol=1e-6 * np.abs(want).max(), err_msg=f"{name} job {ji}")
        # the integer metrics do not depend on the base
        assert torch.equal(res[bi, :, :5], res[0, :, :5])
        assert abs(float(res[bi, 0, 0]) - 2 / int(2 / 0.2)) < 1e-6  # edge homophily = k / int(k / h)


def test_empty_shard_steps_and_gathers():
    """more ranks than jobs (sweep.shard_jobs hands a rank nothing): SweepBatch([]) builds, steps, computes all nine scalars
    and takes part in the result exchange without error - zero rows, same columns"""
    from wdg_amd import sweep
    sb = sweep.SweepBatch(sweep.shard_jobs(sweep.make_jobs([0.2, 0.5], [0], n_nodes=400), 3, 2) and [], n_feat=64, gcn_hidden=16)
    assert sb.jobs == [] and sb.edges == 0
    sb.step()
    sb.step()
    rows = sb.results()
    assert tuple(rows.shape) == (0, sweep.STEP_METRICS)
    sb.prepare_full(epochs=2, sample_max=50)
    sb.launch_full()
    torch.cuda.synchronize()
    assert tuple(sb.full_metrics().shape) == (0, len(sweep.METRIC_NAMES))
    assert [tuple(t.shape) for t in sweep.gather_results(rows, rows.device)] == [(0, sweep.STEP_METRICS)]
    assert sb.spmm_algorithmic_bytes() == 0 and sb.spmm_unique_bytes() == 0


@pytest.mark.parametrize("nine", [False, True])
def test_pipelined_shards_compute_the_sequential_rows(nine):
    """sweep.run_shards: shard b + 1 is uploaded and built on a second HIP stream while shard b's kernels run - the rows of
    every shard must be the rows of the one-shard-at-a-time loop bit for bit (six scalars) / with the same device-drawn node sets
    (nine), in order, for uneven shards and an empty one"""
    from wdg_amd import sweep, synth

    def host_inputs(h_levels, seeds, n_nodes):
        jobs = sweep.make_jobs(h_levels, seeds, k=10, n_nodes=n_nodes)
        feats, inputs = {}, []
        for j in jobs:
            src, dst, lab = synth.regular_graph(j.n_nodes, j.n_classes, j.k, j.h, j.seed)
            feats.setdefault(j.seed, synth.features(j.n_nodes, 100, j.seed))
            inputs.append((src, dst, lab, feats[j.seed]))
        return jobs, inputs

    shards = [host_inputs([0.2, 0.5, 0.8], [3, 4], 600), host_inputs([0.3], [5], 1000), ([], []), host_inputs([0.4, 0.6], [6], 800),
              host_inputs([0.2, 0.9], [7, 8], 600)]
    kw = dict(n_feat=100, nine=nine, epochs=4, sample_max=200)
    one = list(sweep.run_shards(shards, depth=1, **kw))
    for depth in (2, 3):
        piped = list(sweep.run_shards(shards, depth=depth, **kw))
        assert len(piped) == len(one) == len(shards)
        for a, b, (jobs, _) in zip(one, piped, shards):
            assert tuple(a.shape) == tuple(b.shape) == (len(jobs), 9 if nine else sweep.STEP_METRICS)
            np.testing.assert_array_equal(a.numpy(), b.numpy())
    # and a shard's rows are those of its own SweepBatch
    jobs, inputs = shards[0]
    sb = sweep.SweepBatch(jobs, n_feat=100, gcn_hidden=0, inputs=inputs)
    sb.step()
    if nine:
        sb.prepare_full(epochs=4, sample_max=200, base_seed=0)
        sb.launch_full()
        np.testing.assert_array_equal(sb.full_metrics().numpy(), one[0].numpy())
    else:
        np.testing.assert_array_equal(sb.results().cpu().numpy(), one[0].numpy())


@pytest.mark.parametrize("k,seeds", [(2, 10), (10, 5)])
def test_counters_derived_from_the_label_columns_equal_the_edge_pass(k, seeds, monkeypatch):
    """The C2 / C3 shards at full size: every integer counter of the step (totals, compatibility histogram, class degrees,
    the three per-row arrays) derived from H = D^-1 (A + I) onehot inside the LAS launch is bit for bit what
    wdg_edge_label_stats_batched counts over the edges; the step then has one launch fewer."""
    from wdg_amd import sweep, synth
    jobs = sweep.make_jobs(synth.H_LEVELS_10 if k == 2 else synth.H_LEVELS_10_K10, range(seeds), k=k)
    out = {}
    for derive in ("1", "0"):
        monkeypatch.setenv("WDG_SWEEP_DERIVE_COUNTS", derive)
        sb = sweep.SweepBatch(jobs, n_feat=64, gcn_hidden=0)
        assert sb.derive_counts == (derive == "1")
        sb.stats.counters.fill_(-7)  # (the derived path writes, it does not accumulate: stale values must not survive)
        sb.stats.rows.fill_(-7)
        sb.step()
        sb.step()
        torch.cuda.synchronize()
        out[derive] = (sb.stats.totals.clone(), sb.stats.compat.clone(), sb.stats.classdeg.clone(), sb.stats.rows.clone(), sb.results().clone(),
                       sb.las.counts.clone())
    for a, b in zip(out["1"], out["0"]):
        assert torch.equal(a, b)
    assert int(out["1"][0][:, 0].sum()) == sum(j.nnz for j in jobs)


def test_quad_verify_mode_checks_the_table_against_the_csr_kernel(monkeypatch):
    """WDG_QUAD_VERIFY=1: every quad-row table is launched once at construction and compared with the CSR gather kernel; a
    corrupted SELL-16 copy is reported"""
    from wdg_amd import ops, sweep
    jobs = sweep.make_jobs([0.2, 0.7], [0], k=10, n_nodes=2000)
    monkeypatch.setenv("WDG_QUAD_VERIFY", "1")
    sb = sweep.SweepBatch(jobs, n_feat=64, gcn_hidden=0)  # (verifies inside SpmmBatch.__init__)
    assert sb.spmm.quad
    sb.spmm.verify()
    monkeypatch.delenv("WDG_QUAD_VERIFY")
    q = sb.graphs[0].quad
    saved = q["col"][:4096].clone()
    q["col"][:4096] = 0  # the first index chunks now point every entry at column 0
    with pytest.raises(RuntimeError, match="WDG_QUAD_VERIFY"):
        sb.spmm.verify()
    q["col"][:4096] = saved
    sb.spmm.verify()


def test_step_scalars_kernel_equals_the_torch_formulas():
    """SweepBatch.results() (wdg_sweep_scalars_f32: one launch per shard) against results_torch() (the reference's formulas as
    torch operations) on a shard of graphs of different sizes and densities - fp32 both, sums in different orders: 2e-6 relative,
    1e-6 absolute (label informativeness is 2 - a ratio near 2)"""
    from wdg_amd import sweep
    jobs = (sweep.make_jobs([0.05, 0.3, 0.9], [0, 1], k=10, n_nodes=2000) + sweep.make_jobs([0.1, 0.6], [2], k=2, n_nodes=1000)
            + sweep.make_jobs([0.5], [3], k=4, n_nodes=300))
    sb = sweep.SweepBatch(jobs, n_feat=64, gcn_hidden=0)
    sb.step()
    torch.cuda.synchronize()
    got, want = sb.results().cpu().numpy(), sb.results_torch().cpu().numpy()
    assert got.shape == want.shape == (len(jobs), 6) and np.isfinite(got).all()
    np.testing.assert_allclose(got, want, rtol=2e-6, atol=1e-6)
    assert torch.equal(sb.results(), sb.results())


def test_step_scalars_kernel_edge_cases():
    """wdg_sweep_scalars_f32 called on hand-made counters: a class without edges (its NaN term is skipped), a job whose rows are all
    empty beyond the first few, two classes; against the torch formulas on the same counters"""
    from wdg_amd import ops
    from wdg_amd._lib import lib
    j, c, n = 3, 4, 50
    rng = np.random.default_rng(2)
    compat = rng.integers(1, 50, (j, c, c)).astype(np.int64)
    compat[1, 2, :] = 0  # class 2 of job 1 has no edges: rowsum 0 -> NaN -> skipped
    classdeg = compat.sum(2)
    rows = np.zeros((j, 3, n), np.int32)
    nnz = rng.integers(0, 9, (j, n)).astype(np.int32)
    nnz[2, 5:] = 0
    noself = np.maximum(nnz - 1, 0)
    match = (noself * rng.random((j, n))).astype(np.int32)
    rows[:, 0], rows[:, 1], rows[:, 2] = nnz, noself, match
    totals = rng.integers(10, 1000, (j, 6)).astype(np.int64)
    las = rng.integers(0, n, (j, 2)).astype(np.int64)
    las_n = np.full(j, n, np.float32)
    prop = rng.random((j, c)).astype(np.float32)
    prop /= prop.sum(1, keepdims=True)
    dev = [torch.from_numpy(a).cuda() for a in (totals, rows, compat, classdeg, las, las_n, prop)]
    out = torch.empty((j, 6), dtype=torch.float32, device="cuda")
    ops.check(lib.wdg_sweep_scalars_f32(*[t.data_ptr() for t in dev], j, n, c, out.data_ptr(), ops.stream_handle()), "wdg_sweep_scalars_f32")
    tot, k = totals.astype(np.float32), compat.astype(np.float32)
    edge = tot[:, 5] / tot[:, 4]
    f = lambda a: a.astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        hs = (f(match) + (f(nnz) - f(noself))) / f(nnz)
        node = np.where(nnz != 0, hs, 0).sum(1) / (nnz != 0).sum(1)
        terms = np.maximum(np.diagonal(k / k.sum(2, keepdims=True), axis1=1, axis2=2) - prop, 0)
    cls = np.where(np.isnan(terms), 0, terms).sum(1) / (c - 1)
    degsum = f(classdeg.sum(1, keepdims=True))
    pb, pc = f(classdeg) / degsum, k / degsum[:, :, None]
    pb, pc = np.where(pb == 0, 1e-8, pb), np.where(pc == 0, 1e-8, pc)
    s2 = (pb ** 2).sum(1)
    want = np.stack([edge, node, cls, (edge - s2) / (1 - s2), 2 - (pc * np.log(pc)).sum((1, 2)) / (pb * np.log(pb)).sum(1), f(las[:, 0]) / las_n], 1)
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=5e-6, atol=2e-6)
    assert np.isfinite(out.cpu().numpy()).all()


@pytest.mark.parametrize("k,seeds,n_feat,hidden,n_nodes", [(10, 2, 500, 64, 1500), (2, 3, 96, 32, 1500), (10, 1, 36, 0, 1500),
                                                           (10, 1, 72, 32, 3000)])  # (3000 nodes: HALF slabs, 8-feature groups)
def test_tiled_aggregation_output_equals_row_major_bitwise(monkeypatch, k, seeds, n_feat, hidden, n_nodes):
    """The sweep keeps the aggregated features tiled by 16-feature groups (ops.Tiled; wdg_spmm_job.y_group_stride written by the
    quad-row kernel, wdg_mlp2_job.a_group_stride read by the fused transform, the label columns of LAS a strided view): where an
    element is stored does not enter any sum, so every output - Y, the LAS / counter scalars, the GCN-2 logits, and the nine
    scalars behind prepare_full's row-major copy - is the row-major batch's (WDG_SWEEP_TILED_Y=0) bit for bit."""
    from wdg_amd import ops, sweep, synth
    levels = synth.H_LEVELS_10_K10[:4] if k == 10 else synth.H_LEVELS_10[:4]
    jobs = sweep.make_jobs(levels, range(seeds), k=k, n_nodes=n_nodes)
    got = {}
    for tiled in ("1", "0"):
        monkeypatch.setenv("WDG_SWEEP_TILED_Y", tiled)
        sb = sweep.SweepBatch(jobs, n_feat=n_feat, gcn_hidden=hidden)
        # (the fp32-chain transform behind WDG_MLP2_SPLIT=0 reads row-major operands: such a batch stays row-major)
        assert sb.tiled_y == (tiled == "1" and (hidden == 0 or ops.Mlp2Batch.split_kernel()))
        assert isinstance(sb.y_agg[0], ops.Tiled) == sb.tiled_y
        for y in sb.y_agg:
            y.fill_(float("nan"))
        sb.step()
        sb.step()
        torch.cuda.synchronize()
        ys = [y.clone() for y in sb.y]
        agg = [(_rowmajor(y)).clone() for y in sb.y_agg]  # incl. the label and zero columns
        logits = [l.clone() for l in sb.gcn["logits"]] if hidden else []
        rows = sb.results().clone()
        full = None
        if hidden == 64:
            sb.prepare_full(epochs=3, sample_max=200)
            sb.step()
            sb.launch_full()
            full = sb.full_metrics().clone()
        got[tiled] = (ys, agg, logits, rows, full)
    a, b = got["1"], got["0"]
    for ta, tb in zip(a[0] + a[1] + a[2], b[0] + b[1] + b[2]):
        assert not bool(torch.isnan(ta).any()) and torch.equal(ta, tb)
    assert torch.equal(a[3], b[3])
    if a[4] is not None:
        np.testing.assert_array_equal(a[4].numpy(), b[4].numpy())


def test_tiled_output_is_refused_off_the_quad_kernel(monkeypatch):
    """only the quad-row kernel writes a tiled Y: a table that cannot run there refuses it, and so does the single-graph entry"""
    from wdg_amd import ops
    rng = np.random.default_rng(5)
    n, f = 600, 32
    src, dst = rng.integers(0, n, 4000), rng.integers(0, n, 4000)
    g = ops.CsrGraph.from_coo(src, dst, n, None, ops.COO_ADD_SELF_LOOPS)
    x = torch.from_numpy(rng.standard_normal((n, f)).astype(np.float32)).cuda()
    y = ops.Tiled(torch.empty((2, n, 16), device="cuda"))
    ref = torch.empty((n, f), device="cuda")
    ops.SpmmBatch([(g, x, ref, None, None, False)]).launch()
    batch = ops.SpmmBatch([(g, x, y, None, None, False)])
    assert batch.quad
    batch.launch()
    torch.cuda.synchronize()
    assert torch.equal(y.rowmajor(), ref)  # (the tiled launch: the same sums in other places)
    monkeypatch.setenv("WDG_SPMM_NO_QUAD", "1")
    g2 = ops.CsrGraph(g.rowptr, g.col, g.val, n, n)
    with pytest.raises(ValueError):
        ops.SpmmBatch([(g2, x, y, None, None, False)])
