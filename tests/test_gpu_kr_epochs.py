"""The device kernel-regression solver against what the REFERENCE computed inside each epoch (tests/golden/kr_epochs.npz,
recorded inside the reference's own classifier_based_performance_metric call by tests/golden/make_golden_kr.py): same node
sets, per (fixture, classifier, epoch, kernel) the validation accuracy of the regression."""
import numpy as np
import pytest
import torch

from _golden import KR_FIXTURES, dense_features, load, load_kr, p_tolerance, welch_p

pytestmark = pytest.mark.gpu

# validation rows (of 196 - 200; texas 73) by which an epoch's accuracy may differ from what the REFERENCE recorded in that epoch.
# Round 6 (row representatives + the deflating solver, csrc/row_rep.hip, csrc/kernel_reg.hip): the synthetic-sweep fixtures - whose
# pubmed-sample features hold duplicate rows: 39 % of their train blocks were regularised in round 5, 0 - 2 rows off - are
# reproduced EXACTLY, every epoch, both kernels; cora (duplicate aggregated rows under the raw adjacency: every block deflated, one
# still regularised) within 1 row where round 5 needed 4; texas' raw-adjacency kernels (rank deficient beyond their duplicates:
# aggregated rows that are sums of other rows - the deflated block is regularised in the full system's metric) within 1 row.
# One structural exception: a class of DUPLICATE nodes (texas: the leaves of one hub aggregate to the same row) whose train members'
# labels TIE for the majority.  The regression interpolates its train rows, so every validation member of the class is predicted
# the class's mean label - two equal entries, an arg-max decided by rounding for all of them at once (texas, arc-cosine kernel,
# epoch 6: 24 of 73 validation rows; an fp64 pseudo-inverse of the device kernel and one of the host's GEMM kernel disagree there
# as well).  Such an epoch may differ by the number of validation rows in tied classes on top of the bound.
ROWS_OFF = {"real_texas": 1, "real_cora": 1, "real_cora_s200": 0}
ROWS_SYN = 0


def _device_inputs(name):
    """(h_agg, x, labels) on the GPU, built the way the reference call was fed"""
    from wdg_amd import ops
    g = load(name.replace("_s200", ""))
    n = int(g["n_nodes"])
    if name.startswith("syn_"):  # synthetic_plot.py:81-92
        gr = ops.CsrGraph.from_coo(g["adj_row"].astype(np.int64), g["adj_col"].astype(np.int64), n, None, ops.COO_ADD_SELF_LOOPS)
        d = ops.degree_norm(gr, ops.NORM_RW, ops.PREC_F32)["dinv"]
        x = ops.row_l1_normalise(torch.from_numpy(dense_features(g)))
        h = ops.spmm(gr, x, row_scale=d)
    else:                        # homophily_tests.py:133-137: raw adjacency, raw features
        gr = ops.CsrGraph.from_coo(g["adj_row"].astype(np.int64), g["adj_col"].astype(np.int64), n, g["adj_val"])
        x = torch.from_numpy(dense_features(g)).cuda()
        h = ops.spmm(gr, x)
    return h, x, torch.from_numpy(g["labels"]).cuda().to(torch.int32)


@pytest.mark.parametrize("name", KR_FIXTURES)
def test_device_solver_per_epoch_against_the_reference(name):
    from wdg_amd import ops
    kr = load_kr(name)
    h, x, lab = _device_inputs(name)
    gb = ops.GramBatch([h, x])
    gb.launch()
    rows = ROWS_OFF.get(name, ROWS_SYN)
    lab_h = lab.cpu().numpy()

    def tied_validation_rows(rep_h, tr, va):
        """validation rows whose duplicate class has train members with a tied majority label"""
        n_tied = 0
        for u in np.unique(rep_h[va]):
            members = tr[rep_h[tr] == u]
            if members.size >= 2:
                cnt = np.sort(np.bincount(lab_h[members]))[::-1]
                if cnt.size >= 2 and cnt[0] == cnt[1]:
                    n_tied += int((rep_h[va] == u).sum())
        return n_tied
    for clf, kern in (("kernel_reg0", gb.k_linear), ("kernel_reg1", gb.k_arccos)):
        rec = kr[clf]
        problems = []
        for tr, va in rec["node_sets"]:
            tr, va = torch.from_numpy(tr).cuda().to(torch.int32), torch.from_numpy(va).cuda().to(torch.int32)
            problems += [(kern[0], tr, va, lab, gb.rep[0]), (kern[1], tr, va, lab, gb.rep[1])]
        kb = ops.KrBatch(problems, int(lab.max().item()) + 1)
        kb.launch()
        torch.cuda.synchronize()
        acc = kb.accuracy().cpu().numpy().reshape(-1, 2).astype(np.float64)
        n_val = np.array([len(v) for _, v in rec["node_sets"]], np.float64)
        off_g, off_x = (acc[:, 0] - rec["g_results"]) * n_val, (acc[:, 1] - rec["x_results"]) * n_val
        ridged = kb.ridged().cpu().numpy().reshape(-1, 2)
        print(f"[kr epochs] {name} {clf}: rows off (graph) max {np.abs(off_g).max():.0f} mean {np.abs(off_g).mean():.2f}, (features) max "
              f"{np.abs(off_x).max():.0f} mean {np.abs(off_x).mean():.2f}; ridged {ridged.sum(0).tolist()} deflated "
              f"{kb.deflated().cpu().numpy().reshape(-1, 2).sum(0).tolist()} of {len(rec['node_sets'])}")
        lim = np.full(ridged.shape, rows + 0.01)  # [epoch, (graph-aware, features only)]
        if gb.rep[0] is not None:  # (validation members of a duplicate class with a tied mean label share ONE fragile arg-max: see above)
            lim[:, 0] += np.array([tied_validation_rows(gb.rep[0].cpu().numpy(), tr, va) for tr, va in rec["node_sets"]])
            lim[:, 1] += np.array([tied_validation_rows(gb.rep[1].cpu().numpy(), tr, va) for tr, va in rec["node_sets"]])
        assert (np.abs(off_g) <= lim[:, 0]).all() and (np.abs(off_x) <= lim[:, 1]).all(), (name, clf, np.round(off_g), np.round(off_x), ridged.T)
        assert (np.abs(off_g) > rows + 0.01).sum() + (np.abs(off_x) > rows + 0.01).sum() <= 1  # (at most one such epoch per fixture and classifier)
        # the p-value: within what the accuracy bound implies for these accuracies (no fixed 0.15)
        tol = p_tolerance(rec["g_results"], rec["x_results"], float(n_val.min()), max(int(np.abs(off_g).max().round()), int(np.abs(off_x).max().round()), 1))
        assert abs(welch_p(acc[:, 0], acc[:, 1]) - rec["p"]) <= tol, (name, clf, welch_p(acc[:, 0], acc[:, 1]), rec["p"], tol)


def test_solver_limits_are_refused_loudly():
    """more than 8 classes / an empty train set: KrBatch raises instead of launching problems the kernel answers with -1
    (accuracies of -1 / n_val fed the t-test a garbage p-value)"""
    from wdg_amd import ops
    k = torch.eye(64, device="cuda")
    lab = (torch.arange(64, device="cuda") % 10).to(torch.int32)
    tr, va = torch.arange(40, device="cuda", dtype=torch.int32), torch.arange(40, 64, device="cuda", dtype=torch.int32)
    with pytest.raises(ValueError):
        ops.KrBatch([(k, tr, va, lab)], 10)
    with pytest.raises(ValueError):
        ops.KrBatch([(k, tr[:0], va, lab)], 5)
    kb = ops.KrBatch([(k, tr, va, lab % 5)], 5)
    kb.launch()
    assert 0.0 <= float(kb.accuracy()[0]) <= 1.0
    kb.correct.fill_(-1)  # what the kernel writes for a refused problem
    with pytest.raises(RuntimeError):
        kb.accuracy()


def test_ten_class_labels_fall_back_to_the_host_path():
    """classifier_based_performance_metric with 10 classes: the device path declines before consuming the generator, the
    reference's host path (np.linalg.pinv) runs on the same node sets and returns a valid p-value"""
    from wdg_amd.utils import homophily_metrics as hm
    rng = np.random.default_rng(3)
    n, f, c = 300, 24, 10
    lab = torch.from_numpy(np.arange(n) % c)
    x = torch.from_numpy((rng.standard_normal((n, f)) + np.eye(c, f)[lab.numpy()] * 2).astype(np.float32))
    src = rng.integers(0, n, 1500)
    adj = torch.sparse_coo_tensor(torch.from_numpy(np.stack([src, (src + c * rng.integers(1, 5, 1500)) % n])), torch.ones(1500), (n, n)).coalesce()
    assert hm._kernel_regression_on_device(x, adj, lab, 200.0, "kernel_reg1", 3) is None
    res = {}
    for solver in ("device", "host"):
        torch.manual_seed(4)
        hm.LAST_KR_ACCURACIES = None
        res[solver] = hm.classifier_based_performance_metric(x, adj, lab, 200.0, base_classifier="kernel_reg1", epochs=3, solver=solver)[0]
        assert hm.LAST_KR_ACCURACIES is None  # (no device regression ran)
    assert res["device"] == res["host"] and 0.0 <= res["host"] <= 1.0


# Bounds asserted by the at-scale test below, in validation rows (of 200), against the reference's own per-epoch computation on the
# host (sampled Gram by fp32 GEMM + arc-cosine map + np.linalg.pinv = SweepBatch.pinv_accuracies(kernels="host")):
#   blocks the device solver factors as they are: <= 2 (measured: 99.7 % equal, the rest 1 row);
#   blocks it regularises (ridge): >= 98 % within 2 rows, none beyond 4 (measured on all 3 108 flagged blocks of the 8 000: 2 599 equal,
#   472 one row, 36 two, 1 three - profiles/r05_kr_ridge_vs_pinv.json; the three-way comparison on the golden epochs,
#   profiles/r05_kr_three_way.txt, has the ridge within 0 - 2 rows of what the reference recorded on every synthetic fixture).
ROWS_PD, RIDGED_SHARE_WITHIN_2, RIDGED_MAX = 2, 0.98, 4
# Round 6: blocks with duplicate nodes in the train rows are DEFLATED on the device (row representatives, csrc/row_rep.hip + the
# solver's pre-pass) - the pseudo-inverse's answer by a positive definite factorisation; they are held to the bound of the blocks
# factored as they are, and nothing of this test's 8 000 regressions is left to the ridge.
DEFLATED_EQUAL_SHARE = 0.99


def test_batched_sweep_ridge_against_the_pseudo_inverse_at_scale():
    """The batched sweep path (SweepBatch.prepare_full / launch_full / full_metrics) against the reference's solver at the
    reference's epoch count: the five golden synthetic fixtures x 4 independent draws of the node sets = 20 jobs x 2 classifiers x
    100 epochs x 2 kernels = 8 000 regressions on the device (Cholesky; ridge on the blocks it flags), and for EVERY flagged block
    plus a sample of the others the reference's own per-epoch computation on the host (utils/homophily_metrics.py:232-257 +
    :283-297 through SweepBatch.pinv_accuracies: the epoch's sampled Gram by fp32 GEMM, the arc-cosine map, np.linalg.pinv).
    Asserts the documented bounds (ROWS_PD, RIDGED_*), that full_metrics(ridge="pinv") carries the host answer on exactly the
    flagged blocks with p-values inside what the row bound implies, and reports - not asserts - how far np.linalg.pinv applied to
    blocks of the DEVICE kernel lands from both (the reason the patch computes its kernels the reference's way).  The
    distributions are written to gpurun_out/ for profiles/r05_kr_ridge_vs_pinv.json."""
    import json
    import os
    import time
    from _golden import SYN
    from wdg_amd import sweep
    from wdg_amd.utils import util_funcs as uf
    jobs, inputs = [], []
    for i, name in enumerate(SYN):
        g0 = load(name)
        n = int(g0["n_nodes"])
        x = uf.preprocess_features(torch.from_numpy(dense_features(g0))).cpu().numpy()  # synthetic_plot.py:81-83
        jobs.append(sweep.Job(float(name.split("_")[2]), 100 + i, 10 if "_4000_" in name else 2, n, int(g0["labels"].max()) + 1))
        inputs.append((g0["adj_row"].astype(np.int64), g0["adj_col"].astype(np.int64), g0["labels"], x))
    epochs, draws = 100, 4
    report = {"fixtures": list(SYN), "draws": draws, "epochs": epochs, "per_draw": []}
    rng = np.random.default_rng(5)
    d_pd, d_ridged, d_devgram, d_defl, dp_all, failures = [], [], [], [], [], []
    n_regressions = n_flagged = n_deflated = 0
    for draw in range(draws):
        sb = sweep.SweepBatch(jobs, n_feat=inputs[0][3].shape[1], gcn_hidden=0, inputs=inputs)
        sb.prepare_full(epochs=epochs, sample_max=500, base_seed=77 + draw)  # device-drawn sets, as the sweep driver does
        sb.step()
        sb.launch_full()
        torch.cuda.synchronize()
        rows_dev = sb.full_metrics(ridge="device").numpy()
        acc_dev = sb.kr_acc.copy().reshape(-1)                                 # problem order [job, clf, epoch, kernel]
        ridged = sb.kr_ridged_mask().cpu().numpy().reshape(-1)                 # (the same order: every job its own sample here)
        deflated = sb.kr_deflated_mask().cpu().numpy().reshape(-1) & ~ridged
        n_val = float(sb.kr_val.shape[2])
        flagged = np.flatnonzero(ridged)
        others = rng.choice(np.flatnonzero(~ridged & ~deflated), 150, replace=False)
        defl = np.flatnonzero(deflated)
        host_d = sb.pinv_accuracies(defl)                                      # EVERY deflated block the reference's way
        d_defl.append(np.abs(acc_dev[defl].astype(np.float64) - host_d) * n_val)
        n_deflated += len(defl)
        t0 = time.perf_counter()
        host_f = sb.pinv_accuracies(flagged)                                   # the reference's way, kernels included
        t_host = time.perf_counter() - t0
        host_o = sb.pinv_accuracies(others)
        d_ridged.append(np.abs(acc_dev[flagged].astype(np.float64) - host_f) * n_val)
        d_pd.append(np.abs(acc_dev[others].astype(np.float64) - host_o) * n_val)
        src_, host_ = (flagged, host_f) if len(flagged) else (defl, host_d)    # ~120 singular blocks: pinv on the DEVICE kernel's blocks
        some = src_[:: max(1, len(src_) // 120)]
        d_devgram.append(np.abs(sb.pinv_accuracies(some, kernels="device").astype(np.float64) - host_[:: max(1, len(src_) // 120)]) * n_val)
        n_regressions, n_flagged = n_regressions + ridged.size, n_flagged + len(flagged)
        rec = {"ridged": int(len(flagged)), "total": int(ridged.size), "host_seconds_flagged": t_host}
        if draw == 0:  # the patch itself: flagged blocks carry the host answer, the others the device answer; p-values follow
            rows_pinv = sb.full_metrics(ridge="pinv").numpy()
            acc_pinv = sb.kr_acc.copy().reshape(-1)
            if not (np.array_equal(acc_pinv[flagged], host_f) and np.array_equal(np.delete(acc_pinv, flagged), np.delete(acc_dev, flagged))):
                failures.append("ridge='pinv' did not patch exactly the flagged blocks with the host answer")
            if not (sb.kr_ridged == len(flagged) and sb.kr_total == ridged.size and (sb.kr_pinv_seconds > 0 or not len(flagged))
                    and sb.kr_deflated == int(sb.kr.deflated().sum())):
                failures.append(f"counts {sb.kr_ridged} / {sb.kr_total} / {sb.kr_pinv_seconds}")
            rec["pinv_patch_seconds"] = sb.kr_pinv_seconds
            acc4 = acc_pinv.reshape(len(jobs), 2, epochs, 2)
            p_patched = sweep.welch_p_values(acc4[..., 0], acc4[..., 1])
            assert np.allclose(rows_pinv[:, 7:9], p_patched, equal_nan=True)
            dp = np.abs(rows_dev[:, 7:9] - rows_pinv[:, 7:9])
            dp_all.append(dp)
            for ji in range(len(jobs)):
                for ci in range(2):
                    tol = p_tolerance(acc4[ji, ci, :, 0], acc4[ji, ci, :, 1], n_val, RIDGED_MAX if ridged.reshape(acc4.shape)[ji, ci].any() else ROWS_PD, trials=100)
                    if abs(rows_dev[ji, 7 + ci] - rows_pinv[ji, 7 + ci]) > tol:
                        failures.append(f"job {ji} clf {ci}: p {rows_dev[ji, 7 + ci]:.3g} (ridge) vs {rows_pinv[ji, 7 + ci]:.3g} (pinv), tolerance {tol:.3g}")
        report["per_draw"].append(rec)
        del sb
    d_pd, d_ridged, d_devgram, d_defl = np.concatenate(d_pd), np.concatenate(d_ridged), np.concatenate(d_devgram), np.concatenate(d_defl)

    def dist_of(v):
        v = np.round(v).astype(int)
        return {"n": int(v.size), "median": float(np.median(v)) if v.size else 0.0, "max": int(v.max(initial=0)),
                "share_within_2": float((v <= 2).mean()) if v.size else 1.0,
                "histogram_rows": {str(k): int((v == k).sum()) for k in range(int(v.max(initial=0)) + 1) if (v == k).any()}}

    if d_pd.max(initial=0) > ROWS_PD + 0.01:
        failures.append(f"blocks factored as they are: up to {d_pd.max():.1f} rows from the host answer")
    if d_ridged.size and ((d_ridged <= 2.01).mean() < RIDGED_SHARE_WITHIN_2 or d_ridged.max(initial=0) > RIDGED_MAX + 0.01):
        failures.append(f"regularised blocks: {(d_ridged <= 2.01).mean():.3f} within 2 rows, max {d_ridged.max():.1f}")
    if d_defl.size and ((d_defl < 0.5).mean() < DEFLATED_EQUAL_SHARE or d_defl.max(initial=0) > ROWS_PD + 0.01):
        failures.append(f"deflated blocks: {(d_defl < 0.5).mean():.4f} equal to the host answer, max {d_defl.max():.1f} rows")
    if n_deflated < 2000 or n_flagged > 0:  # (the pubmed-sample features hold duplicate rows: ~39 % of these blocks; none may need the ridge)
        failures.append(f"{n_deflated} deflated / {n_flagged} regularised blocks of {n_regressions}")
    report.update(regressions=int(n_regressions), ridged=int(n_flagged), deflated=int(n_deflated),
                  device_deflated_vs_reference_way_all=dist_of(d_defl),
                  device_vs_reference_way_not_flagged_sample=dist_of(d_pd), device_ridge_vs_reference_way_flagged_all=dist_of(d_ridged),
                  pinv_on_device_kernel_blocks_vs_reference_way_flagged_sample=dist_of(d_devgram),
                  abs_dp_ridge_vs_patched={"median": float(np.median(np.concatenate(dp_all))), "max": float(np.concatenate(dp_all).max())},
                  bound_asserted={"rows_not_flagged": ROWS_PD, "flagged_share_within_2_rows": RIDGED_SHARE_WITHIN_2, "flagged_max_rows": RIDGED_MAX,
                                  "p": "_golden.p_tolerance of the row bound on the patched accuracies"},
                  failures=failures)
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, "kr_ridge_vs_pinv.json"), "w") as f:
            json.dump(report, f, indent=1)
    except OSError:
        pass
    assert report["regressions"] == 8000
    assert not failures, "\n".join(failures[:20])
