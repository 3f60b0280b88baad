"""GPU parity of the batched Gaussian naive Bayes (csrc/gnb.hip, ops.GnbBatch) - the GNB branch of the classifier-based metric
(reference: utils/homophily_metrics.py:296-312).  The statistics are compared bit for bit with scikit-learn's own attributes and
with the CPU oracle's restatement, the predictions row by row (the only freedom the kernel has is the rounding of two fp64 sums)."""
import ctypes

import numpy as np
import pytest
import torch
from sklearn.naive_bayes import GaussianNB

from _golden import load

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from wdg_amd import ops as o
    return o


def _features(rng, n, f, zero_cols=True):
    """row-normalised, sparse, non-negative - like the reference's preprocess_features output; some all-zero columns"""
    x = (rng.random((n, f)) ** 6).astype(np.float32)
    x[x < 0.2] = 0
    if zero_cols:
        x[:, ::7] = 0
    x /= np.maximum(x.sum(1, keepdims=True), 1e-12)
    return x.astype(np.float32)


def _ws_statistics(gb, i, f, c):
    """theta, var (fp32, before epsilon), the counts and epsilon's factor of problem i, from the workspace layout of csrc/gnb.hip"""
    per = np.array([int(gb.ops_lib.wdg_gnb_workspace_bytes(int(p[0].shape[1]), c)) for p in gb.keep], np.int64)
    off = int(per[:i].sum())
    raw = gb.ws[off:off + int(per[i])].cpu().numpy()
    head = raw[:256].view(np.int32)
    stats = raw[512:512 + 2 * c * f * 4].view(np.float32).reshape(2, c, f)
    return stats[0], stats[1], head[2:2 + c].copy(), raw[:4].view(np.float32)[0]


CASES = [  # (rows, features, classes, train rows, validation rows, a class without train rows, leading dimension pad)
    (2000, 1433, 5, 300, 200, False, 0), (300, 7, 2, 180, 119, False, 1), (900, 3703, 6, 120, 80, False, 0),
    (700, 257, 16, 400, 203, True, 3), (5000, 64, 3, 2500, 1001, False, 0), (40, 1, 2, 20, 20, False, 0), (64, 300, 4, 33, 0, False, 0)]


@pytest.mark.filterwarnings("ignore::RuntimeWarning")  # (the one-feature case: scikit-learn divides 0 by 0 as well)
@pytest.mark.parametrize("n,f,c,nt,nv,absent,pad", CASES)
def test_gnb_batched_equals_scikit_learn(ops, oracle, n, f, c, nt, nv, absent, pad):
    """two problems per case (different node sets) in one launch: theta / var / counts / epsilon bit for bit scikit-learn's, every
    validation row's predicted class scikit-learn's and the oracle's, the hit counts"""
    from wdg_amd._lib import lib
    rng = np.random.default_rng(n + 31 * f)
    x_full = np.zeros((n, f + pad), np.float32)
    x_full[:, :f] = _features(rng, n, f)
    y = rng.integers(0, c, n).astype(np.int32)
    xd = torch.from_numpy(x_full).cuda()[:, :f]
    yd = torch.from_numpy(y).cuda()
    problems, sets = [], []
    for _rep in range(2):
        perm = rng.permutation(n)
        tr, va = np.sort(perm[:nt]), np.sort(perm[nt:nt + nv])
        if absent:
            tr = tr[y[tr] != 1]  # class 1 never among the train rows: scikit-learn then never predicts it
        sets.append((tr, va))
        problems.append((xd, torch.from_numpy(tr.astype(np.int32)).cuda(), torch.from_numpy(va.astype(np.int32)).cuda(), yd))
    gb = ops.GnbBatch(problems, c, want_pred=True)
    gb.ops_lib = lib
    gb.launch()
    torch.cuda.synchronize()
    x = x_full[:, :f]
    for i, (tr, va) in enumerate(sets):
        sk = GaussianNB().fit(x[tr], y[tr])
        model = oracle.gnb_fit(x[tr], y[tr])
        theta, var, counts, maxvar = _ws_statistics(gb, i, f, c)
        present = sk.classes_.astype(np.int64)
        np.testing.assert_array_equal(counts[present], sk.class_count_.astype(np.int32))
        assert counts.sum() == len(tr) and (np.delete(counts, present) == 0).all()
        assert np.float32(np.float32(1e-9) * maxvar) == sk.epsilon_ == model["epsilon"]
        np.testing.assert_array_equal(theta[present].astype(np.float64), sk.theta_)
        np.testing.assert_array_equal(var[present].astype(np.float64) + np.float64(sk.epsilon_), sk.var_)
        if nv:
            want = sk.predict(x[va])
            np.testing.assert_array_equal(want, oracle.gnb_predict(model, x[va]))
            got = gb.pred[i].cpu().numpy()
            jll = np.sort(sk._joint_log_likelihood(x[va]), axis=1)
            clear = (jll[:, -1] - jll[:, -2]) > 1e-9 * np.abs(jll[:, -1]) if jll.shape[1] > 1 else np.ones(len(va), bool)
            np.testing.assert_array_equal(got[clear], want[clear])  # (a tie at fp64 rounding level may go either way: none seen)
            assert (got == want).mean() > 0.999
            assert int(gb.correct[i].item()) == int((got == y[va]).sum())
            np.testing.assert_allclose(gb.accuracy()[i], np.mean(got == y[va]), rtol=1e-6)
        else:
            assert int(gb.correct[i].item()) == 0 and np.isnan(gb.accuracy()[i])


def test_gnb_constant_features_follow_numpy_semantics(ops):
    """every feature constant among the train rows: all variances 0, epsilon 0, the log likelihoods NaN - np.argmax then answers
    with the first class, and so does the kernel"""
    x = np.ones((50, 9), np.float32)
    y = (np.arange(50) % 3).astype(np.int32)
    tr, va = np.arange(0, 30, dtype=np.int32), np.arange(30, 50, dtype=np.int32)
    with np.errstate(all="ignore"):
        want = GaussianNB().fit(x[tr], y[tr]).predict(x[va])
    gb = ops.GnbBatch([(torch.from_numpy(x).cuda(), torch.from_numpy(tr).cuda(), torch.from_numpy(va).cuda(), torch.from_numpy(y).cuda())], 3,
                      want_pred=True)
    gb.launch()
    np.testing.assert_array_equal(gb.pred[0].cpu().numpy(), want)


def test_gnb_refusals(ops):
    from wdg_amd._lib import WdgError, lib
    x = torch.zeros((8, 4), device="cuda")
    ids, lab = torch.arange(4, dtype=torch.int32, device="cuda"), torch.zeros(8, dtype=torch.int32, device="cuda")
    with pytest.raises(ValueError):
        ops.GnbBatch([(x, ids, ids, lab)], 17)
    with pytest.raises(ValueError):
        ops.GnbBatch([(x, ids[:0], ids, lab)], 2)
    with pytest.raises(ValueError):
        ops.GnbBatch([(x.double(), ids, ids, lab)], 2)
    assert lib.wdg_gnb_batched_f32(ctypes.c_void_p(0), 1, 4, 4, 2, ctypes.c_void_p(0)) != 0      # null table
    assert lib.wdg_gnb_batched_f32(ctypes.c_void_p(0), 0, 4, 4, 2, ctypes.c_void_p(0)) == 0      # nothing to do
    tab = torch.zeros(64, dtype=torch.uint8, device="cuda")
    assert lib.wdg_gnb_batched_f32(ctypes.c_void_p(tab.data_ptr()), 1, 4, 4, 17, ctypes.c_void_p(0)) != 0  # more classes than the kernel holds
    assert lib.wdg_gnb_batched_f32(ctypes.c_void_p(tab.data_ptr()), 1, -1, 4, 2, ctypes.c_void_p(0)) != 0
    assert WdgError is not None


@pytest.mark.parametrize("name", ["texas", "cora", "citeseer", "film"])
def test_gnb_metric_on_device_equals_the_host_path(name, monkeypatch):
    """classifier_based_performance_metric(base_classifier='gnb'): the device call against the reference's own route - scikit-learn
    on the host over the same aggregated features and the same node sets (same torch CPU generator stream): the per-epoch accuracies
    and the p-value are equal"""
    from wdg_amd.utils import homophily_metrics as hm
    from test_gpu_api import _raw
    g0 = load("real_" + name)
    adj_raw, features, labels = _raw(g0)
    accs = []
    orig_mean = torch.mean
    torch.manual_seed(5)
    hm.LAST_GNB_ACCURACIES = None
    p_dev, _ = hm.classifier_based_performance_metric(features, adj_raw, labels, 300.0, base_classifier="gnb", epochs=8)
    assert hm.LAST_GNB_ACCURACIES is not None and hm.LAST_GNB_ACCURACIES.shape == (8, 2)
    dev_acc = hm.LAST_GNB_ACCURACIES.numpy().copy()
    # the host route, recording what it computes per epoch
    class Rec(GaussianNB):
        def predict(self, X):
            out = super().predict(X)
            accs.append(out)
            return out
    import sklearn.naive_bayes as nb
    monkeypatch.setattr(nb, "GaussianNB", Rec)
    monkeypatch.setenv("WDG_GNB_SOLVER", "host")
    torch.manual_seed(5)
    hm.LAST_GNB_ACCURACIES = None
    p_host, _ = hm.classifier_based_performance_metric(features, adj_raw, labels, 300.0, base_classifier="gnb", epochs=8)
    assert hm.LAST_GNB_ACCURACIES is None and len(accs) == 16 and orig_mean is torch.mean
    assert abs(p_dev - p_host) <= 1e-12, (p_dev, p_host)
    # per-epoch hit rates from the recorded host predictions need the epoch's validation labels: re-derive the sets
    from wdg_amd.utils.util_funcs import kernel_regression_epoch_indices
    torch.manual_seed(5)
    sets = kernel_regression_epoch_indices(labels, 300.0, 8)
    lab = labels.flatten().numpy()
    for e, (_tr, va) in enumerate(sets):
        x_pred, g_pred = accs[2 * e], accs[2 * e + 1]  # (the twin fits X first, then X_agg, and predicts in that order)
        assert np.float32(np.mean(g_pred == lab[va.numpy()])) == dev_acc[e, 0]
        assert np.float32(np.mean(x_pred == lab[va.numpy()])) == dev_acc[e, 1]
