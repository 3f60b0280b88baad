"""The oracle's Gaussian naive Bayes (oracle/oracle.py::gnb_fit / gnb_predict) is pinned against the algorithm it restates: the
installed scikit-learn's GaussianNB - the third-party dependency the reference calls (utils/homophily_metrics.py:296-312) -,
attribute by attribute and bit for bit, on float32 inputs shaped like the reference's (row-normalised sparse features)."""
import numpy as np
import pytest
from sklearn.naive_bayes import GaussianNB

from _golden import dense_features, load


def _check(oracle, x, y, tr, va):
    sk = GaussianNB().fit(x[tr], y[tr])
    m = oracle.gnb_fit(x[tr], y[tr])
    np.testing.assert_array_equal(m["classes"], sk.classes_)
    np.testing.assert_array_equal(m["theta"], sk.theta_)
    np.testing.assert_array_equal(m["var"], sk.var_)
    np.testing.assert_array_equal(m["prior"], sk.class_prior_)
    assert m["epsilon"] == sk.epsilon_ and type(sk.epsilon_) is np.float32  # (numpy 2: float32(1e-9) * the fp32 variance)
    np.testing.assert_array_equal(oracle.gnb_joint_log_likelihood(m, x[va]), sk._joint_log_likelihood(x[va]))
    np.testing.assert_array_equal(oracle.gnb_predict(m, x[va]), sk.predict(x[va]))


@pytest.mark.parametrize("n,f,c", [(300, 500, 5), (60, 9, 2), (500, 1433, 7), (90, 3, 4)])
def test_gnb_oracle_equals_scikit_learn_on_synthetic_features(oracle, n, f, c):
    rng = np.random.default_rng(n + f)
    x = (rng.random((n, f)) ** 5).astype(np.float32)
    x[x < 0.3] = 0
    x[:, ::4] = 0  # (features that are constant inside every class: variance = epsilon)
    x /= np.maximum(x.sum(1, keepdims=True), 1e-12)
    y = rng.integers(0, c, n)
    perm = rng.permutation(n)
    _check(oracle, x.astype(np.float32), y, np.sort(perm[:int(0.6 * n)]), np.sort(perm[int(0.6 * n):]))


def test_gnb_oracle_equals_scikit_learn_on_the_reference_features(oracle):
    g0 = load("real_texas")
    x = dense_features(g0, "featl1_data") if "featl1_data" in g0 else None
    if x is None:
        pytest.skip("fixture without dense features")
    y = np.asarray(g0["labels"]).reshape(-1)
    rng = np.random.default_rng(3)
    perm = rng.permutation(x.shape[0])
    _check(oracle, x.astype(np.float32), y, np.sort(perm[:110]), np.sort(perm[110:]))


def test_gnb_accuracies_restates_the_epoch_loop(oracle):
    """gnb_accuracies(x, x_agg, labels, node_sets) = per epoch scikit-learn on the aggregated and the raw features"""
    rng = np.random.default_rng(9)
    x = rng.random((120, 20)).astype(np.float32)
    xa = (x + np.roll(x, 1, 0)).astype(np.float32) / 2
    y = rng.integers(0, 3, 120)
    sets = [(rng.permutation(120)[:70], rng.permutation(120)[:40]) for _ in range(3)]
    g, xr = oracle.gnb_accuracies(x, xa, y, sets)
    for e, (tr, va) in enumerate(sets):
        tr, va = np.sort(tr), np.sort(va)
        assert g[e] == np.float32(np.mean(GaussianNB().fit(xa[tr], y[tr]).predict(xa[va]) == y[va]))
        assert xr[e] == np.float32(np.mean(GaussianNB().fit(x[tr], y[tr]).predict(x[va]) == y[va]))
