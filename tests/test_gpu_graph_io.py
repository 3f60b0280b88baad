"""Fixture / loader layer on the GPU: bit-packed features are expanded (and row-normalised) by wdg_unpack_bits_f32."""
import numpy as np
import pytest
import torch

from _golden import load

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,f", [(183, 1703), (64, 32), (7, 1), (300, 33), (5, 4000)])
def test_unpack_bits_kernel(n, f):
    from wdg_amd import graph_io, ops
    rng = np.random.default_rng(n * f)
    x = (rng.random((n, f)) < 0.1).astype(np.float32)
    x[0] = 0  # an empty row: stays 0 under the normalisation (inf -> 0 guard)
    words = torch.from_numpy(graph_io.pack_bits(x).astype(np.int32))
    got = ops.unpack_bits(words, f).cpu().numpy()
    np.testing.assert_array_equal(got, x)
    rowsum = x.sum(1, keepdims=True)
    scale = np.divide(np.float32(1), rowsum, out=np.zeros_like(rowsum), where=rowsum != 0).astype(np.float32)
    np.testing.assert_array_equal(ops.unpack_bits(words, f, row_normalise=True).cpu().numpy(), x * scale)


def test_container_to_device_matches_golden_pipeline(tmp_path, oracle):
    """texas fixture -> container (bit-packed features) -> device: the normalised features equal the golden
    `preprocess_features` output and the CSR feeds the aggregation."""
    from wdg_amd import graph_io, ops
    g = load("real_texas")
    n, f = int(g["n_nodes"]), int(g["n_feat"])
    x = np.zeros((n, f), np.float32)
    rows = np.repeat(np.arange(n), np.diff(g["feat_indptr"]))
    x[rows, g["feat_indices"]] = g["feat_data"]
    rowptr, col = graph_io._undirected_binary_csr(g["adj_row"], g["adj_col"], n)
    path = str(tmp_path / "texas.wdgg")
    graph_io.save_graph(path, rowptr, col, g["labels"], x)
    graph, feats, labels = graph_io.load_device(path, row_normalise=True)
    want = np.zeros((n, f), np.float32)
    want[rows, g["feat_indices"]] = g["featn_data"]  # the reference's preprocess_features(features)
    np.testing.assert_allclose(feats.cpu().numpy(), want, rtol=1e-6, atol=0)
    np.testing.assert_array_equal(labels.cpu().numpy(), g["labels"])
    y = ops.spmm(graph, feats, use_values=False)
    ref = oracle.spmm_csr(rowptr, col, np.ones(len(col), np.float32), feats.cpu().numpy())
    np.testing.assert_allclose(y.cpu().numpy(), ref, rtol=1e-5, atol=1e-7)


def test_cli_runs_from_container(tmp_path):
    """homophily_tests.py counterpart fed from a .wdgg container: same scalars as from the golden .npz."""
    from wdg_amd import graph_io
    from wdg_amd import homophily_tests as cli
    g = load("real_texas")
    n, f = int(g["n_nodes"]), int(g["n_feat"])
    x = np.zeros((n, f), np.float32)
    x[np.repeat(np.arange(n), np.diff(g["feat_indptr"])), g["feat_indices"]] = g["feat_data"]
    rowptr, col = graph_io._undirected_binary_csr(g["adj_row"], g["adj_col"], n)
    path = str(tmp_path / "texas.wdgg")
    graph_io.save_graph(path, rowptr, col, g["labels"], x)
    for metric, key, tol in (("node_homo", "m_node_homo", 1e-6), ("adj_homo", "m_adj_homo", 1e-5), ("agg_homo_soft", "m_agg_soft", None)):
        got = float(cli.main(["--dataset_name", path, "--homophily_metric", metric]))
        if tol is None:
            assert abs(got - float(g[key])) <= 2 * 1.01 / n
        else:
            assert got == pytest.approx(float(g[key]), rel=tol, abs=1e-6)
