"""Fixture / loader layer, CPU side: the converters reproduce what the reference's own loaders return
(golden fixtures captured from `full_load_data_large` / `torch.load`), and the binary container round-trips."""
import os

import numpy as np
import pytest

from _golden import load

REF = "/root/reference"
needs_ref = pytest.mark.skipif(not os.path.isdir(REF), reason="raw dataset files live in the reference checkout")


def _coo_of(golden):
    n = int(golden["n_nodes"])
    key = np.unique(golden["adj_row"].astype(np.int64) * n + golden["adj_col"])
    return key // n, key % n


def _dense_features(golden):
    n, f = int(golden["n_nodes"]), int(golden["n_feat"])
    x = np.zeros((n, f), np.float32)
    rows = np.repeat(np.arange(n), np.diff(golden["feat_indptr"]))
    x[rows, golden["feat_indices"]] = golden["feat_data"]
    return x


@needs_ref
@pytest.mark.parametrize("name", ["texas", "film"])
def test_geom_gcn_converter_matches_reference_loader(name):
    from wdg_amd import graph_io
    d = os.path.join(REF, "new_data", name)
    rowptr, col, x, y = graph_io.read_geom_gcn(os.path.join(d, "out1_graph_edges.txt"),
                                               os.path.join(d, "out1_node_feature_label.txt"), film=(name == "film"))
    g = load("real_" + name)
    rows, cols = graph_io.csr_to_coo(rowptr, col)
    want_r, want_c = _coo_of(g)
    np.testing.assert_array_equal(rows, want_r)
    np.testing.assert_array_equal(cols, want_c)
    np.testing.assert_array_equal(y, g["labels"])
    np.testing.assert_array_equal(x.astype(np.float32), _dense_features(g))


@needs_ref
@pytest.mark.parametrize("name", ["cora", "citeseer"])
def test_planetoid_converter_matches_reference_loader(name):
    from wdg_amd import graph_io
    rowptr, col, x, y = graph_io.read_planetoid(os.path.join(REF, "data"), name)
    g = load("real_" + name)
    rows, cols = graph_io.csr_to_coo(rowptr, col)
    want_r, want_c = _coo_of(g)
    np.testing.assert_array_equal(rows, want_r)
    np.testing.assert_array_equal(cols, want_c)
    np.testing.assert_array_equal(y, g["labels"])
    np.testing.assert_array_equal(x, _dense_features(g))


@needs_ref
def test_synthetic_pt_and_topology_converters():
    from wdg_amd import graph_io
    rowptr, col, n = graph_io.read_synthetic_pt(os.path.join(REF, "data_synthesis", "800", "0.5", "adj_0.5_0.pt"))
    g = load("syn_800_0.5_0")
    assert n == int(g["n_nodes"])
    rows, cols = graph_io.csr_to_coo(rowptr, col)
    order = np.lexsort((g["adj_col"], g["adj_row"]))
    np.testing.assert_array_equal(rows, g["adj_row"][order])
    np.testing.assert_array_equal(cols, g["adj_col"][order])
    rowptr, col, n = graph_io.read_edge_tsv(os.path.join(REF, "new_data", "chameleon", "out1_graph_edges.txt"))
    t = load("topo_chameleon")
    assert n == int(t["n_nodes"])
    rows, cols = graph_io.csr_to_coo(rowptr, col)
    want_r, want_c = _coo_of(t)
    np.testing.assert_array_equal(rows, want_r)
    np.testing.assert_array_equal(cols, want_c)


@pytest.mark.parametrize("n,f,binary", [(37, 1, True), (64, 32, True), (100, 45, True), (183, 1703, True), (50, 33, False), (0, 8, True)])
def test_container_roundtrip(tmp_path, n, f, binary):
    from wdg_amd import graph_io
    rng = np.random.default_rng(n + f)
    src, dst = rng.integers(0, max(n, 1), 3 * n), rng.integers(0, max(n, 1), 3 * n)
    rowptr, col = graph_io._undirected_binary_csr(src, dst, n) if n else (np.zeros(1, np.int32), np.zeros(0, np.int32))
    labels = rng.integers(0, 4, n)
    x = (rng.random((n, f)) < 0.2).astype(np.float32) if binary else rng.standard_normal((n, f)).astype(np.float32)
    path = str(tmp_path / "g.wdgg")
    graph_io.save_graph(path, rowptr, col, labels, x, n_classes=4)
    g = graph_io.load_graph(path)
    assert g["feature_kind"] == (graph_io.FEAT_BITS if binary else graph_io.FEAT_F32)
    assert (g["n_nodes"], g["n_feat"], g["n_classes"]) == (n, f, 4)
    np.testing.assert_array_equal(g["rowptr"], rowptr)
    np.testing.assert_array_equal(g["col"], col)
    np.testing.assert_array_equal(g["labels"], labels)
    np.testing.assert_array_equal(g["features"], x)
    if binary and n:
        assert os.path.getsize(path) < 4 * n * f / 8 + 4 * (len(col) + 2 * n + 1) + 4 * n * ((f + 31) // 32) + 128
        words = graph_io.pack_bits(x)
        assert words.shape == (n, (f + 31) // 32) and words.dtype == np.dtype("<u4")
        assert int(words[0, 0]) & 1 == int(x[0, 0] != 0)  # bit 0 of word 0 = feature 0
    with pytest.raises(ValueError):
        open(path, "r+b").write(b"XXXX")
        graph_io.load_graph(path)
