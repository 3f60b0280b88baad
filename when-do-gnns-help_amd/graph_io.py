"""Fixture / loader layer: converters from the reference's on-disk formats to a compact binary
graph container, and the container's host and device readers.

The reference reads its datasets through networkx, pickles and torch.save files (`utils/util_funcs.py:49-97` Planetoid
pickles, `:289-337` the `new_data/<name>/out1_*.txt` TSV pair, `synthetic_plot.py:81-92` `data_synthesis/*.pt`).  The
converters restate what those loaders RETURN (node order, undirected binary adjacency, label vector, feature matrix),
without networkx, so that a dataset can be converted once where the raw files live and fed to the GPU box as one file:

    container (".wdgg", little endian):
        magic "WDGG", u32 version, i64 n_nodes, i64 nnz, i64 n_feat, i64 n_classes, u32 feature_kind, u32 reserved
        i32 rowptr[n+1], i32 col[nnz]            CSR pattern, rows sorted by column (binary adjacency)
        i32 labels[n]
        features: kind 0 = none; 1 = fp32 dense [n, n_feat];
                  2 = bit-packed binary [n, ceil(n_feat / 32)] u32 words, bit j of word w = feature 32 w + j
Binary bag-of-words features (cora, citeseer, texas, film, ...) shrink 32x; `load_device` unpacks them on the GPU
(`wdg_unpack_bits_f32`, optionally fused with the row-L1 normalisation of `preprocess_features`,
`utils/util_funcs.py:39-46`).
"""
import os
import pickle
import struct

import numpy as np

MAGIC = b"WDGG"
VERSION = 1
FEAT_NONE, FEAT_F32, FEAT_BITS = 0, 1, 2


# ------------------------------------------------------------------------------------------------ helpers
def _undirected_binary_csr(src, dst, n):
    """What `nx.adjacency_matrix` of an undirected simple graph holds: every edge in both directions, duplicates merged,
    a self-loop once; int32 CSR with sorted rows."""
    src, dst = np.asarray(src, np.int64), np.asarray(dst, np.int64)
    a = np.concatenate([src, dst])
    b = np.concatenate([dst, src])
    key = np.unique(a * n + b)
    rows, cols = key // n, key % n
    rowptr = np.zeros(n + 1, np.int64)
    np.add.at(rowptr, rows + 1, 1)
    return np.cumsum(rowptr).astype(np.int32), cols.astype(np.int32)


def csr_to_coo(rowptr, col):
    return np.repeat(np.arange(len(rowptr) - 1, dtype=np.int32), np.diff(rowptr)), np.asarray(col, np.int32)


# ------------------------------------------------------------------------------------------------ converters
def read_geom_gcn(edges_path, features_path, film=False, film_width=932):
    """`new_data/<name>/out1_graph_edges.txt` + `out1_node_feature_label.txt` -> (rowptr, col, features u8 [n, F], labels).

    Follows `utils/util_funcs.py:289-337`: the graph is undirected; only nodes that occur in an edge exist, in
    ascending id order (`sorted(G.nodes())`); `film` stores the indices of the set features instead of a 0/1 list."""
    feats, labels = {}, {}
    with open(features_path) as f:
        f.readline()
        for line in f:
            node, feat, lab = line.rstrip().split("\t")
            node = int(node)
            assert node not in feats
            if film:
                v = np.zeros(film_width, np.uint8)
                v[np.array(feat.split(","), dtype=np.uint16)] = 1
            else:
                v = np.array(feat.split(","), dtype=np.uint8)
            feats[node], labels[node] = v, int(lab)
    e = np.loadtxt(edges_path, dtype=np.int64, skiprows=1, delimiter="\t", ndmin=2)
    nodes = np.unique(e)
    index = {int(v): i for i, v in enumerate(nodes)}
    src = np.fromiter((index[int(v)] for v in e[:, 0]), np.int64, e.shape[0])
    dst = np.fromiter((index[int(v)] for v in e[:, 1]), np.int64, e.shape[0])
    rowptr, col = _undirected_binary_csr(src, dst, len(nodes))
    x = np.stack([feats[int(v)] for v in nodes])
    y = np.array([labels[int(v)] for v in nodes], np.int64)
    return rowptr, col, x, y


def read_edge_tsv(edges_path):
    """Topology only (squirrel / chameleon ship no feature file in the reference checkout): -> (rowptr, col, n)."""
    e = np.loadtxt(edges_path, dtype=np.int64, skiprows=1, delimiter="\t", ndmin=2)
    nodes = np.unique(e)
    remap = np.searchsorted(nodes, e)
    rowptr, col = _undirected_binary_csr(remap[:, 0], remap[:, 1], len(nodes))
    return rowptr, col, len(nodes)


def read_planetoid(data_dir, name):
    """`data/ind.<name>.{x,y,tx,ty,allx,ally,graph,test.index}` -> (rowptr, col, features fp32 dense, labels).

    Follows `load_data` + `full_load_data_large` (`utils/util_funcs.py:49-97,208-211`): test rows are put back in graph
    order, citeseer's isolated test nodes become zero rows, the adjacency is the undirected simple graph of the
    neighbour lists (node order = key order of the dict), labels = argmax of the one-hot rows."""
    import scipy.sparse as sp
    objs = []
    for part in ("x", "y", "tx", "ty", "allx", "ally", "graph"):
        with open(os.path.join(data_dir, f"ind.{name}.{part}"), "rb") as f:
            objs.append(pickle.load(f, encoding="latin1"))
    x, y, tx, ty, allx, ally, graph = objs
    test_idx = np.array([int(l.strip()) for l in open(os.path.join(data_dir, f"ind.{name}.test.index"))])
    test_sorted = np.sort(test_idx)
    if name == "citeseer":
        full = np.arange(test_idx.min(), test_idx.max() + 1)
        tx_ext = sp.lil_matrix((len(full), x.shape[1]))
        tx_ext[test_sorted - test_sorted.min(), :] = tx
        ty_ext = np.zeros((len(full), y.shape[1]))
        ty_ext[test_sorted - test_sorted.min(), :] = ty
        tx, ty = tx_ext, ty_ext
    feats = sp.vstack((allx, tx)).tolil()
    feats[test_idx, :] = feats[test_sorted, :]
    lab = np.vstack((ally, ty))
    lab[test_idx, :] = lab[test_sorted, :]
    keys = list(graph.keys())
    index = {k: i for i, k in enumerate(keys)}
    for nbrs in graph.values():  # neighbours that are not keys are appended in first-seen order, as add_edges_from does
        for v in nbrs:
            if v not in index:
                index[v] = len(index)
    src = np.fromiter((index[k] for k, nbrs in graph.items() for _ in nbrs), np.int64)
    dst = np.fromiter((index[v] for nbrs in graph.values() for v in nbrs), np.int64)
    rowptr, col = _undirected_binary_csr(src, dst, len(index))
    return rowptr, col, np.asarray(feats.todense(), np.float32), np.argmax(lab, axis=-1).astype(np.int64)


def read_synthetic_pt(adj_path):
    """`data_synthesis/<k*400>/<h>/adj_<h>_<s>.pt` (torch sparse COO fp64, directed, no self-loops) -> (rowptr, col, n);
    labels are `arange(n) // (n / 5)` (`synthetic_plot.py:83`)."""
    import torch
    a = torch.load(adj_path, map_location="cpu").coalesce()
    idx = a.indices().numpy()
    n = a.shape[0]
    order = np.lexsort((idx[1], idx[0]))
    rows, cols = idx[0][order], idx[1][order]
    rowptr = np.zeros(n + 1, np.int64)
    np.add.at(rowptr, rows + 1, 1)
    return np.cumsum(rowptr).astype(np.int32), cols.astype(np.int32), n


# ------------------------------------------------------------------------------------------------ the container
def pack_bits(x):
    """0/1 matrix [n, F] -> u32 words [n, ceil(F / 32)], bit j of word w = feature 32 w + j."""
    x = np.asarray(x)
    n, f = x.shape
    words = (f + 31) // 32
    if n == 0:
        return np.zeros((0, words), "<u4")
    padded = np.zeros((n, words * 32), np.uint8)
    padded[:, :f] = x != 0
    return np.packbits(padded.reshape(n, words, 32), axis=2, bitorder="little").view("<u4").reshape(n, words)


def unpack_bits(words, n_feat):
    w = np.ascontiguousarray(words, dtype="<u4")
    if w.shape[0] == 0:
        return np.zeros((0, n_feat), np.float32)
    bits = np.unpackbits(w.view(np.uint8).reshape(w.shape[0], -1), axis=1, bitorder="little")
    return bits[:, :n_feat].astype(np.float32)


def save_graph(path, rowptr, col, labels, features=None, n_classes=None, pack=None):
    """Write the container.  `pack=None` bit-packs the features when every entry is 0 or 1."""
    rowptr, col = np.asarray(rowptr, "<i4"), np.asarray(col, "<i4")
    labels = np.asarray(labels, "<i4")
    n = len(rowptr) - 1
    assert labels.shape == (n,) and rowptr[-1] == len(col)
    kind, n_feat, payload = FEAT_NONE, 0, b""
    if features is not None:
        features = np.asarray(features)
        n_feat = features.shape[1]
        if pack is None:
            pack = bool(np.isin(features, (0, 1)).all())
        if pack:
            kind, payload = FEAT_BITS, pack_bits(features).tobytes()
        else:
            kind, payload = FEAT_F32, np.ascontiguousarray(features, "<f4").tobytes()
    c = int(n_classes) if n_classes is not None else (int(labels.max()) + 1 if n else 0)
    with open(path, "wb") as f:
        f.write(MAGIC + struct.pack("<Iqqqq II", VERSION, n, len(col), n_feat, c, kind, 0))
        f.write(rowptr.tobytes() + col.tobytes() + labels.tobytes() + payload)


def load_graph(path, unpack=True):
    """-> dict(rowptr, col, labels, features | feature_words, n_nodes, n_feat, n_classes, feature_kind); host arrays."""
    with open(path, "rb") as f:
        blob = f.read()
    if blob[:4] != MAGIC:
        raise ValueError(f"{path}: not a WDGG graph container")
    version, n, nnz, n_feat, c, kind, _ = struct.unpack_from("<Iqqqq II", blob, 4)
    if version != VERSION:
        raise ValueError(f"{path}: container version {version}, this reader understands {VERSION}")
    off = 4 + struct.calcsize("<Iqqqq II")
    rowptr = np.frombuffer(blob, "<i4", n + 1, off); off += 4 * (n + 1)
    col = np.frombuffer(blob, "<i4", nnz, off); off += 4 * nnz
    labels = np.frombuffer(blob, "<i4", n, off); off += 4 * n
    out = dict(rowptr=rowptr, col=col, labels=labels, n_nodes=n, n_feat=n_feat, n_classes=c, feature_kind=kind)
    if kind == FEAT_F32:
        out["features"] = np.frombuffer(blob, "<f4", n * n_feat, off).reshape(n, n_feat)
    elif kind == FEAT_BITS:
        words = np.frombuffer(blob, "<u4", n * ((n_feat + 31) // 32), off).reshape(n, (n_feat + 31) // 32)
        out["feature_words"] = words
        if unpack:
            out["features"] = unpack_bits(words, n_feat)
    elif kind != FEAT_NONE:
        raise ValueError(f"{path}: unknown feature kind {kind}")
    return out


def load_device(path, row_normalise=False):
    """Container -> (ops.CsrGraph, features fp32 on the GPU | None, labels int64 on the GPU).  Bit-packed features are
    uploaded packed and expanded by `wdg_unpack_bits_f32`; `row_normalise` fuses `preprocess_features`' row-L1 scaling
    (rows summing to 0 stay 0: the reference's inf -> 0 guard)."""
    import torch

    from . import ops
    g = load_graph(path, unpack=False)
    dev = ops.require_gpu()
    graph = ops.CsrGraph(torch.from_numpy(g["rowptr"].copy()).to(dev), torch.from_numpy(g["col"].copy()).to(dev), None,
                         g["n_nodes"], g["n_nodes"])
    labels = torch.from_numpy(g["labels"].astype(np.int64)).to(dev)
    feats = None
    if g["feature_kind"] == FEAT_BITS:
        words = torch.from_numpy(g["feature_words"].astype(np.int32)).to(dev)  # same bits, torch has no uint32 arithmetic
        feats = ops.unpack_bits(words, g["n_feat"], row_normalise=row_normalise)
    elif g["feature_kind"] == FEAT_F32:
        feats = torch.from_numpy(g["features"].copy()).to(dev)
        if row_normalise:
            feats = ops.row_l1_normalise(feats)
    return graph, feats, labels


def main(argv=None):
    """python -m wdg_amd.graph_io <kind> <out.wdgg> <inputs...>
         geom-gcn  <out> <out1_graph_edges.txt> <out1_node_feature_label.txt> [--film]
         planetoid <out> <data_dir> <name>
         topology  <out> <out1_graph_edges.txt>          (no features / labels: labels all 0)
         synthetic <out> <adj_<h>_<s>.pt> [features.npy] (labels = arange(n) // (n / 5))"""
    import argparse
    p = argparse.ArgumentParser(description=main.__doc__, formatter_class=argparse.RawTextHelpFormatter)
    p.add_argument("kind", choices=["geom-gcn", "planetoid", "topology", "synthetic"])
    p.add_argument("out")
    p.add_argument("inputs", nargs="+")
    p.add_argument("--film", action="store_true")
    a = p.parse_args(argv)
    if a.kind == "geom-gcn":
        rowptr, col, x, y = read_geom_gcn(a.inputs[0], a.inputs[1], film=a.film)
    elif a.kind == "planetoid":
        rowptr, col, x, y = read_planetoid(a.inputs[0], a.inputs[1])
    elif a.kind == "topology":
        rowptr, col, n = read_edge_tsv(a.inputs[0])
        x, y = None, np.zeros(n, np.int64)
    else:
        rowptr, col, n = read_synthetic_pt(a.inputs[0])
        x = np.load(a.inputs[1]) if len(a.inputs) > 1 else None
        y = np.arange(n) // (n // 5)
    save_graph(a.out, rowptr, col, y, x)
    print(f"{a.out}: {len(rowptr) - 1} nodes, {len(col)} stored entries, "
          f"{0 if x is None else x.shape[1]} features, {os.path.getsize(a.out)} bytes")


if __name__ == "__main__":
    main()
