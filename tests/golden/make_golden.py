#!/usr/bin/env python3
"""Golden-vector generator: runs the REAL reference (read-only /root/reference) on CPU.

This script is the only place the reference's Python is imported.  It runs in the
build container (where /root/reference exists), never on the GPU box, and writes
small `.npz` fixtures (inputs + expected outputs) next to itself.  Nothing from
the reference's source text is stored - only data the reference computed.

Missing third-party modules are stubbed exactly as SURVEY.md Appendix C describes
(dgl, torch_geometric, torch_scatter, ogb, google_drive_downloader).

    python tests/golden/make_golden.py            # regenerate all fixtures
"""
import os
import sys
import types

import numpy as np
import scipy.sparse as sp
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


# --------------------------------------------------------------------------- stubs
def _install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    mod("dgl")

    def to_undirected(edge_index, num_nodes=None):
        ei = torch.cat([edge_index, edge_index.flip(0)], dim=1)
        n = int(ei.max()) + 1 if num_nodes is None else num_nodes
        key = torch.unique(ei[0] * n + ei[1])
        return torch.stack([key // n, key % n])

    def to_scipy_sparse_matrix(edge_index, edge_attr=None, num_nodes=None):
        row, col = edge_index.cpu().numpy()
        n = int(edge_index.max()) + 1 if num_nodes is None else num_nodes
        return sp.coo_matrix((np.ones(row.shape[0]), (row, col)), shape=(n, n))

    tg = mod("torch_geometric")
    tgu = mod("torch_geometric.utils", to_undirected=to_undirected)
    tguc = mod("torch_geometric.utils.convert", to_scipy_sparse_matrix=to_scipy_sparse_matrix)
    tg.utils = tgu
    tgu.convert = tguc

    def scatter_add(src, index, dim=-1, out=None):
        return out.scatter_add_(dim, index, src)

    mod("torch_scatter", scatter_add=scatter_add)
    ogb = mod("ogb")
    ogb.nodeproppred = mod("ogb.nodeproppred", NodePropPredDataset=None)
    mod("google_drive_downloader", GoogleDriveDownloader=None)


def _import_reference():
    _install_stubs()
    sys.path.insert(0, REF)
    os.chdir(REF)
    from utils import util_funcs, homophily_metrics, homophily_plot  # noqa
    return util_funcs, homophily_metrics, homophily_plot


# --------------------------------------------------------------------------- helpers
def _f(x):
    """python/torch/numpy scalar -> float64"""
    if isinstance(x, torch.Tensor):
        return float(x.detach().cpu().double())
    return float(x)


def _dense_to_csr_parts(x):
    m = sp.csr_matrix(np.asarray(x))
    m.sort_indices()
    return m.indptr.astype(np.int32), m.indices.astype(np.int32), m.data.astype(np.float32)


def _sample_rows(n, k, seed=1234):
    rng = np.random.default_rng(seed)
    return np.sort(rng.choice(n, size=min(k, n), replace=False)).astype(np.int32)


def _y_summary(Y, rows):
    Y = Y.detach().cpu()
    return dict(
        y_rows=Y[torch.as_tensor(rows, dtype=torch.long)].numpy().astype(np.float32),
        y_rowsum=Y.double().sum(1).numpy(),
        y_colsum=Y.double().sum(0).numpy(),
        y_fro=np.float64(torch.linalg.norm(Y.double())),
    )


# --------------------------------------------------------------------------- real graphs
def golden_real(uf, hm, name, n_rows=48):
    print(f"[golden] real dataset {name}")
    adj_raw, features, labels = uf.full_load_data_large(name)
    adj_raw = adj_raw.coalesce()
    N = labels.shape[0]
    out = {}
    ri = adj_raw.indices().numpy()
    out["adj_row"], out["adj_col"] = ri[0].astype(np.int32), ri[1].astype(np.int32)
    out["adj_val"] = adj_raw.values().numpy().astype(np.float32)
    out["labels"] = labels.numpy().astype(np.int64)
    out["n_nodes"] = np.int64(N)
    fp, fi, fd = _dense_to_csr_parts(features.numpy())
    out["feat_indptr"], out["feat_indices"], out["feat_data"] = fp, fi, fd
    out["n_feat"] = np.int64(features.shape[1])

    # homophily_tests.py:80 - dense row-normalised features (normalize_tensor)
    feat_n = uf.normalize_tensor(features)
    # expected values at the stored sparsity pattern
    rr = np.repeat(np.arange(N), np.diff(fp))
    out["featn_data"] = feat_n.numpy()[rr, fi].astype(np.float32)
    # large path: f.normalize(features, p=1, dim=1)  (homophily_tests.py:94)
    feat_l1 = torch.nn.functional.normalize(features, p=1, dim=1)
    out["featl1_data"] = feat_l1.numpy()[rr, fi].astype(np.float32)

    rows = _sample_rows(N, n_rows)
    out["sample_rows"] = rows

    # small path (homophily_tests.py:83-85): normalize_tensor(eye + dense) -> to_sparse
    for tag, symmetric in (("rw", 0), ("sym", 1)):
        a = uf.normalize_tensor(torch.eye(N) + adj_raw.to_dense(), symmetric=symmetric).to_sparse().coalesce()
        ai = a.indices().numpy()
        out[f"small_{tag}_row"], out[f"small_{tag}_col"] = ai[0].astype(np.int32), ai[1].astype(np.int32)
        out[f"small_{tag}_val"] = a.values().numpy().astype(np.float32)
        Y = torch.spmm(a, feat_n)
        for k, v in _y_summary(Y, rows).items():
            out[f"small_{tag}_{k}"] = v
        if tag == "rw":
            a_rw = a

    # large path (homophily_tests.py:98-104): scipy fp64 normalisers -> torch COO fp32
    sp_adj = sys.modules["torch_geometric.utils.convert"].to_scipy_sparse_matrix(adj_raw.indices(), num_nodes=N)
    for tag, fn in (("rw", uf.row_normalized_adjacency), ("sym", uf.sys_normalized_adjacency)):
        a = uf.sparse_mx_to_torch_sparse_tensor(fn(sp_adj))
        # un-coalesced COO exactly as sparse_mx_to_torch_sparse_tensor returns it
        ai = a._indices().numpy()
        out[f"large_{tag}_row"], out[f"large_{tag}_col"] = ai[0].astype(np.int32), ai[1].astype(np.int32)
        out[f"large_{tag}_val"] = a._values().numpy().astype(np.float32)
        Y = torch.spmm(a, feat_l1)
        for k, v in _y_summary(Y, rows).items():
            out[f"large_{tag}_{k}"] = v

    # metrics on the rw-normalised adjacency with self loops (homophily_tests.py:112-116)
    adj = a_rw
    m = {}
    m["node_homo"] = _f(hm.node_homophily(adj, labels))
    m["edge_homo"] = _f(hm.edge_homophily(adj, labels))
    onehot = torch.eye(int(labels.max()) + 1)[labels]
    m["edge_homo_onehot_quirk"] = _f(hm.edge_homophily(adj, onehot))
    m["class_homo"] = _f(hm.our_measure(adj.indices(), labels))  # edge-index input
    m["adj_homo"] = _f(hm.adjusted_homo(adj, labels))
    m["label_info"] = _f(hm.label_informativeness(adj, labels))
    p, p_bar, pc = hm.class_distribution(adj, labels)
    out["cd_p"], out["cd_p_bar"], out["cd_pc"] = p.numpy(), p_bar.numpy(), pc.numpy()
    H = hm.compact_matrix_edge_idx(adj.indices(), labels)
    out["compat_H"] = H.numpy()
    m["ge_homo"] = _f(hm.generalized_edge_homophily(adj, feat_n, labels))
    # aggregation homophily on the RAW adjacency (homophily_tests.py:119-132)
    m["agg_soft"] = 2 * _f(hm.similarity(onehot, adj_raw, onehot, hard=None, LP=1, idx_train=None)) - 1
    m["agg_hard"] = 2 * _f(hm.similarity(onehot, adj_raw, onehot, hard=1, LP=1, idx_train=None)) - 1
    # sampled branch (idx_train bool mask), mask stored
    torch.manual_seed(0)
    idx_train, _, _ = uf.random_disassortative_splits(labels, labels.max() + 1, 0.3)
    out["las_mask"] = idx_train.numpy()
    m["agg_soft_masked"] = _f(hm.similarity(onehot, adj_raw, onehot, hard=None, LP=1, idx_train=idx_train))
    m["agg_hard_masked"] = _f(hm.similarity(onehot, adj_raw, onehot, hard=1, LP=1, idx_train=idx_train))
    # similarity with real-valued features (row-normalised X) on the normalised adjacency
    m["sim_feat_soft"] = _f(hm.similarity(feat_n, adj, onehot, hard=None, LP=1, idx_train=None))
    m["sim_feat_hard"] = _f(hm.similarity(feat_n, adj, onehot, hard=1, LP=1, idx_train=None))

    # splits (util_funcs.py:454-475) under a fixed seed
    torch.manual_seed(7)
    tr, va, te = uf.random_disassortative_splits(labels, labels.max() + 1)
    out["split_train"], out["split_val"], out["split_test"] = tr.numpy(), va.numpy(), te.numpy()

    # GNTK kernels (homophily_metrics.py:232-257) on a stored sample, raw adj + raw features
    smp = _sample_rows(N, 96, seed=99).astype(np.int64)
    out["gntk_sample"] = smp
    for nl in (0, 1):
        KG, KX = hm.gntk_homophily_(features, adj_raw, smp, nl)
        out[f"gntk_KG_l{nl}"] = KG.numpy().astype(np.float32)
        out[f"gntk_KX_l{nl}"] = KX.numpy().astype(np.float32)

    # classifier-based metric, seeded (homophily_tests.py:133-137); p-values depend on the RNG stream
    for clf in ("kernel_reg0", "kernel_reg1", "gnb"):
        torch.manual_seed(11)
        pval, _ = hm.classifier_based_performance_metric(features, adj_raw, labels, 200.0,
                                                         base_classifier=clf, epochs=6)
        m[f"cpm_{clf}_seed11_e6_s200"] = float(pval)

    for k, v in m.items():
        out["m_" + k] = np.float64(v)
    np.savez_compressed(os.path.join(HERE, f"real_{name}.npz"), **out)
    print("   ", {k: round(v, 8) for k, v in m.items()})


# --------------------------------------------------------------------------- synthetic graphs
def golden_syn(uf, hp, k_dir, h, seed, n_rows=32):
    print(f"[golden] synthetic {k_dir}/{h}/{seed}")
    # synthetic_plot.py:81-92
    feats_raw = torch.load(f"./data_synthesis/features/pubmed/pubmed_{seed}.pt").clone().detach().float()
    features = torch.tensor(uf.preprocess_features(feats_raw)).clone().detach()
    adj_sp = torch.load(f"./data_synthesis/{k_dir}/{h}/adj_{h}_{seed}.pt").coalesce()
    adj_raw = adj_sp.to_dense().clone().detach().float()
    label = torch.load(f"./data_synthesis/{k_dir}/{h}/label_{h}_{seed}.pt").to_dense().clone().detach().float()
    N = adj_raw.shape[0]
    adj = torch.tensor(uf.normalize(adj_raw + torch.eye(N)))

    out = {}
    ai = adj_sp.indices().numpy()
    out["adj_row"], out["adj_col"] = ai[0].astype(np.int32), ai[1].astype(np.int32)
    out["labels"] = torch.argmax(label, 1).numpy().astype(np.int64)
    out["n_nodes"] = np.int64(N)
    fp, fi, fd = _dense_to_csr_parts(feats_raw.numpy())
    out["feat_indptr"], out["feat_indices"], out["feat_data"] = fp, fi, fd
    out["n_feat"] = np.int64(feats_raw.shape[1])
    rr = np.repeat(np.arange(N), np.diff(fp))
    out["featn_data"] = features.numpy()[rr, fi].astype(np.float32)

    an = adj.to_sparse().coalesce()
    ni = an.indices().numpy()
    out["norm_row"], out["norm_col"] = ni[0].astype(np.int32), ni[1].astype(np.int32)
    out["norm_val"] = an.values().numpy().astype(np.float32)

    rows = _sample_rows(N, n_rows)
    out["sample_rows"] = rows
    Y = torch.spmm(adj, features)  # dense x dense, as the reference does on this path
    out.update(_y_summary(Y, rows))

    lab = torch.argmax(label, 1)
    m = {}
    m["edge_homo"] = _f(hp.edge_homophily(adj, label))
    m["node_homo"] = _f(hp.node_homophily(adj, lab))
    m["class_homo"] = _f(hp.our_measure(adj, lab))
    m["soft_las"] = _f(hp.similarity(label, adj, label, NTK=None, hard=None, LP=1))
    m["hard_las"] = _f(hp.similarity(label, adj, label, NTK=None, hard=1, LP=1))
    m["adj_homo"] = _f(hp.adjusted_homo(adj, label))
    m["label_info"] = _f(hp.label_informativeness(adj, label))
    m["ge_homo"] = _f(hp.generalized_edge_homophily(adj, features, label))
    for clf in ("kernel_reg0", "kernel_reg1"):
        torch.manual_seed(5)
        m[f"cpm_{clf}_seed5_e4_s500"] = float(hp.classifier_based_performance_metric(
            features, adj, lab, sample_max=500, base_classifier=clf, epochs=4))
    for k, v in m.items():
        out["m_" + k] = np.float64(v)
    np.savez_compressed(os.path.join(HERE, f"syn_{k_dir}_{h}_{seed}.npz"), **out)
    print("   ", {k: round(v, 8) for k, v in m.items()})


def main():
    uf, hm, hp = _import_reference()
    torch.set_num_threads(8)
    for name in ("cora", "citeseer", "film", "texas"):
        golden_real(uf, hm, name)
    for k_dir, h, seed in (("800", 0.5, 0), ("800", 0.05, 0), ("800", 0.9, 1), ("4000", 0.2, 0), ("4000", 0.15, 2)):
        golden_syn(uf, hp, k_dir, h, seed)


if __name__ == "__main__":
    main()
