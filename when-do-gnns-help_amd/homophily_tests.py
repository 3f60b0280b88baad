#!/usr/bin/env python3
"""GPU counterpart of the reference's `homophily_tests.py`: same flags, same branching.

    python -m wdg_amd.homophily_tests --dataset_name cora --homophily_metric agg_homo_soft [--symmetric 1]

What is kept from the reference script (`homophily_tests.py:46-139`):
  * flag names `--dataset_name --symmetric --sample_max --base_classifier --homophily_metric --no-cuda`
    (`--no-cuda` is parsed and, as in the reference, never read - there is no CPU path here anyway);
  * edge/node/class/adjusted/label-informativeness metrics run on the NORMALISED adjacency WITH self loops
    (`:83-85,112-113`); `edge_homo` is handed one-hot labels (`:114-116`);
  * aggregation homophily and the classifier metrics use the RAW adjacency and raw features (`:120,135`,
    SURVEY Q4), ten repetitions and the `2 s - 1` map (`:126-132`), a 10 000-node class-balanced sample above
    that size.
What differs: `class_homo` passes the edge index (the reference passes the sparse tensor and crashes);
data come from `.npz` / `.wdgg` graph files (`--data_dir`, default `$WDG_DATA_DIR` or `./data`, where the reference keeps
its datasets; graph_io.py converts the reference's formats) because the reference's downloads are out of scope.
"""
import argparse
import os

import numpy as np
import torch

from . import ops
from .utils import homophily_metrics as hm
from .utils import util_funcs as uf

BASE_CLASSIFIERS = ['kernel_reg0', 'kernel_reg1', 'gnb']
METRIC_LIST = {  # reference: homophily_tests.py:32-44
    "node_homo": lambda adj, labels: hm.node_homophily(adj, labels),
    "edge_homo": lambda adj, labels: hm.edge_homophily(adj, labels),
    "class_homo": lambda adj, labels: hm.our_measure(adj, labels),
    "node_hom_generalized": lambda adj, features, labels: hm.generalized_edge_homophily(adj, features, labels),
    "agg_homo_soft": lambda x: np.mean(x),
    "agg_homo_hard": lambda x: np.mean(x),
    "adj_homo": lambda adj, labels: hm.adjusted_homo(adj, labels),
    "label_info": lambda adj, labels: hm.label_informativeness(adj, labels),
    "kernel_reg0_based_homo": lambda *a, **k: hm.classifier_based_performance_metric(*a, **k),
    "kernel_reg1_based_homo": lambda *a, **k: hm.classifier_based_performance_metric(*a, **k),
    "gnb_based_homo": lambda *a, **k: hm.classifier_based_performance_metric(*a, **k),
}
LARGE_NODES = 20000  # the reference switches to its scipy path for the LINKX datasets (`LARGE_DATASETS`, :30)


def load_npz_graph(path):
    """-> (adj sparse COO fp32 [N,N] on CPU, features dense fp32, labels int64): what `full_load_data_large` returns.
    File keys: adj_row, adj_col[, adj_val], labels, and either dense `features` or CSR `feat_indptr/indices/data`."""
    if path.endswith(".wdgg"):  # the binary container of graph_io.py
        from . import graph_io
        g = graph_io.load_graph(path)
        rows, cols = graph_io.csr_to_coo(g["rowptr"], g["col"])
        idx = torch.from_numpy(np.vstack([rows, cols]).astype(np.int64))
        adj = torch.sparse_coo_tensor(idx, torch.ones(idx.shape[1]), (g["n_nodes"], g["n_nodes"]))
        return adj, torch.from_numpy(np.array(g["features"], np.float32)), torch.from_numpy(g["labels"].astype(np.int64))
    z = np.load(path)
    n = int(z["labels"].shape[0])
    idx = torch.from_numpy(np.vstack([z["adj_row"], z["adj_col"]]).astype(np.int64))
    val = torch.from_numpy(z["adj_val"].astype(np.float32)) if "adj_val" in z else torch.ones(idx.shape[1])
    adj = torch.sparse_coo_tensor(idx, val, (n, n))
    if "features" in z:
        feats = torch.from_numpy(z["features"].astype(np.float32))
    else:
        f = int(z["n_feat"])
        feats = torch.zeros((n, f))
        rr = np.repeat(np.arange(n), np.diff(z["feat_indptr"]))
        feats[torch.from_numpy(rr), torch.from_numpy(z["feat_indices"].astype(np.int64))] = torch.from_numpy(z["feat_data"])
    return adj, feats, torch.from_numpy(z["labels"].astype(np.int64))


def run(dataset_path, homophily_metric, symmetric=0, sample_max=500, base_classifier='kernel_reg1'):
    device = torch.device("cuda:0")
    adj_raw, features_raw, labels_raw = load_npz_graph(dataset_path)
    nnodes = labels_raw.shape[0]
    if nnodes < LARGE_NODES:  # homophily_tests.py:78-86 (fp32 torch coefficients)
        features = uf.normalize_tensor(features_raw.to(device))
        adj = uf.normalized_adjacency_csr(adj_raw, symmetric=int(symmetric), prec=0).to_torch_sparse()
    else:                      # homophily_tests.py:87-110 (fp64 scipy coefficients, F.normalize on features)
        features = ops.row_l1_normalise(features_raw.to(device), use_abs=True)  # f.normalize(features, p=1, dim=1)
        adj = uf.normalized_adjacency_csr(adj_raw, symmetric=int(symmetric), prec=1).to_torch_sparse()
    labels = labels_raw.to(device).flatten()

    if homophily_metric in ("node_homo", "label_info", "adj_homo"):
        return METRIC_LIST[homophily_metric](adj, labels)
    if homophily_metric == "class_homo":
        return METRIC_LIST[homophily_metric](adj.coalesce().indices(), labels)
    if homophily_metric == "edge_homo":
        onehot = torch.eye(int(labels.max()) + 1, device=device)[labels]
        return METRIC_LIST[homophily_metric](adj, onehot)
    if homophily_metric == "node_hom_generalized":
        return METRIC_LIST[homophily_metric](adj, features, labels)
    if homophily_metric in ("agg_homo_soft", "agg_homo_hard"):
        las = np.zeros(10)
        is_hard = 1 if homophily_metric.endswith("hard") else None
        num_sample = 10000
        label_onehot = torch.eye(int(labels_raw.max()) + 1)[labels_raw]
        for i in range(10):
            idx_train = None
            if nnodes >= num_sample:
                idx_train, _, _ = uf.random_disassortative_splits(labels_raw, labels_raw.max() + 1, num_sample / nnodes)
            las[i] = 2 * float(hm.similarity(label_onehot, adj_raw, label_onehot, hard=is_hard, LP=1, idx_train=idx_train)) - 1
        return METRIC_LIST[homophily_metric](las)
    base = homophily_metric.partition("_based")[0]
    return METRIC_LIST[homophily_metric](features_raw, adj_raw, labels_raw, sample_max, base_classifier=base, epochs=100)


def main(argv=None):
    p = argparse.ArgumentParser(formatter_class=argparse.RawTextHelpFormatter)
    p.add_argument('--no-cuda', action='store_true', default=False, help='accepted for compatibility; ignored')
    p.add_argument('--dataset_name', type=str, required=True, help='name of <data_dir>/real_<name>.npz, or a path to an .npz / .wdgg (graph_io container) graph')
    p.add_argument('--data_dir', type=str, default=os.environ.get("WDG_DATA_DIR", "./data"),
                   help='directory holding real_<name>.npz graph files (default: $WDG_DATA_DIR or ./data, the reference\'s data directory)')
    p.add_argument('--symmetric', type=float, default=0, help='1 for symmetric renormalized adj, 0 for random walk renormalized adj')
    p.add_argument('--sample_max', type=float, default=500, help='maxinum number of samples used in gntk')
    p.add_argument('--base_classifier', type=str, default='kernel_reg1', choices=BASE_CLASSIFIERS)
    p.add_argument('--homophily_metric', required=True, choices=list(METRIC_LIST.keys()))
    args = p.parse_args(argv)
    path = (args.dataset_name if args.dataset_name.endswith((".npz", ".wdgg"))
            else os.path.join(args.data_dir, f"real_{args.dataset_name}.npz"))
    lvl = run(path, args.homophily_metric, args.symmetric, args.sample_max, args.base_classifier)
    if isinstance(lvl, tuple):
        lvl = lvl[0]
    print(f"The Homophily level of given dataset {args.dataset_name} is {lvl} using metric {args.homophily_metric}")
    return lvl


if __name__ == "__main__":
    main()
