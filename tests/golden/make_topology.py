#!/usr/bin/env python3
"""Topology fixtures for BASELINE config C4 (squirrel, chameleon): data conversion only.

The reference's loader (`utils/util_funcs.py:289-337`) reads `new_data/<name>/out1_graph_edges.txt` (TSV, one header
line, `src<TAB>dst`), builds an UNDIRECTED networkx graph and takes the adjacency ordered by sorted node id; the node
feature/label files of these two datasets are missing from the checkout, so only the topology can be a
fixture.  Labels come from `squirrel_target.csv`-style files when present, else are absent.  Run in the build
container:  python tests/golden/make_topology.py
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/new_data"

for name in ("squirrel", "chameleon"):
    e = np.loadtxt(os.path.join(REF, name, "out1_graph_edges.txt"), dtype=np.int64, skiprows=1)
    n = int(e.max()) + 1
    key = np.unique(np.concatenate([e[:, 0] * n + e[:, 1], e[:, 1] * n + e[:, 0]]))  # undirected, loops kept once
    row, col = (key // n).astype(np.int32), (key % n).astype(np.int32)
    deg = np.bincount(row, minlength=n)
    print(name, "N", n, "nnz", row.shape[0], "loops", int((row == col).sum()), "max deg", int(deg.max()), "mean", deg.mean())
    np.savez_compressed(os.path.join(HERE, f"topo_{name}.npz"), adj_row=row, adj_col=col, n_nodes=np.int64(n))
