"""Helpers for reading tests/golden/*.npz (data only; produced by make_golden.py)."""
import glob
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REAL = ["cora", "citeseer", "film", "texas"]
SYN = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "syn_*.npz")))


def load(name):
    return dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))


def dense_features(g, key="feat_data"):
    n, f = int(g["n_nodes"]), int(g["n_feat"])
    x = np.zeros((n, f), np.float32)
    rr = np.repeat(np.arange(n), np.diff(g["feat_indptr"]))
    x[rr, g["feat_indices"]] = g[key]
    return x
