#!/usr/bin/env python3
"""Kernel-regression metric on one synthetic sweep graph (N = 2000, F = 500): host LAPACK pinv, as the
reference does it, against solver="device" (batched symmetric eigendecomposition on the GPU)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from wdg_amd import synth
from wdg_amd.utils import homophily_metrics as hm

n, f, epochs = 2000, 500, int(os.environ.get("EPOCHS", "4"))
for h in (0.2, 0.8):
    src, dst, labels = synth.regular_graph(n, 5, 2, h, 0)
    x = torch.from_numpy(synth.features(n, f, 0, labels=labels))
    adj = torch.sparse_coo_tensor(torch.from_numpy(__import__("numpy").stack([src, dst])), torch.ones(src.shape[0]), (n, n)).coalesce()
    lab = torch.from_numpy(labels)
    for solver in ("host", "device"):
        torch.manual_seed(3)
        p, secs = hm.classifier_based_performance_metric(x, adj, lab, 5000.0, base_classifier="kernel_reg1", epochs=epochs, solver=solver)
        print(f"h={h}: solver={solver:6s} p={p:.4f}  {secs / epochs * 1e3:8.1f} ms per epoch ({epochs} epochs)", flush=True)
