"""Edge / label statistics, LAS and the per-edge cosine (reference: utils/homophily_plot.py:43-231 and
utils/homophily_metrics.py:164-187): one pass over the pattern per graph, or job tables for a sweep shard."""
import ctypes
import os

import numpy as np
import torch

from . import _lib
from ._lib import c_void_p, check, lib, require_gpu, stream_handle
from ._rt import *  # noqa: F401,F403  (the flag values of include/wdg.h)
from ._rt import _dev, _h2d, _ld, _ptr, _table
from ._lib import StatsJob


# ------------------------------------------------------------------------------------------- edge/label stats
def edge_label_stats(g, labels, n_classes=None, per_row=True):
    """One pass over the pattern -> dict of exact integer tensors (see wdg_edge_label_stats)."""
    dev = g.device
    labels = _dev(labels, torch.int32, dev)
    if labels.shape[0] != g.n_rows:
        raise ValueError("edge_label_stats: one label per node expected")
    c = int(n_classes) if n_classes is not None else (int(labels.max().item()) + 1 if labels.numel() else 0)
    n = g.n_rows
    st = dict(totals=torch.empty(6, dtype=torch.int64, device=dev),
              compat=torch.empty((c, c), dtype=torch.int64, device=dev),
              classdeg=torch.empty(c, dtype=torch.int64, device=dev))
    if per_row:
        for k in ("row_nnz", "row_nnz_noself", "row_match_noself"):
            st[k] = torch.empty(n, dtype=torch.int32, device=dev)
    check(lib.wdg_edge_label_stats(_ptr(g.rowptr), _ptr(g.col), _ptr(labels), n, c, _ptr(st["totals"]),
                                   _ptr(st.get("row_nnz")), _ptr(st.get("row_nnz_noself")),
                                   _ptr(st.get("row_match_noself")), _ptr(st["compat"]), _ptr(st["classdeg"]),
                                   stream_handle()), "wdg_edge_label_stats")
    st["n_classes"] = c
    return st


class StatsBatch:
    """Job table for wdg_edge_label_stats_batched; outputs live in pooled tensors zeroed by one memset."""

    def __init__(self, graphs, labels_list, n_classes):
        dev = require_gpu()
        self.n_jobs, self.c = len(graphs), int(n_classes)
        c = self.c
        # one pool, three views: a launch zeroes the counters with a single memset
        self.counters = torch.zeros(self.n_jobs * (6 + c * c + c), dtype=torch.int64, device=dev)
        self.totals = self.counters[:self.n_jobs * 6].view(self.n_jobs, 6)
        self.compat = self.counters[self.n_jobs * 6:self.n_jobs * (6 + c * c)].view(self.n_jobs, c, c)
        self.classdeg = self.counters[self.n_jobs * (6 + c * c):].view(self.n_jobs, c)
        self.max_rows = max([g.n_rows for g in graphs], default=0)
        self.rows = torch.zeros((self.n_jobs, 3, max(self.max_rows, 1)), dtype=torch.int32, device=dev)
        self.labels = [_dev(l, torch.int32, dev) for l in labels_list]
        self.keep = graphs
        # the table by column arithmetic (a structured array with the descriptor's layout): the outputs are slices of pools at
        # regular strides, so only the graphs' own pointers are read one by one
        tab = np.zeros(self.n_jobs, np.dtype(StatsJob))
        idx = np.arange(self.n_jobs, dtype=np.int64)
        tab["rowptr"] = [g.rowptr.data_ptr() for g in graphs]
        tab["col"] = [g.col.data_ptr() for g in graphs]
        tab["labels"] = [l.data_ptr() for l in self.labels]
        tab["totals"] = self.totals.data_ptr() + 8 * 6 * idx
        tab["compat"] = self.compat.data_ptr() + 8 * c * c * idx
        tab["classdeg"] = self.classdeg.data_ptr() + 8 * c * idx
        row_stride = 4 * self.rows.shape[2]
        for k, name in enumerate(("row_nnz", "row_nnz_noself", "row_match_noself")):
            tab[name] = self.rows.data_ptr() + row_stride * (3 * idx + k)
        tab["n_rows"] = [g.n_rows for g in graphs]
        tab["n_classes"] = c
        self.table = _h2d(tab.view(np.uint8), dev) if self.n_jobs else torch.empty(0, dtype=torch.uint8)

    def zero(self):
        """the counters must be zero when the kernel starts (launch() does it; callers that want the memset off a
        dependency chain call zero() earlier and launch(zero=False))"""
        self.counters.zero_()

    def launch(self, zero=True):
        if zero:
            self.counters.zero_()
        check(lib.wdg_edge_label_stats_batched(_ptr(self.table), self.n_jobs, self.max_rows, self.c, stream_handle()),
              "wdg_edge_label_stats_batched")


class LasBatch:
    """Job table for wdg_las_batched_f32: soft / hard LAS counts of many graphs in one launch (3 kernels)."""

    def __init__(self, entries, n_classes, counts=None, row_scales=None):
        """entries: list of (H [n,F] fp32 device, labels int32 device [n]).
        counts (a StatsBatch over the same graphs) + row_scales (each graph's D^-1 coefficients): the launch also derives that
        batch's integer counters from H = D^-1 (A + I) onehot(labels) - every node's neighbour-class counts ride in H already -
        instead of a second pass over the edges (include/wdg.h, wdg_las_job.counts; needs the fused one-workgroup path:
        self.derives_counts tells whether it applies)."""
        dev = require_gpu()
        self.keep = (entries, counts, row_scales)
        self.n_jobs, self.c = len(entries), int(n_classes)
        self.counts = torch.zeros((self.n_jobs, 2), dtype=torch.int64, device=dev)
        self.n = _h2d(np.array([h.shape[0] for h, _ in entries], np.float32), dev)
        sizes = [lib.wdg_las_workspace_bytes(h.shape[0], h.shape[1], self.c) for h, _ in entries]
        offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
        self.ws = torch.empty(int(offs[-1]) + 256, dtype=torch.uint8, device=dev)
        tab = np.zeros(self.n_jobs, np.dtype(_lib.LasJob))
        idx = np.arange(self.n_jobs, dtype=np.int64)
        tab["H"] = [h.data_ptr() for h, _ in entries]
        tab["labels"] = [lab.data_ptr() for _, lab in entries]
        tab["count_out"] = self.counts.data_ptr() + 16 * idx
        tab["workspace"] = self.ws.data_ptr() + offs[:-1]
        tab["ldh"] = [_ld(h) for h, _ in entries]
        tab["n"] = [h.shape[0] for h, _ in entries]
        tab["F"] = [h.shape[1] for h, _ in entries]
        tab["C"] = self.c
        self.max_n = int(tab["n"].max()) if self.n_jobs else 0
        self.max_f = int(tab["F"].max()) if self.n_jobs else 0
        self.derives_counts = bool(counts is not None and row_scales is not None and self.n_jobs and counts.n_jobs == self.n_jobs
                                   and all(h.shape[1] == self.c for h, _ in entries)
                                   and lib.wdg_las_fused_eligible(self.max_n, self.max_f, self.c))
        if self.derives_counts:
            tab["counts"] = counts.table.data_ptr() + ctypes.sizeof(StatsJob) * idx
            tab["row_scale"] = [r.data_ptr() for r in row_scales]
        self.table = _h2d(tab.view(np.uint8), dev) if self.n_jobs else torch.empty(0, dtype=torch.uint8)

    def launch(self):
        check(lib.wdg_las_batched_f32(_ptr(self.table), self.n_jobs, self.max_n, self.max_f, self.c, stream_handle()),
              "wdg_las_batched_f32")


# ------------------------------------------------------------------------------------------- per-edge cosine
def edge_cosine(g, x, entries=None, skip_self=True):
    """fp32 cosine similarity of the endpoints of every stored entry (or of the listed entry ids); wdg_edge_cosine_f32."""
    dev = g.device
    x = _dev(x, torch.float32, dev)
    entries = _dev(entries, torch.int32, dev)
    n = int(entries.shape[0]) if entries is not None else g.nnz
    out = torch.empty(n, dtype=torch.float32, device=dev)
    check(lib.wdg_edge_cosine_f32(_ptr(g.rowptr), _ptr(g.col), _ptr(entries), n, _ptr(x), _ld(x), g.n_rows,
                                  x.shape[1], int(skip_self), _ptr(out), stream_handle()), "wdg_edge_cosine_f32")
    return out


# ------------------------------------------------------------------------------------------- LAS
def las(h, labels, n_classes, rows=None, want_weights=False):
    """-> (soft_count, hard_count, n, W|None): device-side label-aggregation similarity (wdg_las_f32)."""
    dev = require_gpu()
    h = _dev(h, torch.float32, dev)
    labels = _dev(labels, torch.int32, dev)
    rows = _dev(rows, torch.int32, dev)
    n = int(rows.shape[0]) if rows is not None else int(h.shape[0])
    f, c = int(h.shape[1]), int(n_classes)
    w = torch.empty((n, c), dtype=torch.float64, device=dev) if want_weights else None
    cnt = torch.empty(2, dtype=torch.int64, device=dev)
    ws_bytes = lib.wdg_las_workspace_bytes(n, f, c)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    check(lib.wdg_las_f32(_ptr(h), _ld(h), _ptr(labels), _ptr(rows), n, f, c, _ptr(w), _ptr(cnt), _ptr(ws), ws_bytes,
                          stream_handle()), "wdg_las_f32")
    return cnt, n, w
