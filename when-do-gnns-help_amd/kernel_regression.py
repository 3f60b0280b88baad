"""The kernel-regression metric on the device (reference: utils/homophily_metrics.py:190-349): batched Grams with the fused
arc-cosine map, the generalized edge homophily gathered from a Gram, the device sampler of the node sets and the batched solver."""
import ctypes
import os

import numpy as np
import torch

from . import _lib
from ._lib import c_void_p, check, lib, require_gpu, stream_handle
from ._rt import *  # noqa: F401,F403  (the flag values of include/wdg.h)
from ._rt import _dev, _h2d, _ld, _ptr, _table


# ------------------------------------------------------------------------------------------- kernel-regression metric
def deflation_enabled():
    """WDG_KR_DEFLATE=0: no row representatives - the solver answers exactly singular train blocks with its ridge (round 5)"""
    return os.environ.get("WDG_KR_DEFLATE", "1") not in ("0", "")


class RowRepBatch:
    """Job table for wdg_row_rep_batched: per matrix the map row -> smallest bit-identical row (`rep[i]`, int32 [n]) - what the
    kernel-regression solver deflates duplicate nodes with (KrBatch(rep=...); csrc/row_rep.hip)."""

    def __init__(self, dense=None, csr=None, out=None):
        """dense: list of A [n, F] fp32 device tensors (or ops.Tiled) - or csr: list of (CsrGraph, row_scale | None, use_values)
        out: another RowRepBatch over matrices of the same row counts whose `rep` tensors this one fills"""
        dev = require_gpu()
        if (dense is None) == (csr is None):
            raise ValueError("RowRepBatch: dense or csr")
        self.source = 0 if dense is not None else 1
        items = dense if dense is not None else csr
        self.keep = items
        ns = [int(a.shape[0]) for a in dense] if dense is not None else [int(g.n_rows) for g, _rs, _uv in csr]
        self.n_jobs, self.max_n = len(ns), max(ns, default=0)
        off = np.concatenate([[0], np.cumsum(ns)]).astype(np.int64)
        if out is not None:
            if [int(r.shape[0]) for r in out.rep] != ns:
                raise ValueError("RowRepBatch(out=...): the other batch's maps have other lengths")
            self.rep = out.rep
        else:
            pool = torch.empty(int(off[-1]), dtype=torch.int32, device=dev)
            self.rep = [pool[int(off[i]):int(off[i + 1])] for i in range(len(ns))]
        self.hash_ws = torch.empty(max(int(off[-1]), 1), dtype=torch.int64, device=dev)
        arr = (_lib.RowRepJob * self.n_jobs)()
        for i, (job, it) in enumerate(zip(arr, items)):
            job.rep_out, job.hash_ws, job.n = self.rep[i].data_ptr(), self.hash_ws.data_ptr() + 8 * int(off[i]), ns[i]
            if dense is not None:
                if it.dtype != torch.float32 or it.stride(1) != 1:
                    raise ValueError("RowRepBatch: A must be fp32 with unit inner stride")
                job.A, job.lda, job.F = it.data_ptr(), _ld(it), int(it.shape[1])
                job.a_group_stride = it.group_stride if isinstance(it, Tiled) else 0
            else:
                g, rs, use_values = it
                job.rowptr, job.col = g.rowptr.data_ptr(), g.col.data_ptr()
                job.val = g.val.data_ptr() if (use_values and g.val is not None and not getattr(g, "unit_values", False)) else 0
                job.row_scale = 0 if rs is None else rs.data_ptr()
        self.table = _table(arr)

    def launch(self):
        check(lib.wdg_row_rep_batched(_ptr(self.table), self.n_jobs, self.max_n, self.source, stream_handle()), "wdg_row_rep_batched")


class GramBatch:
    """Job table for wdg_gram_map_batched_f32: K = map(A A^T) of every A of a batch (all nodes), linear and / or arc-cosine."""

    def __init__(self, mats, linear=True, arccos=True, out=None):
        """mats: list of A [n, F] fp32 device tensors (unit inner stride; or ops.Tiled - 16-column groups, the sweep's aggregated
        features - when the split-operand kernels run: `tiled_ok()`) -> self.k_linear[i], self.k_arccos[i] ([n, n] or None)
        out: another GramBatch over matrices of the same row counts whose output buffers this one writes (a sweep's next feature
        base on the same graphs: every table that holds those addresses stays valid)"""
        dev = require_gpu()
        self.keep = mats
        self.n_jobs = len(mats)
        self.max_n = max([a.shape[0] for a in mats], default=0)
        if out is not None:
            if [a.shape[0] for a in mats] != [k.shape[0] for k in out.norm2] or (linear and out.k_linear[:1] == [None]) or (arccos and out.k_arccos[:1] == [None]):
                raise ValueError("GramBatch(out=...): the other batch's outputs have other shapes")
            self.norm2, self.k_linear, self.k_arccos = out.norm2, out.k_linear if linear else [None] * len(mats), out.k_arccos if arccos else [None] * len(mats)
        else:
            # (one allocation per matrix: 16-MB blocks come back from torch's caching allocator in microseconds; one 1-GB block per
            # output kind - tried in round 5 - is returned to the driver between shards and costs 5 ms per allocation)
            self.norm2 = [torch.empty(a.shape[0], dtype=torch.float32, device=dev) for a in mats]
            self.k_linear = [torch.empty((a.shape[0], a.shape[0]), dtype=torch.float32, device=dev) if linear else None for a in mats]
            self.k_arccos = [torch.empty((a.shape[0], a.shape[0]), dtype=torch.float32, device=dev) if arccos else None for a in mats]
        arr = (_lib.GramJob * self.n_jobs)()
        for job, a, n2, kl, ka in zip(arr, mats, self.norm2, self.k_linear, self.k_arccos):
            if a.dtype != torch.float32 or a.stride(1) != 1:
                raise ValueError("GramBatch: A must be fp32 with unit inner stride")
            if isinstance(a, Tiled) and not self.tiled_ok():
                raise ValueError("GramBatch: a tiled A needs the split-operand kernels (WDG_GRAM_SPLIT=0 is set)")
            job.a_group_stride = a.group_stride if isinstance(a, Tiled) else 0
            job.A, job.norm2 = a.data_ptr(), n2.data_ptr()
            job.K_linear = 0 if kl is None else kl.data_ptr()
            job.K_arccos = 0 if ka is None else ka.data_ptr()
            job.lda, job.ldk, job.n, job.F = _ld(a), a.shape[0], a.shape[0], a.shape[1]
        self.table = _table(arr)
        # the kernel family is chosen with the table and named at every launch (1 = WDG_KERNEL_SPLIT, 2 = WDG_KERNEL_CHAIN,
        # 4 = WDG_OPERAND_TILED): see Mlp2Batch
        self.flags = (1 if self.tiled_ok() else 2) | (4 if any(isinstance(a, Tiled) for a in mats) else 0)
        # rep[i]: row -> smallest bit-identical row of A_i (launch() fills it beside the Gram): the solver's deflation maps
        self.row_rep = (RowRepBatch(dense=mats, out=getattr(out, "row_rep", None)) if (deflation_enabled() and mats) else None)
        self.rep = self.row_rep.rep if self.row_rep is not None else [None] * len(mats)

    @staticmethod
    def tiled_ok():
        """the launcher's own rule (csrc/kernel_reg.hip, wdg_gram_map_batched_f32): split-operand kernels unless WDG_GRAM_SPLIT=0"""
        e = os.environ.get("WDG_GRAM_SPLIT")
        try:
            return True if e is None else int(e) != 0
        except ValueError:
            return False

    def launch(self):
        check(lib.wdg_gram_map_batched_flags_f32(_ptr(self.table), self.n_jobs, self.max_n, self.flags, stream_handle()),
              "wdg_gram_map_batched_flags_f32")
        if self.row_rep is not None:
            self.row_rep.launch()


_PROP_SCRATCH = {}  # (stream, node counts) -> (T, T^T) scratch of PropagatedGram: launches on one stream run in order


class PropagatedGram:
    """The kernels of the AGGREGATED features of many graphs WITHOUT a dense product per graph: Y = A_hat X gives Y Y^T = A_hat (X X^T)
    A_hat^T, so K_linear(Y) = A_hat K_linear(X) A_hat^T - two batched aggregations with n "features" over the half Gram of the raw
    features (2 nnz n flops each instead of n^2 F: 4 .. 18 x less work for the reference's feature bases, F = 932 .. 3 703), a
    second one reading the first one's output transposed (ops.Transposed) - and the finish pass (mirror + arc-cosine map,
    wdg_gram_finish_batched_f32).
    Same interface as GramBatch (`k_linear[i]`, `k_arccos[i]`, `norm2[i]`, `launch()`); entries agree with GramBatch over the
    aggregated features to fp32 rounding (tests/test_gpu_kernels.py)."""

    def __init__(self, problems, linear=True, arccos=True):
        """problems: list of (CsrGraph g with a SELL-16 copy, row_scale|None, col_scale|None, K_linear_X [n, n] fp32 device tensor:
        the raw features' half Gram, as GramBatch leaves it; problems that share a feature matrix pass the SAME tensor)"""
        from .aggregate import SpmmBatch
        dev = require_gpu()
        self.keep = problems
        self.n_jobs = len(problems)
        ns = [p_[0].n_rows for p_ in problems]
        self.max_n = max(ns, default=0)
        if any(p_[3].shape != (n, n) or p_[0].n_cols != n for p_, n in zip(problems, ns)):
            raise ValueError("PropagatedGram: square graphs and [n, n] kernels expected")
        self.norm2 = [torch.empty(n, dtype=torch.float32, device=dev) for n in ns]
        # U = A_hat T^T lands in the K_linear buffers (the finish pass works in place); T and T^T are scratch
        self.k_linear = [torch.empty((n, n), dtype=torch.float32, device=dev) for n in ns]
        self.k_arccos = [torch.empty((n, n), dtype=torch.float32, device=dev) if arccos else None for n in ns]
        # T and T^T are scratch between the launches of ONE launch(): kept per (stream, shapes) across batches - a sweep builds a
        # batch like this per feature base and shard, 2 x 70 allocations of 16 MB each time otherwise
        # the second product reads T TRANSPOSED where the first wrote it (WDG_SELL16_X_TRANSPOSED: the slab staging transposes 16 rows
        # of T at a time: 3.72 -> 3.15 ms per 70 graphs of 2000 nodes); WDG_PROP_TRANSPOSE=1: round 5's first form, a transpose pass
        # between the products into a second scratch matrix per graph
        self.transpose_pass = os.environ.get("WDG_PROP_TRANSPOSE", "0") == "1"
        # (the job tables below hold T's addresses, so the scratch belongs to the stream current HERE: launch() checks that it runs on
        # that stream - two batches of equal shapes built on one stream and launched on two would race on T without an error)
        self.owner_stream = torch.cuda.current_stream().cuda_stream
        key = (self.owner_stream, tuple(ns), self.transpose_pass)
        if key in _PROP_SCRATCH:
            _PROP_SCRATCH[key] = _PROP_SCRATCH.pop(key)  # (most recently used last: the dict is the LRU order)
        else:
            if len(_PROP_SCRATCH) >= 4:
                _PROP_SCRATCH.pop(next(iter(_PROP_SCRATCH)))  # the least recently used key
            _PROP_SCRATCH[key] = ([torch.empty((n, n), dtype=torch.float32, device=dev) for n in ns],
                                  [torch.empty((n, n), dtype=torch.float32, device=dev) if self.transpose_pass else None for n in ns])
        self._t, self._tt = _PROP_SCRATCH[key]
        self.first = SpmmBatch([(g, kx, t, rs, cs, False) for (g, rs, cs, kx), t in zip(problems, self._t)])
        self.second = SpmmBatch([(g, tt if self.transpose_pass else Transposed(t), u, rs, cs, False)
                                 for (g, rs, cs, _kx), t, tt, u in zip(problems, self._t, self._tt, self.k_linear)])
        self.tr_table = None
        if self.transpose_pass:
            tr = (_lib.TransposeJob * self.n_jobs)()
            for job, t, tt, n in zip(tr, self._t, self._tt, ns):
                job.src, job.dst, job.ld_src, job.ld_dst, job.rows, job.cols = t.data_ptr(), tt.data_ptr(), n, n, n, n
            self.tr_table = _table(tr)
        fin = (_lib.GramJob * self.n_jobs)()
        for job, u, n2, ka, n in zip(fin, self.k_linear, self.norm2, self.k_arccos, ns):
            job.A, job.norm2, job.K_linear = u.data_ptr(), n2.data_ptr(), u.data_ptr()
            job.K_arccos = 0 if ka is None else ka.data_ptr()
            job.lda, job.ldk, job.n, job.F, job.a_group_stride = n, n, n, n, 0
        self.fin_table = _table(fin)
        self.linear = linear  # (K_linear is produced either way: it is the propagated quantity)
        # duplicate rows of Y = A_hat X without Y: identical rows of A_hat (columns, order, values, row scale) - bit-identical rows of
        # the reference's A_hat X whatever X holds.  (Rows that coincide only through duplicate FEATURE rows are not found here: such a
        # train block falls to the solver's ridge and is counted as ridged.)
        self.row_rep = (RowRepBatch(csr=[(g, rs, False) for g, rs, _cs, _kx in problems])  # (pattern + scales: what the products read)
                        if (deflation_enabled() and problems) else None)
        self.rep = self.row_rep.rep if self.row_rep is not None else [None] * len(problems)

    def launch(self):
        if torch.cuda.current_stream().cuda_stream != self.owner_stream:
            raise _lib.WdgError("PropagatedGram.launch: launched on another stream than the one it was built on (its T scratch is shared, "
                           "in stream order, with the other batches of that stream)")
        self.first.launch()       # T_j = A_hat_j K_linear(X)
        if self.transpose_pass:
            check(lib.wdg_transpose_batched_f32(_ptr(self.tr_table), self.n_jobs, self.max_n, self.max_n, stream_handle()), "wdg_transpose_batched_f32")
        self.second.launch()      # U_j = A_hat_j T_j^T
        check(lib.wdg_gram_finish_batched_f32(_ptr(self.fin_table), self.n_jobs, self.max_n, stream_handle()), "wdg_gram_finish_batched_f32")
        if self.row_rep is not None:
            self.row_rep.launch()


class EdgeGramBatch:
    """Job table for wdg_edge_gram_mean_batched_f32: mean edge cosine (generalized edge homophily) of many graphs from the Grams
    of their feature matrices."""

    def __init__(self, problems):
        """problems: list of (CsrGraph, K_linear [n, n], norm2 [n]) -> self.mean [n_problems] fp64 after launch()"""
        dev = require_gpu()
        self.keep = problems
        self.n_jobs = len(problems)
        self.max_rows = max([p[0].n_rows for p in problems], default=0)
        self.mean = torch.zeros(max(self.n_jobs, 1), dtype=torch.float64, device=dev)
        self.ws_bytes = lib.wdg_edge_gram_workspace_bytes(self.n_jobs, self.max_rows)
        self.ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=dev)
        arr = (_lib.EdgeGramJob * self.n_jobs)()
        for i, (job, (g, k, n2)) in enumerate(zip(arr, problems)):
            job.rowptr, job.col, job.K_linear, job.norm2 = g.rowptr.data_ptr(), g.col.data_ptr(), k.data_ptr(), n2.data_ptr()
            job.mean_out = self.mean.data_ptr() + 8 * i
            job.ldk, job.n_rows = _ld(k), g.n_rows
        self.table = _table(arr)

    def launch(self):
        check(lib.wdg_edge_gram_mean_batched_f32(_ptr(self.table), self.n_jobs, self.max_rows, _ptr(self.ws), self.ws_bytes,
                                                 stream_handle()), "wdg_edge_gram_mean_batched_f32")


def kr_split_sizes(labels, sample_max):
    """Per class: how many members an epoch's sample holds (s_c) and how many of those train (t_c) - the sizes the
    reference's two random_disassortative_splits calls produce (utils/homophily_metrics.py:269-279 with utils/util_funcs.py:
    454-475; Python's banker's `round`, classes counted as max label + 1, a class smaller than its share gives all it has).
    labels: host int array -> (s_c, t_c) int32 [C]"""
    labels = np.asarray(labels).reshape(-1)
    n = labels.shape[0]
    c = int(labels.max()) + 1 if n else 0
    n_c = np.bincount(labels[labels >= 0], minlength=c).astype(np.int64)
    if n <= sample_max:
        s_c = n_c.copy()
    else:
        s_c = np.minimum(n_c, int(round((sample_max / n) * (n / c))))
    present = np.flatnonzero(s_c)
    c2 = int(present.max()) + 1 if present.size else 1       # labels_sample.max() + 1
    t_c = np.minimum(s_c, int(round(0.6 * (int(s_c.sum()) / c2))))
    return s_c.astype(np.int32), t_c.astype(np.int32)


_KR_SAMPLE_DTYPE = np.dtype([("labels", "<u8"), ("sample_per_class", "<u8"), ("train_per_class", "<u8"), ("train_out", "<u8"), ("val_out", "<u8"),
                             ("seed", "<u8"), ("n", "<i4"), ("n_classes", "<i4"), ("n_sets", "<i4"), ("first_set", "<i4"), ("train_stride", "<i4"),
                             ("val_stride", "<i4")])
assert _KR_SAMPLE_DTYPE.itemsize == ctypes.sizeof(_lib.KrSampleJob)


class KrSets:
    """Job table for wdg_kr_sample_sets: the (train, validation) node sets of every epoch of many (graph, classifier) pairs,
    drawn on the device in one launch (Philox4x32-10 keyed per pair; include/wdg.h documents the generator).
    self.train [pairs, epochs, n_train], self.val [pairs, epochs, n_val] int32, ascending ids; pairs whose graphs differ in
    class sizes are padded to the widest (self.n_train / self.n_val hold the true lengths)."""

    def __init__(self, entries, epochs, out=None):
        """entries: list of (labels int32 device [n], s_c, t_c (kr_split_sizes), seed int)
        out: another KrSets of the same shape whose train / val tensors this one fills (other seeds for the same label vectors)"""
        dev = require_gpu()
        self.keep = entries
        self.n_pairs, self.epochs = len(entries), int(epochs)
        # (the table by column arithmetic: entries that share a label vector - the jobs of a sweep - share its per-class arrays)
        n = self.n_pairs
        sums = {}
        for _l, s_, t, _seed in entries:
            if id(t) not in sums:
                sums[id(t)] = (int(np.sum(t)), int(np.sum(s_)) - int(np.sum(t)), len(s_))
        self.n_train = np.fromiter((sums[id(e[2])][0] for e in entries), np.int64, n)
        self.n_val = np.fromiter((sums[id(e[2])][1] for e in entries), np.int64, n)
        n_cls = np.fromiter((sums[id(e[2])][2] for e in entries), np.int64, n)
        if n and int(n_cls.max()) > 64:
            raise ValueError("KrSets: more than 64 classes")
        self.train_stride, self.val_stride = int(self.n_train.max(initial=0)), int(self.n_val.max(initial=0))
        ts, vs = max(self.train_stride, 1), max(self.val_stride, 1)
        # (zeros only where sets are padded to a common stride: the kernel writes every entry of a set)
        alloc = torch.zeros if (n and (int(self.n_train.min()) != ts or int(self.n_val.min()) != vs)) else torch.empty
        if out is not None:
            if tuple(out.train.shape) != (n, self.epochs, ts) or tuple(out.val.shape) != (n, self.epochs, vs):
                raise ValueError("KrSets(out=...): the other table's sets have another shape")
            self.train, self.val = out.train, out.val
        else:
            self.train = alloc((n, self.epochs, ts), dtype=torch.int32, device=dev)
            self.val = alloc((n, self.epochs, vs), dtype=torch.int32, device=dev)
        self.max_n = max([int(e[0].shape[0]) for e in entries], default=0)
        tables, off_of, parts, off = {}, np.zeros(n, np.int64), [], 0
        for i, (_l, s_, t, _seed) in enumerate(entries):  # one [s_c | t_c] block per distinct pair of arrays
            key = (id(s_), id(t))
            if key not in tables:
                tables[key] = off
                parts += [np.asarray(s_, np.int32), np.asarray(t, np.int32)]
                off += 2 * len(s_)
            off_of[i] = tables[key]
        self.class_tables = _h2d(np.concatenate(parts) if parts else np.zeros(0, np.int32), dev)
        tab = np.zeros(n, _KR_SAMPLE_DTYPE)
        idx = np.arange(n, dtype=np.int64)
        tab["labels"] = np.fromiter((e[0].data_ptr() for e in entries), np.int64, n)
        tab["sample_per_class"] = self.class_tables.data_ptr() + 4 * off_of
        tab["train_per_class"] = self.class_tables.data_ptr() + 4 * (off_of + n_cls)
        tab["train_out"] = self.train.data_ptr() + 4 * idx * self.epochs * ts
        tab["val_out"] = self.val.data_ptr() + 4 * idx * self.epochs * vs
        tab["seed"] = np.fromiter((int(e[3]) & 0xFFFFFFFFFFFFFFFF for e in entries), np.uint64, n)
        tab["n"] = np.fromiter((int(e[0].shape[0]) for e in entries), np.int64, n)
        tab["n_classes"], tab["n_sets"], tab["first_set"] = n_cls, self.epochs, idx * self.epochs
        tab["train_stride"], tab["val_stride"] = ts, vs
        self.table = _h2d(tab.view(np.uint8), dev) if n else torch.empty(0, dtype=torch.uint8)

    def launch(self):
        check(lib.wdg_kr_sample_sets(_ptr(self.table), self.n_pairs, self.n_pairs * self.epochs, self.max_n, stream_handle()),
              "wdg_kr_sample_sets")


_KR_JOB_DTYPE = np.dtype([("K", "<u8"), ("train", "<u8"), ("val", "<u8"), ("labels", "<u8"), ("correct_out", "<u8"), ("flags_out", "<u8"),
                          ("ldk", "<i8"), ("n_train", "<i4"), ("n_val", "<i4"), ("n_classes", "<i4"), ("reserved", "<i4"), ("rep", "<u8"), ("ws", "<u8")])
assert _KR_JOB_DTYPE.itemsize == ctypes.sizeof(_lib.KrJob)


class KrBatch:
    """Job table for wdg_kernel_regress_batched_f32: many (kernel, train rows, validation rows) problems in one launch."""

    MAX_TRAIN = 320
    MAX_CLASSES = 8  # KR_MAX_C of csrc/kernel_reg.hip: the right-hand sides a problem's workgroup carries

    def __init__(self, problems, n_classes):
        """problems: list of (K [n, n] fp32 device, train int32 device [nt], val int32 device [nv], labels int32 device [n]
        [, rep int32 device [n] | None: the row representatives of the matrix K was computed from - GramBatch.rep[i]; the solver
        then deflates duplicate nodes instead of regularising the block])
        -> self.correct [n_problems] int32 after launch().  Shapes the solver does not hold (more than 8 classes, more than
        320 or fewer than 1 train rows) raise here: the kernel would answer them with the sentinel -1, and an accuracy of
        -1 / n_val fed to the t-test is a silently wrong p-value (callers with such label sets take the host path)."""
        self.keep = problems
        n = len(problems)
        col = lambda f: np.fromiter((f(p_) for p_ in problems), np.int64, n)  # noqa: E731
        self._build(col(lambda p_: p_[0].data_ptr()), col(lambda p_: _ld(p_[0])), col(lambda p_: p_[1].data_ptr()),
                    col(lambda p_: p_[2].data_ptr()), col(lambda p_: p_[3].data_ptr()), col(lambda p_: p_[1].shape[0]),
                    col(lambda p_: p_[2].shape[0]), n_classes,
                    col(lambda p_: p_[4].data_ptr() if len(p_) > 4 and p_[4] is not None else 0))

    @classmethod
    def from_arrays(cls, k_ptr, ldk, train_ptr, val_ptr, labels_ptr, n_train, n_val, n_classes, keep=None, rep_ptr=None):
        """the same table from per-problem numpy columns (device addresses and sizes): a sweep shard's 20 000 problems are
        described by arithmetic on a few base pointers, not by 20 000 tensor objects"""
        self = cls.__new__(cls)
        self.keep = keep
        self._build(*(np.asarray(a, np.int64) for a in (k_ptr, ldk, train_ptr, val_ptr, labels_ptr, n_train, n_val)), n_classes,
                    None if rep_ptr is None else np.asarray(rep_ptr, np.int64))
        return self

    def _build(self, k_ptr, ldk, train_ptr, val_ptr, labels_ptr, n_train, n_val, n_classes, rep_ptr=None):
        dev = require_gpu()
        n = self.n_jobs = int(k_ptr.shape[0])
        if n and not 1 <= int(n_classes) <= self.MAX_CLASSES:
            raise ValueError(f"KrBatch: {n_classes} classes, the solver holds 1..{self.MAX_CLASSES}")
        if n and not (1 <= int(n_train.min()) and int(n_train.max()) <= self.MAX_TRAIN):
            raise ValueError(f"KrBatch: {int(n_train.min())}..{int(n_train.max())} train rows, the solver holds blocks of 1..{self.MAX_TRAIN}")
        if n and int(ldk.max()) >= 65536:  # (the solver's prediction gathers address a kernel matrix by 32-bit element offsets)
            raise ValueError(f"KrBatch: kernel matrices of leading dimension {int(ldk.max())}, the solver addresses up to 65 535")
        self.correct = torch.zeros(max(n, 1), dtype=torch.int32, device=dev)
        self.flags = torch.zeros(max(n, 1), dtype=torch.int32, device=dev)  # bit 0: the ridge refactorisation ran
        self.n_val = _h2d(n_val.astype(np.float32), dev)
        tab = np.zeros(n, _KR_JOB_DTYPE)
        tab["K"], tab["train"], tab["val"], tab["labels"] = k_ptr, train_ptr, val_ptr, labels_ptr
        tab["correct_out"] = self.correct.data_ptr() + 4 * np.arange(n, dtype=np.int64)
        tab["flags_out"] = self.flags.data_ptr() + 4 * np.arange(n, dtype=np.int64)
        tab["ldk"], tab["n_train"], tab["n_val"], tab["n_classes"] = ldk, n_train, n_val, int(n_classes)
        # row representatives -> the deflating entry point: a workspace for EVERY problem of the table (problems without maps: identity)
        self.ws = None
        tab["rep"] = 0 if rep_ptr is None else rep_ptr
        if n and bool((tab["rep"] != 0).any()):
            per = int(lib.wdg_kr_deflate_workspace_bytes(int(n_val.max())))
            self.ws = torch.empty(n * per, dtype=torch.uint8, device=dev)
            tab["ws"] = self.ws.data_ptr() + per * np.arange(n, dtype=np.int64)
        # (timing-only diagnostics of the blocked solver: honoured only by a library built with -DWDG_KR_ABLATION, and never
        # mistaken for a result - accuracy() refuses)
        self.ablate = int(os.environ.get("WDG_KR_ABLATE", "0"))
        tab["reserved"] = self.ablate
        self.table = _h2d(tab.view(np.uint8), dev) if n else torch.empty(0, dtype=torch.uint8)

    def launch(self):
        if self.ws is not None:
            check(lib.wdg_kernel_regress_deflated_batched_f32(_ptr(self.table), self.n_jobs, stream_handle()), "wdg_kernel_regress_deflated_batched_f32")
        else:
            check(lib.wdg_kernel_regress_batched_f32(_ptr(self.table), self.n_jobs, stream_handle()), "wdg_kernel_regress_batched_f32")

    def ridged(self):
        """[n_problems] bool: the train block was rank deficient in fp32 and was solved with the rounding-level ridge"""
        return (self.flags[:self.n_jobs] & 1).bool()

    def deflated(self):
        """[n_problems] bool: duplicate train rows were merged / zero rows dropped before the factorisation (row representatives)"""
        return (self.flags[:self.n_jobs] & 2).bool()

    def accuracy(self):
        """[n_problems] fp32 hit rate on the validation rows; raises when the kernel refused a problem (sentinel -1)"""
        if getattr(self, "ablate", 0):
            raise _lib.WdgError("KrBatch: WDG_KR_ABLATE is set - the launch was a timing-only ablation, its accuracies mean nothing")
        correct = self.correct[:self.n_jobs]
        if self.n_jobs and bool((correct < 0).any().item()):
            raise _lib.WdgError("wdg_kernel_regress_batched_f32 refused a problem (shape outside the solver's limits)")
        return correct.to(torch.float32) / self.n_val


# ------------------------------------------------------------------------------------------- Gaussian naive Bayes (the GNB classifier)
_GNB_JOB_DTYPE = np.dtype([("X", "<u8"), ("train", "<u8"), ("val", "<u8"), ("labels", "<u8"), ("ws", "<u8"), ("correct", "<u8"),
                           ("pred", "<u8"), ("ldx", "<i8"), ("n_train", "<i4"), ("n_val", "<i4"), ("F", "<i4"), ("n_classes", "<i4")])
assert _GNB_JOB_DTYPE.itemsize == ctypes.sizeof(_lib.GnbJob)


class GnbBatch:
    """Job table for wdg_gnb_batched_f32: many (feature matrix, train rows, validation rows) Gaussian-naive-Bayes problems in one call -
    the GNB branch of classifier_based_performance_metric (utils/homophily_metrics.py:296-312), scikit-learn's arithmetic (csrc/gnb.hip)."""

    MAX_CLASSES = 16

    def __init__(self, problems, n_classes, want_pred=False):
        """problems: list of (X [n, F] fp32 device, row-major, train int32 device [nt], val int32 device [nv], labels int32 device [n]);
        the statistics are summed over the train rows IN THE ORDER GIVEN (the reference's boolean masks: ascending ids).
        -> self.correct [n_problems] int32 after launch(); want_pred: self.pred[i] int32 [nv] too.
        Raises for what scikit-learn raises for or the kernel does not hold: no train rows, no features, more than 16 classes."""
        dev = require_gpu()
        self.keep = problems
        n = self.n_jobs = len(problems)
        self.n_classes = int(n_classes)
        if n and not 1 <= self.n_classes <= self.MAX_CLASSES:
            raise ValueError(f"GnbBatch: {n_classes} classes, the kernel holds 1..{self.MAX_CLASSES}")
        for x, tr, va, lab in problems:
            if x.dim() != 2 or x.dtype != torch.float32 or x.stride(1) != 1 or not x.is_cuda:
                raise ValueError("GnbBatch: X must be a row-major fp32 device matrix")
            if tr.dtype != torch.int32 or va.dtype != torch.int32 or lab.dtype != torch.int32:
                raise ValueError("GnbBatch: int32 node ids and labels expected")
            if tr.shape[0] < 1 or x.shape[1] < 1:
                raise ValueError("GnbBatch: a problem needs at least one train row and one feature")
            if lab.shape[0] != x.shape[0]:
                raise ValueError("GnbBatch: one label per row of X")
        col = lambda f: np.fromiter((f(p_) for p_ in problems), np.int64, n)  # noqa: E731
        feats, n_val = col(lambda p_: p_[0].shape[1]), col(lambda p_: p_[2].shape[0])
        self.max_feat, self.max_val = int(feats.max(initial=0)), int(n_val.max(initial=0))
        self.correct = torch.zeros(max(n, 1), dtype=torch.int32, device=dev)
        self.n_val = n_val
        ws_each = np.array([int(lib.wdg_gnb_workspace_bytes(int(f), self.n_classes)) for f in feats], np.int64)
        ws_off = np.concatenate([[0], np.cumsum(ws_each)]).astype(np.int64)
        self.ws = torch.empty(max(int(ws_off[-1]), 256), dtype=torch.uint8, device=dev)
        self.pred = None
        tab = np.zeros(n, _GNB_JOB_DTYPE)
        tab["X"], tab["ldx"] = col(lambda p_: p_[0].data_ptr()), col(lambda p_: _ld(p_[0]))
        tab["train"], tab["val"], tab["labels"] = col(lambda p_: p_[1].data_ptr()), col(lambda p_: p_[2].data_ptr()), col(lambda p_: p_[3].data_ptr())
        tab["ws"] = self.ws.data_ptr() + ws_off[:-1]
        tab["correct"] = self.correct.data_ptr() + 4 * np.arange(n, dtype=np.int64)
        if want_pred:
            v_off = np.concatenate([[0], np.cumsum(n_val)]).astype(np.int64)
            pool = torch.full((max(int(v_off[-1]), 1),), -1, dtype=torch.int32, device=dev)
            self.pred = [pool[int(v_off[i]):int(v_off[i + 1])] for i in range(n)]
            tab["pred"] = pool.data_ptr() + 4 * v_off[:-1]
        tab["n_train"], tab["n_val"], tab["F"], tab["n_classes"] = col(lambda p_: p_[1].shape[0]), n_val, feats, self.n_classes
        self.table = _h2d(tab.view(np.uint8), dev) if n else torch.empty(0, dtype=torch.uint8)

    def launch(self):
        check(lib.wdg_gnb_batched_f32(_ptr(self.table), self.n_jobs, self.max_feat, self.max_val, self.n_classes, stream_handle()),
              "wdg_gnb_batched_f32")

    def accuracy(self):
        """[n_problems] float32 (host): hits / validation rows, as `torch.mean(pred.eq(labels[idx_val]).float())` gives them
        (utils/homophily_metrics.py:311-312); NaN for a problem without validation rows"""
        hits = self.correct[:self.n_jobs].cpu().numpy().astype(np.float32)
        with np.errstate(invalid="ignore", divide="ignore"):
            return hits / self.n_val.astype(np.float32)
