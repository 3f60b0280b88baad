"""Torch-tensor front end of the C ABI (include/wdg.h): device CSR container + one Python function per kernel.

PyTorch is plumbing here (device memory, the current HIP stream); every computation below is a hand-written
gfx950 kernel in csrc/.  Nothing in this file runs on the CPU, and nothing falls back.
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib
from ._lib import SpmmItem, SpmmJob, StatsJob, c_void_p, check, lib, require_gpu, stream_handle

# flags / modes of include/wdg.h
COO_SYMMETRISE, COO_BINARISE, COO_ADD_SELF_LOOPS, COO_DROP_SELF_LOOPS, COO_KEEP_DUPLICATES = 1, 2, 4, 8, 16
NORM_RW, NORM_SYM = 0, 1
PREC_F32, PREC_F64 = 0, 1
ACT_NONE, ACT_RELU = 0, 1
ABLATE_BITS = 0  # diagnostics: scripts/ablate_*.py set this to wdg_spmm_job.reserved timing-ablation bits (results are wrong then)


def _ptr(t):
    return c_void_p(0 if t is None else t.data_ptr())


def _dev(t, dtype, dev):
    """tensor / ndarray / list -> contiguous device tensor of `dtype` (no copy when already there)."""
    if t is None:
        return None
    if not isinstance(t, torch.Tensor):
        t = torch.as_tensor(np.asarray(t))
    return t.to(device=dev, dtype=dtype).contiguous()


def _ld(t):
    """Leading dimension of a row-major matrix for the C ABI.  torch / numpy report an arbitrary stride for a dimension
    of size 1 (a [1, K] view of a [K, 1] array has stride(0) == 1): with one row any value >= the row length is valid."""
    return t.stride(0) if t.shape[0] > 1 else max(int(t.stride(0)), int(t.shape[1]))


class _PinnedArena:
    """One page-locked buffer per process, handed out as a ring: a slice is reused only after the copy that last read it has
    completed (an event per copy; by the time the ring comes round the copy is long done).  torch's own pinned allocator
    cannot reuse a block while its copy is queued behind kernels, and every NEW pinned block is a hipHostMalloc - measured:
    an occasional 90 ms in the middle of a shard's table uploads."""

    def __init__(self, nbytes=128 << 20):
        self.buf = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
        self.size, self.off, self.pending = nbytes, 0, []  # pending: (start, end, event), in issue order

    def take(self, nbytes):
        n = (nbytes + 255) & ~255
        if self.off + n > self.size:
            self.off = 0
        a, b = self.off, self.off + n
        while self.pending and self.pending[0][0] < b and a < self.pending[0][1]:
            self.pending.pop(0)[2].synchronize()
        self.off = b
        return a, self.buf[a:a + nbytes]

    def issued(self, start, nbytes):
        ev = torch.cuda.Event()
        ev.record()
        self.pending.append((start, start + ((nbytes + 255) & ~255), ev))


_ARENA = None
_H2D_MAX_BYTES = int(float(os.environ.get("WDG_H2D_MAX_MB", "8")) * (1 << 20))  # larger arrays: the plain (blocking, pageable) copy


def _h2d(host, dev=None):
    """A host array (job table, offsets, labels, a feature matrix) -> device tensor WITHOUT blocking the host on what the stream
    has queued: through the page-locked arena and a non-blocking copy.  (`tensor.to(dev)` from pageable memory returns only when
    the copy has run, i.e. after every kernel queued before it - a shard's ~70 small uploads then serialise the host with the
    build kernels.)  Arrays of more than WDG_H2D_MAX_MB (8) MB take the plain blocking copy: the wide bases' 30-MB feature matrices
    through a single-threaded memcpy and a 128-MB ring cost the whole sweep 10 % (0.80 -> 0.88 s)."""
    global _ARENA
    dev = dev or require_gpu()
    t = torch.from_numpy(host) if isinstance(host, np.ndarray) else host
    t = t.contiguous()
    nbytes = t.numel() * t.element_size()
    if nbytes == 0:
        return torch.empty(t.shape, dtype=t.dtype, device=dev)
    if _ARENA is None:
        _ARENA = _PinnedArena()
    if nbytes > _H2D_MAX_BYTES:
        return t.to(dev)
    start, piece = _ARENA.take(nbytes)
    p = piece.view(t.dtype).view(t.shape)
    # (numpy's memcpy, not Tensor.copy_: torch parallelises a host copy of a few MB over every hardware thread of the host -
    # measured 90 - 180 ms of thread wake-up on a 128-thread box for a 4-MB feature matrix)
    np.copyto(p.numpy(), t.numpy())
    out = p.to(dev, non_blocking=True)
    _ARENA.issued(start, nbytes)
    return out


def _table(arr):
    """ctypes array of job descriptors -> device bytes (an empty table stays a host tensor: its pointer is NULL)"""
    return _h2d(torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)) if len(arr) else torch.empty(0, dtype=torch.uint8)


class CsrGraph:
    """Device-resident CSR adjacency: int32 rowptr[n_rows+1], int32 col[nnz], optional fp32 val[nnz].

    The layout every kernel consumes: row-major sorted, duplicates already merged - what the
    reference gets from `.coalesce()` on a torch COO tensor, with 4-byte instead of 8-byte indices.
    """

    def __init__(self, rowptr, col, val, n_rows, n_cols):
        self.rowptr, self.col, self.val = rowptr, col, val
        self.n_rows, self.n_cols = int(n_rows), int(n_cols)
        self.quad = None  # SELL-16 copy (dict) for the quad-row kernel, built on demand; False = decided against
        self.band = None  # band plan (dict) for the band kernel, built on demand; False = not applicable
        self.narrow_ws = None  # packed-source workspace of the narrow kernel (one per graph: calls on one stream)
        self.narrow_parts = None  # {parts: split positions of every row} when the packed sources exceed an XCD's L2
        self._unit_values = None if val is not None else True  # every stored value == 1 (checked once, on first use)

    @property
    def unit_values(self):
        """True when every stored value is exactly 1 (a binary adjacency, what the reference's loaders build): products then
        skip the value stream - a + 1 * x and a + x are the same bits, and the entries are 4 bytes instead of 8.  One
        reduction + host read-back per graph, on first use (like the other one-time plans: not inside a stream capture)."""
        if self._unit_values is None:
            self._unit_values = bool((self.val == 1).all().item()) if self.val.numel() else True
        return self._unit_values

    QUAD_SLAB_COLS = 2528  # columns of X the quad-row kernel holds in LDS at once (one column block)

    def ensure_band(self):
        """Build the band plan (wdg_csr_band_plan): rows by length, hub count, cost cuts.  One-time per graph, one host sync."""
        if self.band is not None:
            return self.band is not False
        if self.n_rows == 0 or self.nnz == 0 or self.n_cols == 0:
            self.band = False
            return False
        dev = self.device
        perm = torch.empty(int(lib.wdg_csr_band_perm_len(self.n_rows)), dtype=torch.int32, device=dev)
        cuts = torch.empty(24, dtype=torch.int32, device=dev)
        ws_bytes = lib.wdg_csr_band_plan_workspace_bytes(self.n_rows)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        n_hub = ctypes.c_int32(0)
        check(lib.wdg_csr_band_plan(_ptr(self.rowptr), self.n_rows, _ptr(perm), _ptr(cuts), ctypes.byref(n_hub), _ptr(ws), ws_bytes,
                                    stream_handle()), "wdg_csr_band_plan")
        self.band = dict(perm=perm, cuts=cuts, n_hub=int(n_hub.value))
        return True

    def prefers_band(self, n_feat):
        """the band kernel (L2 gathers, a wave per row) rather than the quad-row kernel (LDS slabs) for a single aggregation:
        wide features and either more columns than one 16-feature LDS slab holds or rows too long for 16-row slices (> 128 entries).
        WDG_SPMM_BAND=1 / 0 forces / forbids it."""
        force = os.environ.get("WDG_SPMM_BAND", "")
        if force == "0" or n_feat < 16:
            return False
        if not self.ensure_band():
            return False
        if force not in ("", "0"):
            return True
        # (2 529 .. 5 056 columns fit one block of 32-byte rows since round 4, but ONE graph there is still the band kernel's: Cora
        # 21 us against 105, a 4000-node sweep graph 17 against 42 - a slab per feature group is worth staging for a batch)
        return n_feat >= 64 and (self.n_cols > self.QUAD_SLAB_COLS or self.band["n_hub"] > 0)

    QUAD_MAX_BLOCKS = 4  # column blocks of <= 2528 columns the quad-row kernel sweeps (csrc/spmm_quad.hip)

    def ensure_quad(self, max_padding=4.0):
        """Build the SELL-16 copy (wdg_csr_to_sell16_*) that the quad-row SpMM consumes.  One-time per graph; False for
        graphs of more than 4 column blocks (10 112 columns) or whose slices would pad too much (the decision is kept)."""
        if self.quad is not None:
            return self.quad is not False
        if quad_disabled() or self.n_rows == 0 or self.nnz == 0:
            self.quad = False
            return False
        block_cols = lib.wdg_sell16_block_cols(self.n_cols)
        n_blocks = (max(self.n_cols, 1) + block_cols - 1) // block_cols
        if n_blocks > self.QUAD_MAX_BLOCKS:
            self.quad = False
            return False
        dev = self.device
        max_entries = int(lib.wdg_sell16_max_entries(self.n_rows))
        ext = torch.empty(2 * (n_blocks * max_entries + 1), dtype=torch.int32, device=dev)
        rows = torch.empty(16 * max_entries, dtype=torch.int32, device=dev)
        perm = torch.empty((self.n_rows + 15) // 16 * 16, dtype=torch.int32, device=dev)  # padding slots repeat the last row
        ws_bytes = lib.wdg_sell16_workspace_bytes(self.n_rows, self.n_cols)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        check(lib.wdg_csr_to_sell16_count(_ptr(self.rowptr), _ptr(self.col), self.n_rows, self.n_cols, _ptr(perm), _ptr(ext),
                                          _ptr(rows), _ptr(ws), ws_bytes, stream_handle()), "wdg_csr_to_sell16_count")
        ext_host = ext.cpu().numpy().reshape(-1, 2)  # the one host sync of the build: sizes the index arrays
        chunks, word = int(ext_host[-1, 0]), int(ext_host[-1, 1])
        n_entries, split = word & 0x3fffffff, bool(word & (1 << 30))
        tasks = n_entries * n_blocks
        if chunks * 256 > max_padding * self.nnz + 256 * tasks:
            self.quad = False  # very skewed rows: the CSR kernels are the better fit (remembered)
            return False
        # (+ 2 chunks of slack: the kernel requests an entry's two chunks unconditionally)
        q_col = torch.zeros((chunks + 2) * 256, dtype=torch.int32, device=dev)
        q_val = torch.zeros((chunks + 2) * 256, dtype=torch.float32, device=dev) if self.val is not None else None
        check(lib.wdg_csr_to_sell16_fill(_ptr(self.rowptr), _ptr(self.col), _ptr(self.val), self.n_rows, self.n_cols,
                                         _ptr(rows), _ptr(ext), n_entries, _ptr(q_col), _ptr(q_val), stream_handle()),
              "wdg_csr_to_sell16_fill")
        # per (block, entry) width: the cost model of SpmmBatch reads it (flags masked off)
        widths = (ext_host[:tasks, 1] & 0x3fffffff).reshape(n_blocks, n_entries).copy()
        self.quad = dict(ext=ext[:2 * (tasks + 1)], col=q_col, val=q_val, perm=perm, rows=rows[:16 * n_entries],
                         block_cols=block_cols, n_blocks=n_blocks, n_entries=n_entries, n_su=n_entries // 4, split=split,
                         widths=widths, chunks=chunks, n_slices=n_entries, half=lib.wdg_sell16_row_bytes(self.n_cols) == 32)
        return True

    @property
    def nnz(self):
        return int(self.col.shape[0])

    @property
    def device(self):
        return self.rowptr.device

    # -- constructors ---------------------------------------------------------------------------
    @staticmethod
    def from_coo(src, dst, n, val=None, flags=0):
        """COO edge list (any integer dtype, host or device) -> CSR on the GPU via wdg_coo_to_csr_i32."""
        dev = require_gpu()
        src, dst = _dev(src, torch.int64, dev), _dev(dst, torch.int64, dev)
        val = _dev(val, torch.float32, dev)
        e, n = int(src.shape[0]), int(n)
        cap = lib.wdg_coo_to_csr_capacity(e, n, flags)
        rowptr = torch.empty(n + 1, dtype=torch.int32, device=dev)
        col = torch.empty(cap, dtype=torch.int32, device=dev)
        out = torch.empty(cap, dtype=torch.float32, device=dev)
        nnz = torch.zeros(1, dtype=torch.int64, device=dev)
        ws_bytes = lib.wdg_coo_to_csr_workspace_bytes(e, n, flags)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        check(lib.wdg_coo_to_csr_i32(_ptr(src), _ptr(dst), _ptr(val), e, n, flags, _ptr(rowptr), _ptr(col), _ptr(out),
                                     _ptr(nnz), _ptr(ws), ws_bytes, stream_handle()), "wdg_coo_to_csr_i32")
        k = int(nnz.item())  # the one host sync of graph construction
        if k < 0:
            raise IndexError("edge index out of range for a graph of %d nodes" % n)
        return CsrGraph(rowptr, col[:k], out[:k], n, n)

    @staticmethod
    def from_torch_sparse(a, flags=0):
        """torch sparse COO (coalesced or not; fp32/fp64 values) -> CSR.  Duplicates are summed like `.coalesce()`."""
        idx = a._indices()
        return CsrGraph.from_coo(idx[0], idx[1], a.shape[0], a._values(), flags)

    @staticmethod
    def from_dense(a):
        """Dense [N,M] fp32 -> CSR of its non-zero entries (wdg_dense_to_csr_*)."""
        dev = require_gpu()
        a = _dev(a, torch.float32, dev)
        n, m = a.shape
        rowptr = torch.empty(n + 1, dtype=torch.int32, device=dev)
        ws_bytes = lib.wdg_scan_workspace_bytes(n)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        check(lib.wdg_dense_to_csr_count(_ptr(a), _ld(a), n, m, _ptr(rowptr), _ptr(ws), ws_bytes,
                                         stream_handle()), "wdg_dense_to_csr_count")
        nnz = int(rowptr[-1].item())
        col = torch.empty(nnz, dtype=torch.int32, device=dev)
        val = torch.empty(nnz, dtype=torch.float32, device=dev)
        check(lib.wdg_dense_to_csr_fill(_ptr(a), _ld(a), n, m, _ptr(rowptr), _ptr(col), _ptr(val),
                                        stream_handle()), "wdg_dense_to_csr_fill")
        return CsrGraph(rowptr, col, val, n, m)

    @staticmethod
    def from_scipy(mx, flags=0):
        coo = mx.tocoo()
        return CsrGraph.from_coo(coo.row, coo.col, coo.shape[0], coo.data, flags)

    @staticmethod
    def from_scipy_csr(mx):
        """scipy sparse of ANY shape (N x F feature matrices included) -> device CSR by uploading `tocsr()`'s arrays
        (duplicates summed, rows sorted by scipy: the same canonical form the COO builder produces)."""
        dev = require_gpu()
        csr = mx.tocsr().copy()
        csr.sum_duplicates()
        csr.sort_indices()
        return CsrGraph(_dev(csr.indptr, torch.int32, dev), _dev(csr.indices, torch.int32, dev),
                        _dev(csr.data, torch.float32, dev), csr.shape[0], csr.shape[1])

    @staticmethod
    def from_any(a, flags=0):
        """Accept what the reference's functions are handed: torch sparse / dense tensors, scipy matrices, CsrGraph."""
        if isinstance(a, CsrGraph):
            return a
        if isinstance(a, torch.Tensor):
            if a.layout == torch.sparse_coo:
                g = getattr(a, "_wdg_csr", None)  # tagged by to_torch_sparse()
                if g is not None and flags == 0 and (g.n_rows, g.n_cols) == tuple(a.shape) and g.nnz == a._nnz():
                    return g
                return CsrGraph.from_torch_sparse(a, flags)
            if a.dim() == 2 and a.shape[0] == 2 and not a.is_floating_point():
                raise TypeError("edge-index tensors need an explicit node count: use CsrGraph.from_coo")
            g = CsrGraph.from_dense(a)
            return g if flags == 0 else g.rebuild(flags)
        if hasattr(a, "tocoo"):
            return CsrGraph.from_scipy(a, flags)
        raise TypeError(f"cannot build a CSR graph from {type(a)}")

    # -- views ------------------------------------------------------------------------------------
    def row_indices(self):
        """int64 row id of every stored entry (device), i.e. COO row vector in coalesced order."""
        counts = (self.rowptr[1:] - self.rowptr[:-1]).to(torch.int64)
        return torch.repeat_interleave(torch.arange(self.n_rows, device=self.device), counts)

    def rebuild(self, flags):
        return CsrGraph.from_coo(self.row_indices(), self.col, self.n_rows, self.val, flags)

    def transpose(self):
        """A^T as CSR (the backward pass of a directed graph needs it)."""
        g = CsrGraph.from_coo(self.col, self.row_indices(), self.n_cols, self.val, 0)
        g.n_cols = self.n_rows
        return g

    def to_torch_sparse(self):
        idx = torch.stack([self.row_indices(), self.col.to(torch.int64)])
        val = self.val if self.val is not None else torch.ones(self.nnz, device=self.device)
        t = torch.sparse_coo_tensor(idx, val, (self.n_rows, self.n_cols)).coalesce()
        # the API twins hand this tensor straight back to functions that need the CSR: from_any() finds it here instead of
        # running the COO -> CSR build again (the tensor is a view of the same pattern; .coalesce() / arithmetic drop the tag)
        t._wdg_csr = self
        return t

    def with_values(self, val):  # (neither SELL copy is shared: both hold values)
        return CsrGraph(self.rowptr, self.col, val, self.n_rows, self.n_cols)  # SELL copy (holds values) not shared


_PACK_STAGING = {}  # device index -> a ring of page-locked int32 staging tensors, grown on demand, reused shard after shard


def _host_pack_coo(coos, lens, node_ptr_h, e_total, dev):
    """-> (src, dst) int32 device tensors holding the shard's edge lists as ids of the block-diagonal union, or None when the
    inputs are not plain contiguous host arrays of one integer width (the caller then takes the torch path).  Raises IndexError
    for an id outside its graph."""
    arrs = [(c[0], c[1]) for c in coos]
    kinds = {a.dtype for pair in arrs for a in pair if isinstance(a, np.ndarray)}
    if len(kinds) != 1 or not all(isinstance(a, np.ndarray) and a.flags.c_contiguous and a.ndim == 1 for pair in arrs for a in pair):
        return None
    kind = kinds.pop()
    if kind not in (np.dtype(np.int64), np.dtype(np.int32)):
        return None
    G = len(coos)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    ring = _PACK_STAGING.setdefault(key, {"next": 0, "bufs": [None] * 3})  # (three: pipelined shards keep two uploads in flight)
    slot = ring["next"]
    ring["next"] = (slot + 1) % len(ring["bufs"])
    stage = ring["bufs"][slot]
    if stage is None or stage.numel() < 2 * e_total:
        stage = ring["bufs"][slot] = torch.empty(int(2 * e_total * 1.25) + 1024, dtype=torch.int32).pin_memory()
    elif getattr(stage, "_busy", None) is not None:
        stage._busy.synchronize()  # the copy that last read this buffer (three shards ago) has left it
    ptrs = ctypes.c_void_p * G
    sp, dp = ptrs(*[a.ctypes.data for a, _b in arrs]), ptrs(*[b.ctypes.data for _a, b in arrs])
    lens_a = np.asarray(lens, np.int64)
    nptr = np.ascontiguousarray(node_ptr_h, np.int32)
    bad = ctypes.c_int32(0)
    host = stage.numpy()
    threads = int(os.environ.get("WDG_HOST_PACK_THREADS", "8"))
    check(lib.wdg_host_pack_coo_i32(sp, dp, lens_a.ctypes.data, nptr.ctypes.data, G, kind.itemsize, host[:e_total].ctypes.data,
                                    host[e_total:2 * e_total].ctypes.data, ctypes.byref(bad), threads), "wdg_host_pack_coo_i32")
    if bad.value:
        raise IndexError("edge index out of range for its graph")
    both = stage[:2 * e_total].to(dev, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    stage._busy = ev  # the next shard may not overwrite the staging buffer before this copy has left it
    return both[:e_total], both[e_total:]


class GraphBatch:
    """A sweep shard's graphs built together: ONE COO -> CSR build of their block-diagonal union (wdg_coo_blockdiag_offset,
    wdg_coo_to_csr_i32, wdg_csr_split_blockdiag), the SELL-16 copies of all of them in five more launches
    (wdg_csr_to_sell16_count_batched / _fill_batched) and ONE host read-back for the whole shard - against ~20 launches and two
    host syncs per graph through CsrGraph.from_coo + ensure_quad (the cold path of a one-pass sweep: synthetic_plot.py:78-109
    visits every graph once).  The results are bit for bit the per-graph builds' (tests/test_gpu_batched_build.py).

    .graphs: list of CsrGraph (views into the shard's pooled arrays; .quad set when `quad`); degree_norm(): every graph's
    degrees / coefficients from one launch over the union."""

    def __init__(self, coos, flags=0, quad=True, quad_values=False, max_padding=4.0):
        """coos: list of (src, dst, n) or (src, dst, n, val): host arrays or tensors, node ids local to each graph."""
        dev = require_gpu()
        G = len(coos)
        self.flags = flags
        ns = [int(c[2]) for c in coos]
        es = [int(len(c[0])) for c in coos]
        node_ptr_h = np.concatenate([[0], np.cumsum(ns)]).astype(np.int64)
        edge_ptr_h = np.concatenate([[0], np.cumsum(es)]).astype(np.int64)
        self.n_total, e_total = int(node_ptr_h[-1]), int(edge_ptr_h[-1])
        if self.n_total >= (1 << 31) - 1:
            raise ValueError("GraphBatch: more than 2^31 nodes in one shard")
        any_val = any(len(c) > 3 and c[3] is not None for c in coos)

        def cat(parts, dtype):
            if all(isinstance(p_, torch.Tensor) for p_ in parts):
                return torch.cat([p_.to(dev, dtype) for p_ in parts]) if parts else torch.empty(0, dtype=dtype, device=dev)
            host = np.concatenate([np.asarray(p_.cpu() if isinstance(p_, torch.Tensor) else p_) for p_ in parts]) if parts else np.empty(0)
            return torch.from_numpy(np.ascontiguousarray(host)).to(device=dev, dtype=dtype)

        # host arrays of one integer width: packed by the library's host threads straight into a page-locked int32 buffer, ids
        # already those of the block-diagonal union (wdg_host_pack_coo_i32) - one upload of 4-byte indices instead of numpy
        # concatenation + two pageable int64 uploads + the offset kernel
        packed = None
        if G and e_total and os.environ.get("WDG_SWEEP_HOST_PACK", "1") != "0":
            packed = _host_pack_coo(coos, es, node_ptr_h, e_total, dev)
        if packed is None:
            src, dst = cat([c[0] for c in coos], torch.int64), cat([c[1] for c in coos], torch.int64)
        val = None
        if any_val:
            val = cat([(c[3] if len(c) > 3 and c[3] is not None else np.ones(es[i], np.float32)) for i, c in enumerate(coos)], torch.float32)
        node_ptr = _h2d(node_ptr_h.astype(np.int32), dev)
        edge_ptr = _h2d(edge_ptr_h, dev)
        st = stream_handle()
        # everything the host wants to know afterwards, in one buffer: [bad, nnz of the union, nnz per graph ...]
        info = torch.zeros(2 + max(G, 1), dtype=torch.int64, device=dev)
        bad = torch.zeros(2, dtype=torch.int32, device=dev)
        if packed is None:
            check(lib.wdg_coo_blockdiag_offset(_ptr(src), _ptr(dst), _ptr(edge_ptr), _ptr(node_ptr), G, e_total, _ptr(bad), st),
                  "wdg_coo_blockdiag_offset")
        else:
            src, dst = packed  # (range-checked on the host: _host_pack_coo raised already)
        cap = lib.wdg_coo_to_csr_capacity(e_total, self.n_total, flags)
        self.rowptr = torch.empty(self.n_total + 1, dtype=torch.int32, device=dev)
        self.col = torch.empty(cap, dtype=torch.int32, device=dev)
        self.val = torch.empty(cap, dtype=torch.float32, device=dev)
        ws_bytes = lib.wdg_coo_to_csr_workspace_bytes(e_total, self.n_total, flags)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        build = lib.wdg_coo_to_csr_i32 if packed is None else lib.wdg_coo32_to_csr_i32
        check(build(_ptr(src), _ptr(dst), _ptr(val), e_total, self.n_total, flags, _ptr(self.rowptr), _ptr(self.col),
                    _ptr(self.val), c_void_p(info.data_ptr() + 8), _ptr(ws), ws_bytes, st), "wdg_coo_to_csr_i32")
        # per-graph buffers of the SELL-16 build, pooled; the job table's rowptr / col / val are filled in by the split kernel
        self.rowptr_pool = torch.zeros(self.n_total + G, dtype=torch.int32, device=dev)  # (zeros: a graph of no nodes keeps rowptr = [0])
        quad = quad and not quad_disabled() and G > 0
        jobs = (_lib.Sell16Job * max(G, 1))()
        if quad:
            n_blocks = [(max(n, 1) + lib.wdg_sell16_block_cols(n) - 1) // lib.wdg_sell16_block_cols(n) for n in ns]
            quad = max(n_blocks) <= CsrGraph.QUAD_MAX_BLOCKS and max(ns) <= 16384
        if quad:
            max_e = [int(lib.wdg_sell16_max_entries(n)) for n in ns]
            ext_len = [2 * (nb * m + 1) for nb, m in zip(n_blocks, max_e)]
            r64 = lambda v: (v + 63) // 64 * 64  # noqa: E731  (every graph's slice of a pool starts 256-byte aligned, like its own allocation)
            ext_off = np.concatenate([[0], np.cumsum([r64(v) for v in ext_len])]).astype(np.int64)
            rows_off = np.concatenate([[0], np.cumsum([r64(16 * m) for m in max_e])]).astype(np.int64)
            perm_len = [(n + 15) // 16 * 16 for n in ns]
            perm_off = np.concatenate([[0], np.cumsum([r64(v) for v in perm_len])]).astype(np.int64)
            ws_len = [(int(lib.wdg_sell16_workspace_bytes(n, n)) + 255) // 256 * 256 for n in ns]
            ws_off = np.concatenate([[0], np.cumsum(ws_len)]).astype(np.int64)
            ext = torch.empty(int(ext_off[-1]), dtype=torch.int32, device=dev)
            rows = torch.empty(int(rows_off[-1]), dtype=torch.int32, device=dev)
            perm = torch.empty(int(perm_off[-1]), dtype=torch.int32, device=dev)
            qws = torch.empty(int(ws_off[-1]) + 256, dtype=torch.uint8, device=dev)
            for g_, job in enumerate(jobs[:G]):
                job.q_perm, job.q_ext = perm.data_ptr() + 4 * int(perm_off[g_]), ext.data_ptr() + 4 * int(ext_off[g_])
                job.q_rows, job.workspace = rows.data_ptr() + 4 * int(rows_off[g_]), qws.data_ptr() + int(ws_off[g_])
                job.n_rows = job.n_cols = ns[g_]
            table = _table(jobs) if G else None
        else:
            table = None
        check(lib.wdg_csr_split_blockdiag(_ptr(self.rowptr), _ptr(self.col), _ptr(self.val), _ptr(node_ptr), G, self.n_total,
                                          _ptr(self.rowptr_pool), c_void_p(info.data_ptr() + 16), _ptr(table), st),
              "wdg_csr_split_blockdiag")
        if quad:
            check(lib.wdg_csr_to_sell16_count_batched(_ptr(table), G, max(ns), max(ns), st), "wdg_csr_to_sell16_count_batched")
        # ---- the shard's ONE host read-back: the info block and, behind it in the same buffer, the pool of extents (64 KB for 50
        #      graphs: the widths price the aggregation's tape cut, the tails size the index arrays) - one blocking copy
        info[0:1].copy_(bad[0:1])
        if quad:
            both = torch.cat([info.view(torch.int32), ext]).cpu().numpy()
            info_h, ext_h = both[:2 * info.numel()].view(np.int64), both[2 * info.numel():]
        else:
            info_h = info.cpu().numpy()
        if info_h[0] != 0 or info_h[1] < 0:
            raise IndexError("edge index out of range for its graph")
        nnz_g = info_h[2:2 + G]
        base = np.concatenate([[0], np.cumsum(nnz_g)]).astype(np.int64)
        self.node_ptr_host = node_ptr_h
        self.graphs = []
        for g_ in range(G):
            o = int(node_ptr_h[g_]) + g_
            self.graphs.append(CsrGraph(self.rowptr_pool[o:o + ns[g_] + 1], self.col[int(base[g_]):int(base[g_ + 1])],
                                        self.val[int(base[g_]):int(base[g_ + 1])], ns[g_], ns[g_]))
        if not quad:
            return
        want, chunks_g = [], []
        for g_ in range(G):
            tail = ext_h[int(ext_off[g_]) + ext_len[g_] - 2:int(ext_off[g_]) + ext_len[g_]]
            chunks, word = int(tail[0]), int(tail[1])
            n_entries, tasks = word & 0x3fffffff, (word & 0x3fffffff) * n_blocks[g_]
            ok = ns[g_] > 0 and nnz_g[g_] > 0 and chunks * 256 <= max_padding * nnz_g[g_] + 256 * tasks
            want.append(ok)
            chunks_g.append(chunks if ok else 0)
        # (+ 2 chunks of slack per graph: the kernel requests an entry's two chunks unconditionally)
        qoff = np.concatenate([[0], np.cumsum([(c + 2) * 256 if w else 0 for c, w in zip(chunks_g, want)])]).astype(np.int64)
        q_col = torch.zeros(int(qoff[-1]), dtype=torch.int32, device=dev)
        q_val = torch.zeros(int(qoff[-1]), dtype=torch.float32, device=dev) if quad_values else None
        for g_, job in enumerate(jobs[:G]):
            gr = self.graphs[g_]
            job.rowptr, job.col, job.val = gr.rowptr.data_ptr(), gr.col.data_ptr() if gr.nnz else 0, gr.val.data_ptr() if gr.nnz else 0
            job.q_col = q_col.data_ptr() + 4 * int(qoff[g_]) if want[g_] else 0
            job.q_val = q_val.data_ptr() + 4 * int(qoff[g_]) if (want[g_] and quad_values) else 0
        table = _table(jobs)
        check(lib.wdg_csr_to_sell16_fill_batched(_ptr(table), G, max(ns), max(ns), st), "wdg_csr_to_sell16_fill_batched")
        self._keep = (table, qws)
        for g_, gr in enumerate(self.graphs):
            if not want[g_]:
                gr.quad = False
                continue
            word = int(ext_h[int(ext_off[g_]) + ext_len[g_] - 1])
            n_entries, split = word & 0x3fffffff, bool(word & (1 << 30))
            tasks = n_entries * n_blocks[g_]
            eh = ext_h[int(ext_off[g_]):int(ext_off[g_]) + ext_len[g_]].reshape(-1, 2)
            widths = (eh[:tasks, 1] & 0x3fffffff).reshape(n_blocks[g_], n_entries).copy()
            a, b = int(qoff[g_]), int(qoff[g_ + 1])
            gr.quad = dict(ext=ext[int(ext_off[g_]):int(ext_off[g_]) + 2 * (tasks + 1)], col=q_col[a:b],
                           val=q_val[a:b] if quad_values else None,
                           perm=perm[int(perm_off[g_]):int(perm_off[g_]) + perm_len[g_]], rows=rows[int(rows_off[g_]):int(rows_off[g_]) + 16 * n_entries],
                           block_cols=int(lib.wdg_sell16_block_cols(ns[g_])), n_blocks=n_blocks[g_], n_entries=n_entries,
                           n_su=n_entries // 4, split=split, widths=widths, chunks=chunks_g[g_], n_slices=n_entries,
                           half=lib.wdg_sell16_row_bytes(ns[g_]) == 32)

    def degree_norm(self, mode=NORM_RW, prec=PREC_F32, use_values=True):
        """-> list (one dict per graph, like ops.degree_norm) of views into the union's arrays: one launch for the shard"""
        dev = self.rowptr.device
        n = self.n_total
        out = dict(rowsum=torch.empty(n, dtype=torch.float32, device=dev), cnt=torch.empty(n, dtype=torch.int32, device=dev),
                   dinv=torch.empty(n, dtype=torch.float32, device=dev), dinv64=torch.empty(n, dtype=torch.float64, device=dev))
        check(lib.wdg_degree_norm(_ptr(self.rowptr), _ptr(self.val if use_values else None), n, mode, prec, _ptr(out["rowsum"]),
                                  _ptr(out["cnt"]), _ptr(out["dinv"]), _ptr(out["dinv64"]), stream_handle()), "wdg_degree_norm")
        p = self.node_ptr_host
        return [{k: v[int(p[g_]):int(p[g_ + 1])] for k, v in out.items()} for g_ in range(len(self.graphs))]


# ------------------------------------------------------------------------------------------- normalisation
def degree_norm(g, mode=NORM_RW, prec=PREC_F32, use_values=True):
    """-> dict(rowsum fp32[N], cnt int32[N], dinv fp32[N], dinv64 fp64[N]) ; wdg_degree_norm."""
    dev = g.device
    n = g.n_rows
    out = dict(rowsum=torch.empty(n, dtype=torch.float32, device=dev), cnt=torch.empty(n, dtype=torch.int32, device=dev),
               dinv=torch.empty(n, dtype=torch.float32, device=dev), dinv64=torch.empty(n, dtype=torch.float64, device=dev))
    val = g.val if use_values else None
    check(lib.wdg_degree_norm(_ptr(g.rowptr), _ptr(val), n, mode, prec, _ptr(out["rowsum"]), _ptr(out["cnt"]),
                              _ptr(out["dinv"]), _ptr(out["dinv64"]), stream_handle()), "wdg_degree_norm")
    return out


def normalise_values(g, mode=NORM_RW, prec=PREC_F32):
    """A_hat's stored values as the reference materialises them (D^-1 A or D^-1/2 A D^-1/2) -> new CsrGraph."""
    d = degree_norm(g, mode, prec)
    out = torch.empty(g.nnz, dtype=torch.float32, device=g.device)
    if g.nnz == 0:  # nothing stored: nothing to scale (empty tensors have no device pointer to hand over)
        return g.with_values(out)
    check(lib.wdg_normalise_values(_ptr(g.rowptr), _ptr(g.col), _ptr(g.val), g.n_rows, mode, prec, _ptr(d["dinv"]),
                                   _ptr(d["dinv64"]), _ptr(out), stream_handle()), "wdg_normalise_values")
    return g.with_values(out)


def row_l1_normalise(x, use_abs=False):
    dev = require_gpu()
    x = _dev(x, torch.float32, dev)
    y = torch.empty_like(x)
    check(lib.wdg_row_l1_normalise_f32(_ptr(x), _ld(x), _ptr(y), _ld(y), x.shape[0], x.shape[1],
                                       int(use_abs), stream_handle()), "wdg_row_l1_normalise_f32")
    return y


def unpack_bits(words, n_feat, row_normalise=False):
    """[N, ceil(F / 32)] int32 words of bit-packed 0/1 features (graph_io.pack_bits) -> dense fp32 [N, F] on the GPU."""
    dev = require_gpu()
    words = _dev(words, torch.int32, dev)
    out = torch.empty((words.shape[0], int(n_feat)), dtype=torch.float32, device=dev)
    check(lib.wdg_unpack_bits_f32(_ptr(words), _ld(words), words.shape[0], int(n_feat), int(row_normalise), _ptr(out),
                                  _ld(out), stream_handle()), "wdg_unpack_bits_f32")
    return out


# ------------------------------------------------------------------------------------------- aggregation
NARROW_MIN_ENTRIES = int(os.environ.get("WDG_NARROW_MIN_ENTRIES", 1 << 15))  # below: the general families (round 2: 2^18 - chameleon's
# 65 019 entries then took the gather kernel for its C = 5 logits aggregation: 85 us against the narrow kernel's 20)


def _fill_job(job, g, x, y, row_scale, col_scale, use_values=True, band=False):
    job.rowptr, job.col = g.rowptr.data_ptr(), g.col.data_ptr()
    job.val = g.val.data_ptr() if (use_values and g.val is not None and not g.unit_values) else 0
    job.row_scale = 0 if row_scale is None else row_scale.data_ptr()
    job.col_scale = 0 if col_scale is None else col_scale.data_ptr()
    job.X, job.Y = x.data_ptr(), y.data_ptr()
    job.ldx, job.ldy = _ld(x), _ld(y)
    job.n_rows, job.n_cols, job.n_feat = g.n_rows, g.n_cols, x.shape[1]
    job.reserved = ABLATE_BITS  # 0 on every product path; only scripts/ablate_*.py assign the module variable
    wants_val = bool(job.val)
    q = g.quad
    if q and (not wants_val or q["val"] is not None):
        job.q_ext, job.q_col, job.q_perm = q["ext"].data_ptr(), q["col"].data_ptr(), q["perm"].data_ptr()
        job.q_rows = q["rows"].data_ptr()
        job.q_val = q["val"].data_ptr() if (wants_val and q["val"] is not None) else 0
        job.q_block_cols, job.q_n_blocks = q["block_cols"], q["n_blocks"]
        job.q_n_entries, job.q_flags = q["n_entries"], (1 if q["split"] else 0) | (2 if q["half"] else 0)
    else:
        job.q_ext = job.q_col = job.q_val = job.q_perm = job.q_rows = 0
        job.q_block_cols = job.q_n_blocks = job.q_n_entries = job.q_flags = 0
    if band and g.band:
        job.band_perm, job.band_cuts, job.band_n_hub = g.band["perm"].data_ptr(), g.band["cuts"].data_ptr(), g.band["n_hub"]
        # (the single-graph entry point prefers a split-form SELL-16 copy: this call asked for the band kernel)
        job.q_ext = job.q_col = job.q_val = job.q_perm = job.q_rows = 0
        job.q_block_cols = job.q_n_blocks = job.q_n_entries = job.q_flags = 0
    else:
        job.band_perm = job.band_cuts = 0
        job.band_n_hub = 0
    job.band_reserved = 0
    return job


def quad_disabled():
    """WDG_SPMM_NO_QUAD=1: keep every aggregation off the quad-row kernel (the CSR slab / gather kernels: A/B comparisons, tests)"""
    return os.environ.get("WDG_SPMM_NO_QUAD", "0") not in ("", "0")


def _sharing_groups(entries):
    groups = {}
    for i, (g, x, *_rest) in enumerate(entries):
        groups.setdefault((x.data_ptr(), _ld(x), g.n_cols, x.shape[1]), []).append(i)
    return list(groups.values())


def _dma_ok(job):
    """WDG_SPMM_DMA_OK contract of include/wdg.h for one job descriptor."""
    return (not job.col_scale and (job.X or 0) % 16 == 0 and (job.Y or 0) % 16 == 0 and job.ldx % 4 == 0
            and job.ldy % 4 == 0 and job.n_feat % 4 == 0)


def spmm(g, x, row_scale=None, col_scale=None, use_values=True, out=None):
    """Y = diag(row_scale) A diag(col_scale) X on the GPU (wdg_spmm_csr_f32 / _bf16 by x.dtype)."""
    dev = require_gpu()
    if x.dtype not in (torch.float32, torch.bfloat16):
        x = x.to(torch.float32)
    x = x.to(dev)
    if x.stride(1) != 1:
        x = x.contiguous()
    if x.shape[0] != g.n_cols:
        raise ValueError(f"spmm: X has {x.shape[0]} rows, adjacency has {g.n_cols} columns")
    y = out if out is not None else torch.empty((g.n_rows, x.shape[1]), dtype=torch.float32, device=dev)
    row_scale, col_scale = _dev(row_scale, torch.float32, dev), _dev(col_scale, torch.float32, dev)
    if x.shape[1] <= 8 and g.nnz >= NARROW_MIN_ENTRIES and os.environ.get("WDG_SPMM_NARROW", "1") != "0" and g.ensure_band():
        # few features on a large graph: packed sources, lanes split the entries (csrc/spmm_narrow.hip)
        job = _fill_job(SpmmJob(), g, x, y, row_scale, col_scale, use_values, band=True)
        ws_bytes = lib.wdg_spmm_narrow_workspace_bytes(g.n_rows, g.n_cols)
        if g.narrow_ws is None or g.narrow_ws.numel() < ws_bytes:
            g.narrow_ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        # 16-byte source rows (<= 4 features, or bf16 sources without a column scale) halve the table: fewer column ranges
        col_bytes = int(lib.wdg_spmm_narrow_col_bytes(x.shape[1], int(x.dtype == torch.bfloat16), int(col_scale is not None)))
        parts = int(lib.wdg_spmm_narrow_parts(g.n_cols, col_bytes))
        if g.narrow_parts is None:
            g.narrow_parts = {}
        if parts > 1 and parts not in g.narrow_parts:  # one-time per graph and range count: every row's split positions
            pp = torch.empty(g.n_rows * (parts - 1), dtype=torch.int32, device=dev)
            check(lib.wdg_spmm_narrow_plan(_ptr(g.rowptr), _ptr(g.col), g.n_rows, g.n_cols, parts, _ptr(pp), stream_handle()), "wdg_spmm_narrow_plan")
            g.narrow_parts[parts] = pp
        part_ptr = g.narrow_parts[parts] if parts > 1 else None
        fn = lib.wdg_spmm_narrow_bf16 if x.dtype == torch.bfloat16 else lib.wdg_spmm_narrow_f32
        check(fn(ctypes.byref(job), _ptr(part_ptr), _ptr(g.narrow_ws), ws_bytes, stream_handle()), "wdg_spmm_narrow")
        return y
    band = x.dtype == torch.float32 and g.prefers_band(x.shape[1])  # one-time plan -> band kernel (wide features, skew)
    if not band and x.shape[1] >= 8:
        g.ensure_quad()  # one-time SELL-16 copy -> quad-row kernel (<= 10 112 columns); else the CSR slab / gather kernels
    job = _fill_job(SpmmJob(), g, x, y, row_scale, col_scale, use_values, band=band)
    fn = lib.wdg_spmm_csr_bf16 if x.dtype == torch.bfloat16 else lib.wdg_spmm_csr_f32
    check(fn(ctypes.byref(job), stream_handle()), "wdg_spmm_csr")
    return y


# measured cost of a 16-row slice as a function of its entries per row (scripts/calibrate_quad.py, ns per slice and wave on
# homogeneous batches of N = 2000 graphs, F = 512): flat while the slice's store bounds it, then ~43 ns per entry (LDS)
_QUAD_COST_W = np.array([0, 3, 7, 11, 13, 15, 17, 21, 26, 34, 41, 51, 67], np.float64)
_QUAD_COST_NS = np.array([1590, 1590, 1619, 1668, 1756, 1768, 1908, 1980, 2076, 2457, 2667, 3027, 3780], np.float64)


def _quad_unit_cost(widths):
    """modelled cost of the super-units (4 slices = 64 rows each) of a graph from its slices' entries per row (summed over
    column blocks): the measured table above, extended linearly; WDG_QUAD_ALPHA / WDG_QUAD_WMIN (entries) select the
    two-parameter model max(width, wmin) + alpha instead (experiments)"""
    w = widths.sum(0).astype(np.float64)
    if "WDG_QUAD_ALPHA" in os.environ or "WDG_QUAD_WMIN" in os.environ:
        per_slice = np.maximum(w, float(os.environ.get("WDG_QUAD_WMIN", "8"))) + float(os.environ.get("WDG_QUAD_ALPHA", "4"))
    else:
        per_slice = np.where(w <= _QUAD_COST_W[-1], np.interp(w, _QUAD_COST_W, _QUAD_COST_NS),
                             _QUAD_COST_NS[-1] + 43.0 * (w - _QUAD_COST_W[-1]))
    return per_slice.reshape(-1, 4).sum(1)


def _quad_unit_costs(widths_list):
    """_quad_unit_cost for many graphs in one pass over the concatenation of their slices (a cold shard prices 50 graphs: one
    interpolation instead of 50) -> list of per-graph arrays"""
    if not widths_list:
        return []
    sizes = [w.shape[1] for w in widths_list]
    allw = np.concatenate([w.sum(0) for w in widths_list])[None, :] if len(widths_list) > 1 else widths_list[0]
    cost = _quad_unit_cost(allw)
    cuts = np.cumsum([sz // 4 for sz in sizes])[:-1]
    return np.split(cost, cuts)


# what a phase costs a workgroup beside its super-units: staging the slab, two barriers, the pipeline's prologue and the
# ragged end of the 16 waves (measured as the extra time of XCDs whose segment holds two phases)
_QUAD_PHASE_NS = 0.0


def _quad_cut(cum, g_off, n_seg, phase, shares=None):
    """cut positions [n_seg + 1] of the tape (cum = cumulative super-unit cost, g_off = phase-group boundaries) such that
    segment s's cost + `phase` per phase group it touches is shares[s] of the whole (equal shares by default): the smallest
    such bound, by bisection"""
    n_units = len(cum) - 1
    w = np.full(n_seg, 1.0) if shares is None else np.asarray(shares, np.float64) * n_seg / float(np.sum(shares))
    if (phase <= 0 and shares is None) or n_units == 0:
        cuts = np.searchsorted(cum, cum[-1] * np.arange(1, n_seg) / n_seg, side="left")
        return np.maximum.accumulate(np.concatenate([[0], np.clip(cuts, 0, n_units), [n_units]]))

    def fill(bound):
        cuts, a = [0], 0
        for s_ in range(n_seg):
            cap = bound * w[s_]
            while a < n_units:
                cap -= phase
                if cap <= 0:
                    break
                gi = int(np.searchsorted(g_off, a, side="right") - 1)
                gb = int(g_off[gi + 1])
                fit = int(np.searchsorted(cum, cum[a] + cap, side="right") - 1)
                if fit < gb:
                    a = max(fit, a)
                    break
                cap -= cum[gb] - cum[a]
                a = gb
            cuts.append(a)
        return cuts

    lo, hi = 0.0, (cum[-1] / n_seg + phase * (len(g_off) + 1) + cum[-1] / max(n_units, 1) * 2) / max(float(w.min()), 1e-3)
    for _ in range(50):
        mid = 0.5 * (lo + hi)
        if fill(mid)[-1] >= n_units:
            hi = mid
        else:
            lo = mid
    cuts = fill(hi)
    cuts[-1] = n_units
    return np.maximum.accumulate(np.asarray(cuts, np.int64))


QUAD_MULTI_ITEM_SU = 64  # super-units per item of a table whose graphs have several column blocks (csrc/spmm_quad.hip: Q_MAXU / 4 x 16 waves)


def _quad_segments(entries, order, n_feat, cus=256, phase_ns=None, shares=None):
    """Cut the tape of super-units (64 rows) of the jobs (in table order `order`) into 8 x S segments of equal modelled cost
    and split every segment into items; -> (items [(first_job, n_jobs, unit_begin, unit_end)], seg_ptr, n_segments).

    The tape is the concatenation of the PHASE GROUPS' super-units: a phase group is a run of consecutive jobs that aggregate
    the same X (same X, ldx, n_cols, n_feat, col_scale); an item is a range of one phase group's super-units."""
    half = len(order) > 0 and entries[order[0]][0].quad["half"]  # (32-byte slab rows: feature groups of 8; all jobs or none)
    n_groups = (n_feat + 7) // 8 if half else (n_feat + 15) // 16
    per_xcd = max(cus // 8, 1)
    keys = []
    for i in order:
        g, x, _y, _rs, cs = entries[i][:5]
        keys.append((x.data_ptr(), _ld(x), g.n_cols, x.shape[1], 0 if cs is None else cs.data_ptr()))
    groups, costs, multi = [], [], False  # groups: (first position in `order`, n_jobs, n_units)
    unit_costs = _quad_unit_costs([entries[i][0].quad["widths"] for i in order])  # (per position in `order`)
    pos = 0
    while pos < len(order):
        end = pos + 1
        while end < len(order) and keys[end] == keys[pos]:
            end += 1
        multi = multi or any(entries[order[k]][0].quad["n_blocks"] > 1 for k in range(pos, end))
        seq = np.concatenate(unit_costs[pos:end]) if end - pos > 1 else unit_costs[pos]
        groups.append((pos, end - pos, len(seq)))
        costs.append(seq)
        pos = end
    n_units = sum(g[2] for g in groups)
    cum = np.concatenate([[0.0], np.cumsum(np.concatenate(costs))]) if costs else np.zeros(1)
    # segments per XCD: one (segment, feature group) pair per workgroup when there are fewer groups than workgroups per XCD,
    # else the S in 1..4 that leaves the least idle tail; never more segments than 8-super-unit bundles
    if n_groups >= per_xcd:
        subs = min(range(1, 5), key=lambda s_: (-(-s_ * n_groups // per_xcd) / (s_ * n_groups / per_xcd), s_))
    else:
        subs = -(-per_xcd // n_groups)
    forced = os.environ.get("WDG_QUAD_SUBS")
    if forced:
        subs = int(forced)
    subs = max(1, min(subs, max(1, n_units // (8 * 8))))
    if multi:  # a wave keeps at most 4 super-units (Q_MAXU = 16 slices) across the column blocks: items of <= 64 super-units
        subs = max(subs, -(-n_units // (8 * QUAD_MULTI_ITEM_SU)))
    n_seg = 8 * subs
    g_off = np.concatenate([[0], np.cumsum([g[2] for g in groups])]).astype(np.int64)
    if phase_ns is None:
        phase_ns = float(os.environ.get("WDG_QUAD_PHASE_NS", _QUAD_PHASE_NS))
    phase = float(phase_ns) * 16  # (the table prices a wave, a workgroup has 16)
    if shares is not None and len(shares) != n_seg:
        shares = None
    cuts = _quad_cut(cum, g_off, n_seg, phase, shares)
    items, seg_ptr = [], [0]
    for s_ in range(n_seg):
        a, b = int(cuts[s_]), int(cuts[s_ + 1])
        seg_items = []
        while a < b:
            gi = int(np.searchsorted(g_off, a, side="right") - 1)
            end = min(b, int(g_off[gi + 1]))
            if multi:  # an item of a several-block table lies inside one job (the kernel's wave keeps that job's slices)
                first_pos, nj_, _n = groups[gi]
                jb = np.concatenate([[0], np.cumsum([entries[order[first_pos + t]][0].quad["n_su"] for t in range(nj_)])]) + int(g_off[gi])
                end = min(end, a + QUAD_MULTI_ITEM_SU, int(jb[np.searchsorted(jb, a, side="right")]))
            first, nj, _n = groups[gi]
            items.append((first, nj, a - int(g_off[gi]), end - int(g_off[gi])))
            seg_items.append((float(cum[end] - cum[a]), len(items) - 1))
            a = end
        # the phases of a segment run shortest first: staging a slab costs 8 us while the memory system is quiet and 20 - 55 us
        # once the launch's stores have filled the write path (scripts/dev/stamps_quad_phases.py: the later the switch, the dearer)
        if len(seg_items) > 1 and os.environ.get("WDG_QUAD_PHASE_ORDER", "1") != "0" and not multi:
            first_item = seg_items[0][1]
            reordered = [items[i] for _c, i in sorted(seg_items)]
            items[first_item:first_item + len(reordered)] = reordered
        seg_ptr.append(len(items))
    return items, seg_ptr, n_seg


class SpmmBatch:
    """Job table for the batched aggregation: many graphs, one launch.  Built once, launched many times.

    Tables whose graphs all carry a SELL-16 copy (<= 10 112 columns, F >= 8) run on the quad-row kernel
    (wdg_spmm_quad_batched_f32: the tape of 16-row units cut into equal-cost segments, graphs that aggregate the same X
    adjacent so that a workgroup stages X once per run); the rest on wdg_spmm_batched_f32 (WDG_SPMM_NO_QUAD=1: all)."""

    def __init__(self, entries):
        """entries: list of (CsrGraph, X, Y, row_scale|None, col_scale|None, use_values)."""
        dev = require_gpu()
        self.keep = entries  # tensors must outlive the table
        arr = (SpmmJob * len(entries))()
        self.max_rows = self.max_cols = self.max_feat = 0
        any_val, dma_ok = False, len(entries) > 0
        for g, x, *_ in entries:
            if x.dtype != torch.float32 or x.stride(1) != 1:
                raise ValueError("SpmmBatch: X must be fp32 with unit inner stride")
        feats = {e[1].shape[1] for e in entries}
        # <= 8 features read in place as one or two float4 per source row (the sweep's logits aggregation): narrow kernel
        need_ld = 8 if max(feats, default=0) > 4 else 4
        self.narrow = (len(entries) > 0 and max(feats) <= 8 and os.environ.get("WDG_SPMM_NARROW", "1") != "0"
                       and all(e[1].data_ptr() % 16 == 0 and _ld(e[1]) % 4 == 0 and _ld(e[1]) >= need_ld for e in entries))
        self.quad = (len(entries) > 0 and not self.narrow and not quad_disabled() and min(feats) >= 8
                     and all(e[0].ensure_quad() for e in entries)
                     and all((not (e[5] and e[0].val is not None)) or e[0].quad["val"] is not None for e in entries)
                     # graphs of 2529 .. 5056 columns carry copies over 32-byte slab rows: a quad table holds them only, or none
                     and len({e[0].quad["half"] for e in entries}) == 1)
        # the kernels start jobs in table order: most stored entries first, so the long jobs do not end up in the tail
        order = sorted(range(len(entries)), key=lambda i: -entries[i][0].nnz)
        if os.environ.get("WDG_SPMM_ORDER") == "0":
            order = list(range(len(entries)))
        if self.quad:
            # graphs that aggregate the same X adjacent (largest first inside a group, groups by total entries)
            groups = _sharing_groups(entries)
            groups.sort(key=lambda grp: -sum(entries[i][0].nnz for i in grp))
            order = [i for grp in groups for i in sorted(grp, key=lambda i: -entries[i][0].nnz)]
        for job, (g, x, y, rs, cs, uv) in zip(arr, (entries[i] for i in order)):
            _fill_job(job, g, x, y, rs, cs, uv)
            any_val = any_val or bool(job.val)
            dma_ok = dma_ok and _dma_ok(job)
            self.max_rows, self.max_cols = max(self.max_rows, g.n_rows), max(self.max_cols, g.n_cols)
            self.max_feat = max(self.max_feat, x.shape[1])
        self.n_jobs = len(entries)
        self.table = _table(arr)
        self.edges = sum(e[0].nnz for e in entries)
        self.flags = (SPMM_ANY_VAL if any_val else 0) | (SPMM_DMA_OK if dma_ok else 0)
        if any(e[4] is not None for e in entries):
            self.flags |= SPMM_ANY_COL_SCALE
        if self.quad:
            if all(e[0].n_rows * _ld(e[2]) < (1 << 30) and e[0].quad["chunks"] < (1 << 22) - 2 and e[0].quad["split"] for e in entries):
                self.flags |= SPMM_SMALL_OFFSETS
            if entries[0][0].quad["half"]:
                self.flags |= SPMM_HALF_SLAB
            self.order = order
            self._set_segments(None)
            if os.environ.get("WDG_QUAD_VERIFY", "0") not in ("", "0"):
                self.verify()

    def verify(self, tol=1e-5):
        """WDG_QUAD_VERIFY=1 (or called directly): launch the table once and check every job against the CSR gather kernel.
        The quad-row kernel's fast loop issues its loads and stores from inline asm with hand-counted `s_waitcnt vmcnt`
        (csrc/spmm_quad.hip: q_units_fast) - invisible to the compiler's own bookkeeping; tests/test_abi.py checks the
        generated code of the shipped build, this checks the results on the machine and data at hand.  The two kernels
        sum a row's entries in different orders: agreement to `tol` of the largest entry, not bitwise.  Overwrites Y."""
        if not self.quad:
            return
        self.launch()
        torch.cuda.synchronize()
        for i, (g, x, y, rs, cs, uv) in enumerate(self.keep):
            got = y.clone()
            ref = torch.empty((g.n_rows, x.shape[1]), dtype=torch.float32, device=y.device)
            job = _fill_job(SpmmJob(), g, x, ref, rs, cs, uv)
            job.q_ext = job.q_col = job.q_val = job.q_perm = job.q_rows = 0  # no SELL copies: the CSR families
            job.q_block_cols = job.q_n_blocks = job.q_n_entries = job.q_flags = 0
            check(lib.wdg_spmm_csr_f32(ctypes.byref(job), stream_handle()), "wdg_spmm_csr_f32")
            torch.cuda.synchronize()
            err = float((got[:, :x.shape[1]] - ref).abs().max()) if ref.numel() else 0.0
            scale = max(float(ref.abs().max()) if ref.numel() else 0.0, 1e-30)
            if not err <= tol * scale:
                raise RuntimeError(f"WDG_QUAD_VERIFY: job {i} of the quad-row table differs from the CSR kernel by {err:.3e} "
                                   f"(largest entry {scale:.3e})")

    def _set_segments(self, phase_ns, shares=None):
        """cut the tape (ops._quad_segments; phase_ns: what a phase switch is priced at, None = the default; shares: the
        fraction of the modelled cost every segment gets, None = equal) and upload it"""
        dev = self.table.device
        items, seg_ptr, self.n_segments = _quad_segments(self.keep, self.order, self.max_feat, max(lib.wdg_device_cus(), 8), phase_ns, shares)
        iarr = (SpmmItem * max(len(items), 1))()
        for it, (fj, nj, ub, ue) in zip(iarr, items):
            it.first_job, it.n_jobs, it.unit_begin, it.unit_end = fj, nj, ub, ue
        self.items = _h2d(torch.frombuffer(bytearray(bytes(iarr)), dtype=torch.uint8), dev)
        self.seg_ptr = _h2d(np.asarray(seg_ptr, np.int32), dev)
        self.n_items = len(items)
        self.items_host, self.seg_ptr_host, self.phase_ns, self.shares = items, seg_ptr, phase_ns, shares  # (scripts read them)

    def segment_spans(self, clock):
        """[n_segments] us: how long each segment's workgroups ran in the launch that filled `clock` (launch(clock=...)): the
        latest end of the segment's XCD minus the launch's earliest start (one segment per XCD; else None)"""
        if self.n_segments != 8:
            return None
        t = clock.cpu().numpy().astype(np.float64).reshape(-1, 2) * 10e-3  # 100 MHz -> us
        start = t[:, 0].min()
        return np.array([t[x::8, 1].max() - start for x in range(8)])

    def tune(self, candidates=(0, 6000, 10000, 14000), launches=6):
        """Pick the tape cut by measurement (quad-row tables with more than one phase group only).  Where the phase switches
        of the eight XCDs fall relative to each other decides how dear they are (a slab staged while the other XCDs' stores
        fill the write path takes 20 - 55 us instead of 8), and that interplay is repeatable on a box but not monotone in
        any model parameter (scripts/dev/ab_phase_order.py): so the launch is timed for a few prices of a phase switch and the
        best cut is kept.  Every cut computes the same bits (a row's sum order is fixed by the SELL-16 copy).  Costs
        len(candidates) x launches launches, once per table; the outputs are (re)written with the same values."""
        if not self.quad or self.n_items <= self.n_segments or os.environ.get("WDG_QUAD_TUNE", "1") == "0":
            return None
        best = None
        for ph in candidates:
            self._set_segments(ph)
            self.launch()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(launches):
                self.launch()
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / launches
            if best is None or t < best[0]:
                best = (t, ph)
        self._set_segments(best[1])
        self.tuned = best
        return best

    def new_clock(self):
        """device buffer for launch(clock=...): start / end of every workgroup of the quad-row launch"""
        n = int(lib.wdg_spmm_quad_workgroups(self.n_segments, self.max_feat, self.flags))
        return torch.zeros(2 * n, dtype=torch.int64, device=self.table.device)

    def launch(self, clock=None):
        if clock is not None and self.quad:
            check(lib.wdg_spmm_quad_batched_clocked_f32(_ptr(self.table), self.n_jobs, _ptr(self.items), _ptr(self.seg_ptr),
                                                        self.n_segments, self.max_cols, self.max_feat, self.flags, _ptr(clock),
                                                        stream_handle()), "wdg_spmm_quad_batched_clocked_f32")
            return
        if self.narrow:
            check(lib.wdg_spmm_narrow_batched_f32(_ptr(self.table), self.n_jobs, self.max_rows, self.max_feat, self.flags,
                                                  stream_handle()), "wdg_spmm_narrow_batched_f32")
            return
        if self.quad:
            check(lib.wdg_spmm_quad_batched_f32(_ptr(self.table), self.n_jobs, _ptr(self.items), _ptr(self.seg_ptr),
                                                self.n_segments, self.max_cols, self.max_feat, self.flags, stream_handle()),
                  "wdg_spmm_quad_batched_f32")
            return
        check(lib.wdg_spmm_batched_f32(_ptr(self.table), self.n_jobs, self.max_rows, self.max_cols, self.max_feat,
                                       self.flags, stream_handle()), "wdg_spmm_batched_f32")

    def plan(self):
        if self.narrow:
            return 6, 16, 256
        if self.quad:
            return 5, 16, 1024
        return spmm_plan(self.max_rows, self.max_cols, self.max_feat, self.n_jobs, self.flags)

    def kernel_name(self):
        """name of the kernel this table launches, as rocprofv3 prints it (bench.py / scripts/bench_configs.py)"""
        fam, slab, threads = self.plan()
        val = "true" if self.flags & SPMM_ANY_VAL else "false"
        return {0: f"spmm_slab_kernel<{slab},{threads},float>", 1: "spmm_gather_kernel",
                5: f"spmm_quad_kernel<float,{val},{2 if self.flags & SPMM_HALF_SLAB else (1 if self.max_cols > 2528 else 0)}>",
                6: "spmm_narrow_batched_kernel"}.get(fam, f"family {fam}")


SPMM_ANY_VAL, SPMM_DMA_OK, SPMM_SMALL_OFFSETS, SPMM_ANY_COL_SCALE, SPMM_HALF_SLAB = 2, 4, 8, 16, 32
GEMM_A_VEC4 = 1


def spmm_plan(n_rows, n_cols, n_feat, n_jobs=1, flags=0):
    """(family, width, threads) of the CSR kernels behind wdg_spmm_batched_f32: 0 = LDS column slab, 1 = row gather."""
    slab, threads = ctypes.c_int(0), ctypes.c_int(0)
    fam = lib.wdg_spmm_plan(n_jobs, n_rows, n_cols, n_feat, flags, ctypes.byref(slab), ctypes.byref(threads))
    return fam, slab.value, threads.value


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


class GemmBatch:
    """Job table for wdg_gemm_batched_f32: C_i = act(A_i @ B_i + bias_i), one launch."""

    def __init__(self, entries, relu=False):
        """entries: list of (A [M,K], B [K,N], C [M,N], bias|None) fp32 device tensors (unit inner stride)."""
        dev = require_gpu()
        self.keep = entries
        arr = (_lib.GemmJob * len(entries))()
        self.max_m = self.max_n = self.max_k = 0
        self.flops = 0
        self.flags = GEMM_A_VEC4  # cleared by the first job whose A is not 16-byte aligned with lda % 4 == K % 4 == 0
        for job, (a, b, c, bias) in zip(arr, entries):
            m, k = a.shape
            n = b.shape[1]
            if b.shape[0] != k or tuple(c.shape) != (m, n) or any(t.stride(1) != 1 or t.dtype != torch.float32 for t in (a, b, c)):
                raise ValueError("GemmBatch: shape / layout mismatch")
            job.A, job.B, job.C = a.data_ptr(), b.data_ptr(), c.data_ptr()
            job.bias = 0 if bias is None else bias.data_ptr()
            job.lda, job.ldb, job.ldc = _ld(a), _ld(b), _ld(c)
            job.M, job.N, job.K, job.act = m, n, k, (ACT_RELU if relu else ACT_NONE)
            self.max_m, self.max_n, self.max_k = max(self.max_m, m), max(self.max_n, n), max(self.max_k, k)
            if a.data_ptr() % 16 or job.lda % 4 or k % 4:
                self.flags = 0
            self.flops += 2 * m * n * k
        self.n_jobs = len(entries)
        host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8) if len(entries) else torch.empty(0, dtype=torch.uint8)
        self.table = host.to(dev)

    def launch(self):
        check(lib.wdg_gemm_batched_flags_f32(_ptr(self.table), self.n_jobs, self.max_m, self.max_n, self.max_k, self.flags,
                                             stream_handle()), "wdg_gemm_batched_flags_f32")


class Mlp2Batch:
    """Job table for wdg_mlp2_batched_f32: Z_i = act(A_i W0_i + b0_i) W1_i + b1_i, one launch, one pass over A_i, the hidden
    layer never stored.  `eligible(entries)` says whether the fused kernel takes the shapes (else: two GemmBatch)."""

    MAX_H, MAX_C, MAX_K = 64, 8, 512

    @classmethod
    def eligible(cls, entries):
        for a, w0, b0, w1, b1, z in entries:
            k, h, c = a.shape[1], w0.shape[1], w1.shape[1]
            if h > cls.MAX_H or c > cls.MAX_C or k > cls.MAX_K or k % 4 or k == 0 or a.data_ptr() % 16 or _ld(a) % 4:
                return False
        return len(entries) > 0

    def __init__(self, entries, relu=True):
        """entries: list of (A [M,K], W0 [K,H], b0 [H]|None, W1 [H,C], b1 [C]|None, Z [M,C]) fp32 device tensors."""
        dev = require_gpu()
        if not self.eligible(entries):
            raise ValueError("Mlp2Batch: needs H <= 64, C <= 8, K <= 512, K % 4 == 0, 16-byte aligned rows of A")
        self.keep = entries
        arr = (_lib.Mlp2Job * len(entries))()
        self.max_m = self.max_k = self.max_h = self.max_c = 0
        self.flops = 0
        for job, (a, w0, b0, w1, b1, z) in zip(arr, entries):
            (m, k), h, c = a.shape, w0.shape[1], w1.shape[1]
            if w0.shape[0] != k or w1.shape[0] != h or tuple(z.shape) != (m, c) or \
                    any(t.stride(1) != 1 or t.dtype != torch.float32 for t in (a, w0, w1, z)):
                raise ValueError("Mlp2Batch: shape / layout mismatch")
            job.A, job.W0, job.W1, job.Z = a.data_ptr(), w0.data_ptr(), w1.data_ptr(), z.data_ptr()
            job.b0 = 0 if b0 is None else b0.data_ptr()
            job.b1 = 0 if b1 is None else b1.data_ptr()
            job.lda, job.ldw0, job.ldw1, job.ldz = _ld(a), _ld(w0), _ld(w1), _ld(z)
            job.M, job.K, job.H, job.C, job.act = m, k, h, c, (ACT_RELU if relu else ACT_NONE)
            self.max_m, self.max_k = max(self.max_m, m), max(self.max_k, k)
            self.max_h, self.max_c = max(self.max_h, h), max(self.max_c, c)
            self.flops += 2 * m * h * (k + c)
        self.n_jobs = len(entries)
        self.table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)

    def launch(self):
        check(lib.wdg_mlp2_batched_f32(_ptr(self.table), self.n_jobs, self.max_m, self.max_k, self.max_h, self.max_c,
                                       stream_handle()), "wdg_mlp2_batched_f32")


# ------------------------------------------------------------------------------------------- kernel-regression metric
class GramBatch:
    """Job table for wdg_gram_map_batched_f32: K = map(A A^T) of every A of a batch (all nodes), linear and / or arc-cosine."""

    def __init__(self, mats, linear=True, arccos=True):
        """mats: list of A [n, F] fp32 device tensors (unit inner stride) -> self.k_linear[i], self.k_arccos[i] ([n, n] or None)"""
        dev = require_gpu()
        self.keep = mats
        self.n_jobs = len(mats)
        self.max_n = max([a.shape[0] for a in mats], default=0)
        self.norm2 = [torch.empty(a.shape[0], dtype=torch.float32, device=dev) for a in mats]
        self.k_linear = [torch.empty((a.shape[0], a.shape[0]), dtype=torch.float32, device=dev) if linear else None for a in mats]
        self.k_arccos = [torch.empty((a.shape[0], a.shape[0]), dtype=torch.float32, device=dev) if arccos else None for a in mats]
        arr = (_lib.GramJob * self.n_jobs)()
        for job, a, n2, kl, ka in zip(arr, mats, self.norm2, self.k_linear, self.k_arccos):
            if a.dtype != torch.float32 or a.stride(1) != 1:
                raise ValueError("GramBatch: A must be fp32 with unit inner stride")
            job.A, job.norm2 = a.data_ptr(), n2.data_ptr()
            job.K_linear = 0 if kl is None else kl.data_ptr()
            job.K_arccos = 0 if ka is None else ka.data_ptr()
            job.lda, job.ldk, job.n, job.F = _ld(a), a.shape[0], a.shape[0], a.shape[1]
        self.table = _table(arr)

    def launch(self):
        check(lib.wdg_gram_map_batched_f32(_ptr(self.table), self.n_jobs, self.max_n, stream_handle()), "wdg_gram_map_batched_f32")


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


class KrSets:
    """Job table for wdg_kr_sample_sets: the (train, validation) node sets of every epoch of many (graph, classifier) pairs,
    drawn on the device in one launch (Philox4x32-10 keyed per pair; include/wdg.h documents the generator).
    self.train [pairs, epochs, n_train], self.val [pairs, epochs, n_val] int32, ascending ids; pairs whose graphs differ in
    class sizes are padded to the widest (self.n_train / self.n_val hold the true lengths)."""

    def __init__(self, entries, epochs):
        """entries: list of (labels int32 device [n], s_c, t_c (kr_split_sizes), seed int)"""
        dev = require_gpu()
        self.keep = entries
        self.n_pairs, self.epochs = len(entries), int(epochs)
        self.n_train = np.array([int(np.sum(t)) for _l, _s, t, _seed in entries], np.int64)
        self.n_val = np.array([int(np.sum(s_) - np.sum(t)) for _l, s_, t, _seed in entries], np.int64)
        self.train_stride, self.val_stride = int(self.n_train.max(initial=0)), int(self.n_val.max(initial=0))
        self.train = torch.zeros((self.n_pairs, self.epochs, max(self.train_stride, 1)), dtype=torch.int32, device=dev)
        self.val = torch.zeros((self.n_pairs, self.epochs, max(self.val_stride, 1)), dtype=torch.int32, device=dev)
        self.max_n = max([int(e[0].shape[0]) for e in entries], default=0)
        cls = np.concatenate([np.concatenate([np.asarray(s_, np.int32), np.asarray(t, np.int32)]) for _l, s_, t, _seed in entries]) \
            if entries else np.zeros(0, np.int32)
        self.class_tables = torch.from_numpy(cls).to(dev)
        arr = (_lib.KrSampleJob * self.n_pairs)()
        off = 0
        for i, (job, (lab, s_, t, seed)) in enumerate(zip(arr, entries)):
            c = len(s_)
            if c > 64:
                raise ValueError("KrSets: more than 64 classes")
            job.labels = lab.data_ptr()
            job.sample_per_class = self.class_tables.data_ptr() + 4 * off
            job.train_per_class = self.class_tables.data_ptr() + 4 * (off + c)
            off += 2 * c
            job.train_out, job.val_out = self.train[i].data_ptr(), self.val[i].data_ptr()
            job.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
            job.n, job.n_classes, job.n_sets, job.first_set = int(lab.shape[0]), c, self.epochs, i * self.epochs
            job.train_stride, job.val_stride = self.train.shape[2], self.val.shape[2]
        self.table = _table(arr)

    def launch(self):
        check(lib.wdg_kr_sample_sets(_ptr(self.table), self.n_pairs, self.n_pairs * self.epochs, self.max_n, stream_handle()),
              "wdg_kr_sample_sets")


_KR_JOB_DTYPE = np.dtype([("K", "<u8"), ("train", "<u8"), ("val", "<u8"), ("labels", "<u8"), ("correct_out", "<u8"), ("flags_out", "<u8"),
                          ("ldk", "<i8"), ("n_train", "<i4"), ("n_val", "<i4"), ("n_classes", "<i4"), ("reserved", "<i4")])
assert _KR_JOB_DTYPE.itemsize == ctypes.sizeof(_lib.KrJob)


class KrBatch:
    """Job table for wdg_kernel_regress_batched_f32: many (kernel, train rows, validation rows) problems in one launch."""

    MAX_TRAIN = 320
    MAX_CLASSES = 8  # KR_MAX_C of csrc/kernel_reg.hip: the right-hand sides a problem's workgroup carries

    def __init__(self, problems, n_classes):
        """problems: list of (K [n, n] fp32 device, train int32 device [nt], val int32 device [nv], labels int32 device [n])
        -> self.correct [n_problems] int32 after launch().  Shapes the solver does not hold (more than 8 classes, more than
        320 or fewer than 1 train rows) raise here: the kernel would answer them with the sentinel -1, and an accuracy of
        -1 / n_val fed to the t-test is a silently wrong p-value (callers with such label sets take the host path)."""
        self.keep = problems
        n = len(problems)
        col = lambda f: np.fromiter((f(p_) for p_ in problems), np.int64, n)  # noqa: E731
        self._build(col(lambda p_: p_[0].data_ptr()), col(lambda p_: _ld(p_[0])), col(lambda p_: p_[1].data_ptr()),
                    col(lambda p_: p_[2].data_ptr()), col(lambda p_: p_[3].data_ptr()), col(lambda p_: p_[1].shape[0]),
                    col(lambda p_: p_[2].shape[0]), n_classes)

    @classmethod
    def from_arrays(cls, k_ptr, ldk, train_ptr, val_ptr, labels_ptr, n_train, n_val, n_classes, keep=None):
        """the same table from per-problem numpy columns (device addresses and sizes): a sweep shard's 20 000 problems are
        described by arithmetic on a few base pointers, not by 20 000 tensor objects"""
        self = cls.__new__(cls)
        self.keep = keep
        self._build(*(np.asarray(a, np.int64) for a in (k_ptr, ldk, train_ptr, val_ptr, labels_ptr, n_train, n_val)), n_classes)
        return self

    def _build(self, k_ptr, ldk, train_ptr, val_ptr, labels_ptr, n_train, n_val, n_classes):
        dev = require_gpu()
        n = self.n_jobs = int(k_ptr.shape[0])
        if n and not 1 <= int(n_classes) <= self.MAX_CLASSES:
            raise ValueError(f"KrBatch: {n_classes} classes, the solver holds 1..{self.MAX_CLASSES}")
        if n and not (1 <= int(n_train.min()) and int(n_train.max()) <= self.MAX_TRAIN):
            raise ValueError(f"KrBatch: {int(n_train.min())}..{int(n_train.max())} train rows, the solver holds blocks of 1..{self.MAX_TRAIN}")
        self.correct = torch.zeros(max(n, 1), dtype=torch.int32, device=dev)
        self.flags = torch.zeros(max(n, 1), dtype=torch.int32, device=dev)  # bit 0: the ridge refactorisation ran
        self.n_val = torch.from_numpy(n_val.astype(np.float32)).to(dev)
        tab = np.zeros(n, _KR_JOB_DTYPE)
        tab["K"], tab["train"], tab["val"], tab["labels"] = k_ptr, train_ptr, val_ptr, labels_ptr
        tab["correct_out"] = self.correct.data_ptr() + 4 * np.arange(n, dtype=np.int64)
        tab["flags_out"] = self.flags.data_ptr() + 4 * np.arange(n, dtype=np.int64)
        tab["ldk"], tab["n_train"], tab["n_val"], tab["n_classes"] = ldk, n_train, n_val, int(n_classes)
        # (timing-only diagnostics of the blocked solver: honoured only by a library built with -DWDG_KR_ABLATION, and never
        # mistaken for a result - accuracy() refuses)
        self.ablate = int(os.environ.get("WDG_KR_ABLATE", "0"))
        tab["reserved"] = self.ablate
        self.table = torch.from_numpy(tab.view(np.uint8)).to(dev) if n else torch.empty(0, dtype=torch.uint8)

    def launch(self):
        check(lib.wdg_kernel_regress_batched_f32(_ptr(self.table), self.n_jobs, stream_handle()), "wdg_kernel_regress_batched_f32")

    def ridged(self):
        """[n_problems] bool: the train block was rank deficient in fp32 and was solved with the rounding-level ridge"""
        return (self.flags[:self.n_jobs] & 1).bool()

    def accuracy(self):
        """[n_problems] fp32 hit rate on the validation rows; raises when the kernel refused a problem (sentinel -1)"""
        if getattr(self, "ablate", 0):
            raise _lib.WdgError("KrBatch: WDG_KR_ABLATE is set - the launch was a timing-only ablation, its accuracies mean nothing")
        correct = self.correct[:self.n_jobs]
        if self.n_jobs and bool((correct < 0).any().item()):
            raise _lib.WdgError("wdg_kernel_regress_batched_f32 refused a problem (shape outside the solver's limits)")
        return correct.to(torch.float32) / self.n_val


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


# ------------------------------------------------------------------------------------------- GEMM
def gemm(a, b, bias=None, relu=False, transb=False, out=None):
    """act(A @ B + bias) (or A @ B^T with transb) in exact fp32 on the MFMA pipe (wdg_gemm_f32; products with few output tiles
    and K >= 1024 as partial products over ranges of K, wdg_gemm_splitk_f32: same arithmetic per range, ranges added in order)."""
    dev = require_gpu()
    a, b, bias = _dev(a, torch.float32, dev), _dev(b, torch.float32, dev), _dev(bias, torch.float32, dev)
    m, k = a.shape
    n = b.shape[0] if transb else b.shape[1]
    if (b.shape[1] if transb else b.shape[0]) != k:
        raise ValueError("gemm: inner dimensions differ")
    c = out if out is not None else torch.empty((m, n), dtype=torch.float32, device=dev)
    splits = 1 if transb else int(lib.wdg_gemm_splitk_plan(m, n, k))
    if splits > 1:  # few output tiles, a long K (a GCN's first layer on one wide-feature graph): partial products over ranges of K
        ws_bytes = lib.wdg_gemm_splitk_workspace_bytes(m, n, splits)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        check(lib.wdg_gemm_splitk_f32(_ptr(a), _ld(a), _ptr(b), _ld(b), _ptr(bias), ACT_RELU if relu else ACT_NONE, _ptr(c), _ld(c),
                                      m, n, k, splits, _ptr(ws), ws_bytes, stream_handle()), "wdg_gemm_splitk_f32")
        return c
    check(lib.wdg_gemm_f32(_ptr(a), _ld(a), _ptr(b), _ld(b), int(transb), _ptr(bias),
                           ACT_RELU if relu else ACT_NONE, _ptr(c), _ld(c), m, n, k, stream_handle()),
          "wdg_gemm_f32")
    return c


def gemm_skinny(a, b, bias=None, relu=False, out=None):
    """act(A @ B + bias) for B of <= 8 columns (a classifier head) on wdg_gemm_skinny_f32: the rows of A spread over the whole
    chip, K split over 16 lanes per row.  Within fp32 rounding of gemm() (whose k-ordered chain it does not reproduce bit for
    bit); wider B: gemm()."""
    dev = require_gpu()
    a, b, bias = _dev(a, torch.float32, dev), _dev(b, torch.float32, dev), _dev(bias, torch.float32, dev)
    m, k = a.shape
    n = b.shape[1]
    if b.shape[0] != k:
        raise ValueError("gemm_skinny: inner dimensions differ")
    if n > 8:
        return gemm(a, b, bias=bias, relu=relu, out=out)
    c = out if out is not None else torch.empty((m, n), dtype=torch.float32, device=dev)
    check(lib.wdg_gemm_skinny_f32(_ptr(a), _ld(a), _ptr(b), _ld(b), _ptr(bias), ACT_RELU if relu else ACT_NONE, _ptr(c), _ld(c),
                                  m, n, k, stream_handle()), "wdg_gemm_skinny_f32")
    return c
