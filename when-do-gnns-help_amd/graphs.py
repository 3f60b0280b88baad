"""Device-resident graphs: the CSR container with its one-time plans (SELL-16 copy, band plan, narrow pack), the batched build of a
sweep shard's graphs (one block-diagonal COO -> CSR build, one host read-back) and the degree normalisations.  Every function is
a call into the C ABI (include/wdg.h); nothing computes on the CPU."""
import ctypes
import os

import numpy as np
import torch

from . import _lib
from ._lib import c_void_p, check, lib, require_gpu, stream_handle
from ._rt import *  # noqa: F401,F403  (the flag values of include/wdg.h)
from ._rt import _arena_take, _dev, _h2d, _H2D_MAX_BYTES, _ld, _ptr, _table


def quad_disabled():
    """WDG_SPMM_NO_QUAD=1: keep every aggregation off the quad-row kernel (the CSR slab / gather kernels: A/B comparisons, tests)"""
    return os.environ.get("WDG_SPMM_NO_QUAD", "0") not in ("", "0")


class _BandPlan(dict):
    """perm / cuts of wdg_csr_band_plan; `["n_hub"]` reads cuts[8] back the first time somebody asks (a host decision such as
    `CsrGraph.prefers_band` on a graph of <= 2 528 columns) - the launches themselves pass WDG_BAND_HUB_ON_DEVICE until then"""

    def __missing__(self, key):
        if key not in ("n_hub", "n_long"):
            raise KeyError(key)
        both = self["cuts"][[8, 19]].tolist()  # (one read-back: the hub rows and the rows of more than 128 entries)
        self["n_hub"], self["n_long"] = int(both[0]), int(both[1])
        return self[key]


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
        """Build the band plan (wdg_csr_band_plan): rows by length, hub count, cost cuts.  One-time per graph; nothing is read
        back (the hub count stays in cuts[8]: `band["n_hub"]` fetches it on first use, the kernels read it on the device)."""
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
        # hub threshold: rows longer than this are swept by four waves.  A launch ends with its longest single-wave row: 6 x the mean
        # row length, 32 .. 192 (Cora - mean 4.9, longest 169 - 22.4 -> 17.5 us at 32; squirrel 186 -> 176 us at 192, non-monotone between:
        # 144: 181, 160: 190, 224: 184, 256: 185, 320: 192; chameleon 44 - 45 us from 128 to 192, 47 - 49 above)
        hub_len = int(min(192, max(32, 6 * self.nnz // max(self.n_rows, 1))))
        check(lib.wdg_csr_band_plan_hub(_ptr(self.rowptr), self.n_rows, hub_len, _ptr(perm), _ptr(cuts), _ptr(ws), ws_bytes, stream_handle()),
              "wdg_csr_band_plan_hub")
        self.band = _BandPlan(perm=perm, cuts=cuts, ws=ws, hub_len=hub_len)  # (ws: alive until the plan's kernels have run)
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
        return n_feat >= 64 and (self.n_cols > self.QUAD_SLAB_COLS or self.band["n_long"] > 0)  # (n_long: rows of more than 128 entries)

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


def _host_pack_coo(coos, lens, node_ptr_h, e_total, dev):
    """-> (src, dst) int32 device tensors holding the shard's edge lists as ids of the block-diagonal union, or None when the
    inputs are not plain contiguous host arrays of one integer width (the caller then takes the torch path).  Raises IndexError
    for an id outside its graph."""
    arrs = [(c[0], c[1]) for c in coos]
    for i, (a, b) in enumerate(arrs):  # (the library's host threads read lens[i] entries of BOTH arrays)
        if len(a) != len(b) or len(a) != lens[i]:
            raise ValueError(f"graph {i}: src and dst hold {len(a)} and {len(b)} entries")
    kinds = {a.dtype for pair in arrs for a in pair if isinstance(a, np.ndarray)}
    if len(kinds) != 1 or not all(isinstance(a, np.ndarray) and a.flags.c_contiguous and a.ndim == 1 for pair in arrs for a in pair):
        return None
    kind = kinds.pop()
    if kind not in (np.dtype(np.int64), np.dtype(np.int32)):
        return None
    G = len(coos)
    ptrs = ctypes.c_void_p * G
    sp, dp = ptrs(*[a.ctypes.data for a, _b in arrs]), ptrs(*[b.ctypes.data for _a, b in arrs])
    lens_a = np.asarray(lens, np.int64)
    nptr = np.ascontiguousarray(node_ptr_h, np.int32)
    bad = ctypes.c_int32(0)
    threads = int(os.environ.get("WDG_HOST_PACK_THREADS", "8"))
    nbytes = 8 * e_total
    # packed straight into a slice of the process's page-locked upload ring (round 6: the staging tensors this function used to pin
    # for itself cost a hipHostMalloc of 20 MB each inside a pipeline's first pass); arrays beyond the ring's share: a pinned tensor of
    # their own, as before
    if 0 < nbytes <= _H2D_MAX_BYTES:
        arena, handle, piece = _arena_take(nbytes)
        try:
            host = piece[:nbytes].view(torch.int32)
            base = host.data_ptr()
            check(lib.wdg_host_pack_coo_i32(sp, dp, lens_a.ctypes.data, nptr.ctypes.data, G, kind.itemsize, ctypes.c_void_p(base),
                                            ctypes.c_void_p(base + 4 * e_total), ctypes.byref(bad), threads), "wdg_host_pack_coo_i32")
            if bad.value:
                raise IndexError("edge index out of range for its graph")
            both = host.to(dev, non_blocking=True)
        except BaseException:
            arena.abandon(handle)
            raise
        arena.issued(handle)
        return both[:e_total], both[e_total:]
    stage = torch.empty(max(2 * e_total, 1), dtype=torch.int32).pin_memory()
    host = stage.numpy()
    check(lib.wdg_host_pack_coo_i32(sp, dp, lens_a.ctypes.data, nptr.ctypes.data, G, kind.itemsize, host[:e_total].ctypes.data,
                                    host[e_total:2 * e_total].ctypes.data, ctypes.byref(bad), threads), "wdg_host_pack_coo_i32")
    if bad.value:
        raise IndexError("edge index out of range for its graph")
    both = stage[:2 * e_total].to(dev)  # (blocking: the staging tensor dies with this call)
    return both[:e_total], both[e_total:]


class GraphBatch:
    """A sweep shard's graphs built together: ONE COO -> CSR build of their block-diagonal union (wdg_coo_blockdiag_offset,
    wdg_coo_to_csr_i32, wdg_csr_split_blockdiag), the SELL-16 copies of all of them in five more launches
    (wdg_csr_to_sell16_count_batched / _fill_batched) and ONE host read-back for the whole shard - against ~20 launches and two
    host syncs per graph through CsrGraph.from_coo + ensure_quad (the cold path of a one-pass sweep: synthetic_plot.py:78-109
    visits every graph once).  The results are bit for bit the per-graph builds' (tests/test_gpu_batched_build.py).

    .graphs: list of CsrGraph (views into the shard's pooled arrays; .quad set when `quad`); degree_norm(): every graph's
    degrees / coefficients from one launch over the union."""

    def __init__(self, coos, flags=0, quad=True, quad_values=False, max_padding=4.0, defer=False):
        """coos: list of (src, dst, n) or (src, dst, n, val): host arrays or tensors, node ids local to each graph.
        defer: return once the build's launches are queued (pack, upload, COO -> CSR, split, SELL-16 count) WITHOUT the shard's one
        read-back; the caller does other host work (feature / label uploads: ~2 ms per shard) while the GPU builds and then
        calls finish() - which no longer waits (round 5; the read-back used to idle the host for 0.9 ms per shard)."""
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
        info[0:1].copy_(bad[0:1])
        self._pending = dict(G=G, ns=ns, quad=quad, quad_values=quad_values, max_padding=max_padding, info=info, jobs=jobs, dev=dev,
                             node_ptr_h=node_ptr_h, st=st, keep=(src, dst, val, ws, node_ptr, edge_ptr, bad, table))
        if quad:
            self._pending.update(ext=ext, rows=rows, perm=perm, qws=qws, ext_off=ext_off, rows_off=rows_off, perm_off=perm_off,
                                 ext_len=ext_len, perm_len=perm_len, n_blocks=n_blocks)
        self.graphs = None
        if not defer:
            self.finish()

    def finish(self):
        """the second half of the build: the shard's ONE host read-back, the per-graph views, the SELL-16 fill"""
        if self._pending is None:
            return
        pd, self._pending = self._pending, None
        G, ns, quad, quad_values, max_padding, info, jobs, dev = (pd[k] for k in ("G", "ns", "quad", "quad_values", "max_padding", "info", "jobs", "dev"))
        node_ptr_h, st = pd["node_ptr_h"], pd["st"]
        if quad:
            ext, rows, perm, qws, ext_off, rows_off, perm_off, ext_len, perm_len, n_blocks = (
                pd[k] for k in ("ext", "rows", "perm", "qws", "ext_off", "rows_off", "perm_off", "ext_len", "perm_len", "n_blocks"))
        # ---- the shard's ONE host read-back: the info block and, behind it in the same buffer, the pool of extents (64 KB for 50
        #      graphs: the widths price the aggregation's tape cut, the tails size the index arrays) - one blocking copy
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
