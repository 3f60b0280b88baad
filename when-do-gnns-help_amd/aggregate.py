"""The aggregation Y = diag(r) A diag(c) X (reference: every torch.spmm / torch.mm(adj, X)): single graphs (wdg_spmm_csr_*) and
job tables for a sweep shard (SpmmBatch: the quad-row kernel's tape of super-units cut into equal-cost segments)."""
import ctypes
import os

import numpy as np
import torch

from . import _lib
from ._lib import c_void_p, check, lib, require_gpu, stream_handle
from ._rt import *  # noqa: F401,F403  (the flag values of include/wdg.h)
from ._rt import _dev, _h2d, _ld, _ptr, _table
from ._lib import SpmmItem, SpmmJob
from .graphs import CsrGraph, quad_disabled

ABLATE_BITS = 0  # diagnostics: scripts/dev/ablate_*.py set this to wdg_spmm_job.reserved timing-ablation bits (results are wrong then)


# ------------------------------------------------------------------------------------------- aggregation
NARROW_MIN_ENTRIES = int(os.environ.get("WDG_NARROW_MIN_ENTRIES", 1 << 15))  # below: the general families (round 2: 2^18 - chameleon's
# 65 019 entries then took the gather kernel for its C = 5 logits aggregation: 85 us against the narrow kernel's 20)


def _fill_job(job, g, x, y, row_scale, col_scale, use_values=True, band=False):
    job.rowptr, job.col = g.rowptr.data_ptr(), g.col.data_ptr()
    job.val = g.val.data_ptr() if (use_values and g.val is not None and not g.unit_values) else 0
    job.row_scale = 0 if row_scale is None else row_scale.data_ptr()
    job.col_scale = 0 if col_scale is None else col_scale.data_ptr()
    job.X, job.Y = x.data_ptr(), y.data_ptr()
    if isinstance(y, Tiled):  # Y in 16-feature groups (the quad-row kernel only: SpmmBatch checks)
        if y.cols != x.shape[1] or y.shape[0] != g.n_rows:
            raise ValueError("spmm: a tiled Y must hold exactly the product's rows and columns")
        job.ldx, job.ldy, job.y_group_stride = _ld(x), y.ld, y.group_stride
    else:
        job.ldx, job.ldy, job.y_group_stride = _ld(x), _ld(y), 0
    job.n_rows, job.n_cols, job.n_feat = g.n_rows, g.n_cols, x.shape[1]
    job.reserved = ABLATE_BITS  # 0 on every product path; only scripts/ablate_*.py assign the module variable
    wants_val = bool(job.val)
    q = g.quad
    if q and (not wants_val or q["val"] is not None):
        job.q_ext, job.q_col, job.q_perm = q["ext"].data_ptr(), q["col"].data_ptr(), q["perm"].data_ptr()
        job.q_rows = q["rows"].data_ptr()
        job.q_val = q["val"].data_ptr() if (wants_val and q["val"] is not None) else 0
        job.q_block_cols, job.q_n_blocks = q["block_cols"], q["n_blocks"]
        job.q_n_entries, job.q_flags = q["n_entries"], (1 if q["split"] else 0) | (2 if q["half"] else 0) | (4 if isinstance(x, Transposed) else 0)
    else:
        job.q_ext = job.q_col = job.q_val = job.q_perm = job.q_rows = 0
        job.q_block_cols = job.q_n_blocks = job.q_n_entries = job.q_flags = 0
    if band and g.band:
        job.band_perm, job.band_cuts = g.band["perm"].data_ptr(), g.band["cuts"].data_ptr()
        job.band_n_hub = g.band.get("n_hub", -1)  # (-1 = WDG_BAND_HUB_ON_DEVICE: nobody has read cuts[8] back yet)
        # (the single-graph entry point prefers a split-form SELL-16 copy: this call asked for the band kernel)
        job.q_ext = job.q_col = job.q_val = job.q_perm = job.q_rows = 0
        job.q_block_cols = job.q_n_blocks = job.q_n_entries = job.q_flags = 0
    else:
        job.band_perm = job.band_cuts = 0
        job.band_n_hub = 0
    job.band_reserved = 0
    return job


def _y_span(y, n_rows):
    """floats between the first and the last element of Y (the pipelined loop addresses it with 32-bit byte offsets)"""
    return y.t.shape[0] * y.group_stride + n_rows * y.ld if isinstance(y, Tiled) else n_rows * _ld(y)


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
    # (the cut depends on the graphs, on which of them share an X and on the width - not on the operands' addresses: a sweep lays the
    # same shard's graphs out again for every feature base and for both products of the propagated kernels.  Remembered on the first
    # graph of the table, so the memory dies with the shard's graphs)
    runs, pos = [], 0
    while pos < len(order):
        end = pos + 1
        while end < len(order) and keys[end] == keys[pos]:
            end += 1
        runs.append(end - pos)
        pos = end
    memo_key = (tuple(id(entries[i][0]) for i in order), tuple(runs), n_feat, cus, phase_ns, None if shares is None else tuple(shares),
                os.environ.get("WDG_QUAD_SUBS"), os.environ.get("WDG_QUAD_PHASE_NS"), os.environ.get("WDG_QUAD_PHASE_ORDER"),
                os.environ.get("WDG_QUAD_ALPHA"), os.environ.get("WDG_QUAD_WMIN"))
    memo = entries[order[0]][0].__dict__.setdefault("_seg_memo", {}) if order else {}
    if memo_key in memo:
        return memo[memo_key]
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
    if not multi:
        # every segment split at the phase-group boundaries, all segments at once (a labels-only step has ONE feature group, so
        # 256 segments: the per-segment loop below was 0.8 ms of every fresh table - a third of a rank's start-up at 8 ranks)
        cuts_a = np.asarray(cuts, np.int64)
        edges = np.union1d(cuts_a, g_off[(g_off > cuts_a[0]) & (g_off < cuts_a[-1])])
        lo, hi = edges[:-1], edges[1:]
        seg_of = np.searchsorted(cuts_a, lo, side="right") - 1          # (the LAST segment starting at or before lo: empty segments own nothing)
        grp_of = np.searchsorted(g_off, lo, side="right") - 1
        cost = cum[hi] - cum[lo]
        if os.environ.get("WDG_QUAD_PHASE_ORDER", "1") != "0":           # (the phases of a segment run shortest first: see below)
            perm = np.lexsort((np.arange(len(lo)), cost, seg_of))
            lo, hi, seg_of, grp_of = lo[perm], hi[perm], seg_of[perm], grp_of[perm]
        first = np.array([g[0] for g in groups], np.int64)
        nj = np.array([g[1] for g in groups], np.int64)
        items = list(zip(first[grp_of].tolist(), nj[grp_of].tolist(), (lo - g_off[grp_of]).tolist(), (hi - g_off[grp_of]).tolist()))
        seg_ptr = np.concatenate([[0], np.cumsum(np.bincount(seg_of, minlength=n_seg))]).tolist()
        if len(memo) >= 16:
            memo.clear()
        memo[memo_key] = (items, seg_ptr, n_seg)
        return items, seg_ptr, n_seg
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
    if len(memo) >= 16:
        memo.clear()
    memo[memo_key] = (items, seg_ptr, n_seg)
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
        if any(isinstance(e[2], Tiled) for e in entries) and not self.quad:
            raise ValueError("SpmmBatch: a tiled Y needs the quad-row kernel (SELL-16 copies, F >= 8)")
        if any(isinstance(e[1], Transposed) for e in entries) and not self.quad:
            raise ValueError("SpmmBatch: a transposed X needs the quad-row kernel (SELL-16 copies, F >= 8)")
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
            if all(_y_span(e[2], e[0].n_rows) < (1 << 30) and e[0].quad["chunks"] < (1 << 22) - 2 and e[0].quad["split"] for e in entries):
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
            got = y.rowmajor() if isinstance(y, Tiled) else y.clone()
            ref = torch.empty((g.n_rows, x.shape[1]), dtype=torch.float32, device=y.device)
            xv = x.t.t().contiguous() if isinstance(x, Transposed) else x  # (the CSR kernels read row-major sources only)
            job = _fill_job(SpmmJob(), g, xv, ref, rs, cs, uv)
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

    n_launches = 0  # launches so far (readers that keep a copy of the outputs compare it with the count they copied at)

    def launch(self, clock=None):
        self.n_launches += 1
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


def spmm_plan(n_rows, n_cols, n_feat, n_jobs=1, flags=0):
    """(family, width, threads) of the CSR kernels behind wdg_spmm_batched_f32: 0 = LDS column slab, 1 = row gather."""
    slab, threads = ctypes.c_int(0), ctypes.c_int(0)
    fam = lib.wdg_spmm_plan(n_jobs, n_rows, n_cols, n_feat, flags, ctypes.byref(slab), ctypes.byref(threads))
    return fam, slab.value, threads.value
