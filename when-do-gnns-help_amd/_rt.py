"""Plumbing shared by the front-end modules (graphs / aggregate / stats / kernel_regression / gemm): the flag values of
include/wdg.h, pointer and leading-dimension helpers, and the page-locked arena every small host-to-device copy goes through.
PyTorch is plumbing here (device memory, the current HIP stream); nothing in this package computes on the CPU or falls back."""
__all__ = ["COO_SYMMETRISE", "COO_BINARISE", "COO_ADD_SELF_LOOPS", "COO_DROP_SELF_LOOPS", "COO_KEEP_DUPLICATES", "NORM_RW", "NORM_SYM",
           "PREC_F32", "PREC_F64", "ACT_NONE", "ACT_RELU", "SPMM_ANY_VAL", "SPMM_DMA_OK", "SPMM_SMALL_OFFSETS", "SPMM_ANY_COL_SCALE",
           "SPMM_HALF_SLAB", "GEMM_A_VEC4", "Tiled", "Transposed"]  # what `from ._rt import *` hands the front-end modules: the flag values only

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
    if isinstance(t, (Tiled, Transposed)):
        return t.ld
    return t.stride(0) if t.shape[0] > 1 else max(int(t.stride(0)), int(t.shape[1]))


class _PinnedArena:
    """One page-locked buffer per process, handed out as a ring: a slice is reused only after EVERY copy that read any part of it
    has completed (an event per copy).  torch's own pinned allocator cannot reuse a block while its copy is queued behind
    kernels, and every NEW pinned block is a hipHostMalloc - measured: an occasional 90 ms in the middle of a shard's table
    uploads.

    Slices are handed out in ring order, so positions are kept on a VIRTUAL axis that never wraps (virtual = laps x size + offset;
    a request that does not fit the tail skips it and starts the next lap): `pending` = (virtual start, event) of the copies
    that may still be in flight, in issue order = ascending virtual start.  A new slice [va, vb) lies one lap ahead of exactly
    the entries with start + size < vb, and those are a PREFIX of the list: `take` waits for that prefix and drops it - O(1)
    amortised, no event queries.  (Round 4 tested only whether the OLDEST entry overlaps physically: after a wrap the oldest
    entry lies at the end of the buffer, does not overlap the new low-offset slices, and younger entries were overwritten while
    their copies were still queued.  The first repair scanned and queried every pending event on every request: ~10 000 event
    queries per base-shard, 0.77 -> 1.9 s for the whole sweep.)"""

    def __init__(self, nbytes=128 << 20, buf=None, new_event=None):
        self.buf = torch.empty(nbytes, dtype=torch.uint8).pin_memory() if buf is None else buf
        self.size, self.voff, self.pending = nbytes, 0, []  # voff: virtual position of the next slice; pending: (virtual start, event)
        self._new_event = new_event or self._recorded_event

    @staticmethod
    def _recorded_event():
        ev = torch.cuda.Event()
        ev.record()
        return ev

    def take(self, nbytes):
        """-> (handle, offset, the slice).  The slice's entry joins `pending` HERE, in ring order, with no event yet: `issued(handle)`
        supplies it once the copy is queued - so two threads may fill and queue their slices at the same time (a lock around take
        and around issued only) and `pending` still ascends.  A later request that laps an entry whose copy has not even been
        queued waits for it (a full lap of the ring between take and issued: never in practice)."""
        n = (nbytes + 255) & ~255
        if n > self.size:
            raise ValueError(f"_PinnedArena: {nbytes} bytes do not fit the {self.size}-byte ring")
        if self.voff % self.size + n > self.size:
            self.voff = (self.voff // self.size + 1) * self.size  # the tail is skipped: next lap
        va, vb = self.voff, self.voff + n
        done = 0
        while done < len(self.pending) and self.pending[done][0] + self.size < vb:
            entry = self.pending[done]
            if entry[1] is None:  # (taken by another thread, its copy not queued yet)
                import time
                deadline = time.monotonic() + self.STALL_S
                while entry[1] is None:
                    if time.monotonic() > deadline:
                        # a slice is only ever left unstamped by a bug (abandon() stamps it when the fill or the copy raises): fail
                        # instead of spinning with the ring's lock held, which would hang every later upload of the process
                        raise RuntimeError("_PinnedArena: a slice a full lap behind was taken but its copy was never queued")
                    time.sleep(0)
            if entry[1] is not self.ABANDONED:
                entry[1].synchronize()  # a lap behind the new slice's end: its bytes are about to be overwritten
            done += 1
        if done:
            del self.pending[:done]
        self.voff = vb
        a = va % self.size
        entry = [va, None]
        self.pending.append(entry)
        return entry, a, self.buf[a:a + nbytes]

    def issued(self, handle):
        handle[1] = self._new_event()

    def abandon(self, handle):
        """the fill or the copy of a taken slice failed: no copy will read its bytes, a later lap may overwrite them at once"""
        handle[1] = self.ABANDONED

    ABANDONED = object()
    STALL_S = 30.0


_ARENA = None
_ARENA_LOCK = __import__("threading").Lock()
_H2D_MAX_BYTES = int(float(os.environ.get("WDG_H2D_MAX_MB", "64")) * (1 << 20))  # larger arrays: the plain (blocking, pageable) copy
_H2D_RING_BYTES = int(float(os.environ.get("WDG_H2D_RING_MB", "512")) * (1 << 20))  # two base-shards of the widest base in flight: 2 x 3 x 30 MB
_H2D_THREADS = int(os.environ.get("WDG_H2D_THREADS", "8"))


def _arena_take(nbytes):
    """-> (arena, handle, uint8 slice) of the process's page-locked upload ring (created on first use, under the lock: two threads'
    first uploads must not pin a ring each).  The caller fills the slice, queues its copy and calls arena.issued(handle) - or
    arena.abandon(handle) if either failed."""
    global _ARENA
    with _ARENA_LOCK:  # (sweep.run_shards uploads from a helper thread as well: the ring's bookkeeping under a lock, the copies not)
        if _ARENA is None:
            _ARENA = _PinnedArena(_H2D_RING_BYTES)
        arena = _ARENA
        handle, _start, piece = arena.take(nbytes)
    return arena, handle, piece


def _h2d(host, dev=None):
    """A host array (job table, offsets, labels, a feature matrix) -> device tensor WITHOUT blocking the host on what the stream
    has queued: through the page-locked arena and a non-blocking copy.  (`tensor.to(dev)` from pageable memory returns only when
    the copy has run, i.e. after every kernel queued before it - a shard's ~70 small uploads then serialise the host with the
    build kernels.)  Arrays of a megabyte and more are copied into the arena by the library's host threads (wdg_host_memcpy_mt:
    round 4 sent everything above 8 MB down the blocking pageable path because ONE thread's memcpy of the wide bases' 30-MB
    feature matrices cost more than it saved - and then every such upload waited for the stream's queued regressions: 5 ms a
    piece in the profile of a rank's share of the whole sweep); only arrays above WDG_H2D_MAX_MB (64) take the plain copy."""
    dev = dev or require_gpu()
    t = torch.from_numpy(host) if isinstance(host, np.ndarray) else host
    t = t.contiguous()
    nbytes = t.numel() * t.element_size()
    if nbytes == 0:
        return torch.empty(t.shape, dtype=t.dtype, device=dev)
    if nbytes > _H2D_MAX_BYTES:
        return t.to(dev)
    arena, handle, piece = _arena_take(nbytes)
    try:
        p = piece.view(t.dtype).view(t.shape)
        # (numpy's memcpy / the library's threads, not Tensor.copy_: torch parallelises a host copy of a few MB over every hardware
        # thread of the host - measured 90 - 180 ms of thread wake-up on a 128-thread box for a 4-MB feature matrix)
        if nbytes >= (1 << 20) and _H2D_THREADS > 1:
            # (a thread per 2 MB, at most WDG_H2D_THREADS: starting a thread costs ~30 us, a megabyte of memcpy ~100)
            check(lib.wdg_host_memcpy_mt(c_void_p(piece.data_ptr()), c_void_p(t.data_ptr()), nbytes, max(1, min(_H2D_THREADS, nbytes >> 21))),
                  "wdg_host_memcpy_mt")
        else:
            np.copyto(p.numpy(), t.numpy())
        out = p.to(dev, non_blocking=True)
    except BaseException:
        arena.abandon(handle)  # (an entry left unstamped would hang the request that laps it - with the lock held)
        raise
    arena.issued(handle)  # (no lock: one assignment into the slice's own entry - a take() that waits for it holds the lock)
    return out


class Tiled:
    """A [rows, cols] fp32 matrix stored in 16-column groups: element (r, c) lives at t[c // 16, r, c % 16] of a device tensor
    t [groups, rows, 16] (wdg_spmm_job.y_group_stride / wdg_mlp2_job.a_group_stride of include/wdg.h).  The quad-row aggregation
    writes such a Y - one workgroup (= one feature group) stores inside one contiguous plane instead of 64-byte pieces a whole
    row apart - and the fused transform reads it back; everything else takes `rowmajor()` (a copy)."""

    def __init__(self, t, cols=None):
        if t.dim() != 3 or t.shape[2] != 16 or t.stride(2) != 1 or t.dtype != torch.float32:
            raise ValueError("Tiled: a [groups, rows, 16] fp32 tensor expected")
        self.t, self.cols = t, int(t.shape[0] * 16 if cols is None else cols)
        if not 0 <= self.cols <= t.shape[0] * 16:
            raise ValueError("Tiled: more columns than the groups hold")

    shape = property(lambda self: (int(self.t.shape[1]), self.cols))
    dtype = property(lambda self: self.t.dtype)
    device = property(lambda self: self.t.device)
    ld = property(lambda self: int(self.t.stride(1)))             # floats between consecutive rows inside a group
    group_stride = property(lambda self: int(self.t.stride(0)))   # floats between consecutive groups

    def data_ptr(self):
        return self.t.data_ptr()

    def stride(self, dim):
        return {0: self.ld, 1: 1}[dim]

    def columns(self, cols):
        """the first `cols` columns (a view)"""
        return Tiled(self.t, cols)

    def rowmajor(self):
        """-> a row-major [rows, cols] COPY"""
        g, n, _ = self.t.shape
        out = self.t.permute(1, 0, 2).reshape(n, g * 16)
        if out.data_ptr() == self.t.data_ptr():  # (a single group: reshape returned a view of the tiled storage itself)
            out = out.clone()
        return out[:, :self.cols]

    def fill_(self, v):
        self.t.fill_(v)
        return self


class Transposed:
    """The X operand of an aggregation given TRANSPOSED: `t` is [F, n] (fp32 / bf16, unit inner stride) and stands for the [n, F]
    matrix t^T (WDG_SELL16_X_TRANSPOSED of include/wdg.h; the quad-row kernel only: SpmmBatch checks)."""

    def __init__(self, t):
        if t.dim() != 2 or t.stride(1) != 1:
            raise ValueError("Transposed: a [F, n] tensor with unit inner stride expected")
        self.t = t

    shape = property(lambda self: (int(self.t.shape[1]), int(self.t.shape[0])))
    dtype = property(lambda self: self.t.dtype)
    device = property(lambda self: self.t.device)
    ld = property(lambda self: int(self.t.stride(0)) if self.t.shape[0] > 1 else max(int(self.t.stride(0)), int(self.t.shape[1])))

    def data_ptr(self):
        return self.t.data_ptr()

    def stride(self, dim):
        return {0: 1, 1: 1}[dim]  # (what callers test is the inner stride of the stored rows)


def _table(arr):
    """ctypes array of job descriptors -> device bytes (an empty table stays a host tensor: its pointer is NULL)"""
    return _h2d(torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)) if len(arr) else torch.empty(0, dtype=torch.uint8)


# flags of wdg_spmm_batched_f32 / wdg_spmm_quad_batched_f32 and of the GEMM tables (include/wdg.h)
SPMM_ANY_VAL, SPMM_DMA_OK, SPMM_SMALL_OFFSETS, SPMM_ANY_COL_SCALE, SPMM_HALF_SLAB = 2, 4, 8, 16, 32
GEMM_A_VEC4 = 1
