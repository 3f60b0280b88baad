"""The feature-transform products X W (reference: the nn.Linear / torch.mm of the models): single calls and job tables for a
sweep shard, on the hand-written MFMA kernels of csrc/gemm.hip."""
import ctypes
import os

import numpy as np
import torch

from . import _lib
from ._lib import c_void_p, check, lib, require_gpu, stream_handle
from ._rt import *  # noqa: F401,F403  (the flag values of include/wdg.h)
from ._rt import _dev, _h2d, _ld, _ptr, _table


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
        self.table = _h2d(host, dev) if len(entries) else host

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
            lda = a.ld if isinstance(a, Tiled) else _ld(a)
            if h > cls.MAX_H or c > cls.MAX_C or k > cls.MAX_K or k % 4 or k == 0 or a.data_ptr() % 16 or lda % 4:
                return False
            if isinstance(a, Tiled) and (a.group_stride % 4 or not cls.split_kernel()):
                return False  # (A tiled by 16-column groups: the split-operand kernel reads it, the fp32 chain does not)
        return len(entries) > 0

    @staticmethod
    def split_kernel():
        """the launcher's own rule (csrc/gemm.hip, wdg_mlp2_batched_f32): split-operand products unless WDG_MLP2_SPLIT=0"""
        e = os.environ.get("WDG_MLP2_SPLIT")
        try:
            return True if e is None else int(e) != 0
        except ValueError:
            return False

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
                    any(t.stride(1) != 1 or t.dtype != torch.float32 for t in (w0, w1, z) + (() if isinstance(a, Tiled) else (a,))):
                raise ValueError("Mlp2Batch: shape / layout mismatch")
            job.A, job.W0, job.W1, job.Z = a.data_ptr(), w0.data_ptr(), w1.data_ptr(), z.data_ptr()
            job.b0 = 0 if b0 is None else b0.data_ptr()
            job.b1 = 0 if b1 is None else b1.data_ptr()
            job.lda, job.ldw0, job.ldw1, job.ldz = (a.ld if isinstance(a, Tiled) else _ld(a)), _ld(w0), _ld(w1), _ld(z)
            job.a_group_stride = a.group_stride if isinstance(a, Tiled) else 0
            job.M, job.K, job.H, job.C, job.act = m, k, h, c, (ACT_RELU if relu else ACT_NONE)
            self.max_m, self.max_k = max(self.max_m, m), max(self.max_k, k)
            self.max_h, self.max_c = max(self.max_h, h), max(self.max_c, c)
            self.flops += 2 * m * h * (k + c)
        self.n_jobs = len(entries)
        self.table = _h2d(torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8), dev)
        # the kernel is chosen HERE, with the table, and named at every launch (1 = WDG_KERNEL_SPLIT, 2 = WDG_KERNEL_CHAIN,
        # 4 = WDG_OPERAND_TILED): the environment may change between build and launch, the table's layout does not
        self.flags = (1 if self.split_kernel() else 2) | (4 if any(isinstance(e[0], Tiled) for e in entries) else 0)

    def launch(self):
        check(lib.wdg_mlp2_batched_flags_f32(_ptr(self.table), self.n_jobs, self.max_m, self.max_k, self.max_h, self.max_c, self.flags,
                                             stream_handle()), "wdg_mlp2_batched_flags_f32")


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
