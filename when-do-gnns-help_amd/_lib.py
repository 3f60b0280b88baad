"""ctypes binding of libwdg_hip.so (the C ABI declared in include/wdg.h).

The product path has NO fallback: if the shared library is missing this module raises at import, and
every op raises if no HIP device is present.  PyTorch is imported first so that its bundled
libamdhip64.so (same SONAME as /opt/rocm's) is the one HIP runtime in the process: device pointers
and stream handles of torch tensors are then valid arguments for the kernels.
"""
import ctypes
import os

# HIP maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and streams that share a queue serialise.  A
# sweep keeps two pipelined base-shards, their side streams and the next shard's graph build in flight - 8 streams; on 4 queues the
# reference's whole sweep took 0.40 s instead of 0.30 (round 5: the same pass ran at 0.30 in a process that had created fewer
# streams before it).  Only a default, and only if the HIP runtime has not read it yet (it does at its first call, not at import).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch  # noqa: E402,F401  (loads libamdhip64 before our library needs it)

_HERE = os.path.dirname(os.path.abspath(__file__))
# (WDG_LIB_PATH: another build of the same library - the A/B scripts under scripts/dev/ compare kernel variants with it)
LIB_PATH = os.environ.get("WDG_LIB_PATH") or os.path.join(_HERE, "lib", "libwdg_hip.so")

c_void_p, c_int, c_int32, c_int64, c_size_t = (ctypes.c_void_p, ctypes.c_int, ctypes.c_int32, ctypes.c_int64,
                                               ctypes.c_size_t)
c_uint32 = ctypes.c_uint32


class WdgError(RuntimeError):
    pass


class SpmmJob(ctypes.Structure):
    """mirror of `wdg_spmm_job` (include/wdg.h)"""
    _fields_ = [("rowptr", c_void_p), ("col", c_void_p), ("val", c_void_p), ("row_scale", c_void_p),
                ("col_scale", c_void_p), ("X", c_void_p), ("Y", c_void_p), ("ldx", c_int64), ("ldy", c_int64),
                ("n_rows", c_int32), ("n_cols", c_int32), ("n_feat", c_int32), ("reserved", c_int32),
                ("q_ext", c_void_p), ("q_col", c_void_p), ("q_val", c_void_p), ("q_perm", c_void_p), ("q_rows", c_void_p),
                ("q_block_cols", c_int32), ("q_n_blocks", c_int32), ("q_n_entries", c_int32), ("q_flags", c_int32),
                ("band_perm", c_void_p), ("band_cuts", c_void_p), ("band_n_hub", c_int32), ("band_reserved", c_int32),
                ("y_group_stride", c_int64)]


class SpmmItem(ctypes.Structure):
    """mirror of `wdg_spmm_item` (include/wdg.h)"""
    _fields_ = [("first_job", c_int32), ("n_jobs", c_int32), ("unit_begin", c_int32), ("unit_end", c_int32),
                ("flags", c_int32), ("reserved", c_int32)]


class StatsJob(ctypes.Structure):
    """mirror of `wdg_stats_job` (include/wdg.h)"""
    _fields_ = [("rowptr", c_void_p), ("col", c_void_p), ("labels", c_void_p), ("totals", c_void_p),
                ("compat", c_void_p), ("classdeg", c_void_p), ("row_nnz", c_void_p), ("row_nnz_noself", c_void_p),
                ("row_match_noself", c_void_p), ("n_rows", c_int32), ("n_classes", c_int32)]


# name -> (restype, argtypes); must list every function include/wdg.h declares (tests/test_abi.py checks)
SIGNATURES = {
    "wdg_version": (c_int, []),
    "wdg_last_error": (ctypes.c_char_p, []),
    "wdg_device_cus": (c_int, []),
    "wdg_coo_to_csr_capacity": (c_int64, [c_int64, c_int32, c_int]),
    "wdg_coo_to_csr_workspace_bytes": (c_size_t, [c_int64, c_int32, c_int]),
    "wdg_coo_to_csr_i32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "wdg_coo32_to_csr_i32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "wdg_host_memcpy_mt": (c_int, [c_void_p, c_void_p, c_size_t, c_int]),
    "wdg_host_pack_coo_i32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int, c_void_p, c_void_p, c_void_p, c_int]),
    "wdg_coo_blockdiag_offset": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int64, c_void_p, c_void_p]),
    "wdg_csr_split_blockdiag": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "wdg_csr_to_sell16_count_batched": (c_int, [c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "wdg_csr_to_sell16_fill_batched": (c_int, [c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "wdg_dense_to_csr_count": (c_int, [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_size_t, c_void_p]),
    "wdg_dense_to_csr_fill": (c_int, [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "wdg_scan_workspace_bytes": (c_size_t, [c_int64]),
    "wdg_degree_norm": (c_int, [c_void_p, c_void_p, c_int32, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                c_void_p]),
    "wdg_normalise_values": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int, c_int, c_void_p, c_void_p,
                                     c_void_p, c_void_p]),
    "wdg_row_l1_normalise_f32": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int32, c_int32, c_int, c_void_p]),
    "wdg_unpack_bits_f32": (c_int, [c_void_p, c_int64, c_int32, c_int32, c_int, c_void_p, c_int64, c_void_p]),
    "wdg_spmm_csr_f32": (c_int, [ctypes.POINTER(SpmmJob), c_void_p]),
    "wdg_spmm_csr_bf16": (c_int, [ctypes.POINTER(SpmmJob), c_void_p]),
    "wdg_spmm_batched_f32": (c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_int, c_void_p]),
    "wdg_spmm_plan": (c_int, [c_int32, c_int32, c_int32, c_int32, c_int, ctypes.POINTER(c_int), ctypes.POINTER(c_int)]),
    "wdg_sell16_block_cols": (c_int32, [c_int32]),
    "wdg_sell16_row_bytes": (c_int32, [c_int32]),
    "wdg_sell16_workspace_bytes": (c_size_t, [c_int32, c_int32]),
    "wdg_sell16_max_entries": (c_int64, [c_int32]),
    "wdg_csr_to_sell16_count": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                        c_void_p]),
    "wdg_csr_to_sell16_fill": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_int32, c_void_p,
                                       c_void_p, c_void_p]),
    "wdg_csr_band_plan_workspace_bytes": (c_size_t, [c_int32]),
    "wdg_csr_band_perm_len": (c_int32, [c_int32]),
    "wdg_csr_band_plan": (c_int, [c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "wdg_csr_band_plan_hub": (c_int, [c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "wdg_spmm_narrow_col_bytes": (c_int32, [c_int32, c_int32, c_int32]),
    "wdg_spmm_narrow_parts": (c_int32, [c_int32, c_int32]),
    "wdg_spmm_narrow_workspace_bytes": (c_size_t, [c_int32, c_int32]),
    "wdg_spmm_narrow_plan": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "wdg_spmm_narrow_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "wdg_spmm_narrow_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "wdg_spmm_narrow_batched_f32": (c_int, [c_void_p, c_int32, c_int32, c_int32, c_int, c_void_p]),
    "wdg_spmm_quad_batched_clocked_f32": (c_int, [c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int, c_void_p,
                                                  c_void_p]),
    "wdg_spmm_quad_workgroups": (c_int32, [c_int32, c_int32, c_int]),
    "wdg_debug_clock": (c_int, [c_void_p, c_void_p]),
    "wdg_spmm_quad_batched_f32": (c_int, [c_void_p, c_int32, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int,
                                          c_void_p]),
    "wdg_edge_label_stats": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_void_p, c_void_p, c_void_p]),
    "wdg_edge_label_stats_batched": (c_int, [c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "wdg_sweep_scalars_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32,
                                      c_void_p, c_void_p]),
    "wdg_edge_cosine_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int32, c_int32, c_int,
                                    c_void_p, c_void_p]),
    "wdg_las_workspace_bytes": (c_size_t, [c_int32, c_int32, c_int32]),
    "wdg_las_f32": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p,
                            c_void_p, c_size_t, c_void_p]),
    "wdg_gemm_f32": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p, c_int, c_void_p, c_int64,
                             c_int32, c_int32, c_int32, c_void_p]),
    "wdg_gemm_skinny_f32": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int, c_void_p, c_int64, c_int32, c_int32, c_int32,
                                    c_void_p]),
    "wdg_gemm_splitk_plan": (c_int32, [c_int32, c_int32, c_int32]),
    "wdg_gemm_splitk_workspace_bytes": (c_size_t, [c_int32, c_int32, c_int32]),
    "wdg_gemm_splitk_f32": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int, c_void_p, c_int64, c_int32, c_int32, c_int32,
                                    c_int32, c_void_p, c_size_t, c_void_p]),
    "wdg_gemm_batched_f32": (c_int, [c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "wdg_gemm_batched_flags_f32": (c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_uint32, c_void_p]),
    "wdg_mlp2_batched_f32": (c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "wdg_mlp2_batched_flags_f32": (c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_uint32, c_void_p]),
    "wdg_las_batched_f32": (c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "wdg_las_fused_eligible": (c_int, [c_int32, c_int32, c_int32]),
    "wdg_gram_map_batched_f32": (c_int, [c_void_p, c_int32, c_int32, c_void_p]),
    "wdg_gram_map_batched_flags_f32": (c_int, [c_void_p, c_int32, c_int32, c_uint32, c_void_p]),
    "wdg_transpose_batched_f32": (c_int, [c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "wdg_gram_finish_batched_f32": (c_int, [c_void_p, c_int32, c_int32, c_void_p]),
    "wdg_kernel_regress_max_train": (c_int32, []),
    "wdg_edge_gram_workspace_bytes": (c_size_t, [c_int32, c_int32]),
    "wdg_edge_gram_mean_batched_f32": (c_int, [c_void_p, c_int32, c_int32, c_void_p, c_size_t, c_void_p]),
    "wdg_kernel_regress_batched_f32": (c_int, [c_void_p, c_int32, c_void_p]),
    "wdg_row_rep_batched": (c_int, [c_void_p, c_int32, c_int32, c_int32, c_void_p]),
    "wdg_gnb_workspace_bytes": (c_size_t, [c_int32, c_int32]),
    "wdg_gnb_batched_f32": (c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    "wdg_sweep_pack_f64": (c_int, [c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p]),
    "wdg_kr_deflate_workspace_bytes": (c_size_t, [c_int32]),
    "wdg_kernel_regress_deflated_batched_f32": (c_int, [c_void_p, c_int32, c_void_p]),
    "wdg_kr_sample_sets": (c_int, [c_void_p, c_int32, c_int32, c_int32, c_void_p]),
}


class Sell16Job(ctypes.Structure):
    """mirror of `wdg_sell16_job` (include/wdg.h)"""
    _fields_ = [("rowptr", c_void_p), ("col", c_void_p), ("val", c_void_p), ("q_perm", c_void_p), ("q_ext", c_void_p),
                ("q_rows", c_void_p), ("q_col", c_void_p), ("q_val", c_void_p), ("workspace", c_void_p),
                ("n_rows", c_int32), ("n_cols", c_int32)]


class GramJob(ctypes.Structure):
    """mirror of `wdg_gram_job` (include/wdg.h)"""
    _fields_ = [("A", c_void_p), ("norm2", c_void_p), ("K_linear", c_void_p), ("K_arccos", c_void_p), ("lda", c_int64),
                ("ldk", c_int64), ("n", c_int32), ("F", c_int32), ("a_group_stride", c_int64)]


class TransposeJob(ctypes.Structure):
    """mirror of `wdg_transpose_job` (include/wdg.h)"""
    _fields_ = [("src", c_void_p), ("dst", c_void_p), ("ld_src", c_int64), ("ld_dst", c_int64), ("rows", c_int32), ("cols", c_int32)]


class EdgeGramJob(ctypes.Structure):
    """mirror of `wdg_edge_gram_job` (include/wdg.h)"""
    _fields_ = [("rowptr", c_void_p), ("col", c_void_p), ("K_linear", c_void_p), ("norm2", c_void_p), ("mean_out", c_void_p),
                ("ldk", c_int64), ("n_rows", c_int32), ("reserved", c_int32)]


class KrJob(ctypes.Structure):
    """mirror of `wdg_kr_job` (include/wdg.h)"""
    _fields_ = [("K", c_void_p), ("train", c_void_p), ("val", c_void_p), ("labels", c_void_p), ("correct_out", c_void_p),
                ("flags_out", c_void_p), ("ldk", c_int64), ("n_train", c_int32), ("n_val", c_int32), ("n_classes", c_int32), ("reserved", c_int32),
                ("rep", c_void_p), ("ws", c_void_p)]


class RowRepJob(ctypes.Structure):
    """mirror of `wdg_row_rep_job` (include/wdg.h)"""
    _fields_ = [("A", c_void_p), ("rowptr", c_void_p), ("col", c_void_p), ("val", c_void_p), ("row_scale", c_void_p),
                ("rep_out", c_void_p), ("hash_ws", c_void_p), ("lda", c_int64), ("a_group_stride", c_int64), ("n", c_int32), ("F", c_int32)]


class GnbJob(ctypes.Structure):
    """mirror of `wdg_gnb_job` (include/wdg.h)"""
    _fields_ = [("X", c_void_p), ("train", c_void_p), ("val", c_void_p), ("labels", c_void_p), ("ws", c_void_p), ("correct", c_void_p),
                ("pred", c_void_p), ("ldx", c_int64), ("n_train", c_int32), ("n_val", c_int32), ("F", c_int32), ("n_classes", c_int32)]


class KrSampleJob(ctypes.Structure):
    """mirror of `wdg_kr_sample_job` (include/wdg.h)"""
    _fields_ = [("labels", c_void_p), ("sample_per_class", c_void_p), ("train_per_class", c_void_p), ("train_out", c_void_p),
                ("val_out", c_void_p), ("seed", ctypes.c_uint64), ("n", c_int32), ("n_classes", c_int32), ("n_sets", c_int32),
                ("first_set", c_int32), ("train_stride", c_int32), ("val_stride", c_int32)]


class LasJob(ctypes.Structure):
    """mirror of `wdg_las_job` (include/wdg.h)"""
    _fields_ = [("H", c_void_p), ("labels", c_void_p), ("rows", c_void_p), ("W_out", c_void_p), ("count_out", c_void_p),
                ("workspace", c_void_p), ("ldh", c_int64), ("n", c_int32), ("F", c_int32), ("C", c_int32),
                ("reserved", c_int32), ("counts", c_void_p), ("row_scale", c_void_p)]


class GemmJob(ctypes.Structure):
    """mirror of `wdg_gemm_job` (include/wdg.h)"""
    _fields_ = [("A", c_void_p), ("B", c_void_p), ("bias", c_void_p), ("C", c_void_p), ("lda", c_int64), ("ldb", c_int64),
                ("ldc", c_int64), ("M", c_int32), ("N", c_int32), ("K", c_int32), ("act", c_int32)]

class Mlp2Job(ctypes.Structure):
    """mirror of `wdg_mlp2_job` (include/wdg.h)"""
    _fields_ = [("A", c_void_p), ("W0", c_void_p), ("b0", c_void_p), ("W1", c_void_p), ("b1", c_void_p), ("Z", c_void_p),
                ("lda", c_int64), ("ldw0", c_int64), ("ldw1", c_int64), ("ldz", c_int64),
                ("M", c_int32), ("K", c_int32), ("H", c_int32), ("C", c_int32), ("act", c_int32), ("reserved", c_int32),
                ("a_group_stride", c_int64)]


if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: the HIP extension has not been built.  Run `make` at the repository root "
        "(or `python -c 'import __graft_entry__ as g; g.build()'`).  There is no CPU fallback.")

lib = ctypes.CDLL(LIB_PATH)
for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)  # AttributeError here = header / library mismatch
    _fn.restype, _fn.argtypes = _res, _args


def check(code, what):
    """Turn a negative return code of the C ABI into a Python exception (reference convention: errors raise)."""
    if code != 0:
        msg = lib.wdg_last_error().decode(errors="replace")
        if code == -1:
            raise ValueError(f"{what}: {msg}")
        raise WdgError(f"{what} failed with code {code}: {msg}")


def require_gpu():
    if not torch.cuda.is_available():
        raise WdgError("no HIP device visible: wdg_amd kernels only run on an MI355X (gfx950); there is no CPU path")
    return torch.device("cuda", torch.cuda.current_device())


def stream_handle():
    return c_void_p(torch.cuda.current_stream().cuda_stream)
