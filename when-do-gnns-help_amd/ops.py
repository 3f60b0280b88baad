"""Torch-tensor front end of the C ABI (include/wdg.h): device CSR container + one Python function per kernel.

PyTorch is plumbing here (device memory, the current HIP stream); every computation below is a hand-written
gfx950 kernel in csrc/.  Nothing here runs on the CPU, and nothing falls back.

This module is the one namespace callers use (`from wdg_amd import ops`); the code lives in
  _rt.py                flag values of include/wdg.h, pointer helpers, the page-locked upload arena
  graphs.py             CsrGraph (+ its one-time plans), GraphBatch (a shard's graphs in one build), normalisations
  aggregate.py          spmm, SpmmBatch (the quad-row kernel's tape and its cost cut), spmm_plan
  stats.py              edge / label statistics, LAS, per-edge cosine, their job tables
  gemm.py               gemm, gemm_skinny, GemmBatch, Mlp2Batch
  kernel_regression.py  GramBatch, PropagatedGram, RowRepBatch, EdgeGramBatch, KrSets, KrBatch, GnbBatch
(module-level switches - aggregate.ABLATE_BITS, aggregate.NARROW_MIN_ENTRIES - are set on the module that owns them)."""
from ._lib import check, lib, require_gpu, stream_handle  # noqa: F401
from ._rt import (  # noqa: F401
    ACT_NONE, ACT_RELU, COO_ADD_SELF_LOOPS, COO_BINARISE, COO_DROP_SELF_LOOPS, COO_KEEP_DUPLICATES, COO_SYMMETRISE,
    GEMM_A_VEC4, NORM_RW, NORM_SYM, PREC_F32, PREC_F64, SPMM_ANY_COL_SCALE, SPMM_ANY_VAL, SPMM_DMA_OK,
    SPMM_HALF_SLAB, SPMM_SMALL_OFFSETS, Tiled, Transposed, _dev, _h2d, _H2D_MAX_BYTES, _ld, _PinnedArena, _ptr, _table,
)
from .graphs import (  # noqa: F401
    CsrGraph, degree_norm, GraphBatch, normalise_values, quad_disabled, row_l1_normalise, unpack_bits,
    _host_pack_coo,
)
from .aggregate import (  # noqa: F401
    QUAD_MULTI_ITEM_SU, spmm, spmm_plan, SpmmBatch, _dma_ok, _fill_job, _QUAD_COST_NS,
    _QUAD_COST_W, _quad_cut, _quad_segments, _quad_unit_cost, _quad_unit_costs, _sharing_groups,
)
from .stats import (  # noqa: F401
    edge_cosine, edge_label_stats, las, LasBatch, StatsBatch,
)
from .gemm import (  # noqa: F401
    gemm, gemm_skinny, GemmBatch, Mlp2Batch,
)
from .kernel_regression import (  # noqa: F401
    deflation_enabled, EdgeGramBatch, GnbBatch, GramBatch, kr_split_sizes, KrBatch, KrSets, PropagatedGram, RowRepBatch, _KR_JOB_DTYPE,
)
