"""wdg_amd - MI355X (gfx950) native aggregation / homophily-metric hot path of SitaoLuan/When-Do-GNNs-Help.

Layout (only what the hot path needs):
  csrc/    hand-written HIP kernels + the C ABI of include/wdg.h  -> lib/libwdg_hip.so
  _lib.py  ctypes binding (fails loudly when the library is missing; no CPU fallback)
  ops.py   torch-tensor front end, one namespace: CsrGraph, spmm, edge_label_stats, las, gemm, batched job tables - the code in
           graphs.py / aggregate.py / stats.py / gemm.py / kernel_regression.py (+ _rt.py: flags, pointers, the upload arena)
  utils/   drop-in twins of the reference's utils.util_funcs / utils.homophily_metrics / utils.homophily_plot
"""
__version__ = "0.1.0"

import os as _os_env

# (see _lib.py: a sweep's eight concurrent HIP streams on eight hardware queues instead of the default four; only when unset, and only
# effective if nothing has initialised the HIP runtime yet)
_os_env.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
del _os_env
