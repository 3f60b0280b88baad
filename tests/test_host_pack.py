"""Host-side pieces of the cold (one-pass) sweep path that need no GPU: the library's COO packer and the vectorised tape-cut cost."""
import ctypes

import numpy as np
import pytest


@pytest.mark.parametrize("dtype", [np.int64, np.int32])
@pytest.mark.parametrize("threads", [1, 3, 8])
def test_host_pack_coo_builds_the_block_diagonal_ids(dtype, threads):
    """wdg_host_pack_coo_i32 (include/wdg.h): per-graph COO arrays -> one pair of int32 arrays of block-diagonal ids, any number
    of threads, int64 or int32 inputs, ragged and empty graphs; == the numpy concatenation + offsets ops.GraphBatch used to do"""
    from wdg_amd._lib import lib
    rng = np.random.default_rng(7)
    ns = [5, 1, 0, 2000, 17, 300]
    es = [12, 0, 0, 40000, 3, 999]
    srcs = [rng.integers(0, max(n, 1), e).astype(dtype) for n, e in zip(ns, es)]
    dsts = [rng.integers(0, max(n, 1), e).astype(dtype) for n, e in zip(ns, es)]
    node_ptr = np.concatenate([[0], np.cumsum(ns)]).astype(np.int32)
    lens = np.array(es, np.int64)
    ptrs = ctypes.c_void_p * len(ns)
    sp, dp = ptrs(*[a.ctypes.data for a in srcs]), ptrs(*[a.ctypes.data for a in dsts])
    tot = int(lens.sum())
    o1, o2, bad = np.full(tot, -7, np.int32), np.full(tot, -7, np.int32), ctypes.c_int32(5)
    rc = lib.wdg_host_pack_coo_i32(sp, dp, lens.ctypes.data, node_ptr.ctypes.data, len(ns), np.dtype(dtype).itemsize, o1.ctypes.data,
                                   o2.ctypes.data, ctypes.byref(bad), threads)
    assert rc == 0 and bad.value == 0
    np.testing.assert_array_equal(o1, np.concatenate([a.astype(np.int64) + node_ptr[g] for g, a in enumerate(srcs)]))
    np.testing.assert_array_equal(o2, np.concatenate([a.astype(np.int64) + node_ptr[g] for g, a in enumerate(dsts)]))
    # an id outside its graph: flagged and pushed out of the whole union (never folded into the neighbour's block)
    srcs[4][1] = 17
    dsts[0][3] = -1
    rc = lib.wdg_host_pack_coo_i32(sp, dp, lens.ctypes.data, node_ptr.ctypes.data, len(ns), np.dtype(dtype).itemsize, o1.ctypes.data,
                                   o2.ctypes.data, ctypes.byref(bad), threads)
    first = np.concatenate([[0], np.cumsum(es)])
    assert rc == 0 and bad.value == 1
    assert o1[first[4] + 1] == -1 and o2[first[4] + 1] == -1 and o1[first[0] + 3] == -1 and o2[first[0] + 3] == -1
    assert (o1[first[3]:first[4]] >= node_ptr[3]).all()


def test_host_pack_coo_rejects_bad_arguments():
    from wdg_amd._lib import lib
    bad = ctypes.c_int32(0)
    assert lib.wdg_host_pack_coo_i32(None, None, None, None, 0, 8, None, None, ctypes.byref(bad), 4) == 0  # nothing to do
    assert lib.wdg_host_pack_coo_i32(None, None, None, None, 2, 8, None, None, ctypes.byref(bad), 4) != 0
    assert lib.wdg_host_pack_coo_i32(None, None, None, None, 0, 2, None, None, ctypes.byref(bad), 4) != 0
    assert b"host_pack_coo" in lib.wdg_last_error()


def test_tape_cut_cost_of_many_graphs_equals_the_per_graph_cost():
    """ops._quad_unit_costs (one interpolation over a shard's slices) == ops._quad_unit_cost graph by graph, one and several
    column blocks"""
    from wdg_amd import ops
    rng = np.random.default_rng(0)
    ws = [rng.integers(0, 140, (1, 4 * int(rng.integers(1, 40)))) for _ in range(9)]
    for got, w in zip(ops._quad_unit_costs(ws), ws):
        np.testing.assert_array_equal(got, ops._quad_unit_cost(w))
    two = [rng.integers(0, 70, (2, 8))]
    np.testing.assert_array_equal(ops._quad_unit_costs(two)[0], ops._quad_unit_cost(two[0]))
    assert ops._quad_unit_costs([]) == []
