"""Host logic of the quad-row batch: the tape of 16-row units is cut into equal-cost segments and phases (ops._quad_segments)
such that every unit of every job is covered exactly once, phases never mix feature matrices, and segment costs are level."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class _X:
    def __init__(self, ptr, n, f):
        self.ptr, self.shape = ptr, (n, f)

    def data_ptr(self):
        return self.ptr

    def stride(self, d):
        return self.shape[1] if d == 0 else 1


class _G:
    def __init__(self, n_rows, n_cols, widths):
        """widths: [n_blocks, padded slices] (a multiple of 4 slices = whole super-units)"""
        assert widths.shape[1] % 4 == 0
        self.n_rows, self.n_cols = n_rows, n_cols
        self.quad = dict(widths=widths, n_slices=widths.shape[1], n_entries=widths.shape[1], n_su=widths.shape[1] // 4,
                         n_blocks=widths.shape[0], half=widths.shape[0] == 1 and 2528 < n_cols <= 5056)


def _units_of(entries, order, item):
    """(job position, super-unit) of every unit of an item, in the kernel's order (the jobs' super-units concatenated)"""
    fj, nj, ub, ue = item
    assert nj >= 1 and 0 <= ub < ue
    ns = [entries[order[fj + k]][0].quad["n_su"] for k in range(nj)]
    xs = {entries[order[fj + k]][1].data_ptr() for k in range(nj)}
    assert len(xs) == 1, "an item aggregates one feature matrix"
    offs = np.cumsum([0] + ns)
    assert ue <= offs[-1]
    out = []
    for u in range(ub, ue):
        k = int(np.searchsorted(offs, u, side="right") - 1)
        out.append((fj + k, u - offs[k]))
    return out


def _coverage(entries, order, items, seg_ptr):
    seen = {pos: np.zeros(entries[i][0].quad["n_su"], int) for pos, i in enumerate(order)}
    for item in items:
        for pos, su in _units_of(entries, order, item):
            seen[pos][su] += 1
    for pos in seen:
        assert (seen[pos] == 1).all(), f"job {pos}: every super-unit exactly once"
    assert seg_ptr[0] == 0 and seg_ptr[-1] == len(items) and all(a <= b for a, b in zip(seg_ptr, seg_ptr[1:]))


@pytest.mark.parametrize("n_feat", [512, 64, 16, 2089])
def test_segments_cover_every_unit_once(n_feat):
    from wdg_amd import ops
    rng = np.random.default_rng(3)
    entries = []
    for seed in range(5):
        x = _X(1000 + seed, 2000, n_feat)
        for h in range(10):
            w = np.full((1, 128), int(rng.integers(3, 70)))
            entries.append((_G(2000, 2000, w), x, None, None, None, False))
    order = list(range(len(entries)))
    items, seg_ptr, n_seg = ops._quad_segments(entries, order, n_feat)
    assert n_seg % 8 == 0 and len(seg_ptr) == n_seg + 1
    _coverage(entries, order, items, seg_ptr)
    # level costs: no segment more than 3 % (+ one unit) above the mean
    costs = []
    for s in range(n_seg):
        c = 0.0
        for item in items[seg_ptr[s]:seg_ptr[s + 1]]:
            for pos, su in _units_of(entries, order, item):
                c += ops._quad_unit_cost(entries[order[pos]][0].quad["widths"][:, 4 * su:4 * su + 4])[0]
        costs.append(c)
    assert max(costs) <= 1.03 * np.mean(costs) + 4 * 3800


def test_segments_ragged_and_multiblock():
    from wdg_amd import ops
    rng = np.random.default_rng(4)
    entries = []
    for j in range(7):  # different sizes, own X each, some tiny
        n = int(rng.integers(1, 900))
        sl = (n + 63) // 64 * 4
        entries.append((_G(n, n, rng.integers(0, 40, (1, sl))), _X(j, n, 96), None, None, None, False))
    order = list(range(len(entries)))
    items, seg_ptr, n_seg = ops._quad_segments(entries, order, 96)
    _coverage(entries, order, items, seg_ptr)
    # several column blocks: phases of at most 128 units
    entries = [(_G(5201, 5201, rng.integers(0, 90, (3, 328))), _X(9, 5201, 2089), None, None, None, False)]
    items, seg_ptr, n_seg = ops._quad_segments(entries, [0], 2089)
    _coverage(entries, [0], items, seg_ptr)
    assert all(ue - ub <= 32 for _, _, ub, ue in items)


def test_segment_phases_run_shortest_first_and_shares_shift_the_cuts():
    """a segment that touches two phase groups lists the cheaper phase first (its slab is staged while the launch's stores
    have not yet filled the write path); `shares` moves cost between segments (the feedback balancing of SweepBatch.tune)
    and a phase price shortens the segments that hold two phases - coverage stays exact in every case"""
    from wdg_amd import ops
    rng = np.random.default_rng(8)
    entries = []
    for seed in range(5):
        x = _X(2000 + seed, 2000, 512)
        for h in range(10):
            entries.append((_G(2000, 2000, np.full((1, 128), int(rng.integers(8, 33)))), x, None, None, None, False))
    order = list(range(len(entries)))

    def seg_costs(items, seg_ptr):
        out = []
        for s in range(len(seg_ptr) - 1):
            out.append([sum(ops._quad_unit_cost(entries[order[pos]][0].quad["widths"][:, 4 * su:4 * su + 4])[0]
                            for pos, su in _units_of(entries, order, it)) for it in items[seg_ptr[s]:seg_ptr[s + 1]]])
        return out

    items, seg_ptr, n_seg = ops._quad_segments(entries, order, 512)
    _coverage(entries, order, items, seg_ptr)
    base = seg_costs(items, seg_ptr)
    assert any(len(c) == 2 for c in base)
    assert all(c == sorted(c) for c in base), "phases of a segment in ascending cost"
    # a price per phase: two-phase segments get less of the tape than one-phase segments
    items2, seg_ptr2, _ = ops._quad_segments(entries, order, 512, phase_ns=20000)
    _coverage(entries, order, items2, seg_ptr2)
    priced = seg_costs(items2, seg_ptr2)
    one = [sum(c) for c in priced if len(c) == 1]
    two = [sum(c) for c in priced if len(c) == 2]
    assert one and two and max(two) < min(one)
    # shares: segment 0 is asked to take 30 % more than the others
    shares = np.ones(8)
    shares[0] = 1.3
    items3, seg_ptr3, _ = ops._quad_segments(entries, order, 512, phase_ns=0, shares=shares)
    _coverage(entries, order, items3, seg_ptr3)
    shared = [sum(c) for c in seg_costs(items3, seg_ptr3)]
    assert shared[0] > 1.2 * np.mean(shared[1:])
