"""The page-locked ring of _rt._h2d on the CPU with fake events: a slice may be handed out again only after every copy that read
any byte of it has completed - for uneven request sizes and across wraps (the ring tested the oldest entry only)."""
import numpy as np
import torch

from wdg_amd._rt import _PinnedArena


class FakeEvent:
    """completes when the test says so; counts who waited for it"""

    def __init__(self, log):
        self.done, self.log = False, log

    def synchronize(self):
        self.log.append(self)
        self.done = True

    def query(self):
        return self.done


def _arena(size):
    log, events = [], []

    def new_event():
        events.append(FakeEvent(log))
        return events[-1]

    return _PinnedArena(size, buf=torch.zeros(size, dtype=torch.uint8), new_event=new_event), log, events


def test_the_advisors_example():
    """S = 1024: P3 = [512, 768), then (next lap) P4 = [0, 256), P5 = [256, 512) in flight; a request of 512 does not fit behind
    P5's lap position 768, wraps to [0, 512) and must wait for P4 AND P5 - the oldest entry P3 does not overlap it (it is waited
    for as well: entries complete in issue order, an older one costs nothing)"""
    ar, log, events = _arena(1024)
    h_, a1, _ = ar.take(512); ar.issued(h_)      # [0, 512)   lap 0
    h_, a3, _ = ar.take(256); ar.issued(h_)      # [512, 768) lap 0  (P3)
    events[0].done = True
    h_, a_skip, _ = ar.take(512); ar.issued(h_)  # does not fit the 256-byte tail: lap 1, [0, 512) - waits for the first entry only
    assert (a1, a3, a_skip) == (0, 512, 0) and log == [events[0]]
    log.clear()
    for e in events:
        e.done = True
    ar2, log2, ev2 = _arena(1024)
    for n in (512, 256):                          # lap 0: [0, 512) and P3 = [512, 768)
        h_, a, _ = ar2.take(n); ar2.issued(h_)
    ev2[0].done = True
    h_, a4, _ = ar2.take(256); ar2.issued(h_)    # P4: fits the tail [768, 1024)
    assert a4 == 768
    h_, a5, _ = ar2.take(256); ar2.issued(h_)    # P5: lap 1 [0, 256): one lap ahead of the first entry only
    h_, a6, _ = ar2.take(256); ar2.issued(h_)    # P6: lap 1 [256, 512)
    assert (a5, a6) == (0, 256) and log2 == [ev2[0]]
    log2.clear()
    _h, a, piece = ar2.take(512)                     # lap 1 [512, 1024): overlaps P3 and P4 of lap 0
    assert a == 512 and piece.numel() == 512
    assert ev2[1] in log2 and ev2[2] in log2 and ev2[3] not in log2 and ev2[4] not in log2
    _h, a, _ = ar2.take(512)                       # lap 2 [0, 512): overlaps P5 and P6 - neither is the head's neighbour in the buffer
    assert a == 0 and ev2[3] in log2 and ev2[4] in log2


def test_random_requests_never_reuse_bytes_in_flight():
    rng = np.random.default_rng(0)
    size = 1 << 16
    ar, log, events = _arena(size)
    in_flight = []   # (start, end, event) of copies the "device" has not run yet
    for step in range(4000):
        n = int(rng.integers(1, size // 3))
        h_, a, piece = ar.take(n)
        b = a + ((n + 255) & ~255)
        assert b <= size and piece.numel() == n
        for s, e, ev in in_flight:
            if s < b and a < e:
                assert ev.done, f"step {step}: [{a}, {b}) handed out while the copy of [{s}, {e}) is in flight"
        ar.issued(h_)
        in_flight.append((a, b, events[-1]))
        # the device completes a random prefix of the queue (copies finish in issue order)
        k = int(rng.integers(0, 3))
        for _s, _e, ev in in_flight[:k]:
            ev.done = True
        in_flight = [x for x in in_flight if not x[2].done]
        assert sum(e - s_ for s_, e, _ in in_flight) <= size  # (what is in flight never exceeds one lap)


def test_oversized_request_is_refused():
    ar, _, _ = _arena(1024)
    try:
        ar.take(2048)
    except ValueError:
        return
    raise AssertionError("a request larger than the ring must raise")


def test_two_threads_may_fill_their_slices_at_the_same_time():
    """take() enters the slice into `pending` in ring order with no event yet, issued(handle) supplies it: slices taken by two
    threads and queued in the other order leave `pending` ascending, and a request that laps a slice whose copy has not been
    queued yet waits until it has (and then for its event)."""
    import threading
    import time
    ar, log, events = _arena(1024)
    ha, a, _ = ar.take(256)
    hb, b, _ = ar.take(256)
    ar.issued(hb)              # thread B queues its copy first
    ar.issued(ha)
    assert (a, b) == (0, 256) and [e[0] for e in ar.pending] == [0, 256] and all(e[1] is not None for e in ar.pending)
    hc, c, _ = ar.take(512)    # [512, 1024), taken but not queued yet
    assert c == 512
    got = []

    def lapper():              # lap 1 [0, 768): overlaps a, b (queued) and c (not queued yet)
        for e in events:
            e.done = True
        got.append(ar.take(768)[1])

    t = threading.Thread(target=lapper)
    t.start()
    time.sleep(0.05)
    assert t.is_alive() and not got          # waiting for c's copy to be queued
    ar.issued(hc)
    t.join(5)
    assert not t.is_alive() and got == [0] and events[-1] in log


def test_an_abandoned_slice_does_not_hang_the_request_that_laps_it():
    """the thread that took a slice raised between take() and issued() -> the entry stayed unstamped and the next
    request a lap behind spun for ever with the ring's lock held.  abandon() stamps it; an unstamped one times out with an error."""
    ar, log, events = _arena(1024)
    h_bad, _a, _ = ar.take(512)
    ar.abandon(h_bad)                               # (what _h2d does when the fill or the copy raises)
    h_, _a, _ = ar.take(512); ar.issued(h_)
    h_, a, _ = ar.take(512)                         # laps the abandoned slice: no event to wait for, no spin
    assert a == 0 and log == []
    ar.issued(h_)
    ar2, _log2, _ev2 = _arena(1024)
    ar2.STALL_S = 0.05
    ar2.take(512)                                   # taken, never stamped (a bug by construction)
    h_, _a, _ = ar2.take(512); ar2.issued(h_)
    import pytest
    with pytest.raises(RuntimeError, match="never queued"):
        ar2.take(512)
