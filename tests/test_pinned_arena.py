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
    """S = 1024: P3 = [512, 768), P4 = [0, 256), P5 = [256, 512) in flight; take(512) wraps to [0, 512) and must wait for P4 AND
    P5, although the oldest entry P3 does not overlap"""
    ar, log, events = _arena(1024)
    for n in (256, 256):          # P1, P2 (complete: forgotten at the next take)
        a, _ = ar.take(n)
        ar.issued(a, n)
    for e in events:
        e.done = True
    a3, _ = ar.take(256); ar.issued(a3, 256)      # [512, 768)
    ar.off = 0                                    # (as after a wrap)
    a4, _ = ar.take(256); ar.issued(a4, 256)      # [0, 256)
    a5, _ = ar.take(256); ar.issued(a5, 256)      # [256, 512)
    assert (a3, a4, a5) == (512, 0, 256)
    p3, p4, p5 = events[2:5]
    ar.off = 768
    a, piece = ar.take(512)                       # 768 + 512 > 1024: wraps to [0, 512)
    assert a == 0 and piece.numel() == 512
    assert p4 in log and p5 in log and p3 not in log
    assert [(s, e) for s, e, _ in ar.pending] == [(512, 768)]


def test_random_requests_never_reuse_bytes_in_flight():
    rng = np.random.default_rng(0)
    size = 1 << 16
    ar, log, events = _arena(size)
    in_flight = []   # (start, end, event) of copies the "device" has not run yet
    for step in range(4000):
        n = int(rng.integers(1, size // 3))
        a, piece = ar.take(n)
        b = a + ((n + 255) & ~255)
        assert b <= size and piece.numel() == n
        for s, e, ev in in_flight:
            if s < b and a < e:
                assert ev.done, f"step {step}: [{a}, {b}) handed out while the copy of [{s}, {e}) is in flight"
        ar.issued(a, n)
        in_flight.append((a, b, events[-1]))
        # the device completes a random prefix of the queue (copies finish in issue order)
        k = int(rng.integers(0, 3))
        for _s, _e, ev in in_flight[:k]:
            ev.done = True
        in_flight = [x for x in in_flight if not x[2].done]
        assert len(ar.pending) <= len(in_flight) + 1 + k  # (bounded: completed entries are forgotten)


def test_oversized_request_is_refused():
    ar, _, _ = _arena(1024)
    try:
        ar.take(2048)
    except ValueError:
        return
    raise AssertionError("a request larger than the ring must raise")
