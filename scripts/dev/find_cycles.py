#!/usr/bin/env python3
"""Does a finished SweepBatch die by reference counting, or does it wait for the cyclic collector (dev tool)?  A batch that waits keeps
its GPU memory - gigabytes of kernels per base-shard - until the collector happens to run."""
import gc
import os
import sys
from collections import Counter

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from wdg_amd import sweep, synth

gc.collect()
gc.disable()
jobs = sweep.make_jobs([0.2, 0.5], [0, 1], k=4, n_nodes=600)
sb = sweep.SweepBatch(jobs, n_feat=int(sys.argv[1]) if len(sys.argv) > 1 else 700, gcn_hidden=0, labels_only=len(sys.argv) > 2)
sb.prepare_full(epochs=4, sample_max=300)
sb.step()
sb.launch_full()
rows = sb.full_metrics()
torch.cuda.synchronize()
before = torch.cuda.memory_allocated()
del sb
after = torch.cuda.memory_allocated()
print(f"memory allocated before / after `del sb`: {before / 1e6:.1f} / {after / 1e6:.1f} MB")
gc.set_debug(gc.DEBUG_SAVEALL)
n = gc.collect()
print("unreachable objects found by the collector:", n)
types = Counter(type(o).__name__ for o in gc.garbage)
print(types.most_common(12))
for o in gc.garbage:
    if type(o).__name__ in ("SweepBatch", "SpmmBatch", "GraphBatch", "PropagatedGram", "KrBatch", "CsrGraph", "function", "cell", "dict") and not isinstance(o, dict):
        refs = [type(r).__name__ for r in gc.get_referrers(o) if r is not gc.garbage][:6]
        print(type(o).__name__, getattr(o, "__qualname__", ""), "<-", refs)
