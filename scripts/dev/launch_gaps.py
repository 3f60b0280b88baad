#!/usr/bin/env python3
"""Where the aggregation launch's time goes inside a step, on the device clock: [clock kernel] launch [clock kernel] in
stream order + every workgroup's own start / end -> start latency, workgroup span, tail after the last workgroup."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from wdg_amd import sweep, synth
from wdg_amd.ops import lib, _ptr, stream_handle, check
from wdg_amd import aggregate as _ops
_ops.ABLATE_BITS = int(os.environ.get("WDG_GAPS_ABLATE", "0"))  # (timing-only ablations of scripts/dev/ablate_quad.py)
for k, seeds in ((10, 5), (2, 10)):
    levels = synth.H_LEVELS_10 if k == 2 else synth.H_LEVELS_10_K10
    b = sweep.SweepBatch(sweep.make_jobs(levels, range(seeds), k=k), n_feat=500)
    sp = b.spmm
    clock = sp.new_clock()
    marks = torch.zeros(2, dtype=torch.int64, device="cuda")
    for mode in ("in step", "alone"):
        rows = []
        for it in range(12):
            if mode == "in step":
                b.step()
            else:
                sp.launch(); sp.launch()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            check(lib.wdg_debug_clock(_ptr(marks[0:1]), stream_handle()), "clock")
            sp.launch(clock=clock)
            check(lib.wdg_debug_clock(_ptr(marks[1:2]), stream_handle()), "clock")
            e1.record()
            if mode == "in step":
                b.step_rest()
            torch.cuda.synchronize()
            t = clock.cpu().numpy().astype(np.float64).reshape(-1, 2) * 10e-3
            m = marks.cpu().numpy().astype(np.float64) * 10e-3
            rows.append((t[:, 0].min() - m[0], t[:, 1].max() - t[:, 0].min(), m[1] - t[:, 1].max(), m[1] - m[0], e0.elapsed_time(e1) * 1e3))
        r = np.median(np.array(rows[2:]), 0)
        print(f"k={k} seeds={seeds} {mode:8s}: first workgroup starts {r[0]:5.1f} us after the clock kernel before the launch; workgroups span "
              f"{r[1]:6.1f} us; clock kernel after the launch runs {r[2]:5.1f} us after the last workgroup's end; total {r[3]:6.1f} us "
              f"(events around all three: {r[4]:6.1f} us)", flush=True)
