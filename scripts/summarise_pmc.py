#!/usr/bin/env python3
"""Average the PMC counters of scripts/profile_pmc.sh per launch of each kernel of interest."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out_dir = sys.argv[1]
want = ("spmm_quad", "spmm_band", "spmm_narrow", "narrow_pack", "kr_solve", "gram_map", "gram_split", "mlp2_split", "spmm_slab", "spmm_gather", "edge_stats", "gemm_f32", "gemm_bres", "mlp2_bres", "las_")
acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob(os.path.join(out_dir, "*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        name = r.get("Kernel_Name", "")
        key = next((w for w in want if w in name), None)
        if key is None:
            continue
        short = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {}
for k, counters in acc.items():
    res[k] = {c: sum(v) / len(v) for c, v in counters.items()}
    res[k]["launches"] = max(len(v) for v in counters.values())
for k, c in res.items():
    print(k)
    for name in sorted(c):
        print(f"    {name:32s} {c[name]:18.1f}")
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        f, w = c["FETCH_SIZE"] * 1024, c["WRITE_SIZE"] * 1024
        print(f"    -> raw fetch {f/1e6:.1f} MB, write {w/1e6:.1f} MB per launch; fetch x2 (gfx950 wide-read correction) {2*f/1e6:.1f} MB")
json.dump(res, open(os.path.join(out_dir, "summary.json"), "w"), indent=1)
