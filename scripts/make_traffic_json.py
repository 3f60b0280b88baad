#!/usr/bin/env python3
"""profiles/traffic.json from the PMC summaries of scripts/profile_pmc.sh: one record per sweep configuration (keyed on k,
graphs per launch and feature width) holding the HBM bytes one aggregation launch moved.
usage: make_traffic_json.py <k>[@<nodes>]:<graphs>:<summary.json>:<source note> ... > profiles/traffic.json"""
import json
import sys

recs = []
for arg in sys.argv[1:]:
    k, graphs, path, source = arg.split(":", 3)
    k, _, nodes = k.partition("@")  # "<k>@<nodes>": a shard of other than 2000-node graphs
    nodes = int(nodes) if nodes else 2000
    s = json.load(open(path))
    name = next(n for n in s if n.startswith("spmm_"))
    c = s[name]
    fetch_raw, write = c["FETCH_SIZE"] * 1024, c["WRITE_SIZE"] * 1024
    recs.append({
        "kernel": name, "graphs_per_launch": int(graphs), "n_nodes": nodes, "n_feat": 500, "agg_feat": 512, "k": int(k),
        "fetch_size_raw_bytes": fetch_raw, "fetch_corrected_bytes": 2 * fetch_raw, "write_size_bytes": write,
        "fetch_correction": "x2: gfx950 FETCH_SIZE counts 128-B requests of wide (16 B/lane) coalesced reads as 64 B "
                            "(MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact for 16-B/lane stores",
        "hbm_bytes_per_launch": 2 * fetch_raw + write,
        "source": source, "launches_averaged": c.get("launches"),
        "tcc_hit": c.get("TCC_HIT_sum"), "tcc_miss": c.get("TCC_MISS_sum"),
        "lds_bank_conflict_cycles": c.get("SQ_LDS_BANK_CONFLICT"), "lds_active_cycles": c.get("SQ_LDS_IDX_ACTIVE"),
        "wait_any_frac": (c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]) if c.get("SQ_WAVE_CYCLES") else None,
        "insts_valu": c.get("SQ_INSTS_VALU"), "insts_salu": c.get("SQ_INSTS_SALU"), "insts_lds": c.get("SQ_INSTS_LDS"),
        "note": "F_agg = 512 (500 features + C one-hot label columns + zero columns in one launch).  X (16-feature slices) is "
                "fetched once per group of graphs sharing a seed, so traffic is below the algorithmic figure of SURVEY 8(d), "
                "which counts X once per graph",
    })
json.dump(recs, sys.stdout, indent=1)
print()
