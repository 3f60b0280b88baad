import os, sys, time, cProfile, pstats
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
import torch
from wdg_amd import sweep, synth
def host_inputs(first_seed):
    jobs = sweep.make_jobs(synth.H_LEVELS_10_K10, range(first_seed, first_seed + 5), k=10, n_nodes=2000)
    feats, inputs = {}, []
    for j in jobs:
        src, dst, lab = synth.regular_graph(j.n_nodes, j.n_classes, j.k, j.h, j.seed)
        feats.setdefault(j.seed, synth.features(j.n_nodes, 500, j.seed))
        inputs.append((src, dst, lab, feats[j.seed]))
    return jobs, inputs
shards = [host_inputs(1000 + 5 * b) for b in range(10)]
for _ in range(2):
    list(sweep.run_shards(shards, n_feat=500, nine=False, depth=2, first_seed=1))
pr = cProfile.Profile(); pr.enable()
list(sweep.run_shards(shards, n_feat=500, nine=False, depth=2, first_seed=1))
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
