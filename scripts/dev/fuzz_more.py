"""One-off extended fuzz run of the seeded GPU fuzz tests over many more seeds (dev tool)."""
import os
import sys
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
os.chdir(root)
from oracle import oracle as orc
orc.build()
from wdg_amd import ops
import test_gpu_kernels as T
import time
bad = 0
t0 = time.time()
for seed in range(100, int(sys.argv[1]) if len(sys.argv) > 1 else 260):
    for fn in (T.test_spmm_fuzz_shapes_and_batches, T.test_fuzz_build_stats_las_gemm, T.test_fuzz_resident_gemm_and_mlp2):
        try:
            fn(ops, orc, seed)
        except Exception as e:  # noqa: BLE001
            bad += 1
            print("seed", seed, fn.__name__, "FAILED:", str(e)[:400].replace("\n", " "), flush=True)
print("done, failures:", bad, "in", round(time.time() - t0, 1), "s")
