"""bench.py --gpus N without a launcher (the driver's `python3 bench.py --gpus 8`): the parent starts N workers - one per GPU,
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set - before anything touches the GPU, forwards rank 0's JSON line and returns
non-zero when a worker fails.  A stand-in worker (WDG_BENCH_WORKER) replaces the GPU half here."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tmp_path, worker_src, gpus, extra=()):
    worker = tmp_path / "worker.py"
    worker.write_text(textwrap.dedent(worker_src))
    env = dict(os.environ, WDG_BENCH_WORKER=str(worker), WDG_TEST_DIR=str(tmp_path))
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "3", "--warmup", "1", *extra],
                          env=env, capture_output=True, text=True, timeout=120)


def test_launcher_starts_one_worker_per_gpu_and_forwards_rank0(tmp_path):
    res = _run(tmp_path, """
        import json, os, sys
        rank = int(os.environ["RANK"])
        rec = {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
        rec["argv"] = sys.argv[1:]
        open(os.path.join(os.environ["WDG_TEST_DIR"], f"rank{rank}.json"), "w").write(json.dumps(rec))
        print(json.dumps({"metric": "stub", "rank": rank}))   # every rank prints: only rank 0's line may reach stdout
        """, gpus=2)
    assert res.returncode == 0, res.stderr
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and json.loads(lines[0]) == {"metric": "stub", "rank": 0}
    recs = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(2)]
    for r, rec in enumerate(recs):
        assert rec["RANK"] == str(r) and rec["LOCAL_RANK"] == str(r) and rec["WORLD_SIZE"] == "2"
        assert rec["MASTER_ADDR"] == "127.0.0.1" and rec["MASTER_PORT"].isdigit()
        assert rec["argv"] == ["--gpus", "2", "--steps", "3", "--warmup", "1"]
    assert recs[0]["MASTER_PORT"] == recs[1]["MASTER_PORT"]


def test_launcher_propagates_worker_failure(tmp_path):
    res = _run(tmp_path, """
        import os, sys
        print('{"metric": "stub"}')
        sys.exit(3 if os.environ["RANK"] == "1" else 0)
        """, gpus=2)
    assert res.returncode != 0
    assert "workers failed" in res.stderr


def test_launcher_stops_the_siblings_of_a_dead_worker(tmp_path):
    """one rank dies while the others would wait forever (a rendezvous / barrier): the launcher terminates them and fails"""
    import time
    t0 = time.monotonic()
    res = _run(tmp_path, """
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.stderr.write("rank 1: out of memory\\n")
            sys.exit(7)
        time.sleep(600)
        """, gpus=3)
    assert res.returncode != 0 and time.monotonic() - t0 < 60
    assert "workers failed" in res.stderr and "rank 1: out of memory" in res.stderr


def test_launcher_timeout(tmp_path):
    res = _run(tmp_path, "import time; time.sleep(600)", gpus=2, extra=("--timeout", "2"))
    assert res.returncode != 0 and "timeout" in res.stderr


def test_launcher_refuses_to_spawn_from_a_profiled_process(tmp_path):
    worker = tmp_path / "worker.py"
    worker.write_text("print('{}')")
    env = dict(os.environ, WDG_BENCH_WORKER=str(worker), HSA_TOOLS_LIB="/opt/rocm/lib/librocprofiler-sdk-tool.so")
    env.pop("WORLD_SIZE", None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=60)
    assert res.returncode == 2 and "profiled process" in res.stderr


def test_single_gpu_and_torchrun_paths_do_not_relaunch():
    """--gpus 1, or WORLD_SIZE already set by torch.distributed.run: bench.py is the worker itself"""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert '"WORLD_SIZE" not in os.environ and args.gpus > 1' in src


def test_eight_worker_dry_run_of_the_strong_scaled_sweep(tmp_path):
    """`python bench.py --gpus 8` with stand-in workers (no GPU here): the launcher's eight processes run what bench.py's
    measure_whole runs around the device work - rank 0's job table broadcast (sweep.broadcast_jobs), the sample-aware partition
    (sweep.pairs_of_rank), one exchange of keyed rows (sweep.exchange_rows), MAX over the ranks' clocks - over gloo on 127.0.0.1, while
    EVERY worker burns a host core for its share's duration (eight enqueueing interpreters at once, as on an 8-GPU node).  Asserts
    the assembled table (every key once, from the rank that owns it), that no rank's host section took more than 3 x the quietest
    rank's (host contention in the enqueue path would show here), and that only rank 0's line reaches stdout."""
    res = _run(tmp_path, """
        import json, os, sys, time
        sys.path.insert(0, %r)
        import numpy as np, torch, torch.distributed as dist
        from wdg_amd import sweep, synth
        rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dev = torch.device("cpu")
        levels = [h for h in synth.H_LEVELS_30 if h not in (0.05, 0.1)]
        pairs = sweep.make_jobs(levels, range(10), k=10, n_nodes=2000) if rank == 0 else []
        pairs = sweep.broadcast_jobs(pairs, dev)
        mine = sweep.pairs_of_rank(pairs, world, rank)
        index = {j: i for i, j in enumerate(pairs)}
        n_bases = 6
        dist.barrier()
        t0 = time.perf_counter()
        # the host section of a rank's pass: interpreter-bound table building - here a pure-Python loop of the same order of work
        # (~50 ms), on every rank at once
        acc, t_cpu = 0, time.process_time()
        while time.process_time() - t_cpu < 0.05:
            acc += sum(i * i for i in range(2000))
        host_s = time.perf_counter() - t0
        keys = torch.tensor([index[j] * n_bases + b for j in mine for b in range(n_bases)], dtype=torch.int64)
        rows = torch.stack([keys.double() * 9 + c for c in range(9)], 1) if len(keys) else torch.zeros((0, 9), dtype=torch.float64)
        table = sweep.exchange_rows(keys, rows, len(pairs) * n_bases, dev)
        t = torch.tensor([time.perf_counter() - t0, host_s], dtype=torch.float64)
        both = [torch.zeros(2, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(both, t)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ok = bool(torch.equal(table, torch.arange(len(pairs) * n_bases, dtype=torch.float64)[:, None] * 9 + torch.arange(9, dtype=torch.float64)[None, :]))
        dist.barrier()
        dist.destroy_process_group()
        print(json.dumps({"metric": "dry run", "rank": rank, "rows_ok": ok, "adjacencies": len(mine), "seconds": float(t[0]),
                          "host_s": [float(b[1]) for b in both]}))
        """ % ROOT, gpus=8)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.lstrip().startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["rank"] == 0 and rec["rows_ok"] and 34 <= rec["adjacencies"] <= 36
    host = rec["host_s"]
    assert len(host) == 8 and max(host) <= 3.0 * min(host) + 0.05, host  # (8 vCPUs here: every worker has a core; a serialised section would show 8 x)
