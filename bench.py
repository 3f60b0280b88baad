#!/usr/bin/env python3
"""Headline benchmark: aggregation edges/s on the synthetic homophily sweep the north star quotes its target on
(BASELINE.json configs[2], SURVEY.md 8(d) config C3: the `data_synthesis/4000`-equivalent set).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment launches its own N worker processes (one per
GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set) before anything touches the GPU, forwards rank 0's JSON line and
exits non-zero when a worker fails.

Workload per rank (weak scaling: fixed per-GPU work): the `data_synthesis/4000`-equivalent sweep shard
  10 homophily levels x `--seeds` seeds (default 5: the C3 job set) = 50 graphs, N = 2000 nodes, k = 10 (12 .. 67 stored
  entries per row of A + I), F = 500 fp32 features (the directory name "4000" is k*400, every graph has 2000 nodes -
  SURVEY G3).  `--k 2 --seeds 10` is the `800` set (C2), the round-1 headline; it is emitted as `secondary` when
  `--secondary` is given.
A step = one pass of the hot path over that batch with all inputs resident in HBM, 4 batched launches:
  (1) A_hat [X | onehot(y)] aggregation, A_hat = D^-1 (A + I) fused into the SpMM (csrc/spmm_quad.hip, spmm_quad_kernel:
      graphs of a seed share X; the label columns of the LAS metric ride in the last, otherwise mostly empty feature group)
  (2) LAS counts + the integer edge / label counters derived from the aggregated label columns (csrc/las.hip), on a second
      stream beside (3)
  (3) GCN-2 feature path with per-graph weights relu(Y W0) W1, fused, split-bf16 operands on the matrix pipe (csrc/gemm.hip)
  (4) the logits aggregation A_hat (.) (F = C, csrc/spmm_narrow.hip)
  (WDG_SWEEP_RIDE_LABELS=0 / WDG_SWEEP_FUSED_MLP=0 / WDG_SWEEP_DERIVE_COUNTS=0 restore the separate label aggregation / the
  two GEMM launches / the integer pass over the edges: 7)
`value` = stored entries of A+I aggregated per second over the whole job (all ranks); the dominant kernel's roofline is
measured live with HIP events on the launch stream (sampled: 50 launches over the timed region); a bounded CPU sample of the
same workload, run the way the reference does it, is reported as `cpu_baseline` (rank 0, N=1 only).

Beside the headline the N = 1 line carries (each can be switched off, see --help):
  secondary           the `800` set (k = 2, 100 graphs) through the same step
  sweep_full / sweep_cold   all nine scalars of the sweep job: replayed on a resident batch / every graph visited once
  sweep_whole         the sweep the reference runs (synthetic_plot.py: 6 feature bases x 28 levels x 10 samples = 1 680 jobs, nine
                      scalars), one pass from host arrays to rows on the host.  ALSO AT N > 1: the 280 adjacencies sharded over the
                      ranks (strong scaling), rows exchanged by one all_gather, `seconds` = slowest rank, and rank 0's one-GPU pass of
                      the same sweep in the same session -> `strong_speedup_vs_1gpu` (the north star's ">= 6 x at 8 GPUs")
  configs             BASELINE.json configs[0], [3], [4] as stated + the literal N = 4000 reading of configs[2]: per launch `us`,
                      kernel and roofline fraction (scripts/bench_configs.py holds the workloads)
  train               train + eval sweep (sweep.TrainBatch on the C3 shard): SGC-1 and GCN-2, ms per epoch (hipGraph replay),
                      graphs/s at 200 epochs
  scaling_projection  this GPU timing every rank's shard of the 50-job sweep for world sizes 2 / 4 / 8 (sweep.shard_jobs):
                      a PROJECTION of strong scaling from one device, labelled as such
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
# (what wdg_amd/_lib.py sets at import, here before anything can have initialised the HIP runtime: a sweep's eight concurrent streams
# on eight hardware queues - on the default four they alias and the whole sweep takes 0.40 s instead of 0.30)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s achievable
ROOFLINE_LAUNCHES = 24  # aggregation launches timed by HIP events for `roofline` (>= 20: SURVEY 8(d)), whatever --steps is
FP32_MFMA_PEAK_TF = 157.3  # dense fp32 MFMA peak (v_mfma_f32_32x32x2_f32: 256 flop / clk / CU x 256 CUs x 2.4 GHz; MI355X_MICROARCH.md)


def launch_workers(args, argv):
    """`python bench.py --gpus N` without a launcher: start N fresh workers (one per GPU) from a parent that never touches
    the GPU, forward rank 0's stdout (the JSON line), return non-zero when any worker fails.  All workers are polled: the
    first non-zero exit (or --timeout) terminates the siblings - they would otherwise sit in the rendezvous or a barrier
    until RCCL's own timeout - and every rank's stderr goes to its own file so the failing rank can be read.
    WDG_BENCH_WORKER (tests) names a stand-in worker script."""
    import socket
    import subprocess
    import tempfile
    preload = os.environ.get("LD_PRELOAD", "") + os.environ.get("ROCP_TOOL_LIBRARIES", "") + os.environ.get("HSA_TOOLS_LIB", "")
    if "rocprof" in preload:
        # under rocprofv3 the profiler's preload has initialised the GPU in THIS process already: starting torch workers from
        # it is the exec-after-GPU-init pattern the pool forbids.  Profile one rank: rocprofv3 ... -- python3 bench.py
        print("bench.py: refusing to spawn workers from a profiled process; profile `bench.py --gpus 1`", file=sys.stderr)
        return 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    script = os.environ.get("WDG_BENCH_WORKER", os.path.abspath(__file__))
    logdir = tempfile.mkdtemp(prefix="wdg_bench_")
    procs, errs = [], []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", str(port)))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        errs.append(open(os.path.join(logdir, f"rank{r}.err"), "w+"))
        procs.append(subprocess.Popen([sys.executable, script] + argv, env=env, stderr=errs[-1],
                                      stdout=open(os.path.join(logdir, "rank0.out"), "w+") if r == 0 else subprocess.DEVNULL))
    deadline = time.monotonic() + args.timeout
    rcs = [None] * len(procs)
    failed = None
    while any(rc is None for rc in rcs):
        for r, p in enumerate(procs):
            if rcs[r] is None:
                rcs[r] = p.poll()
        bad = [r for r, rc in enumerate(rcs) if rc not in (None, 0)]
        if (bad or time.monotonic() > deadline) and failed is None:
            failed = bad or ["timeout"]
            for r, p in enumerate(procs):  # the exact children started above, nothing else
                if rcs[r] is None:
                    p.terminate()
            grace = time.monotonic() + 10
            while time.monotonic() < grace and any(p.poll() is None for p in procs):
                time.sleep(0.1)
            for p in procs:
                if p.poll() is None:
                    p.kill()
        time.sleep(0.05)
    out = open(os.path.join(logdir, "rank0.out")).read()
    # exactly the JSON line: libraries (RCCL) write to the workers' stdout as well
    lines = [l for l in out.splitlines() if l.lstrip().startswith("{")]
    sys.stdout.write("".join(l + "\n" for l in lines[-1:]))
    sys.stdout.flush()
    if failed is not None:
        print(f"bench.py: workers failed (first: rank {failed}; return codes {rcs}); stderr of every rank under {logdir}", file=sys.stderr)
        for r in (failed if failed != ["timeout"] else range(len(procs))):
            errs[r].seek(0)
            sys.stderr.write(f"---- rank {r} stderr (tail) ----\n" + errs[r].read()[-2000:] + "\n")
        return 1
    for r, f in enumerate(errs):  # a clean run: the ranks' stderr is forwarded as before
        f.seek(0)
        sys.stderr.write(f.read())
    return 0


def measure(args, k, seeds, steps, warmup, world, rank, dev):
    """Build this rank's shard of the (k, seeds) sweep, time `steps` steps -> dict of raw measurements (every rank)."""
    import torch
    import torch.distributed as dist
    from wdg_amd import sweep, synth
    h_levels = synth.H_LEVELS_10 if k == 2 else synth.H_LEVELS_10_K10
    # rank 0 defines the whole job list and broadcasts it; every rank takes its shard (sweep.shard_jobs: single jobs, by cost)
    # weak scaling (default): `seeds` seeds per rank; strong: the job list is fixed (BASELINE configs[2] literally: 10 h-levels
    # x 5 seeds = 50 jobs) and sharded over however many ranks there are
    n_seeds = seeds if args.scaling == "strong" else seeds * world
    jobs = sweep.make_jobs(h_levels, range(n_seeds), k=k, n_nodes=args.nodes) if rank == 0 else []
    jobs = sweep.broadcast_jobs(jobs, dev)
    mine = sweep.shard_jobs(jobs, world, rank)
    batch = sweep.SweepBatch(mine, n_feat=args.feat)
    # First use of every kernel and of the tail ops (their code objects load lazily: ~15 ms of host time with the GPU idle) BEFORE
    # the tape cut is balanced, so that nothing but the W warm-up steps and a synchronisation lies between the tuner's steps and the
    # clock: behind >= 5 ms of idling the chip runs steps ~10 - 35 of the next burst 8 - 12 % slower (scripts/dev/step_transient.py),
    # which a 20-step timed region would otherwise sit in.
    batch.step()
    sweep.gather_results(batch.results(), dev)

    def spmm_launch_ms(n):  # n aggregation launches inside whole steps, each between two HIP events on the launch stream
        marks = []
        for _ in range(n):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            batch.spmm.launch()
            b.record()
            batch.step_rest()
            marks.append((a, b))
        torch.cuda.synchronize()
        return sorted(a.elapsed_time(b) for a, b in marks)

    # the aggregation launch with the MODELLED tape cut (what a one-pass sweep - sweep_cold, sweep_whole - runs with), reported beside
    # the tuned launch the replayed headline batch uses
    for _ in range(5):
        batch.step()
    untuned_ms = spmm_launch_ms(ROOFLINE_LAUNCHES)
    batch.tune()  # (replayed `steps` times: the feedback-balanced tape cut pays; ~250 untimed steps, 55 ms)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # WDG_BENCH_GRAPH=1 replays everything after the aggregation launch from one captured hipGraph (the launches of both
    # streams -> 1); measured slower than the plain launches (0.517 vs 0.486 ms per step), so off by default
    step_rest = batch.capture_rest() if os.environ.get("WDG_BENCH_GRAPH", "0") == "1" else batch.step_rest
    # HIP events around the aggregation launch of every `stride`-th step (about 50 samples over the timed region): an
    # event pair costs ~20 us of queue markers per step (scripts/dev/time_gaps.py: 0.443 -> 0.423 ms without them), so the
    # roofline figure is sampled instead of taxing every step; all steps run the same launches either way
    # (never more than every fourth step, starting at the third: a run of 20 steps is not taxed on every step, and the launch
    # right behind the synchronisation - 20-25 us of wake-up on top of the kernel - is not one of five samples)
    stride = max(4, steps // 50)
    ev = {s: (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for s in range(min(2, steps - 1), steps, stride)}
    for _ in range(warmup):
        batch.step()
    sync_all()
    t0 = time.perf_counter()
    for s in range(steps):
        if s in ev:
            ev[s][0].record()       # torch's current stream == the stream the kernels are launched on
            batch.spmm.launch()
            ev[s][1].record()
        else:
            batch.spmm.launch()
        step_rest()
    enqueue_s = time.perf_counter() - t0  # host time to enqueue every step (the device runs behind it)
    rows = batch.results()
    gathered = sweep.gather_results(rows, dev)  # the sweep's one exchange step: per-job metric rows (KBs)
    sync_all()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        e = torch.tensor([batch.edges], dtype=torch.int64, device=dev)
        dist.all_reduce(e)
        total_edges = int(e.item())
    else:
        total_edges = batch.edges
    if os.environ.get("WDG_BENCH_DUMP_LAUNCHES"):
        print("launch us by step:", " ".join(f"{s_}:{a.elapsed_time(b) * 1e3:.0f}" for s_, (a, b) in sorted(ev.items())), file=sys.stderr)
    spmm_ms = [a.elapsed_time(b) for a, b in ev.values()]
    in_region = len(spmm_ms)
    # SURVEY 8(d) asks for >= 20 launches whatever --steps is (the driver runs 20 steps: 5 samples at every fourth step): the same steps
    # go on right behind the timed region, every one with an event pair, until ROOFLINE_LAUNCHES launches have been timed
    if in_region < ROOFLINE_LAUNCHES:
        for _ in range(3):
            batch.step()
        extra = spmm_launch_ms(ROOFLINE_LAUNCHES - in_region)
        if os.environ.get("WDG_BENCH_DUMP_LAUNCHES"):
            print("launch us behind the region:", " ".join(f"{v * 1e3:.0f}" for v in extra), file=sys.stderr)
        spmm_ms += extra
    spmm_ms.sort()
    return dict(batch=batch, mine=mine, h_levels=h_levels, elapsed=elapsed, enqueue_s=enqueue_s, total_edges=total_edges,
                n_graphs=sum(g.shape[0] for g in gathered), spmm_ms=spmm_ms, spmm_in_region=in_region, untuned_ms=untuned_ms,
                k=k, seeds=seeds, steps=steps)


def measure_full(args, dev):
    """The whole sweep job of synthetic_plot.py:92-109 for the shard - all nine scalars: a step's aggregation + counters + LAS,
    then the Gram / arc-cosine kernels of every graph, the edge cosines and every kernel regression of every epoch - as
    graphs/s, REPLAYED on a resident batch (steady state; the one-pass figure is `sweep_cold`).  The epochs' node sets are drawn
    by the device sampler inside the timed launches."""
    import torch
    from wdg_amd import sweep, synth
    h_levels = synth.H_LEVELS_10 if args.k == 2 else synth.H_LEVELS_10_K10
    jobs = sweep.make_jobs(h_levels, range(args.seeds), k=args.k, n_nodes=args.nodes)
    sb = sweep.SweepBatch(jobs, n_feat=args.feat, gcn_hidden=0)
    sb.prepare_full(epochs=args.kr_epochs, sample_max=500)
    sb.step()
    sb.launch_full()
    sb.full_metrics()  # (scipy import, pinned staging)
    # five replays, each between two stream events; the median is reported (and the slowest beside it): on a shared host one
    # replay in eight or so is hit by a 40-ms scheduling stall of the enqueueing thread, which a mean over three would double
    reps, marks = 5, []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        sb.step()
        sb.launch_full()
        e1.record()
        marks.append((e0, e1))
    torch.cuda.synchronize()
    per_rep = sorted(a.elapsed_time(b) * 1e-3 for a, b in marks)
    dev_s = per_rep[len(per_rep) // 2]
    t0 = time.perf_counter()
    rows = sb.full_metrics()
    tail_s = time.perf_counter() - t0

    def launch_ms(fn, reps_=3):  # one stage alone on the current stream, HIP events around `reps_` launches
        fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps_):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps_

    # the two kernels that carry the metric's arithmetic, each against the resource that could bound it
    solve_ms, gram_ms = launch_ms(sb.kr.launch), launch_ms(sb.gram.launch)
    nt, nv, c_ = int(sb.kr_train.shape[2]), int(sb.kr_val.shape[2]), 8  # (the solver carries 8 right-hand sides whatever the class count)
    flop_reg = nt ** 3 / 3 + 2 * nt * nt * c_ + 2 * nv * nt * c_
    gathered = (nt * (nt + 1) // 2 + nv * nt) * 4
    n_reg = sb.kr.n_jobs
    n_nodes = args.nodes
    n_gram = len(sb.gram.k_linear)
    gram_flop = n_gram * float(n_nodes) ** 2 * args.feat  # (the lower triangle is computed and mirrored: n^2 F multiply-adds = n^2 F x 2 / 2 flops)
    rl = {"roofline_solver": {"kernel": "kr_solve_blocked_kernel", "bound": "mfma", "regressions": n_reg, "ms_per_launch": solve_ms,
                              "achieved": n_reg * flop_reg / (solve_ms * 1e-3) / 1e12, "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s",
                              "frac": n_reg * flop_reg / (solve_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TF,
                              "flop_per_regression": flop_reg, "train_rows": nt, "validation_rows": nv,
                              "gathered_bytes_per_regression": gathered, "gathered_GBs": n_reg * gathered / (solve_ms * 1e-3) / 1e9,
                              "us_per_regression_and_cu": solve_ms * 1e3 / max(n_reg, 1) * 256,
                              "note": "one regression per CU at a time (144 KB of LDS): a latency chain - one wave factors a 32 x 32 diagonal block while "
                                      "fifteen gather - not a throughput kernel; neither the matrix pipe nor any bandwidth is the bound (DESIGN 4.8: "
                                      "profiles/r05_kr_pmc_summary.txt)"},
          "roofline_gram": {"kernel": f"{'PropagatedGram (2 x spmm_quad + transpose + finish) + ' if sb.gram_route == 'propagate' else ''}gram_split_kernel",
                            "bound": "mfma", "matrices": n_gram, "ms_per_launch": gram_ms, "achieved": gram_flop / (gram_ms * 1e-3) / 1e12,
                            "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": gram_flop / (gram_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TF,
                            "note": f"K = map(A A^T) of {n_gram} matrices of {n_nodes} x {args.feat}, fp32-equivalent flops of the computed triangle; "
                                    "split-bf16 products (six piece products per fp32 product): issue-bound (DESIGN 4.7)"}}
    kr = {**rl, "kr_ridged": sb.kr_ridged, "kr_deflated": sb.kr_deflated, "kr_total": sb.kr_total, "kr_sets": KR_SETS_NOTE[sb.kr_set_mode],
          "kr_ridged_note": "train blocks the device solver found rank deficient at fp32 rounding level and solved with a ridge (within 0 - 2 "
                            "validation rows of the reference's own epochs on the sweep fixtures: profiles/r05_kr_three_way.txt); "
                            "full_metrics(ridge='pinv') solves exactly those again the reference's way on the host (utils/homophily_plot.py:"
                            "296-316) - 0 blocks on these synthetic features, so no cost to report here; the pubmed-sample fixtures flag 39 %: "
                            "tests/test_gpu_kr_epochs.py times the patch"}
    return {**kr, "workload": f"all nine scalars of the sweep job for {len(jobs)} graphs (k={args.k}, {args.seeds} seeds): + generalized edge "
                        f"homophily, KR_L and KR_NL with {args.kr_epochs} epochs each (sample_max 500: 300 train / 200 validation "
                        f"rows per regression, {sb.kr.n_jobs} regressions per batch)",
            "graphs_per_s": len(jobs) / (dev_s + tail_s), "device_ms_per_batch": dev_s * 1e3, "device_ms_slowest_replay": per_rep[-1] * 1e3,
            "replays": reps, "host_tail_ms": tail_s * 1e3,
            "mean_metrics": {n: float(v) for n, v in zip(sweep.METRIC_NAMES, rows.mean(0).tolist())}}


KR_SETS_NOTE = {
    "sample": "common random node sets per sample (WDG_SWEEP_KR_SETS=sample, opt-in): the homophily levels of one sample - same features, "
              "same labels - share each epoch's (train, validation) sets, so every job's accuracies keep the distribution of the "
              "reference's independent draws while the raw features' regression of an epoch is solved once per sample and shard instead "
              "of once per job (kr_total counts the regressions solved); the default draws per job",
    "job": "independent node sets per job (the default, the reference's own draws: utils/homophily_metrics.py:267-283): 400 regressions per job"}


def measure_cold(args, dev):
    """The sweep as the reference runs it (synthetic_plot.py:78-109): every graph is visited ONCE.  Distinct shards, each from
    host COO arrays (what the reference loads from `data_synthesis/*.pt`) to its metric rows on the host, EVERYTHING inside
    the clock: upload, the batched CSR / SELL-16 build, degrees, job tables, the aggregation step, and - nine-scalar variant -
    Gram + maps, the epochs' node sets drawn on the device, every kernel regression and the t-tests.  No tape-cut tuning
    (a one-pass batch takes the modelled cut).  One untimed shard first (code objects, allocator pools)."""
    import numpy as np
    import torch
    from wdg_amd import sweep, synth
    h_levels = synth.H_LEVELS_10 if args.k == 2 else synth.H_LEVELS_10_K10
    n_shards = args.cold_shards

    def host_inputs(first_seed):
        jobs = sweep.make_jobs(h_levels, range(first_seed, first_seed + args.seeds), k=args.k, n_nodes=args.nodes)
        feats, inputs = {}, []
        for j in jobs:
            src, dst, lab = synth.regular_graph(j.n_nodes, j.n_classes, j.k, j.h, j.seed)
            if j.seed not in feats:
                feats[j.seed] = synth.features(j.n_nodes, args.feat, j.seed)
            inputs.append((src, dst, lab, feats[j.seed]))
        return jobs, inputs

    shards = [host_inputs(1000 + args.seeds * b) for b in range(n_shards + 1)]  # (the inputs: like files on disk)
    extra_shards = [host_inputs(1000 + args.seeds * b) for b in range(n_shards + 1, max(n_shards, 12) + 1)]  # (pipelined run: >= 12 shards -
    # the first shard of a pipeline overlaps nothing; the reference's sweep is 280 adjacencies)
    out = {}
    for name, nine in (("six_scalars", False), ("nine_scalars", True)):
        phases = {"build_ms": [], "sample_ms": [], "device_ms": [], "host_tail_ms": [], "total_ms": []}
        graphs = ridged = total = deflated_n = 0
        for b, (jobs, inputs) in enumerate(shards):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            sb = sweep.SweepBatch(jobs, n_feat=args.feat, gcn_hidden=0, inputs=inputs)   # upload + batched build (one read-back)
            if nine:
                sb.prepare_full(epochs=args.kr_epochs, sample_max=500, base_seed=b)      # tables only; nothing drawn yet
            t1 = time.perf_counter()
            sb.step()
            ev = None
            if nine:  # the same launch schedule as the pipelined figure beside it (launch_full: the sampler on its side stream)
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if sb.kr_sets is not None else None
                sb.launch_full(sample_events=ev)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            rows = sb.full_metrics() if nine else sb.results().cpu()
            t3 = time.perf_counter()
            if b == 0:
                continue  # warm-up shard
            graphs += len(jobs)
            ridged, total = ridged + getattr(sb, "kr_ridged", 0), total + getattr(sb, "kr_total", 0)
            deflated_n += getattr(sb, "kr_deflated", 0)
            phases["build_ms"].append((t1 - t0) * 1e3)
            phases["device_ms"].append((t2 - t1) * 1e3)
            phases["host_tail_ms"].append((t3 - t2) * 1e3)
            phases["total_ms"].append((t3 - t0) * 1e3)
            phases["sample_ms"].append(ev[0].elapsed_time(ev[1]) if ev else 0.0)
            del sb
        total_s = sum(phases["total_ms"]) * 1e-3
        out[name] = {"graphs_per_s": graphs / total_s, **{k: sum(v) / len(v) for k, v in phases.items()},
                     **({"kr_ridged": ridged, "kr_deflated": deflated_n, "kr_total": total} if nine else {}),
                     "mean_metrics": {n: float(v) for n, v in zip(sweep.METRIC_NAMES, rows.double().mean(0).tolist())}}
        # the same shards PIPELINED (sweep.run_shards): shard b + 1's uploads and build - on a helper thread and a stream of their own -
        # while shard b's tables are built and its kernels run; clock = first upload to the last shard's rows on the host
        piped = shards[1:] + extra_shards
        dts = []
        for _rep in range(2):  # (two passes over the same shards: the first one also creates the helper thread, the build stream and their
            #                     pools - it is the COLD figure; the second is reported beside it under its own name)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            piped_rows = list(sweep.run_shards(piped, n_feat=args.feat, nine=nine, epochs=args.kr_epochs, sample_max=500, depth=2, first_seed=1))
            dts.append(time.perf_counter() - t0)
        n_graphs = sum(len(j) for j, _ in piped)
        dt = dts[0]
        out[name]["pipelined"] = {"graphs_per_s": n_graphs / dt, "graphs_per_s_second_pass": n_graphs / dts[1],
                                  "ms_per_shard": dt * 1e3 / len(piped), "shards": len(piped), "streams": 2,
                                  "rows_equal_sequential": bool(torch.equal(piped_rows[n_shards - 1].double(), rows.double()))}
    out["workload"] = (f"{n_shards} distinct shards of {len(shards[0][0])} graphs (k={args.k}, {args.seeds} seeds each, N={args.nodes}, "
                       f"F={args.feat}), each visited once: host COO -> batched build -> step -> metric rows on the host; nine_scalars adds "
                       f"Gram + maps, {args.kr_epochs} epochs of device-drawn node sets x 2 classifiers x 2 kernels = "
                       f"{len(shards[0][0]) * args.kr_epochs * 4} regressions per shard, t-tests; per-shard averages, one warm-up shard untimed")
    return out


def whole_inputs(args, pairs_needed=None):
    """the reference's sweep inputs as it finds them on disk (outside every clock): the (level, sample) adjacencies of
    `data_synthesis/4000` (only those in pairs_needed when given: a rank generates its own shard) and, per feature base, one
    feature matrix per sample"""
    from wdg_amd import sweep, synth
    levels = [h for h in synth.H_LEVELS_30 if h not in (0.05, 0.1)]  # (the levels `data_synthesis/4000` holds: SURVEY Appendix B.2)
    samples = list(range(10))
    bases = sweep.BaseSweep.REFERENCE_BASES
    pairs = sweep.make_jobs(levels, samples, k=10, n_nodes=args.nodes)
    need = pairs if pairs_needed is None else pairs_needed
    graphs = {(j.h, j.seed): synth.regular_graph(args.nodes, 5, 10, j.h, j.seed) for j in need}
    feats = [(name, {s_: synth.features(args.nodes, width, 7000 + 100 * bi + s_) for s_ in samples}, 300 if name in ("chameleon", "film") else 500)
             for bi, (name, width) in enumerate(bases)]
    return dict(levels=levels, samples=samples, bases=bases, pairs=pairs, graphs=graphs, feats=feats)


def measure_whole(args, dev, world=1, rank=0):
    """The sweep the reference actually runs (synthetic_plot.py:64-109): 6 feature bases x the homophily levels of
    `data_synthesis/4000` x 10 samples, all nine scalars per job, every job visited ONCE - host COO arrays and host feature
    matrices in, metric rows on the host out, everything in between inside the clock (uploads, the batched graph build - once
    per (level, sample), shared by the six bases -, aggregation at the base's width, Grams, device-drawn node sets, regressions,
    t-tests).  BASELINE.md: ~35 s per job on the reference's CPU path, ~17 h for the sweep.

    world > 1 (`bench.py --gpus N`): STRONG scaling of that sweep - the 280 adjacencies dealt to the ranks by sweep.shard_pairs (contiguous, sample-aware),
    every rank runs the six bases over its share (sweep.whole_sweep_rank), the [jobs, 9] rows are exchanged once
    (sweep.exchange_rows: one all_gather), `seconds` = the slowest rank (barrier, clock, all_reduce MAX) - and rank 0 then runs the
    WHOLE sweep alone on its GPU (the other ranks wait), so that `strong_speedup_vs_1gpu` compares two measurements of one
    session on one box and the assembled table can be checked against the one-GPU rows bit for bit."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from wdg_amd import sweep
    t_in = time.perf_counter()
    from wdg_amd import synth
    pairs_all = sweep.make_jobs([h for h in synth.H_LEVELS_30 if h not in (0.05, 0.1)], range(10), k=10, n_nodes=args.nodes)
    pairs = sweep.broadcast_jobs(pairs_all if rank == 0 else [], dev)  # rank 0's job table, like the step's
    mine = sweep.pairs_of_rank(pairs, world, rank)
    inp = whole_inputs(args, pairs if rank == 0 else mine)  # (rank 0 also runs the one-GPU reference pass: it needs every graph)
    t_in = time.perf_counter() - t_in
    feats, bases, graphs = inp["feats"], inp["bases"], inp["graphs"]
    n_rows = len(pairs) * len(bases)
    # adjacencies per shard on ONE GPU: about --whole-levels-per-shard x samples (70: 16-MB kernels x 4 x 70 = 18 GB), rounded down to
    # WHOLE samples - the job list is sample-major, and a shard that ends inside a sample makes the next shard compute that sample's
    # raw-feature kernels and regressions a second time (56 = 2 samples x 28 levels: 1.5 % faster than 70)
    per_shard = max(1, args.whole_levels_per_shard * len(inp["samples"]) // len(inp["levels"])) * len(inp["levels"])

    def graph_of(j):
        return graphs[(j.h, j.seed)]

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # warm-up outside the clock (code objects, scipy import, torch's allocator pools, the upload ring and the per-stream scratch at
    # the sizes the timed pass will ask for): this rank's first shard-sized piece of its share over EVERY base - each base has its
    # own width, i.e. its own feature-matrix buffers, and the bases of a shard take turns on two prepared batches (sweep.run_bases)
    # ... and then the pass itself once, untimed in `seconds` (the contract's warm-up step: the same work, outside the clock) but
    # reported as `first_pass_seconds`: a first pass grows torch's allocator pools to the two shards the pipeline keeps alive
    # (hipMalloc inside the pass) - a later pass of the same process finds them
    piece = (mine[:per_shard] if world == 1 else mine[:80]) or pairs[:4]
    sweep.whole_sweep_rank(piece, graph_of, feats, 1, 0, epochs=args.kr_epochs, max_pairs_per_shard=len(piece))
    torch.cuda.synchronize()
    t_first = time.perf_counter()
    sweep.whole_sweep_rank(pairs, graph_of, feats, world, rank, epochs=args.kr_epochs, max_pairs_per_shard=per_shard if world == 1 else 80)
    torch.cuda.synchronize()
    t_first = time.perf_counter() - t_first
    per_base = np.zeros(len(bases))
    state = {"t": 0.0}

    def progress(_si, bi, _rows):
        now = time.perf_counter()
        per_base[bi] += now - state["t"]  # (pipelined: what arrived between two fetches, attributed to the base fetched)
        state["t"] = now

    sync_all()
    t0 = state["t"] = time.perf_counter()
    kr_stats = {}
    keys, rows = sweep.whole_sweep_rank(pairs, graph_of, feats, world, rank, epochs=args.kr_epochs,
                                        max_pairs_per_shard=per_shard if world == 1 else 80, progress=progress, stats=kr_stats)
    torch.cuda.synchronize()
    mine_s = time.perf_counter() - t0
    table = sweep.exchange_rows(keys, rows, n_rows, dev)
    sync_all()
    dt = time.perf_counter() - t0
    rec = {}
    kr_counts = torch.tensor([kr_stats.get("kr_ridged", 0), kr_stats.get("kr_total", 0), kr_stats.get("kr_deflated", 0)], dtype=torch.int64, device=dev)
    if world > 1:
        dist.all_reduce(kr_counts)
        t = torch.tensor([dt, mine_s], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0].item())
        per_rank = [torch.zeros(2, dtype=torch.float64, device=dev) for _ in range(world)]
        dist.all_gather(per_rank, torch.tensor([mine_s, float(len(mine))], dtype=torch.float64, device=dev))
        rec["per_rank_s"] = [round(float(x[0].item()), 4) for x in per_rank]
        rec["adjacencies_per_rank"] = [int(x[1].item()) for x in per_rank]
        # the one-GPU pass of the SAME sweep in the same session (rank 0 alone; everybody else waits at the barrier)
        if rank == 0:
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            k1, r1 = sweep.whole_sweep_rank(pairs, graph_of, feats, 1, 0, epochs=args.kr_epochs, max_pairs_per_shard=per_shard)
            torch.cuda.synchronize()
            one = time.perf_counter() - t1
            t1gpu = torch.full((n_rows, rows.shape[1]), float("nan"), dtype=torch.float64)
            t1gpu[k1] = r1
            rec.update(one_gpu_s=one, strong_speedup_vs_1gpu=one / dt,
                       rows_equal_one_gpu=bool(torch.equal(torch.nan_to_num(table, nan=-7.0), torch.nan_to_num(t1gpu, nan=-7.0))))
        dist.barrier()
    mode = os.environ.get("WDG_SWEEP_KR_SETS", "job")
    if world == 1:
        # the same sweep in the OTHER node-set mode, beside the headline figure (the second of two passes).  Default: independent node
        # sets per job - the reference's own draws, utils/homophily_metrics.py:267-283 (672 000 regressions) -; the other mode
        # (WDG_SWEEP_KR_SETS=sample) shares the sets inside a sample and solves 348 000 (an optimisation with a changed estimator:
        # opt-in since round 6)
        other = "sample" if mode == "job" else "job"
        prev = os.environ.get("WDG_SWEEP_KR_SETS")
        os.environ["WDG_SWEEP_KR_SETS"] = other
        try:
            stats_other = {}
            for _rep in range(2):
                stats_other.clear()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                sweep.whole_sweep_rank(pairs, graph_of, feats, 1, 0, epochs=args.kr_epochs, max_pairs_per_shard=per_shard, stats=stats_other)
                torch.cuda.synchronize()
                rec[f"seconds_sets_{other}"] = time.perf_counter() - t1
            rec[f"kr_total_sets_{other}"] = int(stats_other.get("kr_total", 0))
        finally:
            if prev is None:
                os.environ.pop("WDG_SWEEP_KR_SETS", None)
            else:
                os.environ["WDG_SWEEP_KR_SETS"] = prev
    args._whole = dict(inp, seconds=dt)  # (measure_projection shards the same inputs)
    shards_txt = (f"{-(-len(pairs) // per_shard)} shards of {per_shard} adjacencies" if world == 1 else
                  f"the {len(pairs)} adjacencies dealt to {world} ranks by sweep.shard_pairs (one shard per rank), rows exchanged by one all_gather, "
                  "`seconds` = slowest rank incl. the exchange")
    return {"workload": f"synthetic_plot.py's sweep: {len(bases)} feature bases ({', '.join(f'{n} F={w}' for n, w in bases)}) x {len(inp['levels'])} homophily "
                        f"levels x {len(inp['samples'])} samples = {n_rows} jobs, N={args.nodes}, k=10, all nine scalars, {args.kr_epochs} epochs per "
                        f"classifier; {shards_txt}, graphs built once per shard and shared by the six bases; host COO + host features -> rows on "
                        "the host, two HIP streams per rank",
            "jobs": n_rows, "n_gpus": world, "scaling": "strong", "seconds": dt, "graphs_per_s": n_rows / dt,
            "first_pass_seconds": t_first, "first_pass_note": "this rank's untimed first pass of the same sweep in this process (allocator pools still growing)", **rec,
            "kr_ridged": int(kr_counts[0].item()), "kr_total": int(kr_counts[1].item()), "kr_deflated": int(kr_counts[2].item()),
            "feature_duplicates": args.dup_frac,
            "kr_sets_mode": mode, "kr_sets": KR_SETS_NOTE[mode],
            "reuse_inside_a_shard": "what depends on less than a job is computed once (DESIGN 5): the six step scalars of the wide bases (they "
                                    "aggregate the label columns only: graph + labels) by the first of them (WDG_SWEEP_STEP_TWINS=0: every base), "
                                    "the job tables of a base taken over from an earlier base of equal sample_max (WDG_SWEEP_REBIND=0: built "
                                    "per base), the raw features' regressions once per sample (kr_sets); every row equals the stand-alone "
                                    "batch's bit for bit with or without (tests/test_gpu_sweep.py)",
            "ms_per_base_rank0": {n: 1e3 * t for (n, _w), t in zip(bases, per_base)},
            "input_generation_s_outside_clock": t_in, "rows": list(table.shape), "nan_rows": int(torch.isnan(table).any(1).sum()),
            "reference_cpu_estimate": "~35 s per job, ~17 h for the sweep (BASELINE.md)",
            "mean_metrics": {n: float(v) for n, v in zip(sweep.METRIC_NAMES, table.nanmean(0).tolist())}}


def measure_configs(args, dev):
    """BASELINE.json configs[0], [3], [4] as stated (+ the literal N = 4000 reading of configs[2]) in compact form: what each
    launch takes, on which kernel, at what fraction of its roofline.  The workloads are scripts/bench_configs.py's."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import bench_configs as bc
    reps = args.config_reps

    def agg(rec):
        r = rec["roofline"]
        return {"workload": rec["workload"], "kernel": rec["kernel"], "us": round(r["avg_launch_us"], 2), "edges_per_s": rec["edges_per_s"],
                "roofline": {"bound": r["bound"], "frac": round(r["frac"], 4), "achieved": round(r["achieved"], 1), "peak": r["peak"], "unit": r["unit"]},
                # the resource that really binds the launch (L2 gathers / LDS reads), beside SURVEY 8(d)'s HBM accounting above
                **({"binding": {k: (round(v, 4) if isinstance(v, float) else v) for k, v in rec["binding"].items()}} if "binding" in rec else {}),
                **({"whole_step_us": round(rec["whole_step_us"], 1)} if "whole_step_us" in rec else {})}

    def model(rec):
        out = {"workload": rec["workload"], **({"forward_us": round(rec["forward_us"], 2)} if "forward_us" in rec else {}),
               **({"forward_graph_us": round(rec["forward_graph_us"], 2)} if "forward_graph_us" in rec else {}),
               "legs": [{"leg": l["leg"], "us": round(l["us"], 2), "bound": l["roofline"]["bound"], "frac": round(l["roofline"]["frac"], 4)} for l in rec["legs"]]}
        for k in ("metric_us_10_samples", "agg_homo_soft"):
            if k in rec:
                out[k] = rec[k]
        return out

    out = {}
    out["C1 cora aggregation"] = agg(bc.c1(reps))
    c1m = bc.c1_models(reps)
    out["C1 cora SGC-1 (A X) W"], out["C1 cora SGC-1 A (X W)"] = model(c1m[0]), model(c1m[1])
    torch.cuda.empty_cache()
    for name, f in (("squirrel", 2089), ("chameleon", 2325)):
        out[f"C4 {name} aggregation"] = agg(bc.c4(name, f, reps))
        m = bc.c4_models(name, f, reps)
        out[f"C4 {name} GCN-2"], out[f"C4 {name} SGC-1 (A X) W"], out[f"C4 {name} SGC-1 A (X W)"] = model(m[0]), model(m[1]), model(m[2])
        torch.cuda.empty_cache()
    sym, rw = bc.c5(reps)
    out["C5 twitch aggregation sym bf16"], out["C5 twitch aggregation rw bf16"] = agg(sym), agg(rw)
    m = bc.c5_models(reps)
    out["C5 twitch agg-homophily (10 samples)"], out["C5 twitch SGC-1 (A X) W"], out["C5 twitch SGC-1 A (X W)"] = model(m[0]), model(m[1]), model(m[2])
    torch.cuda.empty_cache()
    out["C3-literal N=4000"] = agg(bc.literal("C3-literal", 4000, 10, 5, reps))
    torch.cuda.empty_cache()
    out["C2-literal N=800"] = agg(bc.literal("C2-literal", 800, 2, 10, reps))
    torch.cuda.empty_cache()
    out["C3-scaled N=2^17 F=512"] = agg(bc.scaled(reps))
    torch.cuda.empty_cache()
    return out


def measure_train(args, dev):
    """Train + eval sweep (SURVEY 8(d) "separately train+eval when the model loop exists"; row N4, gnns_on_syn.py:9-154): one
    model per graph of the C3 shard, all graphs per launch, an epoch (train step + evaluation + selection) = one hipGraph."""
    import torch
    from wdg_amd import sweep, synth
    h_levels = synth.H_LEVELS_10 if args.k == 2 else synth.H_LEVELS_10_K10
    jobs = sweep.make_jobs(h_levels, range(args.seeds), k=args.k, n_nodes=args.nodes)
    sb = sweep.SweepBatch(jobs, n_feat=args.feat, gcn_hidden=0)
    for s_ in sb.x:  # features that carry the class (the reference's synthetic features are sampled from real nodes of the class)
        lab = synth.regular_graph(args.nodes, 5, args.k, 0.5, s_)[2]
        sb.x[s_].copy_(torch.from_numpy(synth.features(args.nodes, args.feat, s_, labels=lab)))
    out = {"workload": f"train + eval of one model per graph, {len(jobs)} graphs (k={args.k}, {args.seeds} seeds, N={args.nodes}, F={args.feat}), "
                       f"Adam, 60/20/20 split, an epoch = train step + evaluation + model selection, replayed as one hipGraph; "
                       f"{args.train_epochs} epochs timed, graphs/s quoted at the reference's 200 epochs"}
    for kind, name in (("sgc", "SGC-1"), ("gcn", "GCN-2")):
        tb = sweep.TrainBatch(sb, kind=kind, hidden=64, seed=1)
        tb.run(epochs=3, capture=True)  # (capture + first replays)
        r = tb.run(epochs=args.train_epochs, capture=True)
        ms = r["seconds"] / r["epochs"] * 1e3
        out[name] = {"ms_per_epoch": ms, "graphs_per_s_at_200_epochs": tb.J / (ms * 1e-3 * 200), "edges_per_s": sb.edges / (ms * 1e-3),
                     "mean_test_acc": float(r["test_acc"].mean())}
        del tb
        torch.cuda.empty_cache()
    return out


def measure_gnb(args, dev):
    """The GNB classifier-based metric (homophily_tests.py:133-137 with `gnb_based_homo`; utils/homophily_metrics.py:296-312) on a
    Cora-sized input - 2 709 nodes, 1 433 row-normalised features, 7 classes, sample_max 500, 100 epochs = 200 fits + 200 predictions:
    the device call (node sets on torch's CPU generator like the reference, aggregation, ONE wdg_gnb_batched_f32 call, t-test) beside the
    reference's own route for the same call - scikit-learn on the host, epoch by epoch (WDG_GNB_SOLVER=host; same sets, same p-value)."""
    import numpy as np
    import torch
    from wdg_amd import synth
    from wdg_amd.utils import homophily_metrics as hm
    n, f, c, epochs = 2709, 1433, 7, 100  # (7 x 387 nodes: the generator's classes are contiguous blocks of equal size)
    src, dst, lab = synth.regular_graph(n, c, 4, 0.8, 3)
    x = synth.features(n, f, 11, labels=lab, n_classes=c)
    x = np.abs(x)
    x /= np.maximum(x.sum(1, keepdims=True), 1e-12)
    idx = torch.from_numpy(np.stack([src, dst]).astype(np.int64))
    adj = torch.sparse_coo_tensor(idx, torch.ones(idx.shape[1]), (n, n)).coalesce()
    feats, labels = torch.from_numpy(x.astype(np.float32)), torch.from_numpy(np.asarray(lab, np.int64))
    out = {"workload": f"classifier_based_performance_metric(base_classifier='gnb'): N={n}, F={f}, C={c}, sample_max 500, {epochs} epochs"}
    prev = os.environ.get("WDG_GNB_SOLVER")
    try:
        for mode, reps in (("device", 3), ("host", 1)):
            os.environ["WDG_GNB_SOLVER"] = mode
            best = None
            for _ in range(reps):
                torch.manual_seed(3)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                p, _secs = hm.classifier_based_performance_metric(feats, adj, labels, 500.0, base_classifier="gnb", epochs=epochs)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                best = dt if best is None else min(best, dt)
            out[f"{mode}_ms"], out[f"p_{mode}"] = best * 1e3, float(p)
    finally:
        if prev is None:
            os.environ.pop("WDG_GNB_SOLVER", None)
        else:
            os.environ["WDG_GNB_SOLVER"] = prev
    out["p_equal"] = bool(abs(out["p_device"] - out["p_host"]) <= 1e-12)
    return out


def measure_projection(args, dev, full_ms):
    """Strong scaling PROJECTED from one device: configs[2]'s 50 jobs sharded by sweep.shard_jobs for world sizes 2 / 4 / 8, every
    rank's shard built and stepped on THIS GPU (plain launches and the whole step replayed from one hipGraph; the faster is
    taken, as a rank would).  speedup = T(all jobs on one GPU) / max over ranks T(shard).  No second device is involved: what
    this cannot see is contention for the host (8 processes enqueueing) and the exchange at the end (KBs)."""
    import torch
    from wdg_amd import sweep, synth
    h_levels = synth.H_LEVELS_10 if args.k == 2 else synth.H_LEVELS_10_K10
    jobs = sweep.make_jobs(h_levels, range(args.seeds), k=args.k, n_nodes=args.nodes)
    steps = args.projection_steps

    def time_steps(fn):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

    out = {"note": "PROJECTION from one GPU, not a measurement on N devices", "jobs": len(jobs), "one_gpu_ms_per_step": full_ms, "worlds": {}}
    for w in (2, 4, 8):
        per_rank = []
        for r in range(w):
            mine = sweep.shard_jobs(jobs, w, r)
            sb = sweep.SweepBatch(mine, n_feat=args.feat)
            plain = time_steps(sb.step)
            graph = time_steps(sb.capture_step()) if mine else plain
            per_rank.append({"graphs": len(mine), "plain_ms": plain, "graph_ms": graph})
            del sb
            torch.cuda.empty_cache()
        worst = max(min(p["plain_ms"], p["graph_ms"]) for p in per_rank)
        out["worlds"][str(w)] = {"graphs_per_rank": [p["graphs"] for p in per_rank],
                                 "ms_per_step_per_rank_plain": [round(p["plain_ms"], 4) for p in per_rank],
                                 "ms_per_step_per_rank_graph": [round(p["graph_ms"], 4) for p in per_rank],
                                 "slowest_rank_ms": worst, "projected_strong_speedup": full_ms / worst,
                                 "projected_strong_speedup_plain_launches": full_ms / max(p["plain_ms"] for p in per_rank),
                                 "weak_scaling": f"{w}.0 x by construction (every rank steps its own {len(jobs)}-graph shard; no data-path collective)"}
    out["reading"] = ("the 50-job step is 0.2 ms of work on ONE GPU: a 6-graph shard still pays the step's chain of four dependent launches "
                      "(~90 us: each kernel's dispatch + staging + drain), so the literal strong split of configs[2] cannot reach 6 x whatever "
                      "the kernels do; the sweep the reference runs is 1 680-1 800 jobs, below")
    whole = getattr(args, "_whole", None)
    if whole is not None:
        # the reference's sweep sharded by (level, sample) adjacency - every rank runs the six bases over its adjacencies
        pairs = whole["pairs"]
        ws = {}
        for w in (2, 4, 8):
            secs = []
            for r in range(w):  # EVERY rank's share (round 4 timed one rank at W = 2, 4: a single sample, visibly jitter-prone)
                best = None
                for _rep in range(3):  # (the fastest of three passes: a rank of a real node has run its share's shapes in its warm-up
                    #                     passes - measure_whole makes two before its clock - while here every rank brings new shapes to the
                    #                     allocator (round 6: the first rank timed was 5 ms slower than the seven after it with two passes);
                    #                     and one stall of this shared host's enqueueing thread - 0.14 s seen - is not a rank's time)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    keys, _rows = sweep.whole_sweep_rank(pairs, lambda j: whole["graphs"][(j.h, j.seed)], whole["feats"], w, r,
                                                         epochs=args.kr_epochs)
                    torch.cuda.synchronize()
                    dt_ = time.perf_counter() - t0
                    best = dt_ if best is None else min(best, dt_)
                secs.append(best)
                assert keys.shape[0] == len(sweep.pairs_of_rank(pairs, w, r)) * len(whole["feats"])
            ws[str(w)] = {"ranks_timed": len(secs), "adjacencies_per_rank": [len(sweep.pairs_of_rank(pairs, w, r)) for r in range(w)],
                          "per_rank_s": [round(x, 4) for x in secs], "slowest_rank_s": max(secs),
                          "projected_strong_speedup": whole["seconds"] / max(secs)}
        out["whole_sweep"] = {"jobs": len(pairs) * len(whole["feats"]), "one_gpu_s": whole["seconds"], "worlds": ws,
                              "note": "PROJECTION: each rank's share of the 1 680-job sweep run on this GPU, one after the other (every rank of every world size "
                                      "timed); `bench.py --gpus N` MEASURES the same split: its `sweep_whole.strong_speedup_vs_1gpu`"}
    return out


def traffic_for(n_graphs, n_feat, k, n_nodes=2000):
    """PMC-derived HBM bytes per aggregation launch of the same workload (profiles/traffic.json: one record per k)."""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        tj = json.load(open(tpath))
    except Exception:
        return None
    for rec in (tj if isinstance(tj, list) else [tj]):
        if (rec.get("graphs_per_launch") == n_graphs and rec.get("n_feat") == n_feat and rec.get("k") == k
                and rec.get("n_nodes", 2000) == n_nodes):
            return rec.get("hbm_bytes_per_launch")
    return None


def roofline_of(args, m):
    batch = m["batch"]
    spmm_ms = m["spmm_ms"]
    spmm_avg_ms = sum(spmm_ms) / len(spmm_ms)
    alg = batch.spmm_algorithmic_bytes()
    achieved = alg / (spmm_avg_ms * 1e-3) / 1e9
    traffic = traffic_for(len(m["mine"]), args.feat, m["k"], args.nodes)
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            # the same launch priced by the bytes the chip really moved (PMC): X is fetched once per group of graphs that
            # share it, the algorithmic figure of SURVEY 8(d) counts it once per graph
            "frac_traffic": (traffic / (spmm_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
            "kernel": batch.spmm.kernel_name(),
            "avg_launch_us": spmm_avg_ms * 1e3, "median_launch_us": spmm_ms[len(spmm_ms) // 2] * 1e3,
            "algorithmic_bytes_per_launch": alg, "unique_bytes_per_launch": batch.spmm_unique_bytes(),
            "launches_timed": len(spmm_ms), "launches_in_timed_region": m["spmm_in_region"],
            # the same launch with the modelled tape cut, before SweepBatch.tune() balanced it by feedback (one-pass sweeps run this)
            "untuned_launch_us": sum(m["untuned_ms"]) / len(m["untuned_ms"]) * 1e3,
            "untuned_frac": alg / (sum(m["untuned_ms"]) / len(m["untuned_ms"]) * 1e-3) / 1e9 / HBM_PEAK_GBS}


def workload_text(args, m, world):
    batch = m["batch"]
    n_launches = 3 + (not batch.derive_counts) + (batch.spmm_las is not None) + (1 if batch.gcn["mlp"] is not None else 2)
    return (f"synthetic homophily sweep (data_synthesis/{m['k'] * 400}-equivalent): "
            + (f"{len(m['h_levels'])} h-levels x {m['seeds']} seeds = {len(m['h_levels']) * m['seeds']} graphs/step sharded by job over "
               f"{world} GPU(s) ({len(m['mine'])} on rank 0), " if args.scaling == "strong" else
               f"{len(m['h_levels'])} h-levels x {m['seeds']} seeds = {len(m['mine'])} graphs/GPU/step, ") +
            f"N={args.nodes} nodes, k={m['k']}, F={args.feat} fp32, C=5; step = batched "
            f"D^-1(A+I)X aggregation + edge/label statistics ("
            + ("derived from the label columns inside the LAS launch" if batch.derive_counts else "integer pass over the edges")
            + ") + label aggregation & LAS + "
            f"GCN-2 forward (hidden 64, per-graph weights), {n_launches} launches"
            + (f"; the C one-hot label columns of the LAS metric ride in the feature aggregation "
               f"(F_agg={batch.agg_feat})" if batch.spmm_las is None else ""))


LINE_LIMIT = 6000  # bytes; the driver parses ONE stdout line - r05's 23-KB line came back `parsed: null`


def _short(text, n=160):
    text = str(text)
    return text if len(text) <= n else text[:n - 3] + "..."


def _num(v, digits=6):
    """floats rounded to `digits` significant digits (the line is for a parser, the full precision is in the detail file)"""
    if isinstance(v, float):
        return float(f"{v:.{digits}g}") if v == v and abs(v) != float("inf") else None
    return v


def compact_line(out, detail_path=None):
    """The ONE stdout line: the contract's keys, `roofline` and `cpu_baseline` whole (numbers only + one short text each) and one
    number per side block; everything else (workload prose, per-config records, projections) goes to the detail file.  Pure
    function of the record (tests/test_bench_line.py feeds it a canned r05 record and asserts size and strict JSON)."""
    line = {k: _num(out.get(k)) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                          "scaling", "vs_baseline", "dtype", "data")}
    cfg = out.get("config", {})
    line["config"] = {"workload": _short(cfg.get("workload", ""), 200),
                      **{k: cfg[k] for k in ("graphs_per_step_per_gpu", "edges_per_step_per_gpu") if k in cfg},
                      "parallelism": _short(cfg.get("parallelism", ""), 80)}
    rl = out.get("roofline") or {}
    line["roofline"] = {k: _num(rl.get(k)) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "frac_traffic", "kernel",
                                                     "avg_launch_us", "median_launch_us", "launches_timed", "launches_in_timed_region",
                                                     "algorithmic_bytes_per_launch", "unique_bytes_per_launch", "untuned_launch_us",
                                                     "untuned_frac") if k in rl}
    cb = out.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {**{k: _num(cb.get(k)) for k in ("value", "unit", "cores", "kind", "physical_cores", "value_1_thread",
                                                                "value_physical_cores", "cpu_model") if k in cb},
                                "sample": _short(cb.get("sample", ""), 220)}
    sec = out.get("secondary")
    if sec:
        line["secondary"] = {"ms_per_step": _num(sec.get("ms_per_step")), "value": _num(sec.get("value")),
                             "spmm_us": _num(sec.get("roofline", {}).get("avg_launch_us")), "frac": _num(sec.get("roofline", {}).get("frac"))}
    sw = out.get("sweep_whole")
    if sw:
        line["sweep_whole"] = {k: _num(sw.get(k)) for k in ("jobs", "n_gpus", "scaling", "seconds", "first_pass_seconds", "kr_sets_mode", "seconds_sets_job", "seconds_sets_sample",
                                                            "kr_total", "kr_ridged", "kr_deflated", "one_gpu_s", "strong_speedup_vs_1gpu",
                                                            "rows_equal_one_gpu", "per_rank_s") if k in sw}
    sf = out.get("sweep_full")
    if sf:
        line["sweep_full"] = {"graphs_per_s": _num(sf.get("graphs_per_s")), "device_ms_per_batch": _num(sf.get("device_ms_per_batch")),
                              "solver_ms": _num(sf.get("roofline_solver", {}).get("ms_per_launch")),
                              "solver_frac_mfma": _num(sf.get("roofline_solver", {}).get("frac")),
                              "kr_total": sf.get("kr_total"), "kr_ridged": sf.get("kr_ridged"), "kr_deflated": sf.get("kr_deflated")}
    sc = out.get("sweep_cold")
    if sc:
        line["sweep_cold"] = {name: _num(sc[name].get("pipelined", {}).get("graphs_per_s", sc[name].get("graphs_per_s")))
                              for name in ("six_scalars", "nine_scalars") if name in sc}
    tr = out.get("train")
    if tr:
        line["train"] = {name: _num(tr[name]["ms_per_epoch"]) for name in ("SGC-1", "GCN-2") if name in tr}
        line["train"]["unit"] = "ms/epoch"
    gn = out.get("gnb")
    if gn:  # the GNB metric of one Cora-sized call: [device ms, scikit-learn on the host ms, p-values equal]
        line["gnb_metric_ms"] = [_num(gn.get("device_ms"), 4), _num(gn.get("host_ms"), 4), gn.get("p_equal")]
    cf = out.get("configs")
    if cf:  # per aggregation config: [launch us, HBM fraction by SURVEY 8(d) bytes]
        line["configs"] = {name: [_num(rec["us"], 4), _num(rec["roofline"]["frac"], 3)] for name, rec in cf.items() if "us" in rec and "roofline" in rec}
    pr = out.get("scaling_projection", {}).get("whole_sweep")
    if pr:
        line["projection_8"] = {"kind": "projection from one GPU", "speedup": _num(pr["worlds"].get("8", {}).get("projected_strong_speedup"), 4),
                                "slowest_rank_s": _num(pr["worlds"].get("8", {}).get("slowest_rank_s"), 4)}
    if detail_path:
        line["detail"] = detail_path
    text = json.dumps(line, allow_nan=False)
    if len(text) > LINE_LIMIT:  # never let a side block cost the headline: drop them in this order
        for k in ("gnb_metric_ms", "configs", "sweep_cold", "sweep_full", "projection_8", "train", "secondary", "sweep_whole"):
            line.pop(k, None)
            text = json.dumps(line, allow_nan=False)
            if len(text) <= LINE_LIMIT:
                break
    return text


def write_detail(out):
    """the full record (what rounds 1-5 printed on the line) -> gpurun_out/bench_detail.json; returns the path relative to the repo"""
    rel = os.path.join("gpurun_out", "bench_detail.json")
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, rel), "w") as f:
            json.dump(out, f, indent=1)
        return rel
    except OSError as e:
        print(f"bench.py: could not write {rel}: {e}", file=sys.stderr)
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--seeds", type=int, default=5, help="seeds per rank (x 10 homophily levels = graphs per step)")
    ap.add_argument("--nodes", type=int, default=2000)
    ap.add_argument("--feat", type=int, default=500)
    ap.add_argument("--k", type=int, default=10, help="same-class out-neighbours per node (10 = `4000` set, 2 = `800` set)")
    ap.add_argument("--cpu-budget", type=float, default=12.0, help="seconds of CPU baseline work (0 = skip)")
    ap.add_argument("--secondary", type=int, default=1, help="1: also time the `800` set (k=2, 10 seeds) and report it as `secondary` (N=1 only)")
    ap.add_argument("--full-metrics", type=int, default=1, help="1: also time the whole nine-scalar sweep job batch (adds generalized edge "
                    "homophily and the kernel-regression p-values, --kr-epochs epochs) and report it as `sweep_full` (N=1 only)")
    ap.add_argument("--kr-epochs", type=int, default=100)
    ap.add_argument("--cold", type=int, default=1, help="1: also time the one-pass (cold) sweep - distinct shards from host COO to "
                    "metric rows, everything inside the clock - and report it as `sweep_cold` (N=1 only)")
    ap.add_argument("--cold-shards", type=int, default=3)
    ap.add_argument("--whole", type=int, default=1, help="1: also run the reference's whole sweep (6 feature bases x 28 levels x 10 samples = 1 680 "
                    "jobs, nine scalars, one pass) and report it as `sweep_whole` (N > 1: sharded over the ranks, + rank 0's one-GPU pass -> strong_speedup_vs_1gpu)")
    ap.add_argument("--whole-levels-per-shard", type=int, default=7)
    ap.add_argument("--configs", type=int, default=1, help="1: also time BASELINE configs[0], [3], [4] as stated and the literal N = 4000 "
                    "reading of configs[2] and report them as `configs` (N=1 only)")
    ap.add_argument("--config-reps", type=int, default=10)
    ap.add_argument("--train", type=int, default=1, help="1: also time the train + eval sweep (SGC-1, GCN-2) on the shard and report it as `train` (N=1 only)")
    ap.add_argument("--train-epochs", type=int, default=30)
    ap.add_argument("--gnb", type=int, default=1, help="1: also time the GNB classifier-based metric on a Cora-sized input, device call beside scikit-learn on the host (N=1 only)")
    ap.add_argument("--projection", type=int, default=1, help="1: also step every rank's shard of the 50-job sweep for world sizes 2 / 4 / 8 on "
                    "this GPU and report the projected strong scaling as `scaling_projection` (N=1 only)")
    ap.add_argument("--projection-steps", type=int, default=100)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak", help="weak: --seeds seeds PER RANK (per-GPU work fixed); "
                    "strong: --seeds seeds in all (configs[2] literally: 50 jobs), sharded over the ranks")
    ap.add_argument("--timeout", type=float, default=1500.0, help="self-launched workers (--gpus N > 1) are stopped after this many seconds")
    ap.add_argument("--dup-frac", type=float, default=0.033, help="share of the synthetic feature rows that copy another row of their class - the "
                    "reference's features are real nodes' rows sampled per class with replacement (pubmed sample: 3.3 %%): such rows make "
                    "kernel-regression train blocks exactly singular, which the solver deflates (0: no duplicates, rounds 1 - 5)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_workers(args, sys.argv[1:]))  # the parent never initialises the GPU

    import torch
    import torch.distributed as dist

    from wdg_amd import synth as _synth
    _synth.DUPLICATE_FRACTION = args.dup_frac  # (every feature matrix of every block below)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs an MI355X; there is no CPU path"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)  # RCCL

    m = measure(args, args.k, args.seeds, args.steps, args.warmup, world, rank, dev)
    if rank == 0:
        batch = m["batch"]
        out = {
            "metric": "aggregation edges/sec (whole job; + %HBM roofline of the SpMM kernel)",
            "value": m["total_edges"] * args.steps / m["elapsed"],
            "unit": "edges/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": m["elapsed"] / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload_text(args, m, world),
                       "graphs_per_step_per_gpu": len(m["mine"]), "edges_per_step_per_gpu": batch.edges,
                       "parallelism": f"independent sweep shards x{world} (no data-path collective; job-table broadcast + result all_gather)",
                       # what the step's GCN-2 feature transform computes in (the aggregation itself: fp32 sources, fp32 adds)
                       "feature_transform": ("fp32 chain on v_mfma_f32_32x32x2_f32 (WDG_MLP2_SPLIT=0)" if os.environ.get("WDG_MLP2_SPLIT", "1") == "0" else
                                             "fp32 products from 3 bf16 pieces per operand (6 piece products on v_mfma_f32_16x16x32_bf16, fp32 accumulation; "
                                             "measured error against fp64 below the fp32 chain's - DESIGN 4.7; WDG_MLP2_SPLIT=0 for the chain)"),
                       # where the aggregation's output lives (values and sum orders are the row-major batch's bit for bit)
                       "aggregation_output": ("tiled by 16-feature groups: one contiguous [n, 16] plane per workgroup of the aggregation; the "
                                              "fused transform and LAS read it in place (DESIGN 3, 4.1; WDG_SWEEP_TILED_Y=0: row-major)"
                                              if getattr(batch, "tiled_y", False) else "row-major [n, F_agg]")},
            "graphs_per_s": m["n_graphs"] * args.steps / m["elapsed"],
            "edge_features_per_s": m["total_edges"] * args.feat * args.steps / m["elapsed"],
            "host_enqueue_ms_per_step": m["enqueue_s"] / args.steps * 1e3,
            "roofline": roofline_of(args, m),
        }
    del m
    if world == 1 and args.secondary and not (args.k == 2 and args.seeds == 10):
        torch.cuda.empty_cache()
        steps2 = max(20, args.steps // 2)
        m2 = measure(args, 2, 10, steps2, min(args.warmup, 10), world, rank, dev)
        out["secondary"] = {"workload": workload_text(args, m2, world), "steps": steps2,
                            "value": m2["total_edges"] * steps2 / m2["elapsed"], "unit": "edges/s",
                            "ms_per_step": m2["elapsed"] / steps2 * 1e3,
                            "graphs_per_s": m2["n_graphs"] * steps2 / m2["elapsed"],
                            "roofline": roofline_of(args, m2)}
        del m2
    if world > 1:
        if args.whole:  # the reference's sweep, strong-scaled over the ranks (every rank takes part; rank 0 keeps the record)
            torch.cuda.empty_cache()
            whole = measure_whole(args, dev, world, rank)
            if rank == 0:
                out["sweep_whole"] = whole
        dist.destroy_process_group()  # (before the line: RCCL writes its library path to stdout when it shuts down)
    if world == 1 and args.full_metrics:
        out["sweep_full"] = measure_full(args, dev)
    if world == 1 and args.cold:
        torch.cuda.empty_cache()
        out["sweep_cold"] = measure_cold(args, dev)
    if world == 1 and args.whole:
        torch.cuda.empty_cache()
        out["sweep_whole"] = measure_whole(args, dev)
    if world == 1 and args.configs:
        torch.cuda.empty_cache()
        out["configs"] = measure_configs(args, dev)
    if world == 1 and args.train:
        torch.cuda.empty_cache()
        out["train"] = measure_train(args, dev)
    if world == 1 and args.gnb:
        torch.cuda.empty_cache()
        out["gnb"] = measure_gnb(args, dev)
    if world == 1 and args.projection:
        torch.cuda.empty_cache()
        out["scaling_projection"] = measure_projection(args, dev, out["ms_per_step"])
    if rank == 0:
        if world == 1 and args.cpu_budget > 0:
            from oracle import cpu_ref
            from wdg_amd import sweep, synth
            h_levels = synth.H_LEVELS_10 if args.k == 2 else synth.H_LEVELS_10_K10
            sample = sweep.make_jobs(h_levels, [0], k=args.k, n_nodes=args.nodes)
            out["cpu_baseline"] = cpu_ref.baseline_record(sample, args.feat, args.cpu_budget)
        print(compact_line(out, write_detail(out)), flush=True)


if __name__ == "__main__":
    main()
