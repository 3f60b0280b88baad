"""Homophily sweep driver: many independent (h, seed) graphs per launch, sharded over GPUs.

Counterpart of the loop body of the reference's `synthetic_plot.py:64-137`: a job is
one synthetic graph (homophily level h, seed s) with the feature matrix of its seed; the per-job work is
  degree / normalisation -> A_hat X aggregation -> edge/label statistics -> metric scalars.
The reference runs the jobs one after another in one Python process; here every stage is ONE batched kernel
launch over all jobs resident on the GPU (job tables in device memory), and the job list is sharded statically
over ranks (one process per GPU).  Jobs are independent, so the data path has no collective; ranks exchange
only the job table (broadcast from rank 0) and the per-job result rows (all_gather) through torch.distributed
(RCCL over xGMI on a GPU node, gloo in the CPU tests).
"""
import os
from dataclasses import dataclass

import numpy as np

import torch

from . import synth
from ._lib import WdgError

# the nine scalars of a sweep job (synthetic_plot.py:94-109).  results() returns the first six (one integer pass + LAS: what
# a bench step computes); full_metrics() adds generalized edge homophily and the two kernel-regression p-values
METRIC_NAMES = ("edge_homo", "node_homo", "class_homo", "adj_homo", "label_info", "soft_las", "ge_homo", "kr_l", "kr_nl")
STEP_METRICS = 6


@dataclass(frozen=True)
class Job:
    h: float
    seed: int
    k: int = 2
    n_nodes: int = 2000
    n_classes: int = 5

    @property
    def nnz(self):  # stored entries of A + I
        return self.n_nodes * (synth.out_degree(self.k, self.h) + 1)


def _mix64(*words):
    """splitmix64 over a few integers -> 64-bit key (host arithmetic; seeds the device sampler's Philox per (job, classifier))"""
    x = 0x9E3779B97F4A7C15
    for w in words:
        x = (x ^ (int(w) & 0xFFFFFFFFFFFFFFFF)) & 0xFFFFFFFFFFFFFFFF
        x = (x + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        x ^= x >> 31
    return x


def make_jobs(h_levels, seeds, k=2, n_nodes=2000, n_classes=5):
    """seed-major order: the jobs of one seed (which share a feature matrix) are adjacent."""
    return [Job(float(h), int(s), k, n_nodes, n_classes) for s in seeds for h in h_levels]


# what a job costs a GPU, in units of one stored entry of A + I: the aggregation's measured price per 16-row slice is flat
# (~1.6 us: the slice's 2-KB rows of Y) plus ~43 ns per entry of width (ops._QUAD_COST_*: 1 200 + 38.5 w ns fits the table),
# i.e. w + 31 entries' worth per row; the feature transform and the metric kernels are per-node work as well
JOB_ROW_COST = 32


def job_cost(j):
    return j.nnz + JOB_ROW_COST * j.n_nodes


def shard_jobs(jobs, world_size, rank, slack=0.02):
    """Static partition of the JOB list (jobs (base, h, sample) are independent, synthetic_plot.py:64-78):
    longest-processing-time first over single jobs by job_cost; among the ranks whose load is within `slack` of a rank's
    fair share of the lightest, one that already holds a job of the same seed is preferred (its X is resident there: 4 MB
    saved - a tie-break, never a reason to unbalance).  Deterministic and identical on every rank; a rank may end up with
    no job when there are fewer jobs than ranks.  The shard keeps the jobs of a seed adjacent, in list order."""
    key = (tuple(jobs), int(world_size), float(slack))
    owner = _SHARD_MEMO.get(key)  # (a pure function of the job list: every rank of a run, and every pass of a process, asks again)
    if owner is None:
        owner = _shard_owner(jobs, world_size, slack)
        if len(_SHARD_MEMO) >= 16:
            _SHARD_MEMO.clear()
        _SHARD_MEMO[key] = owner
    mine = [i for i in range(len(jobs)) if owner[i] == rank]
    return [jobs[i] for i in sorted(mine, key=lambda i: (jobs[i].seed, i))]


_SHARD_MEMO = {}

# What an adjacency costs in the NINE-scalar sweep, in microseconds per feature base (round 6, one MI355X: 200 regressions at 0.58 us
# each - the metric's share does not depend on the graph - + the propagated kernels' two aggregations and the edge cosines, which grow
# with the stored entries), and what a SAMPLE costs the rank that touches it (its feature matrices' uploads, Grams, row hashes and the
# 200 raw-feature regressions per base, shared by the sample's levels)
PAIR_COST_US, PAIR_COST_PER_ENTRY_US, SAMPLE_COST_US = 145.0, 0.045 / 59000 * 1000, 2200.0


def shard_pairs(pairs, world_size, rank):
    """Partition of the (homophily level, sample) adjacencies of the nine-scalar sweep (whole_sweep_rank), SAMPLE-AWARE: the list is
    cut into `world_size` CONTIGUOUS ranges of its sample-major order, so a rank touches as few samples as the split allows (two, where
    shard_jobs' LPT over single jobs dealt the reference's 280 adjacencies to 8 ranks with up to three samples each: a third
    sample is a third set of feature uploads, Grams and raw-feature regressions per base - 6 ms of a rank's 50).  The cuts balance
    the modelled cost (PAIR_COST_*: the regressions' flat share + the stored entries' + SAMPLE_COST_US per sample a range touches):
    every cut is moved, in turn, to where the heavier of its two ranges is lightest, until nothing moves.  Deterministic, identical
    on every rank; ranks beyond the list get nothing."""
    order = sorted(range(len(pairs)), key=lambda i: (pairs[i].seed, i))
    n, w = len(order), int(world_size)
    if n == 0 or w <= 1:
        return [pairs[i] for i in order] if rank == 0 else []
    memo_key = (tuple(pairs), w)  # (a pure function of the list: every pass of a process asks again - 4 ms of search each time)
    if memo_key in _PAIRS_MEMO:
        cuts = _PAIRS_MEMO[memo_key]
        return [pairs[i] for i in order[cuts[rank]:cuts[rank + 1]]]
    cost = np.array([PAIR_COST_US + PAIR_COST_PER_ENTRY_US * pairs[i].nnz for i in order])
    seed = np.array([pairs[i].seed for i in order])
    cum = np.concatenate([[0.0], np.cumsum(cost)])
    new_sample = np.concatenate([[1], (seed[1:] != seed[:-1]).astype(np.int64)])
    cum_new = np.concatenate([[0], np.cumsum(new_sample)])  # distinct samples that START before position p

    def range_cost(a, b):
        if b <= a:
            return 0.0
        return cum[b] - cum[a] + SAMPLE_COST_US * (1 + cum_new[b] - cum_new[a + 1])

    cuts = [int(np.searchsorted(cum, cum[-1] * r / w)) for r in range(w + 1)]
    cuts[0], cuts[-1] = 0, n
    for _ in range(64):
        moved = False
        for c in range(1, w):
            lo, hi = cuts[c - 1], cuts[c + 1]
            best = min(range(lo, hi + 1), key=lambda p_: (max(range_cost(lo, p_), range_cost(p_, hi)), abs(p_ - cuts[c])))
            if best != cuts[c]:
                cuts[c], moved = best, True
        if not moved:
            break
    if len(_PAIRS_MEMO) >= 16:
        _PAIRS_MEMO.clear()
    _PAIRS_MEMO[memo_key] = cuts
    return [pairs[i] for i in order[cuts[rank]:cuts[rank + 1]]]


_PAIRS_MEMO = {}


def _shard_owner(jobs, world_size, slack):
    """shard_jobs' partition: the rank of every job"""
    order = sorted(range(len(jobs)), key=lambda i: (-job_cost(jobs[i]), i))
    fair = sum(job_cost(j) for j in jobs) / max(world_size, 1)
    load = [0] * world_size
    seeds = [set() for _ in range(world_size)]
    owner = [0] * len(jobs)
    for i in order:
        j = jobs[i]
        lightest = min(load)
        cands = [r for r in range(world_size) if load[r] <= lightest + slack * fair]
        r = min(cands, key=lambda q: (j.seed not in seeds[q], load[q], q))
        load[r] += job_cost(j)
        seeds[r].add(j.seed)
        owner[i] = r
    # refinement: while it shortens the longest shard, move one of its jobs to - or swap one with - another rank (LPT alone
    # leaves 50 jobs on 8 ranks 14 % apart; the sweep ends with its slowest rank).  Jobs of equal cost are interchangeable,
    # so a round looks at one representative per (rank, cost) - numpy over the distinct costs - and the rounds stop at the
    # first one that gains less than 0.01 % of a fair share (the reference's 1 800-job sweep on 8 ranks: milliseconds).
    cost = np.array([job_cost(j) for j in jobs], dtype=np.int64)
    owner = np.array(owner, dtype=np.int64)
    load = np.array(load, dtype=np.int64)
    for _ in range(min(4 * len(jobs), 256)):
        hi = int(np.lexsort((np.arange(world_size), -load))[0])  # the heaviest rank, lowest id among equals
        mine_hi = np.flatnonzero(owner == hi)
        if mine_hi.size == 0:
            break
        ci, first_i = np.unique(cost[mine_hi], return_index=True)
        best = None  # (new maximum of the pair, job of hi, job of q | -1, q)
        for q in range(world_size):
            if q == hi:
                continue
            theirs = np.flatnonzero(owner == q)
            ck, first_k = np.unique(cost[theirs], return_index=True) if theirs.size else (np.zeros(0, np.int64), np.zeros(0, np.int64))
            ck = np.concatenate([[0], ck])  # (0: a plain move)
            kid = np.concatenate([[-1], theirs[first_k]]) if theirs.size else np.array([-1])
            top = np.maximum(load[hi] - ci[:, None] + ck[None, :], load[q] + ci[:, None] - ck[None, :])
            top = np.where(ck[None, :] < ci[:, None], top, np.iinfo(np.int64).max)
            a_, b_ = np.unravel_index(np.argmin(top), top.shape)
            if top[a_, b_] < load[hi] and (best is None or top[a_, b_] < best[0]):
                best = (int(top[a_, b_]), int(mine_hi[first_i[a_]]), int(kid[b_]), q)
        if best is None or load[hi] - best[0] < 1e-4 * fair:
            break
        _top, i, k, q = best
        dk = 0 if k < 0 else int(cost[k])
        load[hi] += dk - cost[i]
        load[q] += cost[i] - dk
        owner[i] = q
        if k >= 0:
            owner[k] = hi
    return owner.tolist()


def encode_jobs(jobs):
    return torch.tensor([[j.h, j.seed, j.k, j.n_nodes, j.n_classes] for j in jobs], dtype=torch.float64).reshape(-1, 5)


def decode_jobs(t):
    return [Job(float(r[0]), int(r[1]), int(r[2]), int(r[3]), int(r[4])) for r in t.tolist()]


def broadcast_jobs(jobs, device):
    """rank 0's job list -> every rank (one small broadcast; RCCL on GPUs, gloo on CPU)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return jobs
    n = torch.tensor([len(jobs) if dist.get_rank() == 0 else 0], dtype=torch.int64, device=device)
    dist.broadcast(n, 0)
    table = encode_jobs(jobs).to(device) if dist.get_rank() == 0 else torch.empty((int(n.item()), 5), dtype=torch.float64, device=device)
    dist.broadcast(table, 0)
    return decode_jobs(table.cpu())


def gather_results(local_rows, device):
    """[jobs_local, M] result rows of every rank -> list of per-rank tensors on every rank (all_gather, KBs)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [local_rows]
    ws = dist.get_world_size()
    counts = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(ws)]
    dist.all_gather(counts, torch.tensor([local_rows.shape[0]], dtype=torch.int64, device=device))
    m = max(int(c.item()) for c in counts)
    pad = torch.zeros((m, local_rows.shape[1]), dtype=local_rows.dtype, device=device)
    pad[: local_rows.shape[0]] = local_rows.to(device)
    out = [torch.empty_like(pad) for _ in range(ws)]
    dist.all_gather(out, pad)
    return [o[: int(c.item())] for o, c in zip(out, counts)]


def exchange_rows(keys, rows, n_total, device):
    """The sweep's ONE exchange step for keyed rows: every rank contributes rows [m_r, M] with global row ids keys [m_r] (m_r may
    be 0 and differs per rank); every rank returns the assembled [n_total, M] fp64 table, row `key` = the row sent under that
    key, rows nobody sent NaN.  One all_gather of [m_max, 1 + M] fp64 per rank (KBs; RCCL on GPUs, gloo in the CPU tests) -
    the N-rank form of synthetic_plot.py:64-78's result arrays, which the reference fills in place in one process."""
    keys = torch.as_tensor(keys, dtype=torch.float64).reshape(-1, 1)
    rows = torch.as_tensor(rows, dtype=torch.float64)
    if rows.dim() != 2 or rows.shape[0] != keys.shape[0]:
        raise ValueError("exchange_rows: rows must be [len(keys), M] (an empty shard passes [0, M])")
    if keys.numel() and not (0 <= int(keys.min()) and int(keys.max()) < n_total):
        raise ValueError("exchange_rows: key outside [0, n_total)")
    out = None
    for part in gather_results(torch.cat([keys, rows], 1), device):
        part = part.cpu()
        if out is None:
            out = torch.full((n_total, part.shape[1] - 1), float("nan"), dtype=torch.float64)
        if part.shape[0]:
            out[part[:, 0].long()] = part[:, 1:]
    return out


def pairs_of_rank(pairs, world, rank):
    """the adjacencies whole_sweep_rank gives `rank`: shard_pairs (sample-aware; WDG_SWEEP_SHARD=lpt: round 5's shard_jobs)"""
    return shard_pairs(pairs, world, rank) if os.environ.get("WDG_SWEEP_SHARD", "sample") != "lpt" else shard_jobs(pairs, world, rank)


def whole_sweep_rank(pairs, graph_of, bases, world, rank, epochs=100, max_pairs_per_shard=80, depth=2, progress=None, stats=None):
    """This rank's share of the reference's whole sweep (synthetic_plot.py:64-109), STRONG scaling: the (level, sample)
    adjacencies `pairs` (a list of Job; job.seed = the sample) are dealt to the ranks by shard_pairs (contiguous ranges of the
    sample-major list, balanced by modelled cost; WDG_SWEEP_SHARD=lpt: round 5's shard_jobs), every rank runs all feature
    bases over its adjacencies through run_bases (graphs built once per shard and shared by the bases) - no data-path collective.
    graph_of(job) -> (src, dst, labels) host arrays; bases as run_bases takes them.
    -> (keys [m] int64, rows [m, 9] fp64): key = index of the pair in `pairs` x n_bases + base index, ready for exchange_rows.
    A rank keeps one shard while it holds <= max_pairs_per_shard adjacencies (fewer job tables to build), else equal shards.
    progress(shard index, base index, rows): called as every base-shard's rows arrive on the host (bench.py's per-base clock)."""
    mine = pairs_of_rank(pairs, world, rank)
    index = {j: i for i, j in enumerate(pairs)}
    n_shards = max(1, -(-len(mine) // max_pairs_per_shard))
    per = max(1, -(-len(mine) // n_shards))
    shards = [(mine[a:a + per], [graph_of(j) for j in mine[a:a + per]]) for a in range(0, len(mine), per)]
    keys, rows = [], []
    for si, bi, r in run_bases(shards, bases, epochs=epochs, depth=depth, stats=stats):
        keys.append(torch.tensor([index[j] * len(bases) + bi for j in shards[si][0]], dtype=torch.int64))
        rows.append(r)
        if progress is not None:
            progress(si, bi, r)
    if not keys:
        return torch.zeros(0, dtype=torch.int64), torch.zeros((0, len(METRIC_NAMES)), dtype=torch.float64)
    return torch.cat(keys), torch.cat(rows)


class _PrelaunchedGram:
    """A GramBatch that has been LAUNCHED already on another stream (run_bases' feature prologue: a shard's raw-feature uploads and
    Grams are queued before its graphs exist): the first launch() only waits for that launch's event, later ones (a replayed batch)
    launch the table again.  Outputs as GramBatch's."""

    def __init__(self, gx, event=None):
        """event: recorded behind the launch on the stream it ran on; None: it ran on the stream the batch launches on"""
        self.gx, self.event, self.pending = gx, event, True
        self.k_linear, self.k_arccos, self.norm2, self.rep, self.row_rep = gx.k_linear, gx.k_arccos, gx.norm2, gx.rep, gx.row_rep

    def launch(self):
        if self.pending:
            self.pending = False
            if self.event is not None:
                torch.cuda.current_stream().wait_event(self.event)
                self.event = None
        else:
            self.gx.launch()


class _CopiedGram:
    """The raw-feature kernels of a prepared batch (whose job tables hold these buffers' addresses: SweepBatch.rebind_features) filled
    by COPY from a GramBatch launched elsewhere (the feature prologue): 64 MB device to device per base and two samples instead of the
    Gram itself on this stream.  Outputs: the batch's own buffers."""

    def __init__(self, dst, src, event):
        self.dst, self.src, self.event = dst, src, event
        self.k_linear, self.k_arccos, self.norm2, self.rep, self.row_rep = dst.k_linear, dst.k_arccos, dst.norm2, dst.rep, dst.row_rep

    def launch(self):
        if self.event is not None:
            torch.cuda.current_stream().wait_event(self.event)
            self.event = None
        else:
            self.src.launch()  # (a replay: the source table again, on this stream, then the copies)
        for name in ("k_linear", "k_arccos", "norm2", "rep"):
            for d, s_ in zip(getattr(self.dst, name), getattr(self.src, name)):
                if d is not None and s_ is not None:
                    d.copy_(s_, non_blocking=True)


class FeaturePrologue:
    """The raw-feature side of a shard's wide bases - the feature matrices' uploads, their Grams / arc-cosine kernels and row
    representatives - queued on a stream of its own BEFORE the shard's graphs exist: none of it depends on a graph, and a rank's
    share of the sweep otherwise starts with ~13 ms of host work (graph build, the first base's tables) during which the GPU has
    nothing to do (a quarter of a rank's 51 ms at 8 ranks; scripts/dev/rank_phases.py).  items[base index] = (x {seed: device
    tensor}, GramBatch, event) through get()."""

    _POOL = None  # one helper thread per process: the uploads' host copies (150 MB per rank and pass) run beside the main thread

    def __init__(self, bases, which, seeds, stream):
        """which: the base indices, in the order they will be asked for.  The work runs on a helper thread (the copies into the page-
        locked ring - library threads, the interpreter lock released - took 19 ms of a rank's start-up when the main thread made
        them); get(bi) waits until base bi has been QUEUED."""
        from concurrent.futures import ThreadPoolExecutor
        from . import ops
        dev = ops.require_gpu()
        if FeaturePrologue._POOL is None:
            FeaturePrologue._POOL = ThreadPoolExecutor(max_workers=1, thread_name_prefix="wdg-prologue")
        device_index = torch.cuda.current_device()

        def one(bi):
            torch.cuda.set_device(device_index)
            with torch.cuda.stream(stream):
                _name, feats, _sm = bases[bi]
                x = {s_: ops._h2d(np.ascontiguousarray(feats[s_], np.float32), dev) for s_ in seeds}
                gx = ops.GramBatch([x[s_] for s_ in seeds])
                gx.launch()
                ev = torch.cuda.Event()
                ev.record()
                return x, gx, ev

        self.futures = {bi: FeaturePrologue._POOL.submit(one, bi) for bi in which}

    def get(self, bi):
        f = self.futures.get(bi)
        return None if f is None else f.result()

    def drain(self):
        """(a consumer that skipped bases - an exception, a route decided otherwise - must not leave uploads running into freed state)"""
        for f in self.futures.values():
            try:
                f.result()
            except Exception:  # noqa: BLE001  (reported where the base is asked for; here only the thread is joined)
                pass


class _GramPair:
    """the kernels of a shard in GramBatch's slot order - [every job's aggregated features] + [every feature matrix] - when the two
    halves come from different launches (ops.PropagatedGram over the jobs, ops.GramBatch over the raw features, which runs first)"""

    def __init__(self, jobs_part, feats_part):
        self.jobs_part, self.feats_part = jobs_part, feats_part
        self.k_linear = list(jobs_part.k_linear) + list(feats_part.k_linear)
        self.k_arccos = list(jobs_part.k_arccos) + list(feats_part.k_arccos)
        self.norm2 = list(jobs_part.norm2) + list(feats_part.norm2)
        self.rep = list(jobs_part.rep) + list(feats_part.rep)  # (row representatives: the solver's deflation maps, or None)

    def launch(self):
        self.feats_part.launch()
        self.jobs_part.launch()


class SweepBatch:
    """All jobs of this rank, resident in HBM, with prebuilt job tables (one launch per stage per step)."""

    def __init__(self, jobs, n_feat=500, symmetric=0, gcn_hidden=64, inputs=None, share=None, feature_seed=0, tune=False,
                 build=None, labels_only=False, graph_batch=None, x_dev=None):
        """jobs: list of Job (graph + features come from the generator of synth.py) - or, with `inputs`, a list of the same
        length of (src, dst, labels, features [n, n_feat] fp32 numpy) tuples to run instead (real / fixture graphs; jobs that
        share a feature matrix must pass the same array object and carry the same `seed`).
        share: another SweepBatch over the SAME jobs whose graphs (CSR, SELL-16 copy, degrees, labels) this one reuses - the
        feature bases of synthetic_plot.py:64-65 aggregate different feature matrices over the same 300 adjacencies
        (BaseSweep below); feature_seed offsets the generator's feature seed per base.
        build: "batched" (default; WDG_SWEEP_BUILD) - all graphs of the shard through ops.GraphBatch: one COO -> CSR build of
        their block-diagonal union, the SELL-16 copies in five launches, ONE host read-back; "per_graph" - round 2's
        CsrGraph.from_coo + ensure_quad per graph (~20 launches and two host syncs each), kept for A/B runs and tests.
        tune: balance the aggregation's tape cut by feedback (tune() below: ~250 extra steps) - worth it for a batch that is
        replayed many times (training, the replay benchmark); a one-pass sweep takes the modelled cut.
        graph_batch: the shard's ops.GraphBatch (A + I of every job, quad=True) when the caller has queued or finished the build
        already (run_bases builds the NEXT shard's graphs on a stream of their own while this shard's bases run).
        x_dev: {seed: [n, n_feat] fp32 device tensor} - the feature matrices, uploaded already (run_bases' feature prologue).
        labels_only: aggregate the one-hot LABEL columns only (one 16-feature group: everything the six step scalars need); the
        feature matrices are uploaded for the kernel-regression metric, whose aggregated-feature kernels then come by propagation
        (prepare_full: K(A_hat X) = A_hat K(X) A_hat^T) - the reference's sweep over its wide feature bases (F up to 3 703) never
        needs A_hat X itself.  Needs gcn_hidden = 0; `y` aggregates the features on demand (the host re-solve of flagged blocks).

        The aggregated features are kept TILED by 16-feature groups (ops.Tiled: y_agg[i].t is [groups, n, 16]) when everything
        that reads them inside the step can - the label columns of LAS as a strided view, the fused transform through
        wdg_mlp2_job.a_group_stride: a workgroup of the aggregation (one feature group) then stores into one contiguous plane
        per graph instead of 64-byte pieces a row apart (the k = 2 launch 160 -> 133 us, the k = 10 one 98 -> 94).  `y` (a list of
        row-major [n, n_feat] tensors, as before) is then a COPY, refreshed on the first access after an aggregation; prepare_full(), whose
        kernels read row-major operands on the chain variants, refreshes it after every aggregation; TrainBatch("sgc") aggregates
        ONCE and builds its tables on that copy.  WDG_SWEEP_TILED_Y=0: row-major, as in rounds 1 - 3."""
        from . import ops
        self.ops = ops
        self.jobs = list(jobs)
        self.inputs = inputs
        dev = ops.require_gpu()
        self.n_feat = n_feat
        self.symmetric = symmetric
        n_classes = max([j.n_classes for j in self.jobs], default=0)
        # Label columns ride along: the aggregation walks X in 16-feature groups and the last group of F = 500 is three
        # quarters empty, so [X | onehot(labels) | 0] (F_agg = 512) costs the same 32 groups and the separate F = C
        # aggregation launch of the LAS metric disappears.  Needs one label vector per feature matrix (true for the
        # generator: labels = node // class size); WDG_SWEEP_RIDE_LABELS=0 keeps the separate launch.
        ride = os.environ.get("WDG_SWEEP_RIDE_LABELS", "1") != "0" and n_classes > 0
        self.labels_only = bool(labels_only and ride and not gcn_hidden and n_classes <= 16)
        lab_col = 0 if self.labels_only else n_feat  # first label column of [X | onehot | 0]
        self.lab_col = lab_col
        # rows of [X | onehot | 0] and of Y are padded to whole 16-feature groups: every 64-byte row segment an item stores
        # is then 64-byte aligned (one L2 write request instead of two: 220 against 225 us per launch; ..._ALIGN=4 to compare)
        align = int(os.environ.get("WDG_SWEEP_AGG_ALIGN", "16"))
        self.agg_feat = (lab_col + n_classes + align - 1) // align * align if ride else n_feat
        self.alg_feat = lab_col + n_classes if ride else n_feat  # columns that carry data (byte accounting: no padding)
        build = build or os.environ.get("WDG_SWEEP_BUILD", "batched")
        # A STEP TWIN: with labels_only the step aggregates [onehot(labels) | 0] and nothing of the features, so its tables, its
        # outputs and its six scalars depend on the graphs and their labels only.  A batch that shares the graphs of another
        # labels-only batch (the feature bases of one shard, run_bases) shares that batch's whole step - job tables, Y, counters - and
        # uploads its feature matrices for the kernel-regression metric, nothing else; its step() launches nothing (the owner's has
        # run on a stream this one waits for).  WDG_SWEEP_STEP_TWINS=0: every batch its own step (round 4).
        self.step_owner = None
        if (share is not None and self.labels_only and getattr(share, "labels_only", False) and share.agg_feat == self.agg_feat
                and share.symmetric == symmetric and len(share.jobs) == len(self.jobs) and os.environ.get("WDG_SWEEP_STEP_TWINS", "1") != "0"):
            self._init_step_twin(share.step_owner or share, inputs, n_feat, feature_seed, dev, x_dev)
            return
        feats, self.graphs, self.dinv, self.labels, self._y = {}, [], [], [], []
        self.y_agg, seed_labels = [], {}
        self.tiled_y, self._y_rm, self._untile_each_step, self._y_at = False, None, False, -1
        # pass 1 (host only): every job's edge lists and labels; the graph build of the whole shard is QUEUED before the feature
        # matrices are touched, so that the GPU builds (1.8 ms of kernels per 50-graph shard) while the host copies 20 MB of features
        # into the upload ring - GraphBatch.finish(), the shard's one read-back, then finds its data ready
        coos, labs_host, x_hosts, x_feat = [], [], [], {}
        for ji, j in enumerate(self.jobs):
            if inputs is not None:
                src, dst, lab, x_host = inputs[ji]
                lab = np.asarray(lab)
            elif share is not None:
                src = dst = None
                lab, x_host = np.asarray(share.labels_host[ji]).astype(np.int64), None
            else:
                src, dst, lab = synth.regular_graph(j.n_nodes, j.n_classes, j.k, j.h, j.seed)
                x_host = None
            coos.append((src, dst, j.n_nodes))
            labs_host.append(lab)
            x_hosts.append(x_host)
        mode = ops.NORM_SYM if symmetric else ops.NORM_RW
        if share is None and build == "batched":
            # A + I of every graph (synthetic_plot.py:92) in one build, the degrees of all of them in one launch
            self.graph_batch = graph_batch if graph_batch is not None else ops.GraphBatch(coos, ops.COO_ADD_SELF_LOOPS, quad=True, defer=True)
        for ji, j in enumerate(self.jobs):
            lab = labs_host[ji]
            if j.seed not in feats:
                x_host = x_hosts[ji]
                x = x_dev[j.seed] if x_dev is not None else ops._h2d(
                    synth.features(j.n_nodes, n_feat, j.seed + feature_seed) if x_host is None else np.ascontiguousarray(x_host, np.float32), dev)
                if ride:
                    # [X | onehot(labels) | 0]: the label block is laid out on the host and uploaded like any other array (an
                    # index assignment `xa[arange, labels] = 1` on the device SYNCHRONISES the host with the stream - 3 ms a
                    # piece behind queued regressions: a third of the whole sweep's host time in round 5's line profile)
                    tail = np.zeros((j.n_nodes, self.agg_feat - lab_col), np.float32)
                    ok = (lab >= 0) & (lab < self.agg_feat - lab_col)
                    tail[np.flatnonzero(ok), lab[ok]] = 1.0
                    if self.labels_only:
                        x_feat[j.seed] = x
                        xa = ops._h2d(tail, dev)
                    else:
                        xa = torch.empty((j.n_nodes, self.agg_feat), dtype=torch.float32, device=dev)
                        xa[:, :n_feat] = x
                        xa[:, lab_col:] = ops._h2d(tail, dev)
                    x = xa
                feats[j.seed], seed_labels[j.seed] = x, lab
            elif ride and not np.array_equal(seed_labels[j.seed], lab):
                raise ValueError("SweepBatch: jobs of one seed differ in labels; set WDG_SWEEP_RIDE_LABELS=0")
        if share is not None:
            self.graphs, self.dinv = list(share.graphs), list(share.dinv)
        elif build == "batched":
            self.graph_batch.finish()
            self.graphs = self.graph_batch.graphs
            self.dinv = [d["dinv"] for d in self.graph_batch.degree_norm(mode, ops.PREC_F32, use_values=True)]
        else:
            for src, dst, n_nodes in coos:
                g = ops.CsrGraph.from_coo(src, dst, n_nodes, None, ops.COO_ADD_SELF_LOOPS)  # A + I (synthetic_plot.py:92)
                self.graphs.append(g)
                self.dinv.append(ops.degree_norm(g, mode, ops.PREC_F32, use_values=True)["dinv"])
        # labels of every graph: one pooled upload
        lab_ptr = np.concatenate([[0], np.cumsum([len(l) for l in labs_host])]).astype(np.int64)
        lab_pool = ops._h2d(np.concatenate(labs_host).astype(np.int32) if labs_host else np.zeros(0, np.int32), dev)
        self.labels = [lab_pool[int(lab_ptr[i]):int(lab_ptr[i + 1])] for i in range(len(self.jobs))]
        self.labels_host = labs_host
        # Y tiled by 16-feature groups (see the docstring) when every reader inside the step takes it that way: the label columns
        # inside one group, graphs of one size on the quad-row kernel, the fused split-operand transform (or none)
        nodes = {j.n_nodes for j in self.jobs}
        mlp_ok = (not gcn_hidden) or (os.environ.get("WDG_SWEEP_FUSED_MLP", "1") != "0" and ops.Mlp2Batch.split_kernel()
                                      and gcn_hidden <= ops.Mlp2Batch.MAX_H and n_classes <= ops.Mlp2Batch.MAX_C
                                      and 0 < n_feat <= ops.Mlp2Batch.MAX_K and n_feat % 4 == 0)
        self.tiled_y = bool(os.environ.get("WDG_SWEEP_TILED_Y", "1") != "0" and ride and self.agg_feat % 16 == 0 and len(nodes) == 1
                            and (lab_col % 16) + n_classes <= 16 and mlp_ok and not ops.quad_disabled()
                            and all(g.ensure_quad() for g in self.graphs))
        if self.tiled_y:
            n = next(iter(nodes))
            self.y_pool = torch.empty((len(self.jobs), self.agg_feat // 16, n, 16), dtype=torch.float32, device=dev)
            self.y_agg = [ops.Tiled(self.y_pool[i]) for i in range(len(self.jobs))]
        else:
            for j in self.jobs:
                self.y_agg.append(torch.empty((j.n_nodes, self.agg_feat), dtype=torch.float32, device=dev))
                self._y.append(self.y_agg[-1][:, :n_feat])  # the feature part (a view: leading dimension agg_feat)
        self.x_agg = feats                                        # what the aggregation reads: [X | onehot | 0] (labels_only: [onehot | 0])
        self.x = x_feat if self.labels_only else {s: x[:, :n_feat] for s, x in feats.items()}  # the features proper
        self._y_lazy = None
        # A + I has unit values except a doubled pre-existing loop; the generator emits no loops -> pattern only
        entries = [(g, self.x_agg[j.seed], y, d, d if symmetric else None, False)
                   for j, g, y, d in zip(self.jobs, self.graphs, self.y_agg, self.dinv)]
        self.spmm = ops.SpmmBatch(entries)
        # (a tiled Y: every launch of the aggregation leaves the row-major copy stale - `y` compares spmm.n_launches with the count
        # it copied at and untiles lazily, once per aggregation.  No callback into self: a closure over self stored on self.spmm
        # is a reference cycle, and a batch then dies in the cyclic collector - whenever that runs, e.g. inside a stream capture)
        self.n_classes = n_classes
        self.stats = ops.StatsBatch(self.graphs, self.labels, n_classes)
        self.edges = sum(g.nnz for g in self.graphs)

        def scale(d):
            return d if symmetric else None

        # aggregation homophily (soft LAS, synthetic_plot.py:106): H = A_hat Z with one-hot Z, then W = H (H^T Y)
        self.h_las = []
        if ride:
            if self.tiled_y:  # the label columns lie inside one 16-feature group: [n, C] views with a row stride of 16 floats
                g_lab, off = lab_col // 16, lab_col % 16
                self.h_las = [ya.t[g_lab][:, off:off + n_classes] for ya in self.y_agg]
            else:
                self.h_las = [ya[:, lab_col:lab_col + n_classes] for ya in self.y_agg]  # written by the feature aggregation
            self.spmm_las = None
        else:
            las_entries = []
            for j, lab in zip(self.jobs, self.labels):
                onehot = torch.eye(n_classes, device=dev)[lab.long()].contiguous()
                self.h_las.append(torch.empty((j.n_nodes, n_classes), dtype=torch.float32, device=dev))
                las_entries.append((self.graphs[len(las_entries)], onehot, self.h_las[-1]))
            self.spmm_las = ops.SpmmBatch([(g, z, h, d, scale(d), False) for (g, z, h), d in zip(las_entries, self.dinv)])
        # the integer counters of the step (edge / node / class homophily, ...) ride in the label columns as well: H = D^-1 (A + I)
        # onehot holds every node's neighbour-class counts, so the LAS launch derives them in its per-row pass and the separate
        # pass over the edges (wdg_edge_label_stats_batched, 62 us beside nothing when there is no GCN chain) leaves the step.
        # Random-walk normalisation only (a column scale would mix the counts); WDG_SWEEP_DERIVE_COUNTS=0 keeps the edge pass.
        derive = ride and not symmetric and os.environ.get("WDG_SWEEP_DERIVE_COUNTS", "1") != "0"
        self.las = ops.LasBatch(list(zip(self.h_las, self.labels)), n_classes, counts=self.stats if derive else None,
                                row_scales=self.dinv if derive else None)
        self.derive_counts = self.las.derives_counts

        # GCN-2 forward (build-defined model, models.py): logits = A_hat relu((A_hat X) W0) W1, every job its own weights
        self.gcn = None
        n_streams = int(os.environ.get("WDG_SWEEP_STREAMS", "2"))  # see step_rest
        self.side = _side_stream(0) if n_streams >= 2 else None
        self.side2 = _side_stream(1) if n_streams >= 3 else None
        self._fork = torch.cuda.Event()
        if gcn_hidden:
            gen = torch.Generator(device="cpu").manual_seed(1234)
            w0 = [(torch.randn((n_feat, gcn_hidden), generator=gen) * (2.0 / (n_feat + gcn_hidden)) ** 0.5).to(dev) for _ in self.jobs]
            w1 = [(torch.randn((gcn_hidden, n_classes), generator=gen) * (2.0 / (gcn_hidden + n_classes)) ** 0.5).to(dev) for _ in self.jobs]
            hid = [torch.empty((j.n_nodes, gcn_hidden), dtype=torch.float32, device=dev) for j in self.jobs]
            # (rows of 8 floats: the logits aggregation reads a source row as two aligned float4 - ops.SpmmBatch narrow family)
            z2 = [torch.zeros((j.n_nodes, 8 if n_classes <= 8 else n_classes), dtype=torch.float32, device=dev)[:, :n_classes] for j in self.jobs]
            out = [torch.empty((j.n_nodes, n_classes), dtype=torch.float32, device=dev) for j in self.jobs]
            y_in = [ya.columns(n_feat) for ya in self.y_agg] if self.tiled_y else self._y
            mlp = [(y, a, None, b, None, z) for y, a, b, z in zip(y_in, w0, w1, z2)]
            fused = os.environ.get("WDG_SWEEP_FUSED_MLP", "1") != "0" and ops.Mlp2Batch.eligible(mlp)
            if self.tiled_y and not fused:
                raise RuntimeError("SweepBatch: the tiled Y was chosen for a transform that cannot read it (WDG_SWEEP_TILED_Y=0)")
            self.gcn = dict(w0=w0, w1=w1, hid=None if fused else hid, z2=z2, logits=out,
                            # fused: relu(Y W0) W1 in one pass over Y, the hidden layer stays in registers
                            mlp=ops.Mlp2Batch(mlp, relu=True) if fused else None,
                            gemm1=None if fused else ops.GemmBatch([(y, a, h, None) for y, a, h in zip(self._y, w0, hid)], relu=True),
                            gemm2=None if fused else ops.GemmBatch([(h, b, z, None) for h, b, z in zip(hid, w1, z2)]),
                            spmm=ops.SpmmBatch([(g, z, o, d, scale(d), False)
                                                for g, z, o, d in zip(self.graphs, z2, out, self.dinv)]))

        if tune or os.environ.get("WDG_QUAD_TUNE", "") == "1":
            self.tune()

    def _init_step_twin(self, owner, inputs, n_feat, feature_seed, dev, x_dev=None):
        """the rest of __init__ for a step twin (see there): the feature matrices of this base, everything else the owner's"""
        ops = self.ops
        self.step_owner = owner
        for name in ("graphs", "dinv", "labels", "labels_host", "y_agg", "_y", "x_agg", "spmm", "stats", "las", "h_las", "spmm_las",
                     "edges", "n_classes", "tiled_y", "derive_counts", "gcn", "side", "side2", "_fork", "graph_batch", "y_pool"):
            if hasattr(owner, name):
                setattr(self, name, getattr(owner, name))
        self._y_rm, self._untile_each_step, self._y_at, self._y_lazy = None, False, -1, None
        x_feat = {}
        for ji, j in enumerate(self.jobs):
            if j.seed in x_feat:
                continue
            x_host = inputs[ji][3] if inputs is not None else None
            x_feat[j.seed] = x_dev[j.seed] if x_dev is not None else ops._h2d(
                synth.features(j.n_nodes, n_feat, j.seed + feature_seed) if x_host is None else np.ascontiguousarray(x_host, np.float32), dev)
            if inputs is not None and not np.array_equal(np.asarray(inputs[ji][2]), np.asarray(owner.labels_host[ji])):
                raise ValueError("SweepBatch(share=...): the sharing batch's labels differ from the owner's")
        self.x = x_feat

    # -- the aggregated features as row-major matrices ------------------------------------------------------------
    @property
    def y(self):
        """list of [n, n_feat] row-major tensors, one per job (views of the aggregation's output - or, with a tiled Y, of a
        row-major copy: stable addresses, so job tables built on them stay valid; untiled here when an aggregation has run since
        the last copy, i.e. once per aggregation however often the property is read)"""
        if self.labels_only:  # the step aggregates the label columns only: the features' aggregation on demand, once
            if self._y_lazy is None:
                sc = (lambda d: d) if self.symmetric else (lambda d: None)
                self._y_lazy = [self.ops.spmm(g, self.x[j.seed], row_scale=d, col_scale=sc(d)) for j, g, d in zip(self.jobs, self.graphs, self.dinv)]
            return self._y_lazy
        if self.tiled_y:
            if self._y_at != self.spmm.n_launches or self._y_rm is None:
                self.untile()
            return [self._y_rm[i][:, :self.n_feat] for i in range(len(self.jobs))]
        return self._y

    def untile(self):
        """tiled Y -> the row-major copy (one strided copy kernel for the whole shard); no-op for a row-major batch"""
        if not self.tiled_y:
            return
        j, g, n, _ = self.y_pool.shape
        if self._y_rm is None:
            self._y_rm = torch.empty((j, n, g * 16), dtype=torch.float32, device=self.y_pool.device)
        self._y_rm.view(j, n, g, 16).copy_(self.y_pool.permute(0, 2, 1, 3))
        self._y_at = self.spmm.n_launches

    def tune(self, rounds=10, steps=5, confirm=24, finalists=3):
        """Balance the aggregation's eight segments (one per XCD) by what they really cost INSIDE the step.  The modelled cut
        leaves the XCDs 25 - 40 % apart (a segment that holds two phase groups stages a second slab while the other XCDs' stores
        fill the write path: 10 - 55 us, depending on when), and the launch ends with the slowest.  The kernel can record
        every workgroup's start and end on the device clock (wdg_spmm_quad_batched_clocked_f32); here the step is run a few
        times, each segment's share of the modelled cost is scaled by (mean span / its span) ^ 0.7, the tape is cut again,
        and the cuts with the shortest launches (HIP events around the launch, median over `steps` steps) go to a confirmation: the
        modelled cut and the `finalists` best balanced ones are stepped `confirm` times back to back each, the fastest stays (round 6:
        ten rounds and three finalists instead of six and one - with one finalist picked from five-step medians the tuned launch of
        one box varied between 90 and 99 us from run to run).
        rounds x (2 + steps) + confirm + (1 + finalists) x (1 + confirm) steps, once per batch; every cut computes the same bits (a row's sum
        order is fixed by the SELL-16 copy)."""
        sp = self.spmm
        if not sp.quad or sp.n_segments != 8 or sp.n_items <= sp.n_segments or os.environ.get("WDG_QUAD_TUNE", "1") == "0":  # noqa: E501
            return None
        clock = sp.new_clock()
        shares = np.ones(8)
        best = None
        cands = []  # (median launch ms, shares | None, spans) of every round
        for rnd in range(rounds):
            sp._set_segments(0, None if rnd == 0 else shares)
            for _ in range(2):
                self.step()
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
            spans = np.zeros(8)
            for a, b in ev:
                a.record()
                sp.launch(clock=clock)
                b.record()
                self.step_rest()
                torch.cuda.synchronize()
                spans += sp.segment_spans(clock) / steps
            t = sorted(a.elapsed_time(b) for a, b in ev)[steps // 2]
            cands.append((t, None if rnd == 0 else shares.copy(), spans.copy()))
            shares = shares * (spans.mean() / spans) ** 0.7
            shares /= shares.sum() / 8
        best = min(cands, key=lambda c: c[0])
        # Confirmation in the regime the batch will run in: the rounds above synchronise after every step (they read the clocks
        # back), a replay does not.  The modelled cut and the best balanced ones are stepped `confirm` times back to back each; the
        # fastest stays.  (It also leaves the chip in its steady state: behind the tuner's stop-and-go the first ~30 steps of a
        # burst run 5 - 8 % slower - scripts/dev/step_transient.py.)
        balanced = sorted((c for c in cands if c[1] is not None), key=lambda c: c[0])[:max(int(finalists), 1)]
        if balanced and confirm > 0:
            runs = [("modelled", cands[0])] + [(f"balanced{i}", c) for i, c in enumerate(balanced)]
            for _ in range(confirm):  # (an untimed burst first: the candidates are compared in the steady state ...
                self.step()
            burst = {name: 0.0 for name, _c in runs}
            half = max(confirm // 2, 1)
            for name, c in runs + runs[::-1]:  # ... and there and back again: a drift of the chip's state over the bursts cancels)
                sp._set_segments(0, c[1])
                self.step()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(half):
                    self.step()
                b.record()
                torch.cuda.synchronize()
                burst[name] += a.elapsed_time(b) / (2 * half)
            name, c = min(runs, key=lambda r: burst[r[0]])
            sp._set_segments(0, c[1])
            best = (c[0], c[1], c[2], burst)
        else:
            sp._set_segments(0, best[1])
        sp.tuned = best
        return best

    # -- bytes the aggregation must move (fused normalisation: no `val`, + dinv) ---------------
    def spmm_algorithmic_bytes(self):
        tot = 0
        for g in self.graphs:
            n, e, f = g.n_rows, g.nnz, self.alg_feat
            tot += 4 * (n + 1) + 4 * e + 4 * n + 4 * n * f + 4 * n * f
        return tot

    def spmm_unique_bytes(self):
        """same, counting each distinct feature matrix once (graphs of one seed share X)"""
        tot = sum(4 * x.shape[0] * self.alg_feat for x in self.x_agg.values())
        for g in self.graphs:
            tot += 4 * (g.n_rows + 1) + 4 * g.nnz + 4 * g.n_rows + 4 * g.n_rows * self.alg_feat
        return tot

    def step(self):
        """one pass of the hot path over the batch (synthetic_plot.py:92-108 minus the kernel-regression metric):
        feature aggregation, integer edge/label pass, label aggregation + LAS, GCN-2 forward"""
        if self.step_owner is not None:  # a step twin: the owner's step computed these outputs (same graphs, same labels)
            return
        self.spmm.launch()        # Y = A_hat X                       (F = n_feat)   dominant kernel
        if self._untile_each_step:  # (a tiled Y and row-major readers behind the step: prepare_full's Grams, training)
            self.untile()
        self.step_rest()

    def step_rest(self):
        """everything of a step after the feature aggregation (bench.py times that launch separately).

        Two independent chains on two HIP streams: the metric chain (LAS, then the statistics pass) and the GCN-2 forward
        (fused feature transform + the logits aggregation).  The B-resident GEMM occupies 200 of the 256 CUs for ~150 us;
        the metric kernels use the remaining CUs meanwhile: LAS first (large workgroups, done within ~50 us), then the
        statistics kernel, whose remainder overlaps the logits aggregation (scripts/dev/step_timeline.py on a kernel trace).
        Measured per step: one stream 0.47 ms, two streams statistics-first 0.439, three streams 0.425, two streams
        LAS-first 0.408.  WDG_SWEEP_STREAMS=1|3 for the other arrangements."""
        main = torch.cuda.current_stream()
        if self.gcn and self.side is not None:
            self._fork.record(main)  # one marker on the main queue for both side streams
            self.side.wait_event(self._fork)
            if self.side2 is not None:
                self.side2.wait_event(self._fork)
                with torch.cuda.stream(self.side):
                    if not self.derive_counts:
                        self.stats.launch()
                with torch.cuda.stream(self.side2):
                    if self.spmm_las is not None:
                        self.spmm_las.launch()
                    self.las.launch()
            else:
                with torch.cuda.stream(self.side):
                    self._metric_chain()
            self._gcn_chain()
            main.wait_stream(self.side)
            if self.side2 is not None:
                main.wait_stream(self.side2)
        else:
            self._metric_chain()
            if self.gcn:
                self._gcn_chain()

    def capture_rest(self):
        """step_rest() as one hipGraph (all streams, fork / join included): replaying it replaces eight launches and the
        stream dependencies by one graph launch.  Returns the replay callable; outputs land in the same tensors."""
        for _ in range(2):  # lazy state (kernel attributes, code objects) must exist before the capture
            self.step_rest()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            self.step_rest()
        self._graph = graph  # keep alive
        return graph.replay

    def capture_step(self):
        """step() - the feature aggregation AND everything behind it, both streams - as ONE hipGraph.  On a 50-graph shard the
        plain launches are faster (the device is the bound: 0.185 vs 0.20 ms); on the 6 - 7-graph shards of a 50-job sweep
        spread over 8 GPUs the step's 25 - 35 us of device work sit behind ~50 us of host enqueue, which one graph launch
        replaces (bench.py `scaling_projection`).  Returns the replay callable; outputs land in the same tensors, bit for
        bit the plain launches' (tests/test_gpu_sweep.py)."""
        for _ in range(2):
            self.step()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            self.step()
        self._step_graph = graph  # keep alive

        spmm = self.spmm

        def replay():
            spmm.n_launches += 1  # (a replayed aggregation leaves the row-major copy stale like a launched one)
            graph.replay()
        return replay

    def _metric_chain(self):
        if self.derive_counts:
            self.las.launch()       # soft / hard LAS counts + the integer counters, derived from the label columns of Y
            return
        # LAS first: its 100 workgroups of 1024 threads + 51 KiB of LDS need whole free CUs, which the 1 600 small
        # workgroups of the statistics kernel would otherwise occupy for as long as that kernel crawls beside the GEMM
        self.stats.zero()           # (the counters' memset, ahead of LAS instead of between the two kernels)
        if self.spmm_las is not None:
            self.spmm_las.launch()  # H = A_hat onehot(labels)        (F = C; else: columns of the feature aggregation)
        self.las.launch()           # soft / hard LAS counts
        self.stats.launch(zero=False)  # edge / node / class / adjusted homophily, label informativeness counters

    def _gcn_chain(self):
        if self.gcn["mlp"] is not None:
            self.gcn["mlp"].launch()    # relu(Y W0) W1               fp32 MFMA + per-lane second product, one launch
        else:
            self.gcn["gemm1"].launch()  # relu(Y W0)                  fp32 MFMA
            self.gcn["gemm2"].launch()  # (.) W1
        self.gcn["spmm"].launch()   # logits = A_hat (.)           (F = C)

    def _class_proportions(self):
        """[jobs, C] fp32: every job's class proportions (fixed per batch; counted where the labels already are)"""
        if getattr(self, "_class_prop", None) is None:
            st, c = self.stats, self.stats.c
            host = getattr(self, "labels_host", None)
            if host is not None and len(host) == st.n_jobs:
                memo = {}  # (the jobs of a sample share their label array: counted once per distinct array)

                def count(l):
                    if id(l) not in memo:
                        a = np.asarray(l)
                        memo[id(l)] = np.bincount(a[(a >= 0) & (a < c)], minlength=c)[:c]
                    return memo[id(l)]
                cnt = np.stack([count(l) for l in host])
                counts = self.ops._h2d(cnt.astype(np.float32), st.counters.device)  # (never a blocking pageable copy between launches)
            else:
                n = st.max_rows
                lab = torch.stack([torch.nn.functional.pad(l, (0, n - l.shape[0]), value=-1) for l in st.labels]) if st.n_jobs else torch.zeros((0, n))
                counts = torch.stack([(lab == i).sum(1) for i in range(c)], 1).to(torch.float32)
            self._class_prop = (counts / counts.sum(1, keepdim=True)).contiguous()
        return self._class_prop

    def results(self):
        """[jobs, STEP_METRICS] fp32: dense-flavour metric scalars (utils/homophily_plot.py) from the counters - one launch for the
        shard (wdg_sweep_scalars_f32); results_torch() is the same arithmetic as ~50 torch launches (round 2's path, kept as
        the yardstick of tests/test_gpu_sweep.py and for more than 32 classes)."""
        st = self.stats
        if not self.jobs:  # an empty shard (more ranks than jobs): no rows, same columns
            return torch.zeros((0, STEP_METRICS), dtype=torch.float32, device=st.counters.device)
        if not 1 <= st.c <= 32 or os.environ.get("WDG_SWEEP_SCALARS", "kernel") == "torch":
            return self.results_torch()
        ops = self.ops
        out = torch.empty((st.n_jobs, STEP_METRICS), dtype=torch.float32, device=st.counters.device)
        prop = self._class_proportions()
        ops.check(ops.lib.wdg_sweep_scalars_f32(st.totals.data_ptr(), st.rows.data_ptr(), st.compat.data_ptr(), st.classdeg.data_ptr(),
                                                self.las.counts.data_ptr(), self.las.n.data_ptr(), prop.data_ptr(), st.n_jobs,
                                                st.rows.shape[2], st.c, out.data_ptr(), ops.stream_handle()), "wdg_sweep_scalars_f32")
        return out

    def results_torch(self):
        """results() in torch operations (the reference's formulas line by line)"""
        st = self.stats
        tot = st.totals.to(torch.float32)
        edge = tot[:, 5] / tot[:, 4]
        n = st.max_rows
        nnz, noself, match = (st.rows[:, i, :].to(torch.float32) for i in range(3))
        hs = (match + (nnz - noself)) / nnz
        valid = nnz != 0
        node = torch.where(valid, hs, torch.zeros_like(hs)).sum(1) / valid.sum(1)
        k = st.compat.to(torch.float32)
        c = k.shape[1]
        hmat = k / k.sum(2, keepdim=True)
        prop = self._class_proportions()
        terms = torch.clamp(torch.diagonal(hmat, dim1=1, dim2=2) - prop, min=0)
        cls = torch.where(torch.isnan(terms), torch.zeros_like(terms), terms).sum(1) / (c - 1)
        degsum = st.classdeg.sum(1, keepdim=True).to(torch.float32)
        p_bar = st.classdeg.to(torch.float32) / degsum
        pc = k / degsum[:, :, None]
        p_bar = torch.where(p_bar == 0, torch.full_like(p_bar, 1e-8), p_bar)
        pc = torch.where(pc == 0, torch.full_like(pc, 1e-8), pc)
        s2 = (p_bar ** 2).sum(1)
        adj = (edge - s2) / (1 - s2)
        li = 2 - (pc * torch.log(pc)).sum((1, 2)) / (p_bar * torch.log(p_bar)).sum(1)
        soft_las = self.las.counts[:, 0].to(torch.float32) / self.las.n
        return torch.stack([edge, node, cls, adj, li, soft_las], 1)

    # -- the remaining three scalars: generalized edge homophily + the kernel-regression p-values ------
    def prepare_full(self, epochs=100, sample_max=500, seed_of=None, sampler=None, base_seed=0, sets=None, gx=None):
        """Set up the batched kernel-regression metric for every job: the Gram / arc-cosine kernels of the aggregated features
        (per job) and of the raw features (per feature matrix) - all nodes, once -, and, per job x classifier (kernel_reg0 /
        kernel_reg1) x epoch, the train / validation node sets; both kernels of an epoch share its node sets.
        sampler "device" (default; WDG_KR_SAMPLER): the sets of all epochs are drawn by ONE launch (ops.KrSets,
        wdg_kr_sample_sets: Philox4x32-10 keyed by base_seed, job and classifier - the reference's distribution, a documented
        generator) in launch_full(), inside the clock of a cold sweep.  sampler "host" (implied by seed_of): the reference's
        own routine on torch's CPU generator seeded with seed_of(job index, classifier index) (default 1000 job + classifier) -
        bit for bit the reference's sets (the golden tests), ~40 ms of host time per (job, classifier) at 100 epochs.
        sets (device sampler only; WDG_SWEEP_KR_SETS): whose node sets an epoch's regressions use.
          "job" (default, the reference's own draws: utils/homophily_metrics.py:267-283 draws per call) - independent sets per job, keyed
                   by the job's identity: 400 regressions per job at 100 epochs.
          "sample" - COMMON RANDOM SETS per sample (an optimisation, opt-in since round 6: a CHANGED ESTIMATOR is not a default): the jobs of a shard that share a feature matrix AND a label vector (the
                   homophily levels of one sample, synthetic_plot.py:78-91) draw ONE sequence of sets per (sample, classifier), keyed
                   by the sample's identity.  Every job's (X_results, G_results) pair has exactly the distribution of the reference's
                   independent draws (both accuracies of an epoch share the epoch's sets there too, utils/homophily_metrics.py:
                   279-297); only jobs of DIFFERENT homophily levels become correlated, which none of the reference's outputs (mean /
                   std over the ten samples of a level, synthetic_plot.py:112-128) measures.  The raw features' regressions of such jobs
                   are then the same problem - same kernel, same sets, same labels - and are solved ONCE per (sample, classifier,
                   epoch) instead of once per job: 200 + 200 / (levels per shard) regressions per job instead of 400 (the whole sweep:
                   348 000 instead of 672 000, 0.30 instead of 0.47 s on one GPU)."""
        from .utils.util_funcs import kernel_regression_epoch_indices
        ops = self.ops
        dev = self.graphs[0].device if self.graphs else ops.require_gpu()
        seeds = list(self.x)
        J = len(self.jobs)
        self.kr_epochs, self.kr_sample_max = epochs, sample_max
        sampler = "host" if seed_of is not None else (sampler or os.environ.get("WDG_KR_SAMPLER", "device"))
        if J == 0:  # an empty shard
            self.gram, self.ge, self.kr_sets = ops.GramBatch([]), ops.EdgeGramBatch([]), None
            self.kr = ops.KrBatch([], max(self.n_classes, 1))
            return
        if self.n_classes > ops.KrBatch.MAX_CLASSES:
            raise ValueError(f"SweepBatch.prepare_full: {self.n_classes} classes, the device solver holds {ops.KrBatch.MAX_CLASSES}; "
                             "use utils.homophily_metrics.classifier_based_performance_metric (host path) per graph")
        # kernels: [aggregated features of every job] + [raw features of every feature matrix], linear and arc-cosine.
        # Two routes to the aggregated features' kernels (WDG_GRAM_ROUTE=direct|propagate|auto):
        #   direct     the Gram of Y = A_hat X, one dense n x n x F product per job (ops.GramBatch over Y);
        #   propagate  K_linear(Y) = A_hat K_linear(X) A_hat^T: two aggregations with n "features" over the raw features' half Gram
        #              (ops.PropagatedGram) - 2 nnz n flops each instead of n^2 F; auto takes it from F = 640 on (the reference's
        #              bases: every one but pubmed), where the dense product costs 2 - 7 x the propagation (round 5).
        route = os.environ.get("WDG_GRAM_ROUTE", "auto")
        if route not in ("auto", "direct", "propagate"):
            raise ValueError(f"WDG_GRAM_ROUTE={route!r}")
        nodes = {j.n_nodes for j in self.jobs}
        can = len(nodes) == 1 and not ops.quad_disabled() and all(g.ensure_quad() for g in self.graphs)
        self.gram_route = "propagate" if can and (route == "propagate" or (route == "auto" and self.n_feat >= 640) or self.labels_only) else "direct"
        # (a labels-only batch whose graphs cannot take the propagated route - one graph of the shard without a SELL-16 copy: padding
        # beyond 4 x, graphs.py ensure_quad - falls back to the direct route over `y`, the features' aggregation made on demand graph by
        # graph: slower, same rows; round 5 raised here and a single hub-heavy graph aborted the whole sweep - ADVICE r05)
        x_slot = {s: J + i for i, s in enumerate(seeds)}
        if self.gram_route == "propagate":
            # (gx: (GramBatch over self.x in seed order, event) launched already - run_bases' feature prologue)
            if gx is None:
                # the raw features' kernels are launched HERE, while the host goes on building the propagation's and the regressions'
                # tables: they depend on nothing else of the batch (1.7 ms of device work for cora's two matrices that a rank's
                # start-up otherwise leaves the GPU waiting for)
                gx = ops.GramBatch([self.x[s] for s in seeds])
                if os.environ.get("WDG_SWEEP_EARLY_GRAM", "1") != "0":
                    gx.launch()
                    gx = _PrelaunchedGram(gx)
            else:
                gx = _PrelaunchedGram(*gx)
            sym = getattr(self, "symmetric", 0)
            gy = ops.PropagatedGram([(g, d, d if sym else None, gx.k_linear[x_slot[j.seed] - J])
                                     for j, g, d in zip(self.jobs, self.graphs, self.dinv)])
            self.gram = _GramPair(gy, gx)
        else:
            if self.labels_only:
                ys = self.y  # (aggregated on demand, row-major; the step's tables hold the label columns only)
            elif self.tiled_y and ops.GramBatch.tiled_ok():  # the Gram kernels read the tiled aggregation output as it is
                ys = [ya.columns(self.n_feat) for ya in self.y_agg]
            else:  # row-major: with a tiled Y the copy, refreshed behind every aggregation from here on
                ys = self.y
                self._untile_each_step = self.tiled_y
            self.gram = ops.GramBatch([ys[i] for i in range(J)] + [self.x[s] for s in seeds])
        self.ge = ops.EdgeGramBatch([(g, self.gram.k_linear[x_slot[j.seed]], self.gram.norm2[x_slot[j.seed]])
                                     for j, g in zip(self.jobs, self.graphs)])
        # (the jobs of a sweep share a handful of label vectors - a sample's levels carry equal labels in separate arrays: a job's
        # vector is first compared with the distinct ones met so far (np.array_equal: 2 us) and its bytes taken only for a new one -
        # 35 x 2 `tobytes` of 16 KB were 1 ms of a rank's start-up)
        distinct = []  # (array as int64, its bytes, per-class sizes)
        which = []
        for lab in self.labels_host:
            la = np.asarray(lab)
            k = next((i for i, (a, _b, _s) in enumerate(distinct) if a.shape == la.shape and np.array_equal(a, la)), -1)
            if k < 0:
                a64 = np.ascontiguousarray(la, dtype=np.int64)
                distinct.append((a64, a64.tobytes(), ops.kr_split_sizes(la, sample_max)))
                k = len(distinct) - 1
            which.append(k)
        sizes = [distinct[k][2] for k in which]
        sets = sets or os.environ.get("WDG_SWEEP_KR_SETS", "job")
        if sets not in ("sample", "job"):
            raise ValueError(f"prepare_full: sets={sets!r} (sample | job)")
        if sampler != "device":
            sets = "job"  # (the reference's host routine draws per call)
        self.kr_set_mode = sets
        # group[ji]: the set sequence job ji uses; rep[g]: the first job of group g
        if sets == "sample":
            import zlib
            index, group, rep, label_crc = {}, [], [], []
            for ji, j in enumerate(self.jobs):
                # (the label vector enters by VALUE - its int64 bytes' CRC -: the same job given int32 or int64 labels, or equal labels in
                # another array, draws the same sets - ADVICE r05)
                key = (j.seed, j.n_nodes, which[ji])
                if key not in index:
                    index[key] = len(rep)
                    rep.append(ji)
                    label_crc.append(zlib.crc32(distinct[which[ji]][1]))
                group.append(index[key])
            group, rep = np.asarray(group, np.int64), np.asarray(rep, np.int64)
        else:
            group = rep = np.arange(J, dtype=np.int64)
        G = len(rep)
        self.kr_group, self.kr_rep = group, rep
        if sampler == "device":
            # a (group, classifier) pair's key is an IDENTITY (not a position in this shard), so that the same job draws the same
            # sets on whichever rank / in whichever batch it runs: the sample's (seed, size, label vector) - or, sets="job", the
            # job's (seed, h, k, size), so that different jobs draw independent ones
            def key_of(g, clf):
                j = self.jobs[rep[g]]
                if sets == "sample":
                    return _mix64(base_seed, j.seed, j.n_nodes, label_crc[g], 0x5A3D, clf)
                return _mix64(base_seed, j.seed, int(round(j.h * 1e6)), j.k, j.n_nodes, clf)
            self.kr_sets = ops.KrSets([(self.labels[rep[g]], sizes[rep[g]][0], sizes[rep[g]][1], key_of(g, clf))
                                       for g in range(G) for clf in (0, 1)], epochs)
            self._kr_rebind = (sets, [sizes[r_] for r_ in rep], label_crc if sets == "sample" else None)
            train, val = self.kr_sets.train, self.kr_sets.val         # [G * 2, epochs, n_train / n_val], filled by launch_full()
        else:
            self.kr_sets = None
            seed_of = seed_of or (lambda ji, clf: 1000 * ji + clf)
            nt = max([int(t.sum()) for _s, t in sizes], default=1)
            nv = max([int(s_.sum() - t.sum()) for s_, t in sizes], default=1)
            train_h, val_h = np.zeros((J * 2, epochs, max(nt, 1)), np.int32), np.zeros((J * 2, epochs, max(nv, 1)), np.int32)
            rng_state = torch.get_rng_state()
            for ji, lab in enumerate(self.labels_host):
                lab_t = torch.from_numpy(np.asarray(lab)).long()
                for clf in (0, 1):
                    torch.manual_seed(seed_of(ji, clf))
                    for e, (tr, va) in enumerate(kernel_regression_epoch_indices(lab_t, sample_max, epochs)):
                        train_h[2 * ji + clf, e, :tr.shape[0]] = tr.numpy()
                        val_h[2 * ji + clf, e, :va.shape[0]] = va.numpy()
            torch.set_rng_state(rng_state)
            train, val = ops._h2d(train_h, dev), ops._h2d(val_h, dev)
        self.kr_train, self.kr_val = train, val  # [G * 2, epochs, .]: row kr_group[job] * 2 + classifier holds a job's sets
        # the table of the regressions that are SOLVED, described by arithmetic on base addresses:
        #   u = (job * 2 + classifier) * epochs + epoch                      the aggregated features' kernel of every job, then
        #   u = J * 2 * epochs + (group * 2 + classifier) * epochs + epoch   the raw features' kernel of every group (sets="job":
        #                                                                    a group is a job)
        # kr_index[job, classifier, epoch, (0 aggregated | 1 raw)] -> u is what kr_accuracy() / full_metrics() read through
        e_ = np.arange(epochs, dtype=np.int64)
        jc = np.arange(J * 2, dtype=np.int64)                        # job * 2 + classifier
        gc = np.arange(G * 2, dtype=np.int64)                        # group * 2 + classifier
        row_g = (group[jc // 2] * 2 + jc % 2)                        # the sets row of a (job, classifier)
        u_job = np.repeat(jc // 2, epochs)                           # per solved problem: the job whose labels / sizes describe it,
        u_job = np.concatenate([u_job, np.repeat(rep[gc // 2], epochs)])
        u_clf = np.concatenate([np.repeat(jc % 2, epochs), np.repeat(gc % 2, epochs)])
        u_row = np.concatenate([np.repeat(row_g, epochs), np.repeat(gc, epochs)])   # its sets row,
        u_epoch = np.concatenate([np.tile(e_, J * 2), np.tile(e_, G * 2)])          # its epoch
        k_lin = np.array([k.data_ptr() for k in self.gram.k_linear], np.int64)
        k_arc = np.array([k.data_ptr() for k in self.gram.k_arccos], np.int64)
        xs = np.array([x_slot[j.seed] for j in self.jobs], np.int64)
        u_slot = np.concatenate([np.repeat(jc // 2, epochs), np.repeat(xs[rep[gc // 2]], epochs)])  # and its kernel matrix
        k_ptr = np.where(u_clf == 0, k_lin[u_slot], k_arc[u_slot])
        n_nodes = np.array([j.n_nodes for j in self.jobs], np.int64)
        n_tr = np.array([int(t.sum()) for _s, t in sizes], np.int64)
        n_va = np.array([int(s_.sum() - t.sum()) for s_, t in sizes], np.int64)
        lab_ptr = np.array([l.data_ptr() for l in self.labels], np.int64)
        # (the row representatives of the matrix a kernel was computed from: duplicate nodes are deflated, not regularised - DESIGN 4.8)
        rep_of_slot = np.array([0 if r is None else r.data_ptr() for r in self.gram.rep], np.int64)
        self.kr = ops.KrBatch.from_arrays(
            k_ptr, n_nodes[u_job], train.data_ptr() + 4 * (u_row * epochs + u_epoch) * train.shape[2],
            val.data_ptr() + 4 * (u_row * epochs + u_epoch) * val.shape[2], lab_ptr[u_job], n_tr[u_job], n_va[u_job], self.n_classes,
            keep=(train, val, self.gram), rep_ptr=rep_of_slot[u_slot])
        idx = np.empty((J, 2, epochs, 2), np.int64)
        idx[..., 0] = (jc[:, None] * epochs + e_[None, :]).reshape(J, 2, epochs)
        idx[..., 1] = (J * 2 * epochs + row_g[:, None] * epochs + e_[None, :]).reshape(J, 2, epochs)
        self.kr_index = idx
        # a solved problem's name in the per-job enumeration p = ((job * 2 + classifier) * epochs + epoch) * 2 + which (what
        # pinv_accuracies takes): the group's first job stands for a shared raw-features problem
        self.kr_canonical = ((u_job * 2 + u_clf) * epochs + u_epoch) * 2 + (np.arange(u_job.shape[0]) >= J * 2 * epochs)
        self._kr_index_dev = None

    def rebind_features(self, feats_of_seed, n_feat, base_seed, pre=None):
        """The NEXT FEATURE BASE on a prepared labels-only batch (run_bases: synthetic_plot.py:64-65 runs six feature bases over
        the same adjacencies): new raw feature matrices and new node-set keys, every job table kept.  The step's tables never
        held a feature address (labels_only); the raw features' kernels are written into the SAME buffers (GramBatch(out=...)),
        so the propagation's, the edge cosines' and the regressions' tables stay valid; the node sets are drawn into the same
        tensors under the new base's keys.  Computes what a fresh SweepBatch + prepare_full(base_seed=...) over the same inputs
        computes (tests/test_gpu_sweep.py).  Call only when the batch's previous results have been fetched.
        feats_of_seed: {seed: [n, n_feat] fp32 host array}"""
        ops = self.ops
        if not (self.labels_only and getattr(self, "gram_route", None) == "propagate" and self.kr_sets is not None):
            raise ValueError("rebind_features: a prepared labels-only batch on the propagated route with device-drawn sets is expected")
        dev = self.kr.correct.device
        seeds = list(self.x)
        self.n_feat = n_feat
        self._y_lazy = None
        if pre is not None:  # (x {seed: device tensor}, GramBatch, event): uploaded and launched by the feature prologue - copied in
            self.x = {s_: pre[0][s_] for s_ in seeds}
            old = self.gram.feats_part
            self.gram.feats_part = _CopiedGram(old.dst if isinstance(old, _CopiedGram) else old, pre[1], pre[2])
        else:
            self.x = {s_: ops._h2d(np.ascontiguousarray(feats_of_seed[s_], np.float32), dev) for s_ in seeds}
            old = self.gram.feats_part
            gx = ops.GramBatch([self.x[s_] for s_ in seeds], out=old.dst if isinstance(old, _CopiedGram) else old)
            self.gram.feats_part = gx  # (same output tensors: the pair's k_linear / k_arccos / norm2 lists stay as they are)
        mode, sizes_rep, label_crc = self._kr_rebind
        rep = self.kr_rep

        def key_of(g, clf):
            j = self.jobs[rep[g]]
            if mode == "sample":
                return _mix64(base_seed, j.seed, j.n_nodes, label_crc[g], 0x5A3D, clf)
            return _mix64(base_seed, j.seed, int(round(j.h * 1e6)), j.k, j.n_nodes, clf)
        self.kr_sets = ops.KrSets([(self.labels[rep[g]], sizes_rep[g][0], sizes_rep[g][1], key_of(g, clf))
                                   for g in range(len(rep)) for clf in (0, 1)], self.kr_epochs, out=self.kr_sets)

    def kr_accuracy(self):
        """[jobs, 2 classifiers, epochs, (graph-aware, features only)] fp32 device tensor: every job's accuracies, read through
        kr_index from the regressions that were solved (shared raw-features problems appear once per job of their group)"""
        if self._kr_index_dev is None:
            self._kr_index_dev = self.ops._h2d(self.kr_index.reshape(-1), self.kr.correct.device)
        return self.kr.accuracy()[self._kr_index_dev].reshape(self.kr_index.shape)

    def kr_ridged_mask(self):
        """[jobs, 2, epochs, 2] bool device tensor: KrBatch.ridged() through kr_index"""
        if self._kr_index_dev is None:
            self._kr_index_dev = self.ops._h2d(self.kr_index.reshape(-1), self.kr.correct.device)
        return self.kr.ridged()[self._kr_index_dev].reshape(self.kr_index.shape)

    def kr_deflated_mask(self):
        """[jobs, 2, epochs, 2] bool device tensor: KrBatch.deflated() (duplicate train rows merged / zero rows dropped) through kr_index"""
        if self._kr_index_dev is None:
            self._kr_index_dev = self.ops._h2d(self.kr_index.reshape(-1), self.kr.correct.device)
        return self.kr.deflated()[self._kr_index_dev].reshape(self.kr_index.shape)

    def launch_full(self, sample_events=None):
        """the launches of the three extra scalars (after the aggregation: they read Y): Gram + maps, edge cosines, regressions
        (sample_events: an optional pair of torch events recorded around the node-set sampler on the stream it runs on -
        bench.py's `sample_ms`; nothing is recorded when the sets were drawn on the host)"""
        side = None
        if self.kr_sets is not None:
            # the epochs' node sets (device sampler; a host-sampled batch uploaded them in prepare_full) depend on nothing the
            # step produces: drawn on a second stream beside the Grams (0.8 ms of LDS sorting beside 2.7 ms of MFMA tiles)
            cur = torch.cuda.current_stream()
            if os.environ.get("WDG_SWEEP_SIDE_STREAM", "1") != "0":
                if getattr(self, "_side_stream", None) is None:
                    self._side_stream = _side_stream(2)
                side = self._side_stream
                side.wait_stream(cur)  # (the previous batch's regressions have read the old sets)
                with torch.cuda.stream(side):
                    if sample_events:
                        sample_events[0].record()
                    self.kr_sets.launch()
                    if sample_events:
                        sample_events[1].record()
            else:
                if sample_events:
                    sample_events[0].record()
                self.kr_sets.launch()
                if sample_events:
                    sample_events[1].record()
        self.gram.launch()
        self.ge.launch()
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
        self.kr.launch()

    def pinv_accuracies(self, problems, threads=None, kernels="host"):
        """The listed regressions solved the REFERENCE'S way, epoch by epoch (utils/homophily_plot.py:296-316, utils/
        homophily_metrics.py:283-297): `K[val][:, train] @ (np.linalg.pinv(K[train][:, train]) @ onehot[train])` in fp32 on the host
        (numpy's default rcond 1e-15: every singular value of an fp32 block kept), arg-max, hit rate on the validation rows.
        kernels "host" (default): the epoch's kernel computed the reference's way too - the features of the epoch's sample
                 gathered from the device, their Gram by the host's fp32 GEMM, the arc-cosine map of utils/homophily_metrics.py:
                 232-257 in numpy - i.e. exactly what WDG_KR_SOLVER=host computes for that epoch.  This matters on rank-deficient
                 blocks: duplicate nodes give bit-identical rows in a GEMM's Gram, so their null direction is exact and pinv's
                 inverted rounding-level singular value multiplies an exact zero; the device Gram (products from bf16 pieces,
                 operand roles not symmetric) leaves such rows a last bit apart, and pinv then amplifies that bit by 1e7 - measured on
                 the golden epochs (profiles/r05_kr_three_way.txt): up to 28 validation rows from the reference on linear feature
                 kernels with the device Gram, where the device solver's own ridge answer is within 0 - 1.
                 "device": the blocks gathered from the device kernels (one gather and one copy per kernel matrix).
        The pseudo-inverses run on a pool of host threads with one BLAS thread each.
        problems: 1-D array of problem indices p = ((job * 2 + classifier) * epochs + epoch) * 2 + (0 aggregated | 1 raw features).
        -> fp32 accuracies, one per listed problem.  Needs the node sets on the device (after launch_full())."""
        from concurrent.futures import ThreadPoolExecutor
        problems = np.asarray(problems, np.int64).reshape(-1)
        out = np.zeros(problems.shape[0], np.float32)
        if not problems.shape[0]:
            return out
        if kernels not in ("host", "device"):
            raise ValueError(f"pinv_accuracies: kernels={kernels!r}")
        E = self.kr_epochs
        pair, epoch, which = problems // (2 * E), (problems // 2) % E, problems % 2
        ji, clf = pair // 2, pair % 2
        seeds = list(self.x)
        x_slot = {s_: len(self.jobs) + i for i, s_ in enumerate(seeds)}
        slot = np.where(which == 0, ji, np.array([x_slot[j.seed] for j in self.jobs], np.int64)[ji])
        sizes = [self.ops.kr_split_sizes(lab, self.kr_sample_max) for lab in self.labels_host]
        n_tr = np.array([int(t.sum()) for _s, t in sizes], np.int64)
        n_va = np.array([int(s_.sum() - t.sum()) for s_, t in sizes], np.int64)
        c = self.n_classes
        ys = self.y if kernels == "host" else None  # (row-major aggregated features; a tiled Y is copied out once)
        work = []  # host: (position, H of the sample [train rows first], n_train, classifier, labels train, labels val); device: (position, K_tt, K_vt, ...)
        for key in sorted(set(zip(slot.tolist(), clf.tolist(), ji.tolist()))):
            sl, cl, jb = key
            sel = np.flatnonzero((slot == sl) & (clf == cl) & (ji == jb))
            rows_t = torch.from_numpy((int(self.kr_group[jb]) * 2 + cl) * np.ones_like(sel))  # (the sets row of (job, classifier))
            tr = self.kr_train[rows_t, torch.from_numpy(epoch[sel])][:, :n_tr[jb]].long()  # [m, n_train] node ids
            va = self.kr_val[rows_t, torch.from_numpy(epoch[sel])][:, :n_va[jb]].long()
            lab = np.asarray(self.labels_host[jb]).astype(np.int64)
            tr_h, va_h = tr.cpu().numpy(), va.cpu().numpy()
            if kernels == "device":
                kern = (self.gram.k_linear if cl == 0 else self.gram.k_arccos)[sl]
                k_tt = kern[tr[:, :, None], tr[:, None, :]].cpu().numpy()                 # [m, n_train, n_train]
                k_vt = kern[va[:, :, None], tr[:, None, :]].cpu().numpy()
                for m_, pos in enumerate(sel):
                    work.append((int(pos), k_tt[m_], k_vt[m_], None, cl, lab[tr_h[m_]], lab[va_h[m_]]))
            else:
                h = ys[jb] if sl < len(self.jobs) else self.x[seeds[sl - len(self.jobs)]]
                both = torch.cat([tr, va], 1)                                            # [m, n_train + n_val]
                step = max(1, (64 << 20) // max(1, both.shape[1] * h.shape[1] * 4))       # <= 64 MB per copy
                for a in range(0, both.shape[0], step):
                    hs = h[both[a:a + step].reshape(-1)].reshape(-1, both.shape[1], h.shape[1]).cpu().numpy()
                    for m_ in range(hs.shape[0]):
                        work.append((int(sel[a + m_]), hs[m_], None, int(n_tr[jb]), cl, lab[tr_h[a + m_]], lab[va_h[a + m_]]))
        eye = np.eye(c, dtype=np.float32)
        eps, pi = np.float32(1e-8), np.float32(np.pi)

        def sample_kernel(hs, n_layers):  # gntk_homophily_ (utils/homophily_metrics.py:232-257) for the rows of one epoch's sample
            g = hs @ hs.T
            if n_layers == 1:
                d = np.sqrt(np.diag(g))
                nrm = d[:, None] * d[None, :]
                nrm = np.where(nrm > eps, nrm, eps).astype(np.float32)
                with np.errstate(invalid="ignore", divide="ignore"):
                    ac, sq = np.arccos(g / nrm), np.sqrt(np.square(nrm) - np.square(g))
                ac[np.isnan(ac)], sq[np.isnan(sq)] = 0, 0
                g = (np.float32(1) / pi * (g * (pi - ac) + sq)).astype(np.float32)
            return g / np.float32(2)

        def solve(item):
            pos, a, b, nt, cl, lab_t, lab_v = item
            if nt is not None:
                k = sample_kernel(a, cl)
                a, b = k[:nt, :nt], k[nt:, :nt]
            pred = b @ (np.linalg.pinv(a) @ eye[lab_t])
            return pos, np.float32(np.mean(pred.argmax(1) == lab_v))

        # (16 python threads: numpy's pinv holds the interpreter lock between its LAPACK calls - 64 threads measured SLOWER than 16 on
        # a 128-core host; one BLAS thread per problem - left alone every pinv fans out over all cores and the threads trample
        # each other: 37 ms per 300 x 300 block against 8 single-threaded)
        threads = threads or int(os.environ.get("WDG_KR_PINV_THREADS", str(min(16, os.cpu_count() or 1))))
        try:
            from threadpoolctl import threadpool_limits
            limit = threadpool_limits(limits=1)
        except ImportError:  # (no threadpoolctl: correct, only slower)
            limit = None
        try:
            with ThreadPoolExecutor(max_workers=max(1, threads)) as pool:
                for pos, acc_ in pool.map(solve, work):
                    out[pos] = acc_
        finally:
            if limit is not None:
                limit.restore_original_limits()
        return out

    def full_metrics(self, ridge=None):
        """[jobs, 9] fp64: results() + ge_homo + the p-values KR_L (kernel_reg0) and KR_NL (kernel_reg1) of the Welch t-test
        over the epochs' accuracies (scipy on the host for the t distribution, as in the reference).
        ridge: what to do with the train blocks the device solver flags as rank deficient at fp32 rounding level
          "device" (default; WDG_SWEEP_KR_RIDGE) - keep the solver's ridge answers (counted in kr_ridged / kr_total, announced once
                   per shard).  Measured against what the REFERENCE computed in the same epochs (golden fixtures, profiles/
                   r05_kr_three_way.txt): within 0 - 2 validation rows on every synthetic-sweep fixture (flagged: the raw pubmed
                   features' kernels, which hold duplicate rows), within 4 on cora's raw-adjacency kernels;
          "pinv"   - solve exactly those blocks again the reference's way (pinv_accuracies: the epoch's sample gathered from the
                   device, its kernel by the host's fp32 GEMM + the arc-cosine map, np.linalg.pinv on host threads) and patch their
                   accuracies before the t-tests: the reference-equal answer on the sweep path, at ~8 ms of one host core per flagged
                   block (kr_pinv_seconds).  Nothing is flagged on the synthetic feature bases of bench.py (0 of 672 000); the
                   pubmed-sample fixtures flag 39 %."""
        ridge = ridge or os.environ.get("WDG_SWEEP_KR_RIDGE", "device")
        if ridge not in ("device", "pinv"):
            raise ValueError(f"full_metrics: ridge={ridge!r} (device | pinv)")
        self.kr_ridged = self.kr_deflated = self.kr_total = 0
        self.kr_pinv_seconds = 0.0
        if not self.jobs:
            return torch.zeros((0, len(METRIC_NAMES)), dtype=torch.float64)
        nj = len(self.jobs)
        # one copy back: the step's scalars, the edge cosine means, the 4 x epochs accuracies of every job and the count of
        # train blocks the solver had to regularise
        # (one launch - wdg_sweep_pack_f64 - instead of a dozen library launches queued behind the regressions, then the one copy)
        res = self.results().contiguous()
        n_kr = int(self.kr.n_jobs)
        packed_dev = torch.empty(res.numel() + nj + n_kr + 3, dtype=torch.float64, device=res.device)
        ops = self.ops
        ops.check(ops.lib.wdg_sweep_pack_f64(res.data_ptr(), res.numel(), self.ge.mean.data_ptr(), nj, self.kr.correct.data_ptr(),
                                             self.kr.flags.data_ptr(), self.kr.n_val.data_ptr(), n_kr, packed_dev.data_ptr(),
                                             ops.stream_handle()), "wdg_sweep_pack_f64")
        packed = packed_dev.cpu()
        if getattr(self.kr, "ablate", 0):
            raise WdgError("KrBatch: WDG_KR_ABLATE is set - the launch was a timing-only ablation, its accuracies mean nothing")
        if int(packed[-1].item()):
            raise WdgError("wdg_kernel_regress_batched_f32 refused a problem (shape outside the solver's limits)")
        # (rank-deficient train blocks: solved with a rounding-level ridge where the reference's pinv inverts the rounding-level
        # singular values - counted and said once per shard, like utils/homophily_metrics.py does per call)
        # (kr_total: the regressions that were SOLVED - with common sets per sample a shared raw-features problem counts once)
        self.kr_ridged, self.kr_deflated, self.kr_total = int(packed[-2].item()), int(packed[-3].item()), n_kr
        packed = packed[:-3]
        n_base = packed.numel() - nj - self.kr.n_jobs
        base = packed[:n_base].reshape(nj, -1)
        ge = packed[n_base:n_base + nj]
        acc = packed[n_base + nj:].numpy().astype(np.float32)  # the solved problems, in table order (prepare_full)
        if self.kr_ridged and ridge == "pinv":
            import time
            t0 = time.perf_counter()
            flagged = torch.nonzero(self.kr.ridged()).flatten().cpu().numpy()
            acc[flagged] = self.pinv_accuracies(self.kr_canonical[flagged])
            self.kr_pinv_seconds = time.perf_counter() - t0
        elif self.kr_ridged and os.environ.get("WDG_KR_QUIET", "0") in ("", "0"):
            import warnings
            warnings.warn(f"kernel regression: {self.kr_ridged} of {self.kr.n_jobs} train blocks of this shard were rank-deficient at fp32 "
                          "rounding level and keep the device solver's ridge answers: within 0 - 2 validation rows of what the reference "
                          "computed in the same epochs on the synthetic-sweep fixtures, 4 on cora (profiles/r05_kr_three_way.txt); "
                          "full_metrics(ridge='pinv') / WDG_SWEEP_KR_RIDGE=pinv solves them again the reference's way on the host", stacklevel=2)
        acc = acc[self.kr_index]  # -> every job's accuracies (a shared raw-features problem is read by each job of its group)
        self.kr_acc = acc  # [job, classifier, epoch, (graph-aware, features only)]: diagnostics / tests
        pvals = torch.from_numpy(welch_p_values(acc[..., 0], acc[..., 1]))
        return torch.cat([base, ge[:, None], pvals], 1)


def welch_p_values(g_res, x_res):
    """The metric's p-value from the epochs' accuracies (utils/homophily_metrics.py:335-347: scipy's ttest_ind(X_results, G_results,
    equal_var=False), halved or complemented by which side wins more epochs), for arrays [..., epochs] at once.  The same
    arithmetic as scipy.stats.ttest_ind - Welch's statistic, the Welch-Satterthwaite degrees of freedom, the two-sided tail of
    Student's t (scipy.special.stdtr) - without its per-call validation (2 ms per 100 pairs, most of the host tail of a shard)."""
    from scipy.special import stdtr
    g, x = np.asarray(g_res, np.float32), np.asarray(x_res, np.float32)  # (fp32 moments, like scipy on the reference's fp32 lists)
    n = g.shape[-1]
    vx, vg = x.var(-1, ddof=1) / n, g.var(-1, ddof=1) / n
    with np.errstate(divide="ignore", invalid="ignore"):
        t = (x.mean(-1) - g.mean(-1)) / np.sqrt(vx + vg)
        df = (vx + vg) ** 2 / (vx ** 2 / (n - 1) + vg ** 2 / (n - 1))
        df = np.where(np.isnan(df), 1, df)  # (both samples constant: scipy's _unequal_var_ttest_denom sets df = 1, so that different
        # means give t = +-inf -> p = 0 and equal means t = NaN -> p = NaN, which the reference then folds like any p-value)
        p = 2.0 * stdtr(df.astype(np.float64), -np.abs(t).astype(np.float64))
    better = (np.asarray(g_res, np.float32) > np.asarray(x_res, np.float32)).astype(np.float32).mean(-1)
    return np.where(better <= 0.5, p / 2, 1 - p / 2)


def _count_kr(stats, sb):
    if stats is not None:
        stats["kr_ridged"] = stats.get("kr_ridged", 0) + getattr(sb, "kr_ridged", 0)
        stats["kr_total"] = stats.get("kr_total", 0) + getattr(sb, "kr_total", 0)
        stats["kr_deflated"] = stats.get("kr_deflated", 0) + getattr(sb, "kr_deflated", 0)
        stats["kr_pinv_seconds"] = stats.get("kr_pinv_seconds", 0.0) + getattr(sb, "kr_pinv_seconds", 0.0)


_SIDE_STREAMS = {}  # (device, handle of the stream a batch was built on, index) -> its side stream, created once per process


def _side_stream(index):
    """the index-th side stream of batches built on the current stream (a stream per batch cost a hipStreamCreate - and a fresh
    handle for everything keyed by stream - per base-shard; batches on one stream run one after the other anyway)"""
    key = (torch.cuda.current_device(), torch.cuda.current_stream().cuda_stream, index)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream()
    return _SIDE_STREAMS[key]


_PIPE_STREAMS = {}  # (device, depth) -> the pipelining streams, created once per process (stable handles: the per-stream scratch
#                     of ops.PropagatedGram and the upload ring's events stay valid from one sweep to the next)


def _pipe_streams(depth):
    if depth <= 1:
        return [torch.cuda.current_stream()]
    key = (torch.cuda.current_device(), depth)
    if key not in _PIPE_STREAMS:
        _PIPE_STREAMS[key] = [torch.cuda.Stream() for _ in range(depth)]
    return _PIPE_STREAMS[key]


def run_shards(shards, n_feat=500, nine=False, epochs=100, sample_max=500, symmetric=0, depth=2, first_seed=0, stats=None):
    """The one-pass sweep over a sequence of shards (synthetic_plot.py:78-109: every graph visited once), PIPELINED: a generator
    of the shards' metric rows ([jobs, 6] fp32 / [jobs, 9] fp64 on the host), in order.

    shards: an iterable of (jobs, inputs) as SweepBatch takes them (host COO arrays, labels, features).  Shard b runs on HIP
    stream b mod `depth`: its uploads, the batched build (and that build's one read-back, which waits for ITS stream only), the
    step and - nine=True - the Grams, the device-drawn node sets and the regressions are queued, then the host goes on to shard
    b + 1 and fetches shard b's rows only when `depth` shards are in flight.  The host part of a shard (concatenation, uploads,
    job tables: 8-11 ms per 50 graphs) thus overlaps the device part of the one before (15.6 ms with the regressions), where the
    sequential loop of bench.py's `sweep_cold` pays their sum.  depth=1 is that sequential loop.  Every shard computes the same
    bits either way (tests/test_gpu_sweep.py)."""
    from collections import deque
    depth = max(1, int(depth))
    streams = _pipe_streams(depth)
    in_flight = deque()

    def fetch():
        sb, stream = in_flight.popleft()
        with torch.cuda.stream(stream):
            rows = sb.full_metrics() if nine else sb.results().cpu()
        _count_kr(stats if nine else None, sb)
        return rows

    # The NEXT shard's graph build (host pack of the edge lists into the upload ring, upload, COO -> CSR, SELL-16 count: 2 - 3 ms of
    # mostly GIL-free library calls) runs on a HELPER THREAD and a stream of its own while this thread builds the current shard's
    # tables: the cold path is host-bound (6.5 ms of host work per 50-graph shard against 0.2 - 9 ms on the device), and this is
    # the part of it that needs no interpreter.  Measured over ten 50-graph shards, best of three passes (scripts/dev/time_cold_modes.py):
    # six scalars 8 800 -> 9 700 graphs/s, nine scalars 3 360 -> 4 670.  (The shard's feature uploads moved to the helper as well:
    # 6 300 / 4 200 - its 4-MB copies hold the upload ring's lock against this thread's twenty small uploads.)
    # WDG_SWEEP_BUILD_THREAD=0 / depth 1: everything on this thread, in place.
    threaded = (depth > 1 and os.environ.get("WDG_SWEEP_BUILD_THREAD", "1") != "0" and os.environ.get("WDG_SWEEP_BUILD", "batched") == "batched")
    if "WDG_SWEEP_BUILD_THREAD_FORCE" in os.environ:  # (scripts/dev/time_cold_modes.py)
        threaded = os.environ["WDG_SWEEP_BUILD_THREAD_FORCE"] == "1"
    pool = build_stream = None
    if threaded:
        from concurrent.futures import ThreadPoolExecutor
        pool, build_stream = ThreadPoolExecutor(max_workers=1), _side_stream(3)
        dev_index = torch.cuda.current_device()

    def queue_build(shard):
        from . import ops
        torch.cuda.set_device(dev_index)
        with torch.cuda.stream(build_stream):
            return ops.GraphBatch([(i_[0], i_[1], j.n_nodes) for j, i_ in zip(shard[0], shard[1])], ops.COO_ADD_SELF_LOOPS, quad=True, defer=True)

    def submit(shard):
        return pool.submit(queue_build, shard) if (threaded and shard is not None and shard[0] and shard[1] is not None) else None

    try:
        it = iter(shards)
        nxt = next(it, None)
        fut = submit(nxt)
        b = 0
        while nxt is not None:
            (jobs, inputs), cur_fut = nxt, fut
            nxt = next(it, None)
            stream = streams[b % depth]
            gb = None
            if cur_fut is not None:
                gb = cur_fut.result()
                fut = submit(nxt)  # (the helper packs the next shard while this thread builds this one's tables)
                with torch.cuda.stream(build_stream):
                    gb.finish()
                stream.wait_stream(build_stream)
            else:
                fut = submit(nxt)
            with torch.cuda.stream(stream):
                sb = SweepBatch(jobs, n_feat=n_feat, symmetric=symmetric, gcn_hidden=0, inputs=inputs, graph_batch=gb)
                if nine and sb.jobs:
                    sb.prepare_full(epochs=epochs, sample_max=sample_max, base_seed=first_seed + b)
                sb.step()
                if nine and sb.jobs:
                    sb.launch_full()
            in_flight.append((sb, stream))
            b += 1
            if len(in_flight) >= depth:
                yield fetch()
        while in_flight:
            yield fetch()
    finally:
        if pool is not None:
            pool.shutdown(wait=True)


def propagates(n_feat, jobs):
    """whether prepare_full takes the propagated route for the aggregated features' kernels of these jobs on its own (WDG_GRAM_ROUTE,
    width, graphs of one size that fit the quad-row kernel's one-block slabs): the sweep driver then aggregates the label columns only"""
    route = os.environ.get("WDG_GRAM_ROUTE", "auto")
    one_size = len({j.n_nodes for j in jobs}) == 1 and all(j.n_nodes <= 5056 for j in jobs)
    return bool(jobs) and one_size and os.environ.get("WDG_SPMM_NO_QUAD", "0") in ("", "0") and (
        route == "propagate" or (route == "auto" and n_feat >= 640)) and os.environ.get("WDG_SWEEP_RIDE_LABELS", "1") != "0"


def run_bases(shards, bases, epochs=100, symmetric=0, depth=2, first_seed=0, stats=None):
    """The reference's WHOLE sweep (synthetic_plot.py:64-109: 6 feature bases x 30 homophily levels x 10 samples = 1 800 jobs, nine
    scalars each), one pass, PIPELINED like run_shards: a generator of (shard index, base index, rows [jobs, 9] fp64 on the host).

    shards: list of (jobs, graph_inputs) - graph_inputs[i] = (src, dst, labels) host arrays of job i (an adjacency belongs to a
            (homophily level, sample) pair: `job.seed` = the sample);
    bases:  list of (name, features, sample_max) - features[seed] = [n, F_base] fp32 host array of that base's sample `seed`
            (synthetic_plot.py:81-82), sample_max as synthetic_plot.py:66 (300 for chameleon / film, else 500).
    A shard's graphs (CSR, SELL-16 copy, degrees, labels) are built ONCE, by its first base; the other bases' batches share them
    (SweepBatch(share=...)) and aggregate their own feature matrices over them.  Base-shard b runs on HIP stream b mod `depth`
    and its rows are fetched when `depth` are in flight.  Every (shard, base) computes the rows a stand-alone SweepBatch over
    the same inputs computes (tests/test_gpu_sweep.py)."""
    from collections import deque
    depth = max(1, int(depth))
    streams = _pipe_streams(depth)
    in_flight = deque()

    def fetch():
        si, bi, sb, stream = in_flight.popleft()
        with torch.cuda.stream(stream):
            rows = sb.full_metrics()
        _count_kr(stats, sb)
        return si, bi, rows, sb, stream

    # TABLE REUSE inside a shard (round 5).  Bases on the propagated route aggregate the label columns only, so (i) they share ONE
    # step - tables, outputs, six scalars (SweepBatch step twins) - and (ii) a base can take over the prepared batch of an earlier
    # base with the same sample_max whose rows have been fetched: SweepBatch.rebind_features swaps the feature matrices and the
    # node-set keys and keeps every job table (the kernels land in the same buffers).  The bases are visited in an order that puts
    # bases of equal sample_max apart (with two base-shards in flight a batch is free again two visits later); rows carry their base
    # index, and a batch stays on its stream.  WDG_SWEEP_REBIND=0: every base-shard builds its own tables, in the caller's order.
    rebind_ok = os.environ.get("WDG_SWEEP_REBIND", "1") != "0"
    # The NEXT shard's graphs are built ahead, on a stream of their own, as soon as this shard's first base is queued: the build's
    # one read-back (GraphBatch.finish) then finds its data ready.  Built in place it waited behind whatever ran on the GPU at
    # that moment - typically a regression launch that holds every CU for 5 - 9 ms - and the queue ran dry while the host stood
    # still: a fifth of the sweep's wall clock had no kernel running.  WDG_SWEEP_PREFETCH_BUILD=0: build in place.
    shards = list(shards)
    prefetch = (os.environ.get("WDG_SWEEP_PREFETCH_BUILD", "1") != "0" and os.environ.get("WDG_SWEEP_BUILD", "batched") == "batched")
    build_stream = _side_stream(3) if prefetch else None
    queued = {}

    def queue_build(si_):
        jobs_, gi_ = shards[si_]
        if not jobs_:
            return None
        from . import ops
        with torch.cuda.stream(build_stream):
            return ops.GraphBatch([(src, dst, j.n_nodes) for j, (src, dst, _lab) in zip(jobs_, gi_)], ops.COO_ADD_SELF_LOOPS,
                                  quad=True, defer=True)

    # The FEATURE PROLOGUE (round 6, WDG_SWEEP_PROLOGUE=1; OFF by default): the wide bases' feature uploads, raw-feature Grams and row
    # representatives depend on no graph - queued, by a helper thread and on a stream of their own, before the shard's graph build is
    # waited for, they give the GPU work during the ~13 ms in which the host builds a rank's graphs and its first base's tables.
    # Measured at 8 ranks (scripts/dev/rank_phases.py): the device span of a rank's pass shrinks from 37.4 to 33.2 ms, but the first
    # base's launches are queued 4 ms later (the helper's table and upload work competes with the main thread for the interpreter)
    # - 51.0 ms per pass either way; made by the main thread itself the prologue costs 19 ms of host copies up front (65 ms).  The
    # start-up is host-bound table building, not missing device work: kept as a switch, rows identical (tests/test_gpu_sweep.py).
    prologue_on = os.environ.get("WDG_SWEEP_PROLOGUE", "0") not in ("0", "")
    pro_stream = _side_stream(4) if prologue_on else None
    widths = [next(iter(feats.values())).shape[1] if feats else 0 for _name, feats, _sm in bases]
    n = 0
    for si, (jobs, graph_inputs) in enumerate(shards):
        first = first_lo = None
        gb = None
        if prefetch:
            gb = queued.pop(si) if si in queued else queue_build(si)
        pro, early = None, None
        if jobs and (prologue_on or (prefetch and gb is not None)):
            lo_guess = [propagates(w, jobs) for w in widths]  # (what `lo` below becomes when every graph has its SELL-16 copy)
            guess = (_visit_order([(lo_guess[bi], bases[bi][2]) if lo_guess[bi] else ("own", bi) for bi in range(len(bases))])
                     if rebind_ok else list(range(len(bases))))
            wide = [bi for bi in guess if lo_guess[bi]]
            if prologue_on and wide:
                pro_stream.wait_stream(torch.cuda.current_stream())
                pro = FeaturePrologue(bases, wide, sorted({j.seed for j in jobs}), pro_stream)
            elif lo_guess[guess[0]]:
                # the FIRST base's feature matrices go into the upload ring while the GPU builds the shard's graphs: the copies (1.2 ms
                # of library threads for cora's two 11-MB matrices) otherwise sit between the build's read-back and the first launch
                from . import ops
                with torch.cuda.stream(streams[n % depth]):
                    early = (guess[0], {s_: ops._h2d(np.ascontiguousarray(bases[guess[0]][1][s_], np.float32), ops.require_gpu())
                                        for s_ in sorted({j.seed for j in jobs})})
        if prefetch and gb is not None:
            with torch.cuda.stream(build_stream):
                gb.finish()
        # (decided AFTER the graphs are built when they are built ahead: a graph without a SELL-16 copy rules the propagated route out
        # for the whole shard - its bases then aggregate at full width and take the dense Gram, as round 4 did)
        quad_ok = gb is None or all(g.quad for g in gb.graphs)
        lo = [quad_ok and propagates(w, jobs) for w in widths]
        order = _visit_order([(lo[bi], bases[bi][2]) if lo[bi] else ("own", bi) for bi in range(len(bases))]) if rebind_ok else list(range(len(bases)))
        free = {}  # (sample_max, stream) -> a prepared labels-only batch whose rows have been fetched
        for bi in order:
            _name, feats, sample_max = bases[bi]
            stream = streams[n % depth]
            if first is not None and depth > 1:
                stream.wait_stream(first_stream)  # (the shared graphs are built on the first base's stream)
            elif first is None and gb is not None:
                stream.wait_stream(build_stream)   # (... or, built ahead, on the build stream)
            with torch.cuda.stream(stream):
                width = widths[bi]
                key = (sample_max, stream.cuda_stream)
                sb = free.pop(key, None) if (rebind_ok and lo[bi] and jobs) else None
                pre = pro.get(bi) if (pro is not None and lo[bi]) else None  # (x, GramBatch, event) of the prologue
                x_early = early[1] if (early is not None and early[0] == bi and lo[bi] and first is None) else None  # (uploaded on this stream)
                if sb is not None:
                    sb.rebind_features(feats, width, first_seed + 1000 * bi, pre=pre)
                else:
                    inputs = [(src, dst, lab, feats[j.seed]) for j, (src, dst, lab) in zip(jobs, graph_inputs)]
                    sb = SweepBatch(jobs, n_feat=width, symmetric=symmetric, gcn_hidden=0, inputs=inputs,
                                    share=(first_lo or first) if lo[bi] else first, labels_only=lo[bi],
                                    graph_batch=gb if first is None else None, x_dev=pre[0] if pre else x_early)
                    if sb.jobs:
                        # (the node sets are keyed by the base and the job's identity, not by where the job sits: a job draws the
                        # same sets in whichever shard / on whichever rank it runs - the N-rank sweep computes the one-GPU sweep's rows)
                        sb.prepare_full(epochs=epochs, sample_max=sample_max, base_seed=first_seed + 1000 * bi,
                                        gx=(pre[1], pre[2]) if pre else None)
                    sb.step()
                if sb.jobs:
                    sb.launch_full()
            if first is None:
                first, first_stream = sb, stream
                if prefetch and si + 1 < len(shards):
                    queued[si + 1] = queue_build(si + 1)
            if first_lo is None and sb.labels_only:
                first_lo = sb
            in_flight.append((si, bi, sb, stream))
            n += 1
            if len(in_flight) >= depth:
                done = fetch()
                if (rebind_ok and done[0] == si and done[3].labels_only and done[3].jobs and getattr(done[3], "kr_sets", None) is not None
                        and getattr(done[3], "gram_route", None) == "propagate"):
                    free[(done[3].kr_sample_max, done[4].cuda_stream)] = done[3]
                yield done[:3]
        if pro is not None:
            pro.drain()
    while in_flight:
        yield fetch()[:3]


def _visit_order(keys):
    """indices of `keys` in an order that keeps equal keys apart: each time the key with the most entries left that differs from the
    one just taken (ties: first seen) - A A A B B C -> A B A B A C"""
    left = {}
    for i, k in enumerate(keys):
        left.setdefault(k, []).append(i)
    order, prev = [], object()
    while left:
        cands = [k for k in left if k != prev] or list(left)
        k = max(cands, key=lambda k_: (len(left[k_]), -left[k_][0]))
        order.append(left[k].pop(0))
        if not left[k]:
            del left[k]
        prev = k
    return order


class BaseSweep:
    """The feature bases of the reference's sweep (synthetic_plot.py:64-65,78-82: the 300 synthetic adjacencies are paired
    with features sampled from each of six base datasets, 1 800 jobs): one SweepBatch per base over the SAME jobs, all of
    them sharing the graphs of the first (CSR, SELL-16 copy, degrees, labels are built once); every base has its own
    feature width (the aggregation's grid follows it: ceil((F + C) / 16) feature groups; widths over 512 take two GEMM
    launches for the GCN forward instead of the fused transform)."""

    # (name, feature width) of the reference's bases; the real feature tables are absent from the checkout except pubmed's
    REFERENCE_BASES = (("cora", 1433), ("citeseer", 3703), ("pubmed", 500), ("chameleon", 2325), ("squirrel", 2089), ("film", 932))

    def __init__(self, jobs, bases=REFERENCE_BASES, symmetric=0, gcn_hidden=64):
        self.bases = list(bases)
        self.batches = []
        for bi, (_name, width) in enumerate(self.bases):
            self.batches.append(SweepBatch(jobs, n_feat=width, symmetric=symmetric, gcn_hidden=gcn_hidden,
                                           share=self.batches[0] if self.batches else None, feature_seed=1000 * bi))
        self.jobs = self.batches[0].jobs

    def step(self):
        """a step of every base (the integer edge/label pass is repeated per base: 40-60 us beside each base's GEMM)"""
        for b in self.batches:
            b.step()

    def results(self):
        """[bases, jobs, 6]: the step's scalars per base"""
        return torch.stack([b.results() for b in self.batches])

    @property
    def edges(self):
        return sum(b.edges for b in self.batches)


class TrainBatch:
    """Train + evaluate one model per graph for ALL graphs of a shard at once (the sweep's graphs/s
    with the model loop).  Every stage of an epoch is one batched launch over the job tables (aggregations forward and
    through the transposed graphs backward, GEMMs forward and for both gradients); PyTorch supplies the stacked
    parameters, log-softmax / NLL gradient on the stacked logits, the ReLU masks and one Adam over all models.  Shapes
    are static, nothing syncs with the host: an epoch (train step + evaluation + model selection) is captured once and
    replayed as a hipGraph.

        kind "sgc":  logits_j = (A_hat_j X) W_j                    (the aggregation Y_j is computed once)
        kind "gcn":  logits_j = A_hat_j relu(A_hat_j (X W0_j)) W1_j   (hidden 64)
    Per-graph reference with identical arithmetic: models.train_eval_graphed."""

    def __init__(self, sb, kind="gcn", hidden=64, lr=0.01, weight_decay=5e-4, train_frac=0.6, seed=0):
        from .utils.util_funcs import random_disassortative_splits
        ops = sb.ops
        self.sb, self.kind = sb, kind
        jobs = sb.jobs
        J = len(jobs)
        n, c, f = jobs[0].n_nodes, sb.n_classes, sb.n_feat
        if any(j.n_nodes != n for j in jobs):
            raise ValueError("TrainBatch: graphs of one batch must have the same node count")
        dev = sb.graphs[0].device
        self.J, self.n, self.c, self.f, self.h = J, n, c, f, hidden
        gen = torch.Generator(device="cpu").manual_seed(seed)
        labels = torch.stack([l.long() for l in sb.labels])  # [J, n]
        torch.manual_seed(seed)
        tr, va, te = [], [], []
        for j in range(J):  # the reference's split routine per graph (same sizes for every graph: balanced classes)
            a, b, d = random_disassortative_splits(labels[j].cpu(), labels[j].max().cpu() + 1, train_frac)
            tr.append(a.nonzero().flatten()); va.append(b.nonzero().flatten()); te.append(d.nonzero().flatten())
        self.tr, self.va, self.te = (torch.stack(t).to(dev) for t in (tr, va, te))  # [J, n_split] row indices
        self.y_tr, self.y_va, self.y_te = (labels.gather(1, t) for t in (self.tr, self.va, self.te))
        self.labels = labels

        def xavier(*shape):
            bound = (6.0 / (shape[-2] + shape[-1])) ** 0.5
            return ((torch.rand(shape, generator=gen) * 2 - 1) * bound).to(dev)

        self.logits = torch.empty((J, n, c), device=dev)
        self.dlogits = torch.zeros((J, n, c), device=dev)
        graphs_t = [g.transpose() for g in sb.graphs]
        rs = sb.dinv  # A_hat = diag(dinv) (A + I) (random-walk normalisation of the sweep); A_hat^T = (A + I)^T diag(dinv)

        def fwd_spmm(xs, ys):
            return ops.SpmmBatch([(g, x, y, d, None, False) for g, x, y, d in zip(sb.graphs, xs, ys, rs)])

        def bwd_spmm(xs, ys):
            return ops.SpmmBatch([(gt, x, y, None, d, False) for gt, x, y, d in zip(graphs_t, xs, ys, rs)])

        if kind == "sgc":
            sb.spmm.launch()  # Y_j = A_hat_j X, once
            torch.cuda.synchronize()
            ys = sb.y  # (row-major; a tiled Y is copied out here, once: the aggregation above is the only one)
            self.yt = torch.stack([y.t().contiguous() for y in ys])  # [J, F, n] for dW = Y^T dlogits
            self.w = torch.nn.Parameter(xavier(J, f, c))
            self.w.grad = torch.zeros_like(self.w)
            self.params = [self.w]
            self.fwd = [ops.GemmBatch([(ys[j], self.w.data[j], self.logits[j], None) for j in range(J)])]
            self.bwd = [ops.GemmBatch([(self.yt[j], self.dlogits[j], self.w.grad[j], None) for j in range(J)])]
        elif kind == "gcn":
            xt = {s: x.t().contiguous() for s, x in sb.x.items()}  # X^T per seed, for dW0 = X^T dP
            self.w0 = torch.nn.Parameter(xavier(J, f, hidden))
            self.w1 = torch.nn.Parameter(xavier(J, hidden, c))
            self.w0.grad, self.w1.grad = torch.zeros_like(self.w0), torch.zeros_like(self.w1)
            self.params = [self.w0, self.w1]
            z = lambda *s: torch.empty((J,) + s, device=dev)  # noqa: E731
            self.p, self.hid, self.hid_t, self.z = z(n, hidden), z(n, hidden), z(hidden, n), z(n, c)
            self.dz, self.dhid, self.dp, self.w1t = z(n, c), z(n, hidden), z(n, hidden), z(c, hidden)
            xs = [sb.x[j.seed] for j in jobs]
            self.fwd = [ops.GemmBatch([(xs[j], self.w0.data[j], self.p[j], None) for j in range(J)]),   # P = X W0
                        fwd_spmm(self.p, self.hid),                                                      # A_hat P (relu below)
                        ops.GemmBatch([(self.hid[j], self.w1.data[j], self.z[j], None) for j in range(J)]),  # Z = H W1
                        fwd_spmm(self.z, self.logits)]                                                   # logits = A_hat Z
            self.bwd = [bwd_spmm(self.dlogits, self.dz),                                                 # dZ = A_hat^T dlogits
                        ops.GemmBatch([(self.hid_t[j], self.dz[j], self.w1.grad[j], None) for j in range(J)]),  # dW1 = H^T dZ
                        ops.GemmBatch([(self.dz[j], self.w1t[j], self.dhid[j], None) for j in range(J)]),       # dH = dZ W1^T
                        bwd_spmm(self.dhid, self.dp),                                                    # dP = A_hat^T (dH * mask)
                        ops.GemmBatch([(xt[jobs[j].seed], self.dp[j], self.w0.grad[j], None) for j in range(J)])]  # dW0 = X^T dP
        else:
            raise ValueError(f"unknown model kind {kind!r}")
        self.opt = torch.optim.Adam(self.params, lr=lr, weight_decay=weight_decay, capturable=True)
        self.best_val = torch.full((J,), -1.0, device=dev)
        self.best_test = torch.zeros(J, device=dev)
        self.graph = None

    # -- one epoch ---------------------------------------------------------------------------------------------
    def _forward(self):
        if self.kind == "sgc":
            self.fwd[0].launch()
        else:
            self.fwd[0].launch()
            self.fwd[1].launch()
            self.hid.clamp_(min=0)  # relu
            self.fwd[2].launch()
            self.fwd[3].launch()

    def train_step(self):
        # (the forward pass of these weights has been run already: by the previous epoch's evaluation, or by run())
        with torch.no_grad():
            # d(mean NLL over the training rows) / dlogits = (softmax - onehot) / n_train on those rows, 0 elsewhere
            sm = torch.softmax(self.logits.gather(1, self.tr.unsqueeze(-1).expand(-1, -1, self.c)), 2)
            sm.scatter_add_(2, self.y_tr.unsqueeze(-1), torch.full_like(sm[..., :1], -1.0))
            self.dlogits.zero_()
            self.dlogits.scatter_(1, self.tr.unsqueeze(-1).expand(-1, -1, self.c), sm / self.tr.shape[1])
            if self.kind == "sgc":
                self.bwd[0].launch()
            else:
                self.bwd[0].launch()
                self.hid_t.copy_(self.hid.transpose(1, 2))
                self.bwd[1].launch()
                self.w1t.copy_(self.w1.data.transpose(1, 2))
                self.bwd[2].launch()
                self.dhid.mul_(self.hid > 0)
                self.bwd[3].launch()
                self.bwd[4].launch()
        self.opt.step()

    def eval_step(self):
        with torch.no_grad():
            self._forward()
            pred = self.logits.argmax(2)
            v = (pred.gather(1, self.va) == self.y_va).float().mean(1)
            t = (pred.gather(1, self.te) == self.y_te).float().mean(1)
            better = v > self.best_val
            self.best_test.copy_(torch.where(better, t, self.best_test))
            self.best_val.copy_(torch.where(better, v, self.best_val))

    def epoch(self):
        """gradient of the current logits -> Adam step -> forward with the new weights -> evaluation.  The evaluation's
        forward pass is the next epoch's training forward pass (no dropout: the two would be identical)."""
        self.train_step()
        self.eval_step()

    def capture(self):
        """Capture one epoch as a hipGraph (after a warm-up whose effects are rewound); returns the replay callable."""
        saved = [p.detach().clone() for p in self.params]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            with torch.no_grad():
                self._forward()
            for _ in range(2):
                self.epoch()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        with torch.no_grad():
            for p, s in zip(self.params, saved):
                p.copy_(s)
            for st in self.opt.state.values():
                for v in st.values():
                    if torch.is_tensor(v):
                        v.zero_()
            self.best_val.fill_(-1.0)
            self.best_test.zero_()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.epoch()
        with torch.no_grad():
            for p, s in zip(self.params, saved):
                p.copy_(s)
        return self.graph.replay

    def run(self, epochs=200, capture=True):
        """-> dict(val_acc [J], test_acc [J], seconds, graphs_per_s): train + evaluate every model for `epochs` epochs."""
        import time
        step = self.capture() if capture else self.epoch
        with torch.no_grad():
            self._forward()  # logits of the initial weights
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(epochs):
            step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        return dict(val_acc=self.best_val.cpu(), test_acc=self.best_test.cpu(), seconds=dt, graphs_per_s=self.J / dt, epochs=epochs)
