"""CPU tests of the multi-process sweep logic (world_size 2, gloo): job-table broadcast, static sharding and
result all_gather - the N>1 path of bench.py without a GPU.  The per-job evaluation is done by the oracle here
(tests may use it); on a GPU node the same driver code runs the HIP batch instead."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from wdg_amd import sweep, synth


def test_shard_jobs_partitions_the_job_list():
    """every job exactly once, deterministic, balanced by job_cost - at job granularity (a seed may split: X is 4 MB)"""
    for hl, k, n_seeds in ((synth.H_LEVELS_10, 2, 7), (synth.H_LEVELS_10_K10, 10, 5)):
        jobs = sweep.make_jobs(hl, range(n_seeds), k=k)
        for ws in (1, 2, 3, 8):
            shards = [sweep.shard_jobs(jobs, ws, r) for r in range(ws)]
            flat = [j for s in shards for j in s]
            assert sorted(flat, key=lambda j: (j.seed, j.h)) == sorted(jobs, key=lambda j: (j.seed, j.h))
            assert shards == [sweep.shard_jobs(list(jobs), ws, r) for r in range(ws)]
            for s in shards:  # the jobs of a seed stay adjacent inside a shard (they share X), in list order
                assert s == sorted(s, key=lambda j: (j.seed, jobs.index(j)))
            loads = [sum(sweep.job_cost(j) for j in s) for s in shards]
            assert max(loads) <= 1.15 * min(loads), (k, ws, loads)
    assert sweep.decode_jobs(sweep.encode_jobs(jobs)) == jobs


def test_shard_jobs_north_star_split():
    """BASELINE configs[2] literally: 10 h x 5 seeds = 50 jobs over 8 GPUs -> 6 - 7 jobs per rank, loads within 15 %
    (whole-seed dealing gave [10, 10, 10, 10, 10, 0, 0, 0]: three idle GPUs)"""
    jobs = sweep.make_jobs(synth.H_LEVELS_10_K10, range(5), k=10)
    shards = [sweep.shard_jobs(jobs, 8, r) for r in range(8)]
    assert sorted(len(s) for s in shards) == [6, 6, 6, 6, 6, 6, 7, 7]
    loads = [sum(sweep.job_cost(j) for j in s) for s in shards]
    assert max(loads) / min(loads) <= 1.15
    edges = [sum(j.nnz for j in s) for s in shards]
    assert max(edges) / min(edges) <= 1.25  # (the cost has a per-row part: stored entries alone balance less tightly)


def test_shard_pairs_is_a_contiguous_sample_aware_partition():
    """the nine-scalar sweep's partition (whole_sweep_rank): every adjacency exactly once, contiguous in sample-major order, as few
    samples per rank as the split allows (the reference's 280 adjacencies on 8 ranks: two each - shard_jobs' LPT dealt up to three),
    modelled costs within 3 % of each other, deterministic; more ranks than adjacencies: the spare ranks get nothing"""
    levels = [h for h in synth.H_LEVELS_30 if h not in (0.05, 0.1)]
    pairs = sweep.make_jobs(levels, range(10), k=10, n_nodes=2000)
    key = lambda j: (j.seed, levels.index(j.h))  # noqa: E731
    for ws in (1, 2, 3, 4, 5, 8, 16):
        shards = [sweep.shard_pairs(pairs, ws, r) for r in range(ws)]
        assert sum(shards, []) == sorted(pairs, key=key)  # contiguous ranges of the sample-major list, in order
        assert shards == [sweep.shard_pairs(list(pairs), ws, r) for r in range(ws)]
        cost = [sum(sweep.PAIR_COST_US + sweep.PAIR_COST_PER_ENTRY_US * j.nnz for j in m) + sweep.SAMPLE_COST_US * len({j.seed for j in m})
                for m in shards]
        assert max(cost) <= 1.03 * min(cost) or ws > 8, (ws, cost)
        if ws == 8:
            assert all(len({j.seed for j in m}) == 2 for m in shards)
            lpt = [sweep.shard_jobs(pairs, 8, r) for r in range(8)]
            assert sum(len({j.seed for j in m}) for m in lpt) > 16  # (what the sample-aware split saves)
    few = pairs[:3]
    got = [sweep.shard_pairs(few, 5, r) for r in range(5)]
    assert sum(got, []) == few and sum(1 for m in got if m) == 3
    assert sweep.shard_pairs([], 4, 1) == []


def test_shard_jobs_fewer_jobs_than_ranks():
    jobs = sweep.make_jobs([0.2, 0.5], [0], k=2, n_nodes=200)
    shards = [sweep.shard_jobs(jobs, 3, r) for r in range(3)]
    assert sorted(len(s) for s in shards) == [0, 1, 1]
    assert sweep.shard_jobs([], 4, 2) == []
    assert sweep.encode_jobs([]).shape == (0, 5) and sweep.decode_jobs(sweep.encode_jobs([])) == []


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir, h_levels=(0.1, 0.5, 0.9), seeds=4):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as orc
    dev = torch.device("cpu")
    jobs = sweep.make_jobs(list(h_levels), range(seeds), n_nodes=200) if rank == 0 else []
    jobs = sweep.broadcast_jobs(jobs, dev)
    mine = sweep.shard_jobs(jobs, world, rank)
    rows = []
    for j in mine:  # stand-in evaluator: oracle metrics per job, keyed so the gather can be checked
        src, dst, lab = synth.regular_graph(j.n_nodes, j.n_classes, j.k, j.h, j.seed)
        rowptr, col, _ = orc.coo_to_csr(src, dst, j.n_nodes, None, orc.ADD_SELF_LOOPS)
        st = orc.edge_label_stats(rowptr, col, lab, j.n_classes)
        rows.append([j.seed, j.h, orc.edge_homophily_dense(st), orc.node_homophily_dense(st)])
    local = torch.tensor(rows, dtype=torch.float64).reshape(-1, 4)
    gathered = sweep.gather_results(local, dev)
    torch.save(dict(jobs=len(jobs), mine=len(mine), gathered=[g.clone() for g in gathered]),
               os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_welch_p_values_equal_scipys_ttest():
    """sweep.welch_p_values (the host tail of a shard: the reference's ttest_ind + the side rule of utils/homophily_metrics.py:335-347
    for every (job, classifier) at once) against scipy on the reference's fp32 lists: random accuracies, a tie in every epoch,
    two constant samples (NaN propagates), one-sided wins"""
    import warnings
    import numpy as np
    from _golden import welch_p
    from wdg_amd.sweep import welch_p_values
    rng = np.random.default_rng(3)
    g = (rng.integers(100, 200, (40, 2, 100)) / 200).astype(np.float32)
    x = (rng.integers(90, 190, (40, 2, 100)) / 200).astype(np.float32)
    x[3, 0] = g[3, 0]
    g[4, 1], x[4, 1] = 0.5, 0.5
    x[5, 0] = g[5, 0] - 0.2  # the graph-aware regression wins every epoch
    x[6, 1] = g[6, 1] + 0.2
    g[7, 0], x[7, 0] = 1.0, 0.5  # two constant samples with DIFFERENT means: scipy sets df = 1, t = -inf -> p = 0 -> the metric is 1
    g[8, 1], x[8, 1] = 0.25, 0.75
    got = welch_p_values(g, x)
    assert got.shape == (40, 2)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = np.array([[welch_p(g[j, c], x[j, c]) for c in range(2)] for j in range(40)])
    assert np.array_equal(np.isnan(got), np.isnan(want)) and np.isnan(got).sum() == 1
    assert np.nanmax(np.abs(got - want)) < 1e-6
    assert got[5, 0] > 1 - 1e-6 and got[6, 1] < 1e-6  # (the reference's side rule: 1 - p / 2 when the graph-aware side wins)
    assert got[7, 0] == 1.0 and got[8, 1] == 0.0 and want[7, 0] == 1.0 and want[8, 1] == 0.0


def test_two_rank_sweep_gloo(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(tmp_path / f"r{r}.pt") for r in range(world)]
    assert res[0]["jobs"] == res[1]["jobs"] == 12
    assert res[0]["mine"] + res[1]["mine"] == 12 and res[0]["mine"] == 6
    for r in range(world):  # every rank holds everybody's rows after the all_gather
        g = torch.cat(res[r]["gathered"])
        assert g.shape == (12, 4)
        assert sorted((int(a), float(b)) for a, b in g[:, :2].tolist()) == sorted((s, h) for s in range(4) for h in (0.1, 0.5, 0.9))
        for seed, h, eh, nh in g.tolist():
            d = int(2 / h)
            assert abs(eh - 2 / d) < 1e-12 and abs(nh - 3 / (d + 1)) < 1e-6  # generator's known answers
    assert torch.equal(torch.cat(res[0]["gathered"]), torch.cat(res[1]["gathered"]))


def test_three_ranks_two_jobs_one_empty_shard_gloo(tmp_path):
    """more ranks than jobs: one rank's shard is empty - it still takes part in the broadcast and the all_gather"""
    world, port = 3, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), (0.2, 0.5), 1), nprocs=world, join=True)
    res = [torch.load(tmp_path / f"r{r}.pt") for r in range(world)]
    assert [r["jobs"] for r in res] == [2, 2, 2]
    assert sorted(r["mine"] for r in res) == [0, 1, 1]
    for r in range(world):
        assert sorted(g.shape[0] for g in res[r]["gathered"]) == [0, 1, 1]
        g = torch.cat(res[r]["gathered"])
        assert g.shape == (2, 4) and sorted(g[:, 1].tolist()) == [0.2, 0.5]
        for _seed, h, eh, _nh in g.tolist():
            assert abs(eh - 2 / int(2 / h)) < 1e-12


def _exchange_worker(rank, world, port, out_dir, counts):
    """keyed nine-column fp64 rows, uneven shards (counts[rank] rows, possibly none) through sweep.exchange_rows"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_total = sum(counts)
    first = sum(counts[:rank])
    # ranks own interleaved keys (as shard_jobs deals jobs: not contiguous ranges), rows = f(key) so the table can be checked
    perm = np.random.default_rng(5).permutation(n_total)
    keys = torch.from_numpy(np.sort(perm[first:first + counts[rank]]).astype(np.int64))
    rows = keys.double()[:, None] * 10 + torch.arange(9, dtype=torch.float64)[None, :] / 7
    if keys.numel():
        rows[0, 8] = float("nan")  # (a NaN p-value travels like any other value)
    table = sweep.exchange_rows(keys, rows, n_total + 1, torch.device("cpu"))  # (+ 1: one key nobody owns)
    torch.save(dict(keys=keys, table=table), os.path.join(out_dir, f"x{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def _check_exchange(tmp_path, world, counts):
    res = [torch.load(tmp_path / f"x{r}.pt") for r in range(world)]
    n_total = sum(counts)
    t0 = res[0]["table"]
    assert t0.shape == (n_total + 1, 9) and t0.dtype == torch.float64
    for r in range(1, world):  # every rank assembles the same table
        assert torch.equal(torch.nan_to_num(res[r]["table"], nan=-7.0), torch.nan_to_num(t0, nan=-7.0))
    assert torch.isnan(t0[n_total]).all()  # the key nobody sent
    firsts = {int(r["keys"][0]) for r in res if r["keys"].numel()}
    for key in range(n_total):
        want = key * 10 + torch.arange(9, dtype=torch.float64) / 7
        if key in firsts:
            assert torch.isnan(t0[key, 8]) and torch.equal(t0[key, :8], want[:8])
        else:
            assert torch.equal(t0[key], want), key
    assert sorted(int(k) for r in res for k in r["keys"]) == list(range(n_total))


def test_keyed_nine_column_rows_two_ranks_uneven_gloo(tmp_path):
    world, port, counts = 2, _free_port(), (7, 4)
    mp.spawn(_exchange_worker, args=(world, port, str(tmp_path), counts), nprocs=world, join=True)
    _check_exchange(tmp_path, world, counts)


def test_keyed_nine_column_rows_three_ranks_one_empty_gloo(tmp_path):
    world, port, counts = 3, _free_port(), (5, 0, 3)
    mp.spawn(_exchange_worker, args=(world, port, str(tmp_path), counts), nprocs=world, join=True)
    _check_exchange(tmp_path, world, counts)


def test_exchange_rows_single_process_and_bad_key():
    import pytest
    t = sweep.exchange_rows(torch.tensor([2, 0]), torch.tensor([[1.0, 2.0], [3.0, 4.0]]), 3, torch.device("cpu"))
    assert torch.equal(t[0], torch.tensor([3.0, 4.0], dtype=torch.float64)) and torch.isnan(t[1]).all()
    assert torch.equal(t[2], torch.tensor([1.0, 2.0], dtype=torch.float64))
    assert sweep.exchange_rows(torch.zeros(0, dtype=torch.int64), torch.zeros((0, 9)), 0, torch.device("cpu")) is not None
    with pytest.raises(ValueError):
        sweep.exchange_rows(torch.tensor([5]), torch.zeros((1, 2)), 3, torch.device("cpu"))


def test_visit_order_keeps_equal_keys_apart():
    """sweep._visit_order (run_bases: the order the feature bases of a shard are visited in, so that a prepared batch is free
    again when the next base of its kind comes up): a permutation; equal keys never adjacent while another key is left"""
    order_of = sweep._visit_order
    assert order_of(["a", "a", "a", "b", "b", "c"]) == [0, 3, 1, 4, 2, 5]
    assert order_of([("p", 500), ("p", 500), ("own", 2), ("p", 300), ("p", 500), ("p", 300)]) == [0, 3, 1, 2, 4, 5]
    assert order_of(["a", "a", "a"]) == [0, 1, 2] and order_of([]) == []
    import random
    rnd = random.Random(3)
    for _ in range(200):
        keys = [rnd.choice("abcd") for _ in range(rnd.randint(1, 9))]
        order = order_of(keys)
        assert sorted(order) == list(range(len(keys)))
        for a, b, rest in zip(order, order[1:], range(len(order) - 1, 0, -1)):
            if keys[a] == keys[b]:  # only when nothing else was left
                assert all(keys[i] == keys[a] for i in order[order.index(b):])
