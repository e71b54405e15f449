"""CPU: the N>1 exchange of eirgrid_amd.parallel on two gloo ranks (no GPU): shard ranges, the sum all-reduce of the
integer statistics, best-candidate selection + broadcast, and bit-identical policy replicas after the update."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_covers_everything():
    from eirgrid_amd.parallel import shard_range
    for total in (1, 7, 1024, 131072):
        for ws in (1, 2, 3, 8):
            spans = [shard_range(total, r, ws) for r in range(ws)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == total
            for (f0, c0), (f1, _) in zip(spans, spans[1:]):
                assert f0 + c0 == f1
    assert shard_range(131072, 3, 8) == (3 * 16384, 16384)


def test_pick_candidate_ties_to_lowest_index():
    from eirgrid_amd.parallel import pick_candidate
    assert pick_candidate([(1.5, 10), (1.7, 99), (1.7, 42), (-1.0, -1)]) == (2, 1.7, 42)
    assert pick_candidate([(-1.0, -1), (-1.0, -1)]) is None


def _rank_episodes(rank, world):
    """Episodes of this rank from the CPU oracle's tabled mode (test infrastructure)."""
    from eirgrid_amd.engine import HostTables
    from oracle import api as O
    tb = O.OracleTables(HostTables(world), len(world.existing_x))
    eps = []
    for e in range(4):
        st, out = O.run_episode_tabled(tb, O.OracleWeights(), 500 + rank * 4 + e)
        eps.append(out)
    return eps, [O.score_metrics(list(o.metrics)) for o in eps]


def _worker(rank, ws, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        from eirgrid_amd import _native as N, synthetic_world
        from eirgrid_amd.engine import ActionWeights, apply_reduced
        from eirgrid_amd.parallel import exchange_update, pack_candidate
        world = synthetic_world()
        eps, scores = _rank_episodes(rank, world)
        stats = np.zeros(N.STATS_LEN, dtype=np.int64)
        stats[0] = len(eps)           # first batch: no best yet, so the statistics kernel emits only the episode count
        best = int(np.argmax(scores))
        o = eps[best]
        nr, nd = np.array(o.n_run, np.int32), np.array(o.n_def, np.int32)
        rl = np.zeros(N.RUN_CAP, np.uint8); rl[:nr.sum()] = list(o.run_log[:nr.sum()])
        dl = np.zeros(N.DEF_CAP, np.uint8); dl[:nd.sum()] = list(o.def_log[:nd.sum()])
        t = torch.from_numpy(stats.copy())
        summed, cand = exchange_update(t, (scores[best], 500 + rank * 4 + best), lambda: pack_candidate(list(o.metrics), nr, rl, nd, dl),
                                       dist, torch.device("cpu"))
        pol = ActionWeights()
        improved = apply_reduced(pol, summed, cand, noise_seed=1)
        w, dw, _ = pol.tables()
        q.put((rank, int(summed[0]), improved, [float(v) for v in cand[0]], pol.get("iteration_count"),
               pol.get("iterations_without_improvement"), w.tobytes() + dw.tobytes(), [pol.get_list(0, y) for y in range(26)],
               scores, [list(e.metrics) for e in eps]))
    finally:
        dist.destroy_process_group()


def test_two_rank_exchange(built):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    a, b = res
    assert a[1] == b[1] == 8                       # all-reduced episode count
    assert a[2] and b[2]                           # first batch: the candidate becomes the best on both ranks
    assert a[3] == b[3] and a[4] == b[4] == 8 and a[5] == b[5] == 0
    assert a[6] == b[6] and a[7] == b[7]           # bit-identical replicas
    scores, metrics = a[8] + b[8], a[9] + b[9]
    assert a[3] == metrics[int(np.argmax(scores))]   # the winner is the global arg-max (np.argmax: lowest index on ties)


def test_apply_packet_equals_pick_then_apply_reduced(built):
    """eg_policy_apply_packet (C: winner by score, ties to the lowest global index) == pick_candidate + apply_reduced."""
    import numpy as np
    from eirgrid_amd import _native as N
    from eirgrid_amd.engine import ActionWeights, apply_packet, apply_reduced
    from eirgrid_amd.parallel import pack_candidate, parse_candidate, pick_candidate
    rng = np.random.default_rng(11)
    for trial in range(4):
        stats = np.zeros(N.STATS_LEN, np.int64)
        stats[0] = 40; stats[2] = 12
        stats[8:8 + 2 * 26 * 61] = -rng.integers(0, 2**28, 2 * 26 * 61)
        stats[8 + 2 * 26 * 61:] = rng.integers(0, 5, 26 * 15)
        recs = np.zeros((3, N.CANDIDATE_BYTES), np.uint8)
        scores = [0.41, 0.77, 0.77] if trial % 2 == 0 else [0.2, 0.1, 0.3]
        index = [5, 90, 31] if trial < 3 else [-1, -1, -1]
        for r in range(3):
            nr = rng.integers(0, 4, 26).astype(np.int32); nd = rng.integers(0, 3, 26).astype(np.int32)
            rl = np.zeros(N.RUN_CAP, np.uint8); rl[:int(nr.sum())] = rng.integers(0, 61, int(nr.sum()))
            dl = np.zeros(N.DEF_CAP, np.uint8); dl[:int(nd.sum())] = 3 * rng.integers(0, 15, int(nd.sum()))
            metrics = [-1000.0 * (r + 1), 0.7, 3e10 / (1.0 + scores[r]), 1.0]
            recs[r, 0:8] = np.array([scores[r]], np.float64).view(np.uint8)
            recs[r, 8:16] = np.array([index[r]], np.int64).view(np.uint8)
            recs[r, 16:] = pack_candidate(metrics, nr, rl, nd, dl)
        a, b = ActionWeights(), ActionWeights()
        for pol in (a, b):     # a best strategy must exist for the contrast step to do anything
            pol.apply_episode([5e5, 0.6, 9e11, 1.0], np.ones(26, np.int32), np.full(26, 60, np.uint8), np.zeros(26, np.int32), np.zeros(0, np.uint8))
        parsed = [parse_candidate(recs[r]) for r in range(3)]
        win = pick_candidate([(p[0], p[1]) for p in parsed])
        fa = apply_packet(a, stats, recs, noise_seed=trial)
        fb = apply_reduced(b, stats, parsed[win[0]][2] if win else None, noise_seed=trial)
        assert fa == fb
        if trial == 0:
            assert win[0] == 2          # tie on the score: the lower global index wins
        for x, y in zip(a.tables(), b.tables()):
            assert x.tobytes() == y.tobytes()
        assert a.lists(0) == b.lists(0) and a.get("iterations_without_improvement") == b.get("iterations_without_improvement")
