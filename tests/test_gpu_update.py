"""GPU: the batch update path — statistics kernel vs a numpy restatement, the reduced update, and the trainer loop."""
import math

import numpy as np
import pytest
import torch

from eirgrid_amd import _native as N
from eirgrid_amd.engine import ActionWeights, apply_reduced, score_metrics
from eirgrid_amd.parallel import BatchTrainer

pytestmark = pytest.mark.gpu

DEFICIT_SLOT = {24: 0, 21: 1, 36: 2, 33: 3, 27: 4, 0: 5, 3: 6, 12: 7, 30: 8, 15: 9, 6: 10, 9: 11, 39: 12, 42: 13, 60: 14}


def _seed_best(engine):
    pol = ActionWeights()
    first = engine.run_iteration(0, pol, False, 12345)
    pol.apply_episode(first.metrics[0], first.n_run[0], first.run_log[0, :first.n_run[0].sum()], first.n_def[0],
                      first.def_log[0, :first.n_def[0].sum()])
    return pol


def _expected_stats(pol, res):
    """learning.rs:131-255 / :285-352 accumulated over a batch, straight from the definitions."""
    n = len(res.status)
    best = [pol.get_list(0, y) for y in range(26)]; bestd = [pol.get_list(1, y) for y in range(26)]
    s0 = score_metrics([pol.get(k) for k in ("best_net_emissions", "best_opinion", "best_cost", "best_reliability")])
    k = pol.get("iterations_without_improvement"); lr = pol.get("learning_rate")
    thr = 0.1 * max(math.exp(-k / 500.0), 1e-4)
    stag = 1.0 + 0.2 * (k / 10.0) ** 1.8; alr = lr * (1.0 + 0.1 * k)
    pen = np.zeros((26, 61)); mild = np.zeros((26, 61)); dcnt = np.zeros((26, 15), np.int64); nq = 0
    scores = np.zeros(n)
    for e in range(n):
        s = score_metrics(res.metrics[e]); scores[e] = s
        det = (s0 - s) / s0 if s0 > 0 else 0.0
        q = det > thr or k > 800
        nq += q
        run, dfl = res.lists(e, "run"), res.lists(e, "def")
        for y in range(26):
            cb = best[y] + bestd[y]
            if q:
                assert det >= 0          # (a better episode under forced contrast takes the NaN branch: tests/test_gpu_reduced_oracle.py)
                comb = det ** 0.3 * stag
                for j, a in enumerate(run[y] + dfl[y]):
                    if a not in cb:
                        pen[y, a] += math.log(1.0 / (1.0 + alr * 1.5 * comb))
                    elif j < len(cb) and a != cb[j]:
                        mild[y, a] += math.log(1.0 / (1.0 + alr * comb * 0.5))
            for a in dfl[y]:
                if a not in bestd[y] and a in DEFICIT_SLOT:
                    dcnt[y, DEFICIT_SLOT[a]] += 1
    return scores, nq, pen, mild, dcnt


def test_stats_kernel_vs_numpy(engine):
    pol = _seed_best(engine)
    n = 512
    engine.upload_snapshot(pol)
    engine.launch(999, 0, n)
    stats = torch.zeros(N.STATS_LEN, dtype=torch.int64, device="cuda")
    engine.update_stats(stats.data_ptr())
    dev_scores = engine.fetch_scores(n)
    res = engine.fetch(n)
    st = stats.cpu().numpy()
    scores, nq, pen, mild, dcnt = _expected_stats(pol, res)
    assert st[0] == n and st[1] == 0 and st[2] == nq and nq > 0
    np.testing.assert_allclose(dev_scores, scores, rtol=1e-14)
    A = 26 * 61
    np.testing.assert_allclose(st[8:8 + A].reshape(26, 61) / 2.0**32, pen, rtol=0, atol=n * 100 * 2.0**-31)
    np.testing.assert_allclose(st[8 + A:8 + 2 * A].reshape(26, 61) / 2.0**32, mild, rtol=0, atol=n * 100 * 2.0**-31)
    assert (st[8 + 2 * A:].reshape(26, 15) == dcnt).all()
    assert (pen <= 0).all() and pen.min() < 0
    # slot 3: the best score of the batch as a sortable integer (bit pattern + 1), a maximum rather than a sum
    assert int(st[3]) == int(np.float64(dev_scores.max()).view(np.int64)) + 1
    # the statistics are integer sums: launching the same batch again gives the identical buffer
    stats2 = torch.zeros_like(stats)
    engine.launch(999, 0, n); engine.update_stats(stats2.data_ptr()); engine.sync()
    assert torch.equal(stats, stats2)
    # and they are shard-invariant: two half batches add up to the whole
    parts = torch.zeros_like(stats)
    for f in (0, n // 2):
        t = torch.zeros_like(stats)
        engine.launch(999, f, n // 2); engine.update_stats(t.data_ptr()); engine.sync()
        key = max(int(parts[3]), int(t[3]))
        parts += t
        parts[3] = key
    assert torch.equal(parts, stats)


def test_reduced_update_moves_weights_the_right_way(engine):
    pol = _seed_best(engine)
    w0, dw0, _ = pol.tables()
    n = 256
    engine.upload_snapshot(pol); engine.launch(5, 0, n)
    stats = torch.zeros(N.STATS_LEN, dtype=torch.int64, device="cuda")
    engine.update_stats(stats.data_ptr())
    scores = engine.fetch_scores(n)
    b = int(np.argmax(scores))
    improved = apply_reduced(pol, stats.cpu().numpy(), engine.fetch_episode_lists(b), noise_seed=3)
    w1, dw1, _ = pol.tables()
    assert pol.get("iteration_count") == 1 + n
    assert w1.min() >= 1e-4 and w1.max() <= 0.999
    st = stats.cpu().numpy(); A = 26 * 61
    pen = st[8:8 + A].reshape(26, 61)
    # actions that were only penalised went down; best actions (boosted, never penalised) went up
    if st[2] > 0:
        only_pen = (pen < 0)
        best_occ = np.zeros((26, 61), bool)
        # best lists before the update were those of config 1's episode
        assert (w1[only_pen & ~best_occ] <= w0[only_pen & ~best_occ]).all() or improved
    assert improved == (pol.get("iterations_without_improvement") == 0)


def test_trainer_steps_are_deterministic(engine):
    """BatchTrainer in both modes (host updates / device-resident policy): repeatable, and the two modes agree."""
    runs = []
    for device_resident in (False, False, True, True):
        pol = ActionWeights()
        tr = BatchTrainer(engine, pol, 128, 4321, replay_fraction=0.1, device_resident=device_resident)
        flags = [tr.step() for _ in range(6)]
        tr.sync()
        w, dw, _ = pol.tables()
        runs.append((w.tobytes(), dw.tobytes(), pol.get("iteration_count"), pol.lists(0), tr.improvements))
        if not device_resident:
            assert flags[0] is True and sum(flags) == tr.improvements
    assert runs[0] == runs[1] == runs[2] == runs[3]
    assert runs[0][2] == 6 * 128 and runs[0][4] >= 1


def test_fused_packet_matches_separate_kernels(engine):
    """eg_rollout_launch_update (statistics in the rollout epilogue + device-side best pick) == launch + eg_update_stats."""
    from eirgrid_amd.parallel import exchange_packet
    pol = _seed_best(engine)
    n = 384
    engine.upload_snapshot(pol)
    packet = torch.zeros(N.PACKET_BYTES, dtype=torch.uint8, device="cuda")
    engine.launch_update(31, 7000, n, packet.data_ptr())
    stats_f, cand = exchange_packet(packet)
    stats = torch.zeros(N.STATS_LEN, dtype=torch.int64, device="cuda")
    engine.launch(31, 7000, n); engine.update_stats(stats.data_ptr())
    scores = engine.fetch_scores(n)
    assert (stats.cpu().numpy() == stats_f).all()
    b = int(np.argmax(scores))          # first maximum = lowest index on ties
    m, nr, rl, nd, dl = engine.fetch_episode_lists(b)
    assert cand is not None
    assert cand[0].tobytes() == m.tobytes() and (cand[1] == nr).all() and (cand[3] == nd).all()
    assert cand[2][:nr.sum()].tobytes() == rl[:nr.sum()].tobytes() and cand[4][:nd.sum()].tobytes() == dl[:nd.sum()].tobytes()
    host = packet.cpu().numpy()[8 * N.STATS_LEN:]
    assert host[0:8].view(np.float64)[0] == scores[b] and host[8:16].view(np.int64)[0] == 7000 + b


def test_two_rank_bench_rehearsal_on_one_gpu():
    """The N > 1 path of bench.py end to end (sharded global indices, stats all-reduce, candidate gather, identical
    replicas) with two ranks sharing cuda:0 over gloo — RCCL itself needs the driver's multi-GPU node."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--episodes", "256",
           "--backend", "gloo", "--share-gpu", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["value"] > 0 and line["scaling"] == "weak"
    assert line["config"]["last_batch"]["ok"] == 256 and line["config"]["episodes_failed"] == 0
    assert 0.0 < line["roofline"]["frac"] <= 1.0 and line["roofline"]["touched_bytes_per_launch"] > 0


def test_two_rank_bench_rehearsal_at_the_real_shard_size():
    """The same rehearsal at the size the 8-GPU run uses per rank — BASELINE configs[3]'s shard: 16 384 episodes per rank and batch,
    every 10th global index a replay, the sustained (grown-replay) state — two processes sharing cuda:0, gloo in place of RCCL.
    No episode fails, both ranks did the same work and hold the same policy after the last update (identical replica digests):
    what is left to the first real multi-GPU run is RCCL itself."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29537", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--min-seconds", "0.05",
           "--backend", "gloo", "--share-gpu", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    cfg = line["config"]
    assert line["n_gpus"] == 2 and cfg["episodes_per_gpu_per_batch"] == 16384 and cfg["replay_fraction"] == 0.1
    assert cfg["episodes_failed"] == 0 and cfg["last_batch"]["ok"] == 16384 and cfg["last_batch"]["overflow"] == 0
    assert cfg["last_batch"]["replay_episodes"] in (1638, 1639) and cfg["last_batch"]["generators_per_replay_episode"] > 100
    assert len(cfg["replica_digests"]) == 2 and len(set(cfg["replica_digests"])) == 1, cfg["replica_digests"]
    assert line["value"] > 0 and line["scaling"] == "weak"
    # the same batches with the replay hoist on, and the state the 2-rank loop itself reaches (grown at 2 x 16 384 per update on
    # every rank, timed with the exchange in the loop): both paths, nobody fails
    h = line["config2_replay_hoisted"]
    assert h["hoist_served_last_batch"] and h["episodes_failed"] == 0 and h["value"] > 0
    sn = line["sustained_at_n"]
    for key in ("per_episode_replays", "replay_hoisted"):
        assert sn[key]["value"] > 0 and sn[key]["episodes_failed"] == 0 and sn[key]["last_batch"]["ok"] == 16384
    assert sn["replay_hoisted"]["hoist_served_last_batch"]
    assert sn["per_episode_replays"]["replay"] == sn["replay_hoisted"]["replay"] and sn["replay_hoisted"]["replay"]["best_list_len"] > 96


def test_bench_line_contract_on_one_gpu():
    """`python bench.py --steps K --warmup W` prints ONE JSON line with the driver's keys (steps / warmup echoed), the `roofline`
    and `cpu_baseline` objects, and the two secondary objects; no episode fails; every batch started from the same policy."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1", "--episodes", "2048",
                          "--min-seconds", "0.05"], capture_output=True, text=True, timeout=900, cwd=root)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    for key, want in (("metric", "26-year episodes/sec"), ("unit", "episodes/s"), ("n_gpus", 1), ("steps", 3), ("warmup", 1),
                      ("higher_is_better", True), ("scaling", "weak"), ("vs_baseline", None), ("dtype", "f64"), ("data", "synthetic")):
        assert line[key] == want, key
    assert line["value"] > 0 and abs(line["ms_per_step"] - line["ms_per_batch"] * line["batches_per_step"]) < 1e-6 * line["ms_per_step"]
    cfg = line["config"]
    assert "workload" in cfg and cfg["episodes_failed"] == 0 and cfg["last_batch"]["ok"] == 2048 and cfg["policy"].startswith("pinned")
    # the headline is the state the training loop sustains (replay episodes have won and doubled the replayed list) ...
    assert cfg["replay"]["best_list_len"] > 96 and cfg["last_batch"]["generators_per_replay_episode"] > 100
    r = line["roofline"]
    assert r["bound"] and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["achieved"] > 0
    assert "traffic" in r and r["avg_kernel_ms"] > 0 and 0.0 < r["frac_requested"] < r["frac"]
    # counters come from committed profiles and are only quoted for the very library being timed
    from eirgrid_amd import _native as N
    assert r["profiles"]["library"] == N.lib().eg_build_hash().decode()
    if r["traffic"] is None or r["valu_busy"] is None:
        assert r["counters_not_quoted"] and all("was taken on build" in x or "no committed counter profile" in x for x in r["counters_not_quoted"])
    # ... the seeded policy (config 1's episode, replayed: SURVEY §8(d) config 3 read literally) is reported beside it
    g = line["config2_seeded"]
    assert g["value"] > 0 and g["episodes_failed"] == 0 and g["replay"]["best_list_len"] == 28 and g["last_batch"]["generators_per_replay_episode"] == 35.0
    assert line["config1"]["value"] > 0 and line["config1"]["episodes_failed"] == 0
    # ... the same batches with the replay hoist on (the replay episodes computed once): same episodes, served by the hoist
    h = line["config2_replay_hoisted"]
    assert h["replay_hoist"] and h["hoist_served_last_batch"] and h["value"] > 0 and h["episodes_failed"] == 0
    # (the sampled episodes of the last batch are others, and a window of 2 048 indices holds 204 or 205 multiples of ten)
    same = ("ok", "overflow", "other_failures", "generators_per_replay_episode")
    assert all(h["last_batch"][k] == cfg["last_batch"][k] for k in same) and h["replay"] == cfg["replay"] and h["speedup_vs_value"] > 0
    # ... one rank's batch in the state of the 8-GPU loop, and the loop from a fresh policy: per-episode replays and hoisted
    for key in ("config3_one_rank_state", "fresh_policy_steady_state"):
        a, b = line[key]["per_episode_replays"], line[key]["replay_hoisted"]
        assert a["value"] > 0 and b["value"] > 0 and a["episodes_failed"] == 0 and b["episodes_failed"] == 0
        assert all(a["last_batch"][k] == b["last_batch"][k] for k in same)
        assert not a["replay_hoist"] and b["replay_hoist"]
    # the primary ceiling is first in the roofline object, the byte accounting labelled as such
    assert list(r)[:4] == ["bound", "valu_busy", "valu_busy_grids", "issue_slot_frac"] and "accounting" in r and "north_star_40pct_of_hbm" in r
    c = line["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and c["unit"] == "episodes/s" and c["sample"]


def test_grids_of_a_batch_overlap_with_a_process_group_up():
    """The grids of a batch must run side by side also after torch.distributed has brought RCCL (and its streams) up before the
    engine exists.  They once did not: the library's side stream shared a hardware queue with the null stream and a batch took
    1.69 instead of 1.48 ms.  Shown from the library's own events (eg_timing_read_grids), not from wall-clock differences between
    runs: with the collectives forced on one rank, a batch's span (first grid's start to last grid's end) must stay well below
    the grids' own durations added up — serialised grids give span == sum (EIRGRID_SIDE_STREAM=plain reproduces that)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    got = {}
    for extra in ((), ("--force-collectives",)):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29561")
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "5", "--warmup", "2", "--min-seconds", "0.1",
                              "--no-cpu-baseline", "--no-config1", *extra], capture_output=True, text=True, timeout=900, cwd=root, env=env)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
        line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
        assert line["config"]["episodes_failed"] == 0
        got[bool(extra)] = (line["ms_per_batch"], line["roofline"]["grids"])
    print("ms per batch / grids:", got)
    for forced, (ms, grids) in got.items():
        assert grids["span_ms"] < 0.88 * grids["sum_of_grids_ms"], (forced, grids)


def test_train_step_equals_the_stepwise_path(engine):
    """eg_train_step (one library call per step) == upload + launch_update + packet copy + apply, bit for bit."""
    from eirgrid_amd.engine import apply_packet
    from eirgrid_amd.parallel import exchange_packet_raw
    a, b = ActionWeights(), ActionWeights()
    packet = torch.zeros(N.PACKET_BYTES, dtype=torch.uint8, device="cuda")
    n = 192
    for step in range(5):
        mask = ((np.arange(n) % 5) == 1).astype(np.uint8) if step > 0 else None
        fa = engine.train_step(a, 99, step * n, n, mask, noise_seed=1000 + step)
        engine.upload_snapshot(b)
        engine.launch_update(99, step * n, n, packet.data_ptr(), mask)
        stats, cands = exchange_packet_raw(packet)
        fb = apply_packet(b, stats, cands, noise_seed=1000 + step)
        assert fa == fb
        for x, y in zip(a.tables(), b.tables()):
            assert x.tobytes() == y.tobytes()
        assert a.lists(0) == b.lists(0) and a.lists(1) == b.lists(1)
        for name in ("iterations_without_improvement", "iteration_count", "has_best", "best_cost", "best_net_emissions"):
            assert a.get(name) == b.get(name), name


def test_device_resident_training_equals_host_updates(engine, world):
    """eg_policy_push + eg_device_step (rollout, statistics, best pick, k_apply_update, stalled tables: all on the device,
    no host copy) must leave the policy exactly where eg_train_step (host update from the same packets) leaves it —
    weights, best strategy, counters — at every step, through the first improvement, the stalled sampler (> 500) and
    the stagnation noise (> 1200)."""
    from eirgrid_amd.engine import Engine
    dev = Engine(world, device=0)
    try:
        a, b = ActionWeights(), ActionWeights()
        n, period = 160, 4
        dev.push(b)
        improved_steps = 0; max_stall = 0
        steps = 40
        for step in range(steps):
            first = step * n
            mask = ((np.arange(first, first + n) % period) == 0).astype(np.uint8) if a.get("has_best_actions") == 1 else None
            improved_steps += int(engine.train_step(a, 4711, first, n, mask, noise_seed=900 + step))
            dev.device_step(4711, first, n, period, 900 + step)
            dev.pull(b)
            max_stall = max(max_stall, a.get("iterations_without_improvement"))
            for x, y, name in zip(a.tables(), b.tables(), ("weights", "deficit weights", "count weights")):
                assert x.tobytes() == y.tobytes(), f"step {step}: {name} differ"
            assert a.lists(0) == b.lists(0) and a.lists(1) == b.lists(1), f"step {step}: best lists differ"
            for name in ("iterations_without_improvement", "iteration_count", "has_best", "best_cost", "best_net_emissions",
                         "best_opinion", "best_reliability", "has_best_actions", "has_best_deficit_actions"):
                assert a.get(name) == b.get(name), f"step {step}: {name}"
        assert improved_steps >= 1 and max_stall > 1200
        # the same without any pull in between: the device loop is self-contained
        c = ActionWeights(); dev.push(c)
        for step in range(steps):
            dev.debug_fill_lds((0x7FF80000, 1, 0xFFFFFFFF, 17)[step % 4])      # nothing may depend on what LDS held before
            dev.device_step(4711, step * n, n, period, 900 + step)
        dev.pull(c)
        for x, y in zip(a.tables(), c.tables()):
            assert x.tobytes() == y.tobytes()
        assert a.lists(0) == c.lists(0) and a.get("iteration_count") == c.get("iteration_count")
    finally:
        dev.close()


def test_hold_and_rewind_put_the_device_policy_back(world):
    """eg_policy_hold / eg_policy_rewind (what bench.py does before every batch): after a rewind the device-resident policy is
    the held one — tables, best strategy, counters — so the same step gives the same bytes again; only the count of failed
    episodes goes on counting."""
    from eirgrid_amd.engine import Engine
    dev = Engine(world, device=0)
    names = ("iterations_without_improvement", "iteration_count", "has_best", "best_cost", "best_net_emissions", "best_opinion",
             "best_reliability", "has_best_actions", "has_best_deficit_actions", "learning_rate", "exploration_rate")

    def state(p):
        return ([t.tobytes() for t in p.tables()], p.lists(0), p.lists(1), [p.get(k) for k in names])
    try:
        pol = ActionWeights()
        first = dev.run_iteration(0, pol, False, 12345)
        pol.apply_episode(first.metrics[0], first.n_run[0], first.run_log[0, :first.n_run[0].sum()], first.n_def[0],
                          first.def_log[0, :first.n_def[0].sum()])
        dev.push(pol)
        for k in range(3):      # not the pushed state itself: one that on-device updates have produced
            dev.device_step(77, k * 512, 512, 4, 500 + k)
        dev.hold()
        dev.pull(pol); held = state(pol)
        dev.device_step(77, 4096, 512, 4, 600)
        a = dev.fetch(512)
        dev.pull(pol); moved = state(pol)
        assert moved != held
        for k in range(4):
            dev.device_step(77, 8192 + k * 512, 512, 4, 700 + k)
        dev.rewind()
        dev.pull(pol)
        assert state(pol) == held
        dev.device_step(77, 4096, 512, 4, 600)
        b = dev.fetch(512)
        dev.pull(pol)
        assert state(pol) == moved
        for name in ("status", "metrics", "yearly", "n_run", "n_def", "n_gens"):
            assert getattr(a, name).tobytes() == getattr(b, name).tobytes(), name
        for name, count in (("run_log", a.n_run.sum(axis=1)), ("def_log", a.n_def.sum(axis=1)), ("gen_cell", a.n_gens)):      # (the buffers are not cleared between batches)
            live = np.arange(getattr(a, name).shape[1])[None, :] < count[:, None]
            assert (getattr(a, name)[live] == getattr(b, name)[live]).all(), name
    finally:
        dev.close()


def test_rewind_behind_unpulled_updates_still_runs_the_long_replays(world):
    """eg_policy_hold behind on-device updates that nobody has pulled, then eg_policy_rewind and a step — the host
    does not know the held list, so BOTH replay variants must be launched (the long one used to be skipped: the replay episodes'
    records kept the previous batch's bytes and still fed the update).  The device-resident loop grows a best list beyond 96
    actions; after hold / rewind / step WITHOUT a pull every replay episode of the batch must be the tabled oracle's."""
    from eirgrid_amd.engine import Engine, HostTables
    from oracle import api as O
    from tests.helpers import assert_episode_equal, oracle_weights_like
    dev = Engine(world, device=0)
    try:
        pol = ActionWeights()
        first = dev.run_iteration(0, pol, False, 12345)
        pol.apply_episode(first.metrics[0], first.n_run[0], first.run_log[0, :first.n_run[0].sum()], first.n_def[0],
                          first.def_log[0, :first.n_def[0].sum()])
        dev.push(pol)
        n, period = 4096, 10
        for k in range(12):      # replay episodes win and double the replayed list
            dev.device_step(12345, k * n, n, period, 900 + k)
        dev.hold()               # ... and nobody has pulled: the host's idea of the list is the pushed one (28 actions)
        dev.device_step(12345, 50 * n, n, period, 1000)      # something else in the records
        dev.rewind()
        dev.device_step(12345, 60 * n, n, period, 1001)
        res = dev.fetch(n)
        probe = ActionWeights()
        dev.rewind(); dev.pull(probe)      # the held policy, for the oracle (after the batch under test)
        assert sum(len(l) for l in probe.lists(0)) > 96, "the loop was meant to grow a long best list"
        tb = O.OracleTables(HostTables(world), len(world.existing_x))
        idx = np.arange(60 * n, 61 * n)
        reps = np.flatnonzero(idx % period == 0)
        assert (res.status == 0).all()
        for e in list(reps[:3]) + [int(reps[-1]), 1]:
            st, ref = O.run_episode_tabled(tb, oracle_weights_like(probe), 12345 + 60 * n + int(e), replay=bool(idx[e] % period == 0))
            assert_episode_equal(res, int(e), ref, "after hold / rewind without a pull")
    finally:
        dev.close()


def test_rccl_path_of_the_trainer_on_one_rank():
    """The N > 1 device-resident step (eg_device_rollout -> all_reduce -> all_gather_into_tensor -> eg_device_apply, all
    on the stream) with the collectives really issued through RCCL (world size 1, forced) == the single-GPU step."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import os, sys, hashlib
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from eirgrid_amd import synthetic_world
from eirgrid_amd.engine import ActionWeights, Engine
from eirgrid_amd.parallel import BatchTrainer
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29617")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
eng = Engine(synthetic_world(), device=0)
out = []
for forced in (False, True):
    pol = ActionWeights()
    tr = BatchTrainer(eng, pol, 96, 2024, 0, 1, dist, replay_fraction=0.25, force_collectives=forced)
    for _ in range(8): tr.step()
    tr.sync()
    w, dw, _ = pol.tables()
    out.append(hashlib.sha256(w.tobytes() + dw.tobytes() + bytes(sum(pol.lists(0), []))).hexdigest() + ":%%d:%%d" %% (pol.get("iteration_count"), tr.improvements))
print("RESULT", out[0], out[1])
dist.destroy_process_group()
""" % root
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")][0].split()
    assert line[1] == line[2] and line[1].endswith(":768:" + line[1].rsplit(":", 1)[1])


def test_device_update_from_two_rank_packets(engine, world):
    """k_apply_update with two update packets (what an all-gather delivers on two ranks: each rank's shard statistics and
    candidate) == the host update from the summed statistics and both candidate records."""
    from eirgrid_amd.engine import Engine, apply_packet
    dev = Engine(world, device=0)
    try:
        a, b = ActionWeights(), ActionWeights()
        n = 96                                    # episodes per "rank"
        PB, nstat = N.PACKET_BYTES, 8 * N.STATS_LEN
        packets = torch.zeros(2 * PB, dtype=torch.uint8, device="cuda")
        hostp = torch.zeros(PB, dtype=torch.uint8, device="cuda")
        dev.push(b)
        for step in range(10):
            first = step * 2 * n
            # host side: the two shards one after the other against the same snapshot, statistics added, both candidates kept
            engine.upload_snapshot(a)
            stats = np.zeros(N.STATS_LEN, np.int64); cands = []
            for r in range(2):
                engine.launch_update(31, first + r * n, n, hostp.data_ptr(), None)
                h = hostp.cpu().numpy()
                stats += h[:nstat].view(np.int64); cands.append(h[nstat:].copy())
            apply_packet(a, stats, np.stack(cands), noise_seed=70 + step)
            # device side: "rank" r writes packet r, then one update from both
            for r in range(2):
                dev.device_rollout(31, first + r * n, n, 0, packets.data_ptr() + r * PB)
            dev.device_apply(packets.data_ptr(), 2, packets.data_ptr(), 70 + step)
            packets[PB:PB + nstat] = 0            # the second rank's own statistics (its own k_apply_update zeroes them)
            dev.pull(b)
            for x, y in zip(a.tables(), b.tables()):
                assert x.tobytes() == y.tobytes(), f"step {step}"
            assert a.lists(0) == b.lists(0) and a.lists(1) == b.lists(1)
            assert a.get("iterations_without_improvement") == b.get("iterations_without_improvement")
            assert a.get("iteration_count") == b.get("iteration_count") == (step + 1) * 2 * n
    finally:
        dev.close()


def test_device_resident_training_survives_pull_and_push(engine, world, tmp_path):
    """Checkpoint / resume of the device-resident loop: 12 steps in one go == 6 steps, eg_policy_pull, a JSON round trip
    is NOT taken (count weights would be dropped, as in the reference) but a fresh engine and eg_policy_push, 6 more."""
    from eirgrid_amd.engine import Engine
    n, period = 128, 5
    a = ActionWeights(); engine.push(a)
    for step in range(12):
        engine.device_step(97, step * n, n, period, 300 + step)
    engine.pull(a)
    b = ActionWeights(); engine.push(b)
    for step in range(6):
        engine.device_step(97, step * n, n, period, 300 + step)
    engine.pull(b)
    other = Engine(world, device=0)
    try:
        other.push(b)
        for step in range(6, 12):
            other.device_step(97, step * n, n, period, 300 + step)
        other.pull(b)
    finally:
        other.close()
    for x, y in zip(a.tables(), b.tables()):
        assert x.tobytes() == y.tobytes()
    assert a.lists(0) == b.lists(0) and a.lists(1) == b.lists(1)
    for name in ("iterations_without_improvement", "iteration_count", "has_best", "best_cost", "improvement_history_len"):
        assert a.get(name) == b.get(name), name
    # and the checkpoint written from the pulled policy is the one the host-driven loop writes
    pa, pb = tmp_path / "a.json", tmp_path / "b.json"
    a.save_to_file(str(pa)); b.save_to_file(str(pb))
    import json
    ja, jb = json.load(open(pa)), json.load(open(pb))
    for j in (ja, jb):      # timestamps differ
        for rec in j.get("improvement_history") or []:
            rec.pop("timestamp", None)
    assert ja == jb


def test_best_run_record_follows_the_best_episode(engine, world, tmp_path):
    """The on-device update keeps the whole record of the episode that became the best strategy (the reference keeps its
    SimulationResult for the export, multi_simulation.rs:494-508).  Shadow every step on a second engine: same policy,
    same batch, pick the best score on the host; after an improving step the kept record must be that episode's, byte
    for byte, and the summary CSV written from it equals the restatement's."""
    from eirgrid_amd.engine import Engine, score_metrics
    from oracle import csv_export as OC
    dev = Engine(world, device=0)
    try:
        pol = ActionWeights(); dev.push(pol, write_yearly=True)
        assert dev.fetch_best_run() == (0, None)
        n, period = 96, 4
        before = ActionWeights(); improvements = 0; expect = None
        for step in range(16):
            first = step * n
            dev.pull(before)
            mask = ((np.arange(first, first + n) % period) == 0).astype(np.uint8) if before.get("has_best_actions") == 1 else None
            dev.device_step(515, first, n, period, 40 + step)
            dev.pull(pol)
            best_of = lambda p: tuple(p.get(k) for k in ("has_best", "best_net_emissions", "best_opinion", "best_cost"))
            if best_of(pol) != best_of(before):
                improvements += 1
                res = engine.rollout_batch(before, 515, n, first_episode_index=first, replay_mask=mask)
                scores = np.array([score_metrics(res.metrics[e]) if res.status[e] == 0 else -1.0 for e in range(n)])
                expect = (res, int(np.argmax(scores)))                 # argmax returns the lowest index among ties
            state, rec = dev.fetch_best_run()
            if expect is None:
                assert state == 0
                continue
            assert state == 1
            res, e = expect
            for name in ("metrics", "yearly", "status", "n_run", "n_def", "n_act", "n_gens", "n_offsets", "n_draws", "bytes_moved"):
                assert getattr(rec, name)[0].tobytes() == getattr(res, name)[e].tobytes(), (step, name)
            for which in ("run", "def", "act"):
                assert rec.lists(0, which) == res.lists(e, which)
            g = int(res.n_gens[e])
            assert rec.gen_cell[0, :g].tobytes() == res.gen_cell[e, :g].tobytes() and rec.gen_pack[0, :g].tobytes() == res.gen_pack[e, :g].tobytes()
            assert rec.metrics[0].tolist() == [pol.get("best_net_emissions"), pol.get("best_opinion"), pol.get("best_cost"), pol.get("best_reliability")]
        assert improvements >= 2
        # a push on the same context (checkpoint / resume) keeps the record; the export reads it
        dev.push(pol, write_yearly=True)
        state, rec = dev.fetch_best_run()
        assert state == 1
        path = tmp_path / "simulation_summary.csv"
        rec.export_summary_csv(str(path), "stamp")
        assert path.read_bytes().decode() == OC.summary_csv_text(rec.metrics[0], rec.yearly[0], rec.n_act[0], rec.act_log[0], "stamp")
        assert rec.yearly[0, 25, 0] == 2050.0 and rec.yearly[0, 0, 1] > 5e6
        # a winner that ran in another rank's shard: this context only knows about its own last batch
        PB = N.PACKET_BYTES
        packets = torch.zeros(2 * PB, dtype=torch.uint8, device="cuda")
        fresh = Engine(world, device=0)
        try:
            fresh.push(ActionWeights())
            for r in range(2):
                fresh.device_rollout(31, r * n, n, 0, packets.data_ptr() + r * PB)
            cand = [packets[r * PB + 8 * N.STATS_LEN:r * PB + 8 * N.STATS_LEN + 16].cpu().numpy().view(np.float64)[0] for r in range(2)]
            fresh.device_apply(packets.data_ptr(), 2, packets.data_ptr(), 1)
            state, rec = fresh.fetch_best_run()
            assert state == (2 if cand[0] >= cand[1] else 1)
        finally:
            fresh.close()
    finally:
        dev.close()


def test_device_resident_replicas_of_two_processes_stay_identical(tmp_path):
    """Two processes (two ranks sharing cuda:0, gloo standing in for RCCL) run the device-resident multi-rank step —
    eg_device_rollout of the own shard, exchange of the 37 008-byte packets, eg_device_apply on both packets — and the
    host-driven step on the same shards: after 8 steps all four policies must be the same, bit for bit."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import os, sys, hashlib
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from eirgrid_amd import synthetic_world
from eirgrid_amd.engine import ActionWeights, Engine
from eirgrid_amd.parallel import BatchTrainer
torch.cuda.set_device(0)
dist.init_process_group("gloo")
rank = dist.get_rank()
eng = Engine(synthetic_world(), device=0)
out = []
for resident in (True, False):
    pol = ActionWeights()
    tr = BatchTrainer(eng, pol, 96, 2024, rank, 2, dist, replay_fraction=0.25, device_resident=resident)
    for _ in range(8): tr.step()
    tr.sync()
    w, dw, _ = pol.tables()
    out.append(hashlib.sha256(w.tobytes() + dw.tobytes() + bytes(sum(pol.lists(0), []))).hexdigest() + ":%%d:%%d" %% (pol.get("iteration_count"), pol.get("iterations_without_improvement")))
open(os.path.join(%r, "_two_rank_resident_%%d.txt" %% rank), "w").write(out[0] + " " + out[1])
dist.barrier()
dist.destroy_process_group()
""" % (root, str(tmp_path))
    script = os.path.join(str(tmp_path), "_two_rank_resident.py")
    open(script, "w").write(code)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29544", script]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-2500:]
    res = [open(os.path.join(str(tmp_path), "_two_rank_resident_%d.txt" % k)).read().split() for k in (0, 1)]
    assert res[0][0] == res[0][1] == res[1][0] == res[1][1], res
    assert res[0][0].split(":")[1] == str(8 * 2 * 96)
