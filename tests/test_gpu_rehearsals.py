"""GPU: full-size rehearsals of BASELINE configs[2]-[4] on one MI355X.

* configs[4] ("full 100k-iteration training run, checkpoint-compatible with the reference"): eirgrid-hip -n 100000 with a
  checkpoint after every batch, interrupted and resumed, through the replay phase of the last 10 % of the iterations
  (multi_simulation.rs:38-39, :437-465) where every episode replays the best strategy (and grows it).
* configs[3] ("131 072 episodes sharded across 8 GPUs, one exchange per update"): the 8 shards by global episode index on
  one GPU, eg_device_rollout x 8 -> eg_device_apply(n_packets = 8), against the host update from the same packets and
  the tabled oracle on sampled episodes of every shard.
* configs[2] (16 384 episodes, 10 % replays) as a training loop for 50 updates inside a stated time budget, with the failed
  episodes reported rather than dropped."""
import json
import os
import re
import subprocess
import time

import numpy as np
import pytest
import torch

from eirgrid_amd import _native as N
from eirgrid_amd.engine import ActionWeights, Engine, HostTables, apply_packet
from eirgrid_amd.parallel import BatchTrainer
from oracle import api as O
from tests.helpers import assert_episode_equal, oracle_weights_like

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "eirgrid_amd", "eirgrid-hip")
WORLD = os.path.join(ROOT, "tests", "golden", "world_v1.json")


def _cli(ck, *extra):
    t0 = time.time()
    out = subprocess.run([CLI, "--world", WORLD, "-n", "100000", "--batch", "1024", "-i", "1", "--seed", "7", "-c", ck, "-r", "1000", *extra],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    return out.stdout, time.time() - t0


def _state(ck):
    rd = os.path.join(ck, sorted(os.listdir(ck))[0])
    d = json.load(open(os.path.join(rd, "latest_weights.json")))
    for rec in d.get("improvement_history") or []:
        rec.pop("timestamp", None)
    return rd, d


def test_config5_100k_iterations_checkpoint_and_resume(built, tmp_path):
    """100 000 iterations in batches of 1 024, a checkpoint after every batch.  Run A is interrupted once (after 51 200
    iterations) and resumed; run B is interrupted twice (51 200, 80 896).  A load drops the action-count table exactly as the
    reference's loader does (serialization.rs:474: a resumed run samples the action count with the heuristic branch), so
    an uninterrupted run is NOT the same trajectory — in the reference either; what must hold, and is checked: everything
    else a run needs survives the file (weights, best strategy, counters, improvement history), i.e. the second
    interruption of B changes nothing: A == B, byte for byte apart from timestamps, through the replay phase."""
    a, b = str(tmp_path / "a"), str(tmp_path / "b")
    out_a1, t_a1 = _cli(a, "--stop-after", "51200")
    assert "Stopped after 51200 iterations" in out_a1
    rd_a, mid_a = _state(a)
    assert open(os.path.join(rd_a, "checkpoint_iteration.txt")).read() == "51200"
    out_a2, t_a2 = _cli(a)
    assert "(51200 completed, 48800 remaining)" in out_a2
    out_b1, _ = _cli(b, "--stop-after", "51200")
    _, mid_b = _state(b)
    assert mid_a == mid_b, "two runs of the same command line differ after 51 200 iterations"
    out_b2, _ = _cli(b, "--stop-after", "80896")
    assert "Stopped after 80896 iterations" in out_b2
    out_b3, _ = _cli(b)
    assert "(80896 completed, 19104 remaining)" in out_b3
    rd_a, fin_a = _state(a); rd_b, fin_b = _state(b)
    assert fin_a == fin_b, "an extra interruption changed the run"
    for rd in (rd_a, rd_b):
        assert open(os.path.join(rd, "checkpoint_iteration.txt")).read() == "100000"
        assert {"latest_weights.json", "thread_0_weights.json", "best_weights.json"} <= set(os.listdir(rd))
    # the file is the reference's schema and loads through the library's loader
    pol = ActionWeights.load_from_file(os.path.join(rd_a, "latest_weights.json"))
    assert pol.get("has_best") == 1 and pol.get("has_count_weights") == 0
    failed = int(re.search(r"\((\d+) episodes failed\)", out_a2).group(1))
    # iteration_count counts the episodes that finished; the rest are reported, not dropped (failed since the resume)
    assert fin_a["iteration_count"] + failed <= 100000 and fin_a["iteration_count"] >= 100000 - 48800
    assert set(fin_a) >= {"weights", "learning_rate", "best_metrics", "best_weights", "best_actions", "iteration_count",
                          "iterations_without_improvement", "exploration_rate", "deficit_weights", "best_deficit_actions",
                          "optimization_mode", "improvement_history"}                  # learning/serialization.rs:38-51
    best_len = sum(len(v) for v in fin_a["best_actions"].values())
    print(f"config 5: legs {t_a1:.1f} s + {t_a2:.1f} s wall; iteration_count {fin_a['iteration_count']}, failed since resume {failed}, "
          f"best list {best_len} actions, improvements {len(fin_a['improvement_history'] or [])}")
    assert t_a1 + t_a2 < 120.0, "time budget of the 100k-iteration run (two legs) on one MI355X"


def test_cli_run_is_the_same_with_and_without_the_replay_hoist(built, tmp_path):
    """The driver computes the replay iterations of a batch once (the reference's replay phases run the same replay in every
    iteration: core/multi_simulation.rs:38-39, :437-465); `--no-replay-hoist` runs every iteration on its own.  20 480 iterations
    with a forced all-replay run (--force-full-simulation) and the ordinary schedule (replays in the last 10 %): the checkpoint, the
    exported summary and the printed best result are the same bytes either way — and the hoisted all-replay run is the faster one."""
    import hashlib
    for extra in (("--force-full-simulation",), ()):
        got = {}
        for hoist in (True, False):
            ck = str(tmp_path / f"ck_{len(extra)}_{int(hoist)}")
            t0 = time.time()
            out = subprocess.run([CLI, "--world", WORLD, "-n", "20480", "--batch", "1024", "-i", "4", "--seed", "11", "-c", ck, "-r", "1000", "--no-continue",
                                  *extra, *(() if hoist else ("--no-replay-hoist",))], capture_output=True, text=True, timeout=900)
            wall = time.time() - t0
            assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
            rd, state = _state(ck)
            files = {}      # every exported CSV, by its path below the export's time-stamped directory, without the lines that carry that stamp
            stamp = os.listdir(os.path.join(rd, "enhanced_csv"))[0]
            for base, _, names in os.walk(os.path.join(rd, "enhanced_csv", stamp)):
                for nm in names:
                    if nm.endswith(".csv") and nm != "improvement_history.csv":      # (its rows carry wall-clock stamps; the history itself is in `state`)
                        text = [l for l in open(os.path.join(base, nm), encoding="utf-8").read().split("\n") if stamp not in l and "imestamp" not in l]
                        files[os.path.relpath(os.path.join(base, nm), os.path.join(rd, "enhanced_csv", stamp))] = hashlib.sha256("\n".join(text).encode()).hexdigest()
            # the printed best result, without the run's pace and directory
            best = [re.sub(r"\d+ iterations/s", "", l) for l in out.stdout.splitlines() if ("Best" in l or "best" in l) and not l.startswith("Done:")]
            got[hoist] = (state, files, best, wall)
        assert got[True][0] == got[False][0], extra
        assert got[True][1] == got[False][1] and got[True][1], extra
        assert got[True][2] == got[False][2], extra
        print(f"CLI {' '.join(extra) or '(default schedule)'}: {got[True][3]:.2f} s hoisted, {got[False][3]:.2f} s per iteration")


def test_config4_131072_episodes_as_eight_shards(world):
    """One update from 131 072 episodes: 8 shards of 16 384 by global episode index (every 10th index replays the best
    strategy), each rolled out into its own 37 008-byte packet, ONE k_apply_update over the 8 packets — against the host
    update from the same packets (statistics summed as the all-reduce would) and, for sampled episodes of every shard,
    the tabled oracle run from the global index alone (shard invariance)."""
    tb = O.OracleTables(HostTables(world), len(world.existing_x))
    dev = Engine(world, device=0)
    try:
        a, b = ActionWeights(), ActionWeights()
        n, shards, period = 16384, 8, 10
        PB, nstat = N.PACKET_BYTES, 8 * N.STATS_LEN
        packets = torch.zeros(shards * PB, dtype=torch.uint8, device="cuda")
        dev.push(b)
        t0 = time.time()
        for step in range(3):
            first = step * shards * n
            before = oracle_weights_like(a)
            has_best = a.get("has_best_actions") == 1
            for r in range(shards):
                dev.device_rollout(2718, first + r * n, n, period, packets.data_ptr() + r * PB)
                res = dev.fetch(n)
                assert (res.status == 0).all()
                for e in (0, 1, 7, 5003, 16383) + tuple(e for e in range(n) if (first + r * n + e) % period == 0)[:2]:
                    g = first + r * n + e
                    st, ref = O.run_episode_tabled(tb, oracle_weights_like(a) if e == 0 else before.clone(), 2718 + g, replay=bool(has_best and g % period == 0))
                    assert_episode_equal(res, e, ref, f"step {step} shard {r}")
            host = packets.cpu().numpy().reshape(shards, PB)
            stats = host[:, :nstat].copy().view(np.int64).reshape(shards, N.STATS_LEN)
            total = stats.sum(axis=0); total[3] = 0
            assert total[0] == shards * n
            apply_packet(a, total, np.stack([host[r, nstat:] for r in range(shards)]), noise_seed=31 + step)
            dev.device_apply(packets.data_ptr(), shards, packets.data_ptr(), 31 + step)
            for r in range(1, shards):
                packets[r * PB:r * PB + nstat] = 0        # (every rank's own k_apply_update zeroes its own statistics)
            dev.pull(b)
            for x, y in zip(a.tables()[:2], b.tables()[:2]):
                assert x.tobytes() == y.tobytes(), f"step {step}"
            assert a.lists(0) == b.lists(0) and a.lists(1) == b.lists(1)
            for name in ("iterations_without_improvement", "iteration_count", "has_best", "best_cost", "best_net_emissions", "failed_episodes"):
                assert a.get(name) == b.get(name), (step, name)
            assert a.get("iteration_count") == (step + 1) * shards * n
        print(f"config 4: 3 updates of 131 072 episodes (8 shards on one GPU, oracle checks included) in {time.time() - t0:.1f} s")
    finally:
        dev.close()


def test_config3_replay_loop_time_budget_and_failure_accounting(world):
    """configs[2] as a training loop: 16 384 episodes per update, every 10th replaying the best strategy, 50 updates, from a
    fresh policy (the path on which the replayed lists double up to 468 generators per replay episode).
    No episode fails (the lists stay far below the records' 4096 entries).  The loop's pace is printed, and bounded only against
    the pathology it once had: 4.8 ms per update measured once the lists have stopped growing (r02h), 65 ms before the
    heavy-episode path — the bound is 25 ms per update of k_rollout time from the library's own events, not wall clock on a
    shared box."""
    eng = Engine(world, device=0)
    try:
        pol = ActionWeights()
        tr = BatchTrainer(eng, pol, 16384, 12345, replay_fraction=0.1)
        tr.step(); tr.sync()
        eng.timing_reset()
        t0 = time.time()
        for _ in range(50):
            tr.step()
        failed = tr.failed_episodes()
        wall = time.time() - t0
        kernel_ms, launches = eng.timing_read()
        res = eng.fetch(16384)
        rep = (np.arange(50 * 16384, 51 * 16384) % 10) == 0
        ok = res.status == 0
        assert pol.get("iteration_count") + failed == 51 * 16384
        assert int((~ok).sum()) <= failed
        print(f"config 3 loop: 50 updates in {wall:.2f} s ({wall / 50 * 1e3:.1f} ms each), failed episodes {failed}, "
              f"generators per replay episode {res.n_gens[rep & ok].mean():.0f}, per sampled episode {res.n_gens[~rep & ok].mean():.1f}, "
              f"best list {sum(len(l) for l in pol.lists(0))} actions, improvements {tr.improvements}")
        assert failed == 0 and launches == 50
        print(f"k_rollout: {kernel_ms / launches:.2f} ms per update")
        assert kernel_ms / launches < 25.0
    finally:
        eng.close()
