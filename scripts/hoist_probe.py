"""What the replay hoist does to a batch: the sustained state of configs[2] (bench.py's pinned policy) and, optionally, the state of
the 8-GPU loop, timed with the hoist off and on, interleaved, from the library's own events and the wall clock.
  python scripts/hoist_probe.py [episodes=16384] [grow=48] [batches=200] [global_shards=1]
global_shards > 1: the policy is grown at the GLOBAL batch of that many shards (every shard rolled out on this GPU, one update from
all their packets — the 8-shard loop of tests/test_gpu_rehearsals.py), then ONE rank's batch is timed in that state."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from eirgrid_amd import synthetic_world
from eirgrid_amd import _native as N
from eirgrid_amd.engine import ActionWeights, Engine
from eirgrid_amd.parallel import BatchTrainer

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
GROW = int(sys.argv[2]) if len(sys.argv) > 2 else 48
BATCHES = int(sys.argv[3]) if len(sys.argv) > 3 else 200
SHARDS = int(sys.argv[4]) if len(sys.argv) > 4 else 1
FRESH = len(sys.argv) > 5 and sys.argv[5] == "fresh"      # no seeded best strategy: the loop from ActionWeights::new
torch.cuda.set_device(0)
eng = Engine(synthetic_world())
w = ActionWeights()
if not FRESH:
    first = eng.run_iteration(0, w, False, 12345)
    w.apply_episode(first.metrics[0], first.n_run[0], first.run_log[0, :first.n_run[0].sum()], first.n_def[0], first.def_log[0, :first.n_def[0].sum()])


def grow_global(w, shards, steps):
    """`steps` updates of the loop at `shards` x B episodes per update, every shard on this GPU (no RCCL): eg_device_rollout per shard
    into its own packet, ONE eg_device_apply over all of them."""
    eng.push(w)
    packets = torch.zeros(shards * N.PACKET_BYTES, dtype=torch.uint8, device="cuda")
    for s in range(steps):
        for r in range(shards):
            eng.device_rollout(12345, (s * shards + r) * B, B, 10, packets.data_ptr() + r * N.PACKET_BYTES)
        eng.device_apply(packets.data_ptr(), shards, packets.data_ptr(), 12345 + s)
        for r in range(1, shards):      # (every rank's own k_apply_update zeroes its own statistics)
            packets[r * N.PACKET_BYTES:r * N.PACKET_BYTES + 8 * N.STATS_LEN] = 0
    eng.pull(w)


if SHARDS > 1:
    grow_global(w, SHARDS, GROW)
tr = BatchTrainer(eng, w, B, 12345, replay_fraction=0.1)
if SHARDS == 1:
    for _ in range(GROW):
        tr.step()
    tr.sync()
    tr = BatchTrainer(eng, w, B, 12345, replay_fraction=0.1)
tr.pin_policy()
out = {"episodes": B, "grow_batches": GROW, "global_shards": SHARDS, "best_list_len": int(sum(len(l) for l in w.lists(0))), "build_hash": N.lib().eg_build_hash().decode(), "runs": []}


def timed(hoist):
    eng.replay_hoist(hoist)
    for _ in range(10):
        tr.step()
    eng.sync(); eng.timing_reset(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(BATCHES):
        tr.step()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    span, grids, n = eng.timing_read_grids()
    res = eng.fetch(B)
    idx = np.arange((tr.step_index - 1) * B, tr.step_index * B)
    rep = idx % 10 == 0
    return {"hoist": hoist, "ms_per_batch": el / BATCHES * 1e3, "rollout_span_ms": span / n, "sum_of_grids_ms": grids / n,
            "episodes_per_s": B * BATCHES / el, "generators_per_replay_episode": float(res.n_gens[rep].mean()),
            "generators_per_sampled_episode": float(res.n_gens[~rep].mean()), "ok": int((res.status == 0).sum()),
            "served": eng.replay_hoist_stats()[1] if hoist else None}


def timed_no_update(hoist):
    """the same batches without the statistics epilogue and the update (eg_rollout_launch): rollout grids only"""
    eng.replay_hoist(hoist)
    eng.upload_snapshot(w)
    mask = (np.arange(B) % 10 == 0).astype(np.uint8)
    for _ in range(5):
        eng.launch(12345, 0, B, mask)
    eng.sync(); eng.timing_reset()
    for _ in range(50):
        eng.launch(12345, 0, B, mask)
    eng.sync()
    span, grids, n = eng.timing_read_grids()
    return {"hoist": hoist, "no_update": True, "rollout_span_ms": span / n, "sum_of_grids_ms": grids / n}


tr.sync()
for hoist in (False, True):
    print(json.dumps(timed_no_update(hoist)), flush=True)
tr = BatchTrainer(eng, w, B, 12345, replay_fraction=0.1)
tr.pin_policy()
for rep in range(3):
    for hoist in (False, True):
        r = timed(hoist)
        out["runs"].append(r)
        print(json.dumps(r), flush=True)
print(json.dumps(out))
