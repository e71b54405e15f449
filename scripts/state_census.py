"""What a SAMPLED episode costs in the states bench.py pins (the lean grid is the batch's span once the replay episodes are hoisted):
per state the lean grid's duration alone and what its episodes did.  python scripts/state_census.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from eirgrid_amd import synthetic_world
from eirgrid_amd import _native as N
from eirgrid_amd.engine import ActionWeights, Engine
from eirgrid_amd.parallel import BatchTrainer

B = 16384
eng = Engine(synthetic_world())
eng.replay_hoist(True)


def seeded():
    w = ActionWeights()
    first = eng.run_iteration(0, w, False, 12345)
    w.apply_episode(first.metrics[0], first.n_run[0], first.run_log[0, :first.n_run[0].sum()], first.n_def[0], first.def_log[0, :first.n_def[0].sum()])
    return w


def grown(w, shards, steps):
    eng.push(w)
    packets = torch.zeros(shards * N.PACKET_BYTES, dtype=torch.uint8, device="cuda")
    for s in range(steps):
        for r in range(shards):
            eng.device_rollout(12345, (s * shards + r) * B, B, 10, packets.data_ptr() + r * N.PACKET_BYTES)
        eng.device_apply(packets.data_ptr(), shards, packets.data_ptr(), 12345 + s)
        for r in range(1, shards):
            packets[r * N.PACKET_BYTES:r * N.PACKET_BYTES + 8 * N.STATS_LEN] = 0
    eng.pull(w)
    return w


states = {"fresh policy (ActionWeights::new)": ActionWeights(), "seeded": seeded(), "sustained, 1 GPU (48 updates)": grown(seeded(), 1, 48),
          "sustained, 8 shards (30 updates)": grown(seeded(), 8, 30), "fresh policy, 48 updates": grown(ActionWeights(), 1, 48)}
for name, w in states.items():
    eng.timing_reset()
    for rep in range(5):
        res = eng.rollout_batch(w, 12345, B, first_episode_index=1 + 7 * B, write_yearly=True)      # no replay mask: every episode sampled
    ms, n = eng.timing_read()
    g = res.n_gens.astype(float)
    print(f"{name:34s} stall {w.get('iterations_without_improvement'):9.0f}  lean grid {ms / n:6.3f} ms  generators {g.mean():5.1f} (max {g.max():.0f})  "
          f"offsets {res.n_offsets.mean():4.1f}  deficit actions {res.n_def.sum(1).mean():5.1f}  additional {res.n_act.sum(1).mean():5.1f}  "
          f"draws {res.n_draws.mean():6.1f}  chunks/search {res.n_chunks.sum() / g.sum():4.2f}", flush=True)
