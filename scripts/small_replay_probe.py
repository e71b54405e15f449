"""The CLI's all-replay phase (the last 10 % of a run, --force-full-simulation: core/multi_simulation.rs:38-39, :437-465) at its batch
size: 1 024 replay episodes of a LONG best list per batch — which kernel should run them?  k_rollout<1,2> (episode wave + helper wave,
256 registers, 2 waves per SIMD), k_rollout<0,2> (one wave per episode, 128 registers, 4 per SIMD), or the replay hoist.
  python scripts/small_replay_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from eirgrid_amd import synthetic_world
from eirgrid_amd.engine import ActionWeights, Engine
from eirgrid_amd.parallel import BatchTrainer

world = synthetic_world()
eng = Engine(world)
w = ActionWeights()
first = eng.run_iteration(0, w, False, 12345)
w.apply_episode(first.metrics[0], first.n_run[0], first.run_log[0, :first.n_run[0].sum()], first.n_def[0], first.def_log[0, :first.n_def[0].sum()])
eng.replay_hoist(True)
tr = BatchTrainer(eng, w, 16384, 12345, replay_fraction=0.1)
for _ in range(48):
    tr.step()
tr.sync()
eng.close()
print("best list:", sum(len(l) for l in w.lists(0)), "actions")
mask = np.ones(1024, np.uint8)
for label, env, hoist in (("default (helper-wave kernel <1,2>)", None, False), ("one wave per episode <0,2>", "0", False), ("replay hoist", None, True)):
    if env is not None:
        os.environ["EIRGRID_HELPER_WAVES"] = env
    e = Engine(world)
    os.environ.pop("EIRGRID_HELPER_WAVES", None)
    e.replay_hoist(hoist)
    e.upload_snapshot(w)
    for _ in range(5):
        e.launch(12345, 0, 1024, mask)
    e.sync(); e.timing_reset(); torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(50):
        e.launch(12345, 0, 1024, mask)
    e.sync()
    el = time.perf_counter() - t0
    ms, n = e.timing_read()
    res = e.fetch(1024)
    print(f"{label:38s} kernels {ms / n:7.3f} ms per batch, wall {el / 50 * 1e3:7.3f} ms; generators per episode {res.n_gens.mean():.0f}, ok {int((res.status == 0).sum())}", flush=True)
    e.close()
