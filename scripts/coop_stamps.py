"""Where the hoisted replay's wave 0 spends its cycles (diagnostic build: make -C eirgrid_amd/csrc ab AB=coopstamps ABFLAGS=-DEG_COOP_STAMPS,
then EIRGRID_LIB=eirgrid_amd/libeirgrid_hip_ab_coopstamps.so python scripts/coop_stamps.py [grow=48]): the sustained state of configs[2], one
all-replay batch of 64 episodes on an otherwise idle GPU, and the same beside a lean grid (16 384 x 10 %)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from eirgrid_amd import synthetic_world
from eirgrid_amd import _native as N
from eirgrid_amd.engine import ActionWeights, Engine
from eirgrid_amd.parallel import BatchTrainer

GROW = int(sys.argv[1]) if len(sys.argv) > 1 else 48
SHARDS = int(sys.argv[2]) if len(sys.argv) > 2 else 1      # > 1: the policy grown at the global batch of that many shards (hoist_probe.py)
eng = Engine(synthetic_world())
w = ActionWeights()
first = eng.run_iteration(0, w, False, 12345)
w.apply_episode(first.metrics[0], first.n_run[0], first.run_log[0, :first.n_run[0].sum()], first.n_def[0], first.def_log[0, :first.n_def[0].sum()])
if SHARDS > 1:
    eng.replay_hoist(True)      # (the same policies either way — tests/test_gpu_replay_hoist.py — and an eighth of the time)
    eng.push(w)
    packets = torch.zeros(SHARDS * N.PACKET_BYTES, dtype=torch.uint8, device="cuda")
    for s in range(GROW):
        for r in range(SHARDS):
            eng.device_rollout(12345, (s * SHARDS + r) * 16384, 16384, 10, packets.data_ptr() + r * N.PACKET_BYTES)
        eng.device_apply(packets.data_ptr(), SHARDS, packets.data_ptr(), 12345 + s)
        for r in range(1, SHARDS):
            packets[r * N.PACKET_BYTES:r * N.PACKET_BYTES + 8 * N.STATS_LEN] = 0
    eng.pull(w)
else:
    tr = BatchTrainer(eng, w, 16384, 12345, replay_fraction=0.1)
    for _ in range(GROW):
        tr.step()
    tr.sync()
eng.replay_hoist(True)
names = ["set-up", "script (phase 1)", "year changes (requests)", "search: the lane's cells", "search: the wave's two largest", "field updates + barrier", "search: exchange barrier", "search: decision (+ slow path)"]
for label, n, mask in (("alone", 64, np.ones(64, np.uint8)), ("beside a lean grid", 16384, (np.arange(16384) % 10 == 0).astype(np.uint8))):
    for rep in range(2):
        res = eng.rollout_batch(w, 12345, n, replay_mask=mask)
    st = (C.c_uint64 * 8)()
    N.check(N.lib().eg_debug_hoist_stamps(eng.h, st))
    tot = sum(st)
    g = int(res.n_gens[0])
    print(f"{label}: {g} generators, served {eng.replay_hoist_stats()[1]}, {tot} cycles = {tot / 2.4e3:.1f} us at 2.4 GHz (100 MHz counter? see below)")
    for k in range(8):
        print(f"   {names[k]:26s} {st[k]:10d}  {100.0 * st[k] / max(tot, 1):5.1f} %   per placement {st[k] / max(g, 1):8.1f}")
