"""Diagnostic: where does k_replay_solo (csrc/eg_replay_solo.h) spend an episode's cycles?  -DEG_SOLO_STAMPS build:
   make -C eirgrid_amd/csrc ab AB=solostamps ABFLAGS=-DEG_SOLO_STAMPS
   EIRGRID_LIB=eirgrid_amd/libeirgrid_hip_ab_solostamps.so python scripts/solo_stamps.py [per_year] [n_replay] [n_sampled_beside]"""
import os, sys
os.environ["EIRGRID_FETCH_FULL"] = "1"      # (the stamps sit at the end of act_log: whole rows, please)
os.environ["EIRGRID_HELPER_WAVES"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eirgrid_amd import synthetic_world
from eirgrid_amd.engine import ActionWeights, Engine
per_year = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n_rep = int(sys.argv[2]) if len(sys.argv) > 2 else 1638
n_lean = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rng = np.random.default_rng(11)
pol = ActionWeights()
types = [0, 4, 12, 7]
run = [[int(3 * rng.choice(types) + rng.integers(0, 3)) for _ in range(per_year)] for _ in range(26)]
dfl = [[24, 21, 36, 33] for _ in range(26)]
nr = np.array([len(l) for l in run], np.int32); nd = np.full(26, 4, np.int32)
pol.apply_episode([-5e4, 0.7, 4e10, 1.0], nr, np.array([a for l in run for a in l], np.uint8), nd, np.array([a for l in dfl for a in l], np.uint8))
eng = Engine(synthetic_world())
n = n_rep + n_lean
mask = np.zeros(n, np.uint8); mask[:n_rep] = 1
for _ in range(2):
    res = eng.rollout_batch(pol, 321, n, replay_mask=mask)
ms, k = eng.timing_read()
full = res.act_log[:n_rep, -256:].copy().view(np.uint64).astype(np.float64)
st = full[:, :8]
tot = st[:, :7].sum(axis=1).mean()
g = res.n_gens[:n_rep].mean()
print(f"{n_rep} replay episodes of {g:.0f} generators beside {n_lean} sampled ones, status ok {int((res.status == 0).sum())} of {n}; mean episode cycles {tot:.0f}")
for name, col, per in (("set-up", 0, 1), ("script", 1, 1), ("placements: list entry", 2, g), ("placements: search", 3, g), ("placements: field update + list entry", 4, g),
                       ("yearly rows", 5, 26), ("header + statistics epilogue", 6, 1)):
    print(f"  {name:40s} {st[:, col].mean():12.0f} cycles {100 * st[:, col].mean() / tot:5.1f} %   ({st[:, col].mean() / per:.0f} each)")
if full[:, 13].mean() > 0:      # built with -DEG_STAMPS as well: place_heavy's own counters
    n = full[:, 13].mean()
    print(f"  place_heavy: {n:.0f} searches per episode: scan {full[:, 8].mean() / n:.0f} cycles, candidates {full[:, 9].mean() / n:.0f}, exact evaluation + rest {full[:, 10].mean() / n:.0f}; "
          f"chunks scanned per search {full[:, 11].mean() / n:.1f}, candidates {full[:, 12].mean() / n:.2f}")
