"""Diagnostic: where does a heavy (replay) episode spend its cycles?  -DEG_STAMPS build (make -C eirgrid_amd/csrc stamps):
   EIRGRID_LIB=eirgrid_amd/libeirgrid_hip_stamps.so python scripts/heavy_stamps.py [per_year]"""
import os, sys
os.environ["EIRGRID_FETCH_FULL"] = "1"      # (the stamps sit at the end of act_log: whole rows, please)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eirgrid_amd import synthetic_world
from eirgrid_amd.engine import ActionWeights, Engine
per_year = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1638
rng = np.random.default_rng(11)
pol = ActionWeights()
types = list(range(15)) if len(sys.argv) > 3 and sys.argv[3] == "all" else [0, 4, 12, 7]      # "all": every radius class gets searched
run = [[int(3 * rng.choice(types) + rng.integers(0, 3)) for _ in range(per_year)] for _ in range(26)]
nr = np.array([len(l) for l in run], np.int32); nd = np.zeros(26, np.int32)
pol.apply_episode([-5e4, 0.7, 4e10, 1.0], nr, np.array([a for l in run for a in l], np.uint8), nd, np.zeros(0, np.uint8))
eng = Engine(synthetic_world())
mask = np.ones(n, np.uint8)
res = eng.rollout_batch(pol, 321, n, replay_mask=mask)
ms, k = eng.timing_read()
st = res.act_log[:, -256:].copy().view(np.uint64).astype(np.float64)
tot = st[:, 7].mean()
print(f"{n} replay episodes, {res.n_gens.mean():.0f} generators each, status ok {int((res.status == 0).sum())}; kernel {ms / max(k, 1):.2f} ms; mean episode cycles {tot:.0f}")
srch = st[:, 26].mean()
for name, col in (("placement (all searches incl. exact-scan ones)", 1), ("bookkeeping after a search (incl. field update)", 12), ("sampling / replay pick", 2),
                  ("year start aggregates", 0), ("yearly metrics", 4)):
    print(f"  {name:48s} {st[:, col].mean():12.0f} cycles {100 * st[:, col].mean() / tot:5.1f} %")
print(f"  heavy searches per episode {srch:.0f}: scan {st[:, 27].mean() / srch:.0f} cyc/search, candidates+records {st[:, 28].mean() / srch:.0f}, exact evaluation {st[:, 29].mean() / srch:.0f}, "
      f"field update {st[:, 30].mean() / max(res.n_gens.mean(), 1):.0f} cyc/add; chunks scanned/search {st[:, 24].mean() / srch:.1f}, candidates/search {st[:, 25].mean() / srch:.2f}")
print(f"  around the place_heavy call {st[:, 9].mean() / srch:.0f} cyc/search, between() {st[:, 10].mean() / srch:.0f}")
names = {0: "year-start aggregates", 1: "placement search", 2: "sampling / replay pick", 3: "deficit evaluate + nudges", 4: "yearly metrics + stores",
         5: "policy rows -> LDS", 12: "apply: generator bookkeeping", 13: "apply: offset", 14: "year: totals scalars", 15: "year: initial state",
         16: "episode start", 17: "glue: year loop back edge", 18: "glue: before aggregates", 19: "glue: loop top -> sampling", 20: "glue: sampled -> search/offset",
         21: "glue: search -> bookkeeping", 22: "glue: apply -> evaluate (logs)", 23: "phase-1 logs + back edge"}
print("  all slots (cycles per generator added):", {n: round(st[:, i].mean() / res.n_gens.mean()) for i, n in names.items() if st[:, i].mean() / tot >= 0.005})
