"""Diagnostic: where does an episode spend its cycles?  Needs the -DEG_STAMPS build (make -C eirgrid_amd/csrc stamps):
   EIRGRID_LIB=eirgrid_amd/libeirgrid_hip_stamps.so python scripts/stamps.py
Every cycle of an episode is charged to exactly one slot.  Shares only; never quote this build's run time."""
import os, sys
os.environ["EIRGRID_FETCH_FULL"] = "1"      # (the stamps sit at the end of act_log: whole rows, please)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eirgrid_amd import synthetic_world
from eirgrid_amd.engine import ActionWeights, Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
eng = Engine(synthetic_world()); pol = ActionWeights()
eng.upload_snapshot(pol, write_yearly=(os.environ.get('EG_NO_YEARLY') is None)); eng.launch(12345, 0, B); eng.sync()
res = eng.fetch(B)
st = res.act_log[:, -256:].copy().view(np.uint64).astype(np.float64)   # [B, 32]
names = {0: "year-start aggregates", 1: "placement search", 2: "sampling (rng + walks)", 3: "deficit evaluate + nudges",
         4: "yearly metrics + stores", 5: "policy rows -> LDS", 6: "(inside placement) spin on helper flags", 12: "apply: generator bookkeeping",
         13: "apply: offset", 14: "year: totals scalars", 15: "year: initial state", 16: "episode start (tables, seed)",
         17: "glue: year loop back edge", 18: "glue: before aggregates", 19: "glue: loop top -> sampling",
         20: "glue: sampled -> search/offset", 21: "glue: search -> bookkeeping", 22: "glue: apply -> evaluate (logs)",
         23: "phase-1 logs + back edge", 24: "glue: before n_add", 25: "glue: loop exit -> metrics", 26: "episode end"}
tot = st[:, 7].mean()
print(f"B={B}  mean episode cycles {tot:.0f}  (min {st[:,7].min():.0f} max {st[:,7].max():.0f})  gens/ep {res.n_gens.mean():.1f}")
# a launch of B <= 1024 episodes (all resident at once) lasts as long as its slowest episode: the spread of the episodes' durations
cyc = np.sort(st[:, 7])
print(f"  episode cycles: mean {cyc.mean():.0f}  p50 {cyc[len(cyc) // 2]:.0f}  p90 {cyc[int(0.9 * len(cyc))]:.0f}  p99 {cyc[int(0.99 * len(cyc))]:.0f}  max {cyc[-1]:.0f}"
      f"  (max / mean {cyc[-1] / cyc.mean():.2f}, p99 / mean {cyc[int(0.99 * len(cyc))] / cyc.mean():.2f});  "
      f"searches/ep mean {st[:, 11].mean():.1f} max {st[:, 11].max():.0f};  chunks/ep mean {st[:, 8].mean():.1f} max {st[:, 8].max():.0f};  draws/ep mean {res.n_draws.mean():.0f}")
acc = 0.0
for i, n in names.items():
    v = st[:, i].mean(); acc += v
    print(f"  {n:34s} {v:10.0f} cycles  {100 * v / tot:5.1f} %")
print(f"  {'unattributed':34s} {tot - acc:10.0f} cycles  {100 * (tot - acc) / tot:5.1f} %")
print(f"  placement detail: searches/ep {st[:, 11].mean():.1f}  chunks/search {st[:, 8].sum() / st[:, 11].sum():.2f}  "
      f"generator loop {st[:, 9].mean():.0f} cyc/ep ({st[:, 9].sum() / st[:, 8].sum():.0f}/chunk)  reduce+select {st[:, 10].mean():.0f} cyc/ep ({st[:, 10].sum() / st[:, 8].sum():.0f}/chunk)  "
      f"rest (loads, setup, exit test) {(st[:, 1] - st[:, 9] - st[:, 10]).mean():.0f} cyc/ep")
k = int(np.argmax(st[:, 7]))
nm = st[:, 30].sum()
print(f"  helper results merged/ep {st[:, 30].mean():.1f}: episode wave spun {st[:, 6].sum() / nm:.0f} cyc per merge; helper per chunk: load wait {st[:, 27].sum() / nm:.0f}, score {st[:, 28].sum() / nm:.0f}, reduce {st[:, 29].sum() / nm:.0f}")
print(f"  slowest episode {k}: total {st[k, 7]:.0f}, placement {st[k, 1]:.0f}, searches {st[k, 11]:.0f}, chunks {st[k, 8]:.0f}, gens {res.n_gens[k]}, year-start {st[k, 0]:.0f}, sampling {st[k, 2]:.0f}")

# the launch lasts as long as its slowest episode: the same table for the slowest 1 % of the batch
order = np.argsort(st[:, 7])[::-1][:max(1, B // 100)]
sl = st[order]
tot = sl[:, 7].mean()
print(f"slowest {len(order)} episodes: mean cycles {tot:.0f}, gens {res.n_gens[order].mean():.1f}, searches {sl[:, 11].mean():.1f}, chunks/search {sl[:, 8].sum() / sl[:, 11].sum():.2f}, "
      f"actions logged {res.n_act[order].sum(axis=1).mean():.1f} sampled + {res.n_def[order].sum(axis=1).mean():.1f} repair")
for i, n in names.items():
    v = sl[:, i].mean()
    if v / tot >= 0.01: print(f"  {n:34s} {v:10.0f} cycles  {100 * v / tot:5.1f} %")
print(f"  placement detail: generator loop {sl[:, 9].mean():.0f} cyc/ep ({sl[:, 9].sum() / sl[:, 8].sum():.0f}/chunk)  reduce+select {sl[:, 10].mean():.0f} cyc/ep  rest {(sl[:, 1] - sl[:, 9] - sl[:, 10]).mean():.0f} cyc/ep; per search {sl[:, 1].sum() / sl[:, 11].sum():.0f}")
